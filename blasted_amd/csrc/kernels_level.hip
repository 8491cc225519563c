// kernels_level.hip -- level-scheduled (exact, in-order) sweeps.
//
// The reference's `level_sgs` and `async_level_ilu0` types (src/solverops_levels_sgs.cpp:52-123,
// src/solverops_levels_ilu0.cpp:58-105) and its sequential variants (`seqilu0`, `sapilu0`:
// threadedapply = false, src/solverfactory.cpp:88-107) all produce the result of ONE in-order pass over
// the rows.  The reference gets there with a list of row ranges whose rows are mutually independent
// (computeLevels, src/levelschedule.cpp:13-72 -- ranges of CONSECUTIVE rows, so it relies on the caller
// having reordered the matrix) and an `omp parallel for` per range.
//
// Here the levels are the longest-path depths of the dependency DAG itself, so no reordering is needed:
//   level(i) = 1 + max{ level(j) : j < i, A_ij != 0 or A_ji != 0 }      (0 without such j)
// Rows of one level share no stored block in either direction, every row a level-l row reads through its
// lower (upper) part lies in a level < l (> l).  One launch per level, ascending levels for an ascending
// pass and descending levels for a descending one, therefore reproduces the serial pass exactly: each
// row is computed by the same expression from the final values of the rows it depends on.
// On a matrix whose rows are renumbered level by level, the reference's computeLevels returns exactly
// these level boundaries (tests/test_gpu_parity.py::test_level_schedule_matches_reference_levels).
//
// Build (once per pattern, all in HBM): chaotic fixed-point iteration of the definition above with
// atomicMax (pull from the lower neighbours, push to the upper ones, so a structurally non-symmetric
// pattern is symmetrised on the fly) until a pass changes nothing; stable radix sort of the rows by
// level (hipCUB); level boundaries from the sorted keys.
#include <hipcub/hipcub.hpp>

#include "ctx.hpp"
#include "lanes.hpp"
#include "sweep_geo.hpp"

namespace bhip {

namespace {

__global__ __launch_bounds__(256) void level_relax_kernel(const Pattern pat, int *level, int *changed)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= pat.nbrows)
		return;
	const int rbeg = pat.browptr[i], rend = pat.browptr[i + 1];
	int li = __hip_atomic_load(&level[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	int m = li;
	for (int jj = rbeg; jj < rend; jj++) {
		const int j = pat.bcolind[jj];
		if (j < i) {
			const int lj = __hip_atomic_load(&level[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			m = lj + 1 > m ? lj + 1 : m;
		}
	}
	bool ch = false;
	if (m > li) {
		atomicMax(&level[i], m);
		li = m;
		ch = true;
	}
	for (int jj = rbeg; jj < rend; jj++) {
		const int j = pat.bcolind[jj];
		if (j > i && __hip_atomic_load(&level[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= li) {
			atomicMax(&level[j], li + 1);
			ch = true;
		}
	}
	if (ch)
		*changed = 1;
}

__global__ __launch_bounds__(256) void iota_kernel(int *v, int n)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n)
		v[i] = i;
}

// ptr[l] = first position of level l in the sorted key array; ptr[nlevels] = n
__global__ __launch_bounds__(256) void level_bounds_kernel(const int *keys, int n, int nlevels, int *ptr)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n)
		return;
	const int k = keys[i];
	if (i == 0 || keys[i - 1] != k)
		ptr[k] = i;
	if (i == n - 1)
		ptr[nlevels] = n;
}

// One level of a sweep: the rows listed in rows[0..count) are mutually independent.  Lane mapping and
// arithmetic as in kernels_sweep.hip (a group of G lanes per block-row, SUB lanes per block, reduction on
// the VALU); indices come straight from HBM since the rows of a level are scattered.
template <int BS, bool RM, int PART, int POST, int DSRC>
__global__ __launch_bounds__(256) void level_sweep_kernel(const SweepArgs a, const int *__restrict__ rows,
                                                          const int count)
{
	using Ge = Geo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, G = Ge::G, NB = Ge::NB, BS2 = BS * BS;
	constexpr int RPW = Ge::RPW, RSTEP = Ge::RSTEP;
	constexpr bool DIAG_RIDES = PART == PART_UPPER && (DSRC == D_VALS_DIAG || DSRC == D_RECIP_DIAG);

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const int slot = t / SUB, u = t % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(G - 1);

	const int ls = blockIdx.x * RSTEP + wave * RPW + g;
	const bool ok = ls < count;
	const int row = ok ? rows[ls] : 0;
	const int rp0 = a.pat.browptr[row], rp1 = a.pat.browptr[row + 1], dg = a.pat.diagind[row];
	int jbeg = 0, jend = 0;
	if (ok) {
		if (PART == PART_LOWER) {
			jbeg = rp0;
			jend = dg;
		} else if (PART == PART_UPPER) {
			jbeg = DIAG_RIDES ? dg : dg + 1;
			jend = rp1;
		} else if (PART == PART_OFFDIAG || PART == PART_ALL) {
			jbeg = rp0;
			jend = rp1;
		}
	}

	double d = 0.0;
	if (DSRC == D_DBLOCKS && ok && active && slot == 0)
		d = a.dvals[(long)row * BS2 + e];
	double rv = 0.0;
	if (ok && r < BS && a.rhs) {
		rv = a.rhs[(long)row * BS + r];
		if (a.rscale)
			rv *= a.rscale[(long)row * BS + r];
	}

	double acc = 0.0;
	if (PART != PART_NONE) {
		if (active) {
			for (int jj = jbeg + slot; jj < jend; jj += NB) {
				if (PART == PART_OFFDIAG && jj == dg)
					continue;
				const double bv = a.vals[(long)jj * BS2 + e];
				if (DIAG_RIDES && jj == dg) {  // item 0 of the row, block slot 0
					d = (DSRC == D_VALS_DIAG) ? bv : 1.0 / bv;
					continue;
				}
				const int col = a.pat.bcolind[jj];
				acc += bv * a.xin[(long)col * BS + c];
			}
		}
		acc = allreduce_bits<Ge::LOBIT, Ge::HIBIT>(acc);
	}

	double out;
	if (POST == POST_SUB) {
		out = rv - acc;
	} else if (POST == POST_D_SUB || POST == POST_SUB_D) {
		const double w = (POST == POST_D_SUB) ? rv - acc : acc;
		const double wc = __shfl(w, gbase + c, 64);
		const double pr = allreduce_bits<Ge::LOBIT, Ge::HIBIT>((active && slot == 0) ? d * wc : 0.0);
		out = (POST == POST_D_SUB) ? pr : rv - pr;
	} else {
		out = a.a * acc;
		if (a.b != 0.0)
			out += a.b * rv;
	}
	if (ok && slot == 0 && c == 0 && r < BS)
		a.xout[(long)row * BS + r] = out;
}

template <int BS, bool RM, int PART, int POST, int DSRC>
void run_levels(const SweepArgs &a, const LevelSchedule &ls, hipStream_t s)
{
	constexpr int RSTEP = Geo<BS>::RSTEP;
	for (int q = 0; q < ls.nlevels; q++) {
		const int l = a.descending ? ls.nlevels - 1 - q : q;
		const int first = ls.ptr[l], count = ls.ptr[l + 1] - first;
		if (count <= 0)
			continue;
		const unsigned grid = (unsigned)((count + RSTEP - 1) / RSTEP);
		hipLaunchKernelGGL((level_sweep_kernel<BS, RM, PART, POST, DSRC>), dim3(grid), dim3(256), 0, s, a,
		                   ls.rows + first, count);
	}
}

template <int BS, bool RM>
void level_dispatch_ops(const SweepArgs &a, Part part, Post post, DSrc dsrc, const LevelSchedule &ls,
                        hipStream_t s)
{
#define BHIP_CASE(P, Q, D)                                  \
	if (part == P && post == Q && dsrc == D) {              \
		run_levels<BS, RM, P, Q, D>(a, ls, s);              \
		return;                                             \
	}
	BHIP_CASE(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASE(PART_UPPER, POST_D_SUB, D_VALS_DIAG)
	BHIP_CASE(PART_UPPER, POST_D_SUB, D_RECIP_DIAG)
	BHIP_CASE(PART_LOWER, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE(PART_UPPER, POST_SUB_D, D_DBLOCKS)
	BHIP_CASE(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS)
#undef BHIP_CASE
	BHIP_FAIL(BLASTED_HIP_EINVAL, "launch_level_sweep: operator combination not instantiated");
}

template <int BS>
void level_dispatch_layout(const SweepArgs &a, Part part, Post post, DSrc dsrc, const LevelSchedule &ls,
                           hipStream_t s)
{
	if (BS > 1 && a.pat.rowmajor)
		level_dispatch_ops<BS, true>(a, part, post, dsrc, ls, s);
	else
		level_dispatch_ops<BS, false>(a, part, post, dsrc, ls, s);
}

template <typename T>
T *lvl_alloc(size_t count)
{
	void *q = nullptr;
	BHIP_CHECK(hipMalloc(&q, sizeof(T) * (count ? count : 1)));
	return static_cast<T *>(q);
}

}  // namespace

void free_level_schedule(LevelSchedule &ls)
{
	if (ls.rows)
		(void)hipFree(ls.rows);
	if (ls.level)
		(void)hipFree(ls.level);
	ls = LevelSchedule();
}

void build_level_schedule(const Pattern &pat, LevelSchedule &ls, hipStream_t s)
{
	free_level_schedule(ls);
	const int n = pat.nbrows;
	if (n == 0) {
		ls.ptr.assign(1, 0);
		ls.built = true;
		return;
	}
	const unsigned grid = (unsigned)((n + 255) / 256);
	int *level = lvl_alloc<int>(n), *keys = nullptr, *iota = nullptr, *rows = nullptr, *flags = nullptr;
	int *dptr = nullptr;
	void *tmp = nullptr;
	try {
		BHIP_CHECK(hipMemsetAsync(level, 0, sizeof(int) * (size_t)n, s));
		constexpr int BATCH = 8;
		flags = lvl_alloc<int>(BATCH);
		int hflags[BATCH];
		bool fixed = false;
		long passes = 0;
		for (long done = 0; done < (long)n + 2 && !fixed; done += BATCH) {
			BHIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int) * BATCH, s));
			for (int q = 0; q < BATCH; q++)
				hipLaunchKernelGGL(level_relax_kernel, dim3(grid), dim3(256), 0, s, pat, level, flags + q);
			BHIP_CHECK(hipMemcpyAsync(hflags, flags, sizeof(int) * BATCH, hipMemcpyDeviceToHost, s));
			BHIP_CHECK(hipStreamSynchronize(s));
			passes += BATCH;
			for (int q = 0; q < BATCH; q++)
				if (!hflags[q])
					fixed = true;
		}
		if (!fixed)
			BHIP_FAIL(BLASTED_HIP_ERUNTIME, "level schedule: dependency depths did not settle");
		ls.build_passes = passes;

		keys = lvl_alloc<int>(n);
		iota = lvl_alloc<int>(n);
		rows = lvl_alloc<int>(n);
		hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, s, iota, n);
		size_t bytes = 0;
		BHIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, level, keys, iota, rows, n, 0, 31, s));
		BHIP_CHECK(hipMalloc(&tmp, bytes ? bytes : 1));
		BHIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, level, keys, iota, rows, n, 0, 31, s));
		int maxlevel = 0;
		BHIP_CHECK(hipMemcpyAsync(&maxlevel, keys + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
		BHIP_CHECK(hipStreamSynchronize(s));
		const int nlevels = maxlevel + 1;
		dptr = lvl_alloc<int>((size_t)nlevels + 1);
		BHIP_CHECK(hipMemsetAsync(dptr, 0xff, sizeof(int) * ((size_t)nlevels + 1), s));
		hipLaunchKernelGGL(level_bounds_kernel, dim3(grid), dim3(256), 0, s, keys, n, nlevels, dptr);
		ls.ptr.resize((size_t)nlevels + 1);
		BHIP_CHECK(hipMemcpyAsync(ls.ptr.data(), dptr, sizeof(int) * ((size_t)nlevels + 1),
		                          hipMemcpyDeviceToHost, s));
		BHIP_CHECK(hipStreamSynchronize(s));
		BHIP_CHECK(hipGetLastError());
		for (int l = 0; l <= nlevels; l++)
			if (ls.ptr[l] < 0 || (l > 0 && ls.ptr[l] <= ls.ptr[l - 1]))
				BHIP_FAIL(BLASTED_HIP_ERUNTIME, "level schedule: empty level (internal error)");
		ls.nlevels = nlevels;
		ls.rows = rows;
		ls.level = level;
		rows = nullptr;
		level = nullptr;
		ls.built = true;
	} catch (...) {
		for (void *q : {(void *)level, (void *)keys, (void *)iota, (void *)rows, (void *)flags, (void *)dptr, tmp})
			if (q)
				(void)hipFree(q);
		throw;
	}
	for (void *q : {(void *)keys, (void *)iota, (void *)flags, (void *)dptr, tmp})
		if (q)
			(void)hipFree(q);
}

// One exact in-order pass of the operator, in place on a.xout (a.xin must equal a.xout), as
// ls.nlevels launches.  Returns the number of launches.
int launch_level_sweep(const SweepArgs &a, Part part, Post post, DSrc dsrc, const LevelSchedule &ls,
                       hipStream_t s)
{
	if (!ls.built)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "launch_level_sweep: no level schedule");
	if (a.xin != a.xout)
		BHIP_FAIL(BLASTED_HIP_EINVAL, "launch_level_sweep: level sweeps run in place");
	switch (a.pat.bs) {
	case 1: level_dispatch_layout<1>(a, part, post, dsrc, ls, s); break;
	case 2: level_dispatch_layout<2>(a, part, post, dsrc, ls, s); break;
	case 3: level_dispatch_layout<3>(a, part, post, dsrc, ls, s); break;
	case 4: level_dispatch_layout<4>(a, part, post, dsrc, ls, s); break;
	case 5: level_dispatch_layout<5>(a, part, post, dsrc, ls, s); break;
	case 7: level_dispatch_layout<7>(a, part, post, dsrc, ls, s); break;
	case 8: level_dispatch_layout<8>(a, part, post, dsrc, ls, s); break;
	default: BHIP_FAIL(BLASTED_HIP_ENOTIMPL, "block size not instantiated (1,2,3,4,5,7,8)");
	}
	BHIP_CHECK(hipGetLastError());
	return ls.nlevels;
}

}  // namespace bhip
