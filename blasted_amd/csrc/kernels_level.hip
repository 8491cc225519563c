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
// these level boundaries (tests/test_gpu_levels.py::test_level_schedule_bit_exact).
//
// Two ways to run a pass:
//  * launch_level_sweep: one launch per level (kernel boundaries order the levels);
//  * launch_syncfree_sweep: ONE launch per pass.  Workgroup b owns positions [b*C, (b+1)*C) of the
//    level-ordered row list; its waves request their rows' blocks and indices -- none of which depends
//    on other rows -- and then poll the iterate entries they depend on until those stop being the
//    "pending" bit pattern the output vector was filled with.  A row only ever waits for rows at
//    earlier positions, i.e. for workgroups with a smaller id.  Each XCD starts its workgroups in
//    increasing id order, so the unfinished workgroup with the smallest id is always resident and never
//    waits: forward progress by induction (the same in-order-dispatch property rocSPARSE's csrsv relies
//    on).
//    Iterate entries are published and polled with relaxed agent-scope atomics (coherent across the
//    8 XCD L2s); the 8-byte value is its own ready flag, so no fence is needed.  Spins are bounded: a
//    wave that runs out sets an abort flag, every wave leaves, and the caller falls back to per-level
//    launches.  Levels overlap and the matrix stream is prefetched while waiting, so the pass costs
//    about one sweep's traffic plus (number of levels) x (one L2 round trip) of critical path.
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

// The levels of a pattern whose dependencies are all STORED as lower entries (a structurally symmetric pattern),
// in one launch: one lane per row in natural order, a row waits for the levels of its lower neighbours -- pre-filled
// with -1 -- and publishes 1 + their maximum.  Neighbours inside the same wave (the x-line of a stencil: a chain of
// 64) are read through lane exchanges, so a chain inside a wave costs an exchange per link instead of a memory
// round trip; lanes never block (a row of the wave may wait for another), the wave loops until all its rows are
// done.  Workgroups are started in row order, a row waits for earlier rows only.  256^3: one launch instead of 328
// relaxation passes (round 2: 158 ms for the schedule of the headline matrix, 130 of them those passes).
constexpr int LVF_SPIN_LIMIT = 1 << 22;

__global__ __launch_bounds__(256) void level_poll_kernel(const Pattern pat, int *level, int *ctl)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	const int lane = threadIdx.x & 63;
	const int wbase = i - lane;  // first row of this wave
	const bool rowok = i < pat.nbrows;
	const int rbeg = rowok ? pat.browptr[i] : 0;
	const int nlow = rowok ? pat.diagind[i] - rbeg : 0;  // (columns ascend: the lower entries come first)
	int nmax = nlow;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const int o = __shfl_xor(nmax, off, 64);
		nmax = o > nmax ? o : nmax;
	}
	// the first eight lower neighbours live in registers (a chain inside the wave is walked without memory operations)
	constexpr int NREG = 8;
	int cj[NREG];
#pragma unroll
	for (int q = 0; q < NREG; q++)
		cj[q] = q < nlow ? pat.bcolind[rbeg + q] : wbase;
	int lv = rowok ? -1 : 0;
	int mext = 0;          // 1 + the highest level among the neighbours outside the wave seen so far
	unsigned seen = 0u;    // those neighbours (the first 32 entries of the row) that have been seen: not read again
	int spins = 0;
	for (;;) {
		// one attempt per lane; the loops over a row's lower entries are wave-uniform because of the lane exchange in them
		bool ready = lv < 0;
		int m = mext;
		// neighbours inside the wave first (lane exchanges only) ...
#pragma unroll
		for (int q = 0; q < NREG; q++) {
			if (q >= nmax)
				break;
			const int j = cj[q];
			const int inw = __shfl(lv, (j - wbase) & 63, 64);  // its level if it is a row of this wave
			if (q < nlow && lv < 0 && j >= wbase) {
				if (inw < 0)
					ready = false;
				else
					m = inw + 1 > m ? inw + 1 : m;
			}
		}
		// ... and the others are asked for only by a row that has those: along a chain inside the wave that is one
		// lane at a time instead of sixty-four lanes polling on every round
		const bool ask = ready || (spins & 7) == 0;  // (and everybody now and then: what is there early is seen early)
#pragma unroll
		for (int q = 0; q < NREG; q++) {
			if (q >= nmax)
				break;
			const int j = cj[q];
			if (q < nlow && lv < 0 && j < wbase && !((seen >> q) & 1u)) {
				int dj = -1;
				if (ask)
					dj = __hip_atomic_load(&level[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (dj >= 0) {
					mext = dj + 1 > mext ? dj + 1 : mext;
					m = dj + 1 > m ? dj + 1 : m;
					seen |= 1u << q;
				} else
					ready = false;
			}
		}
		for (int q = NREG; q < nmax; q++) {
			const bool has = q < nlow && lv < 0;
			const int j = has ? pat.bcolind[rbeg + q] : wbase;
			const int inw = __shfl(lv, (j - wbase) & 63, 64);
			if (!has)
				continue;
			int dj;
			if (j >= wbase)
				dj = inw;
			else if (q < 32 && ((seen >> q) & 1u))
				continue;
			else {
				dj = __hip_atomic_load(&level[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (dj >= 0) {
					mext = dj + 1 > mext ? dj + 1 : mext;
					if (q < 32)
						seen |= 1u << q;
				}
			}
			if (dj < 0)
				ready = false;
			else
				m = dj + 1 > m ? dj + 1 : m;
		}
		if (lv < 0 && ready) {
			lv = m;
			__hip_atomic_store(&level[i], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (__builtin_amdgcn_ballot_w64(lv < 0) == 0ull)
			return;
		spins++;
		if (spins > LVF_SPIN_LIMIT ||
		    ((spins & 255) == 0 && __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
			if (lane == 0)
				__hip_atomic_store(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
	}
}

// The same fixed point by ONE wave walking the rows in order: row i's level is final once rows < i are, so one
// pass is exact whatever the depth of the dependency graph -- O(nnz) work on a ~2 us dependent step per row.
// The fall-back for patterns whose depth is of the order of the row count (banded, one-dimensional orderings),
// on which the parallel passes above need one pass per level, i.e. O(nbrows * nnz) work.
__global__ __launch_bounds__(64) void level_serial_kernel(const Pattern pat, int *level)
{
	const int lane = threadIdx.x;
	for (int i = 0; i < pat.nbrows; i++) {
		const int rbeg = pat.browptr[i], rend = pat.browptr[i + 1];
		int m = __hip_atomic_load(&level[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		for (int jj = rbeg + lane; jj < rend; jj += 64) {
			const int j = pat.bcolind[jj];
			if (j < i) {
				const int lj = __hip_atomic_load(&level[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				m = lj + 1 > m ? lj + 1 : m;
			}
		}
		for (int off = 32; off > 0; off >>= 1) {
			const int o = __shfl_xor(m, off, 64);
			m = o > m ? o : m;
		}
		if (lane == 0)
			__hip_atomic_store(&level[i], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		for (int jj = rbeg + lane; jj < rend; jj += 64) {
			const int j = pat.bcolind[jj];
			if (j > i)
				atomicMax(&level[j], m + 1);
		}
		// this wave's own stores and atomics must be visible to its next iterations' loads
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
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

// meta[pos] = {row, browptr[row], diagind[row], browptr[row+1]} in level order; lens[0..1] = longest
// strictly-lower / strictly-upper part of any row
__global__ __launch_bounds__(256) void level_meta_kernel(const Pattern pat, const int *rows, int4 *meta,
                                                         int *lens)
{
	const int k = blockIdx.x * 256 + threadIdx.x;
	int nlow = 0, nup = 0;
	if (k < pat.nbrows) {
		const int row = rows[k];
		const int rp0 = pat.browptr[row], rp1 = pat.browptr[row + 1], dg = pat.diagind[row];
		meta[k] = make_int4(row, rp0, dg, rp1);
		nlow = dg - rp0;
		nup = rp1 - dg - 1;
	}
	// one atomic per wave (one per lane made this kernel six times slower than its loads)
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const int a = __shfl_xor(nlow, off, 64), b = __shfl_xor(nup, off, 64);
		nlow = a > nlow ? a : nlow;
		nup = b > nup ? b : nup;
	}
	if ((threadIdx.x & 63) == 0) {
		atomicMax(lens, nlow);
		atomicMax(lens + 1, nup);
	}
}

constexpr unsigned long long SF_PENDING = 0xFFF8DEADBEEF0001ull;  // a NaN payload arithmetic never produces
constexpr int SF_SPIN_LIMIT = 1 << 22;

__global__ __launch_bounds__(256) void sf_fill_kernel(unsigned long long *x, long n, unsigned long long v)
{
	const long i = (long)blockIdx.x * 256 + threadIdx.x;
	if (i < n)
		x[i] = v;
}

// "levelnowait=1" (measurements only, WRONG results): the iterate is pre-filled with zeros instead of the pending
// pattern, so no row ever waits -- what a single-launch pass costs as a pure stream, without its dependencies
int g_sf_nowait = 0;

__device__ __forceinline__ double sf_load(const double *p)
{
	const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
	                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return __longlong_as_double((long long)v);
}

__device__ __forceinline__ void sf_store(double *p, double v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
	                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool sf_pending(double v)
{
	return (unsigned long long)__double_as_longlong(v) == SF_PENDING;
}

// One exact pass as a single launch (see the file comment).  a.xout: the iterate being produced,
// pre-filled with SF_PENDING; a.xin: the previous iterate, read for the stored blocks that are not
// dependencies of the pass direction (relaxation only).  ctl[1] = abort flag.
// A wave owns ST consecutive steps of RPW positions of the level-ordered row list (a workgroup 4 * ST
// * RPW consecutive positions, workgroups in grid order).  Every block, index and right-hand side of
// all ST steps is requested before the first wait; then the wave polls what is still pending, step by
// step, and commits a row as soon as its group of lanes has everything.
// KF = block passes held in registers: every row part must fit KF * NB blocks (checked by the launcher).
template <int BS, bool RM, int PART, int POST, int DSRC, int KF, int ST>
__global__ __launch_bounds__(256) void sf_sweep_kernel(const SweepArgs a, const int4 *__restrict__ meta,
                                                       const int4 *__restrict__ head, const int count, int *ctl)
{
	using Ge = Geo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, G = Ge::G, NB = Ge::NB, BS2 = BS * BS;
	constexpr int RPW = Ge::RPW;
	constexpr bool DIAG_RIDES = PART == PART_UPPER && (DSRC == D_VALS_DIAG || DSRC == D_RECIP_DIAG);
	constexpr unsigned long long GMASK = G == 64 ? ~0ull : ((1ull << G) - 1ull);

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / G, t = lane % G;
	const int slot = t / SUB, u = t % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(G - 1);
	const bool desc = a.descending != 0;
	const long base = ((long)blockIdx.x * 4 + wave) * (ST * RPW);

	// head (level-ordered copies only): the column indices of a row part's first four blocks, addressable
	// from the position alone, so the polls of those blocks start together with the block loads
	constexpr bool HEAD8 = KF * NB > 4;  // second half of the head only where a lane can reach it
	int4 mt[ST], hd[ST], hd2[ST];
	bool okk[ST];
#pragma unroll
	for (int st = 0; st < ST; st++) {
		const long pos = base + st * RPW + g;
		okk[st] = pos < count;
		mt[st] = make_int4(0, 0, 0, 0);
		hd[st] = make_int4(0, 0, 0, 0);
		hd2[st] = make_int4(0, 0, 0, 0);
		if (okk[st]) {
			const long p = desc ? count - 1 - pos : pos;
			mt[st] = meta[p];
			if (head) {
				hd[st] = head[2 * p];
				if (HEAD8)
					hd2[st] = head[2 * p + 1];
			}
		}
	}
	// the rows' blocks and indices: independent of every other row, all in flight before any wait
	double bv[ST][KF], xv[ST][KF], d[ST], rv[ST];
	int xo[ST][KF];
	unsigned dep[ST];
#pragma unroll
	for (int st = 0; st < ST; st++) {
		const int row = mt[st].x, rp0 = mt[st].y, dg = mt[st].z, rp1 = mt[st].w;
		int jbeg = 0, jend = 0;
		if (okk[st]) {
			if (PART == PART_LOWER) {
				jbeg = rp0;
				jend = dg;
			} else if (PART == PART_UPPER) {
				jbeg = DIAG_RIDES ? dg : dg + 1;
				jend = rp1;
			} else {
				jbeg = rp0;
				jend = rp1;
			}
		}
		d[st] = 0.0;
		dep[st] = 0u;
#pragma unroll
		for (int k = 0; k < KF; k++) {
			const int jj = jbeg + slot + k * NB;
			bv[st][k] = 0.0;
			xv[st][k] = 0.0;
			xo[st][k] = 0;
			if (active && jj < jend && !(PART == PART_OFFDIAG && jj == dg)) {
				const double v = a.vals[(long)jj * BS2 + e];
				if (DIAG_RIDES && jj == dg) {
					d[st] = (DSRC == D_VALS_DIAG) ? v : 1.0 / v;
				} else {
					const int idx = jj - (PART == PART_LOWER ? rp0 : dg);  // item number inside the copy's row
					int col;
					if (head && idx < 4)
						col = idx == 0 ? hd[st].x : (idx == 1 ? hd[st].y : (idx == 2 ? hd[st].z : hd[st].w));
					else if (HEAD8 && head && idx < 8)
						col = idx == 4 ? hd2[st].x : (idx == 5 ? hd2[st].y : (idx == 6 ? hd2[st].z : hd2[st].w));
					else
						col = a.pat.bcolind[jj];
					bv[st][k] = v;
					xo[st][k] = col * BS + c;
					if (PART == PART_OFFDIAG && (desc ? col < row : col > row))
						xv[st][k] = a.xin[xo[st][k]];  // not a dependency of this direction: previous iterate
					else
						dep[st] |= 1u << k;
				}
			}
		}
		if (DSRC == D_DBLOCKS && okk[st] && active && slot == 0)
			d[st] = a.dvals[(long)row * BS2 + e];
		rv[st] = 0.0;
		if (okk[st] && r < BS && a.rhs) {
			rv[st] = a.rhs[(long)row * BS + r];
			if (a.rscale)
				rv[st] *= a.rscale[(long)row * BS + r];
		}
	}

	bool done[ST];
#pragma unroll
	for (int st = 0; st < ST; st++)
		done[st] = !okk[st];
	int spins = 0;
	for (;;) {
		bool alldone = true;
#pragma unroll
		for (int st = 0; st < ST; st++) {
			if (__builtin_amdgcn_ballot_w64(!done[st]) == 0ull)
				continue;  // wave-uniform: this step is finished
#pragma unroll
			for (int k = 0; k < KF; k++) {
				if (dep[st] & (1u << k)) {
					const double v = sf_load(a.xout + xo[st][k]);
					if (!sf_pending(v)) {
						xv[st][k] = v;
						dep[st] &= ~(1u << k);
					}
				}
			}
			const unsigned long long rb = __builtin_amdgcn_ballot_w64(dep[st] == 0u);
			const bool gready = ((rb >> gbase) & GMASK) == GMASK;
			if (__builtin_amdgcn_ballot_w64(gready && !done[st]) != 0ull) {  // wave-uniform: something to commit
				double acc = 0.0;
#pragma unroll
				for (int k = 0; k < KF; k++)
					acc += bv[st][k] * (gready ? xv[st][k] : 0.0);
				acc = allreduce_bits<Ge::LOBIT, Ge::HIBIT>(acc);
				double out;
				if (POST == POST_SUB) {
					out = rv[st] - acc;
				} else {
					const double w = (POST == POST_D_SUB) ? rv[st] - acc : acc;
					const double wc = __shfl(w, gbase + c, 64);
					const double pr =
					    allreduce_bits<Ge::LOBIT, Ge::HIBIT>((active && slot == 0) ? d[st] * wc : 0.0);
					out = (POST == POST_D_SUB) ? pr : rv[st] - pr;
				}
				if (!done[st] && gready && slot == 0 && c == 0 && r < BS)
					sf_store(a.xout + (long)mt[st].x * BS + r, out);
				done[st] = done[st] || gready;
			}
			if (!done[st])
				alldone = false;
		}
		if (__builtin_amdgcn_ballot_w64(!alldone) == 0ull)
			return;
		spins++;
		if (spins > SF_SPIN_LIMIT ||
		    ((spins & 255) == 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
			if (lane == 0)
				__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		__builtin_amdgcn_s_sleep(1);
	}
}

// Short row parts (up to 4 block passes): one row step per wave (8 waves per SIMD resident) instead of 4
// prefetched steps -- 1.7-1.8x faster exact solves on the Poisson pattern at every block size; 0 = the
// prefetching variant (tuning).
int g_sf_one_step = 1;

template <int BS, bool RM, int PART, int POST, int DSRC>
bool sf_launch_kf(const SweepArgs &a, const LevelSchedule &ls, int need, hipStream_t s, const int4 *meta,
                  const int4 *head)
{
	constexpr int NB = Geo<BS>::NB, RPW = Geo<BS>::RPW;
	const int passes = (need + NB - 1) / NB;
#define BHIP_KF(K, STEPS)                                                                                   \
	if (passes <= K) {                                                                                      \
		const long per_wg = 4L * STEPS * RPW;                                                               \
		const long grid = ((long)ls.count + per_wg - 1) / per_wg;                                           \
		if (grid > 0)                                                                                       \
			hipLaunchKernelGGL((sf_sweep_kernel<BS, RM, PART, POST, DSRC, K, STEPS>), dim3((unsigned)grid), \
			                   dim3(256), 0, s, a, meta, head, ls.count, ls.ctl);                           \
		return true;                                                                                        \
	}
	if (g_sf_one_step) {
		BHIP_KF(4, 1)
	} else {
		BHIP_KF(4, 4)
	}
	BHIP_KF(8, 2)  // relaxation passes at 256^3: 15.4 ms with two steps per wave, 18.2 ms with one
	BHIP_KF(16, 1)
#undef BHIP_KF
	return false;
}

template <int BS, bool RM>
bool sf_dispatch_ops(const SweepArgs &a, Part part, Post post, DSrc dsrc, const LevelSchedule &ls, hipStream_t s,
                     const int4 *meta, const int4 *head)
{
	const int lo = ls.max_lower, up = ls.max_upper;
#define BHIP_CASE(P, Q, D, NEED)                                       \
	if (part == P && post == Q && dsrc == D)                           \
		return sf_launch_kf<BS, RM, P, Q, D>(a, ls, NEED, s, meta, head);
	BHIP_CASE(PART_LOWER, POST_SUB, D_NONE, lo)
	BHIP_CASE(PART_UPPER, POST_D_SUB, D_VALS_DIAG, up + 1)
	BHIP_CASE(PART_UPPER, POST_D_SUB, D_RECIP_DIAG, up + 1)
	BHIP_CASE(PART_LOWER, POST_D_SUB, D_DBLOCKS, lo)
	BHIP_CASE(PART_UPPER, POST_SUB_D, D_DBLOCKS, up)
	BHIP_CASE(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS, lo + up + 1)
#undef BHIP_CASE
	BHIP_FAIL(BLASTED_HIP_EINVAL, "launch_syncfree_sweep: operator combination not instantiated");
}

template <int BS>
bool sf_dispatch_layout(const SweepArgs &a, Part part, Post post, DSrc dsrc, const LevelSchedule &ls,
                        hipStream_t s, const int4 *meta, const int4 *head)
{
	if (BS > 1 && a.pat.rowmajor)
		return sf_dispatch_ops<BS, true>(a, part, post, dsrc, ls, s, meta, head);
	return sf_dispatch_ops<BS, false>(a, part, post, dsrc, ls, s, meta, head);
}

// level-ordered storage: per-position block counts of the two triangles
__global__ __launch_bounds__(256) void level_counts_kernel(const Pattern pat, const int *rows, int *cl, int *cu)
{
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= pat.nbrows)
		return;
	const int row = rows[k];
	const int dg = pat.diagind[row];
	cl[k] = dg - pat.browptr[row];
	cu[k] = pat.browptr[row + 1] - dg;
}

__global__ __launch_bounds__(256) void level_posof_kernel(const int *rows, int n, int *posof)
{
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k < n)
		posof[rows[k]] = k;
}

// x_nat[row] = x_perm[position of row], one double per thread
__global__ __launch_bounds__(256) void level_unpermute_kernel(const int *rows, long n, int bs, const double *xperm,
                                                              double *xnat)
{
	const long i = (long)blockIdx.x * 256 + threadIdx.x;
	if (i < n * bs) {
		const long p = i / bs;
		xnat[(long)rows[p] * bs + (i - p * bs)] = xperm[i];
	}
}

// column indices in level order + the descriptors of the two copies
__global__ __launch_bounds__(256) void level_cols_kernel(const Pattern pat, const int *rows, const int *lptr,
                                                         const int *uptr, int *lcol, int *ucol, int4 *lmeta,
                                                         int4 *umeta, int4 *lhead, int4 *uhead, const int *posof,
                                                         int *lcolp, int *ucolp, int4 *lheadp, int4 *uheadp)
{
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= pat.nbrows)
		return;
	const int row = rows[k];
	const int rp0 = pat.browptr[row], rp1 = pat.browptr[row + 1], dg = pat.diagind[row];
	const int l0 = lptr[k], u0 = uptr[k];
	for (int jj = rp0; jj < dg; jj++) {
		lcol[l0 + (jj - rp0)] = pat.bcolind[jj];
		lcolp[l0 + (jj - rp0)] = posof[pat.bcolind[jj]];
	}
	for (int jj = dg; jj < rp1; jj++) {
		ucol[u0 + (jj - dg)] = pat.bcolind[jj];
		ucolp[u0 + (jj - dg)] = posof[pat.bcolind[jj]];
	}
	lmeta[k] = make_int4(row, l0, l0 + (dg - rp0), 0);
	umeta[k] = make_int4(row, 0, u0, u0 + (rp1 - dg));
	// the first eight column indices of each copy, addressable from the position alone (two int4 each)
	int hl[8], hu[8];
	for (int q = 0; q < 8; q++) {
		hl[q] = (rp0 + q < dg) ? pat.bcolind[rp0 + q] : -1;
		hu[q] = (dg + q < rp1) ? pat.bcolind[dg + q] : -1;
	}
	lhead[2 * k] = make_int4(hl[0], hl[1], hl[2], hl[3]);
	lhead[2 * k + 1] = make_int4(hl[4], hl[5], hl[6], hl[7]);
	uhead[2 * k] = make_int4(hu[0], hu[1], hu[2], hu[3]);
	uhead[2 * k + 1] = make_int4(hu[4], hu[5], hu[6], hu[7]);
	for (int q = 0; q < 8; q++) {
		hl[q] = hl[q] >= 0 ? posof[hl[q]] : -1;
		hu[q] = hu[q] >= 0 ? posof[hu[q]] : -1;
	}
	lheadp[2 * k] = make_int4(hl[0], hl[1], hl[2], hl[3]);
	lheadp[2 * k + 1] = make_int4(hl[4], hl[5], hl[6], hl[7]);
	uheadp[2 * k] = make_int4(hu[0], hu[1], hu[2], hu[3]);
	uheadp[2 * k + 1] = make_int4(hu[4], hu[5], hu[6], hu[7]);
}

// values of the two triangles into level order: 16 lanes move one row, 8 bytes per lane and step
template <int BS2>
__global__ __launch_bounds__(256) void level_permute_kernel(const Pattern pat, const int *rows, const int *lptr,
                                                            const int *uptr, const double *__restrict__ vals,
                                                            double *__restrict__ lvals, double *__restrict__ uvals)
{
	const int k = blockIdx.x * 16 + (threadIdx.x >> 4);
	const int t = threadIdx.x & 15;
	if (k >= pat.nbrows)
		return;
	const int row = rows[k];
	const long rp0 = pat.browptr[row], rp1 = pat.browptr[row + 1], dg = pat.diagind[row];
	const long nl = (dg - rp0) * BS2, nu = (rp1 - dg) * BS2;
	const double *const srcl = vals + rp0 * BS2;
	const double *const srcu = vals + dg * BS2;
	if (lvals) {  // (either triangle may be left out)
		double *const dstl = lvals + (long)lptr[k] * BS2;
		for (long q = t; q < nl; q += 16)
			dstl[q] = srcl[q];
	}
	if (uvals) {
		double *const dstu = uvals + (long)uptr[k] * BS2;
		for (long q = t; q < nu; q += 16)
			dstu[q] = srcu[q];
	}
}


template <typename T>
T *lvl_alloc(size_t count)
{
	void *q = nullptr;
	BHIP_CHECK(tracked_malloc(&q, sizeof(T) * (count ? count : 1)));
	return static_cast<T *>(q);
}

}  // namespace

// tuning ("levelfast=0|1"): the level-schedule build starts with the dependency-polling launch (default) or not
int g_level_fast = 1;
void set_level_fast(int on)
{
	g_level_fast = on;
}

// tuning ("levelserial=N"): parallel relaxation passes of the level-schedule build before the in-order fall-back
long g_level_serial_after = 4096;
void set_level_serial_after(long n)
{
	g_level_serial_after = n > 0 ? n : 1;
}

void set_syncfree_one_step(int on)
{
	g_sf_one_step = on;
}

void free_level_schedule(LevelSchedule &ls)
{
	if (ls.rows)
		(void)tracked_free(ls.rows);
	if (ls.level)
		(void)tracked_free(ls.level);
	if (ls.meta)
		(void)tracked_free(ls.meta);
	if (ls.ctl)
		(void)tracked_free(ls.ctl);
	if (ls.f4_desc)
		(void)tracked_free(ls.f4_desc);
	for (void *q : {(void *)ls.lptr, (void *)ls.uptr, (void *)ls.lcol, (void *)ls.ucol, (void *)ls.lmeta,
	                (void *)ls.umeta, (void *)ls.lhead, (void *)ls.uhead, (void *)ls.posof, (void *)ls.lcolp,
	                (void *)ls.ucolp, (void *)ls.lheadp, (void *)ls.uheadp})
		if (q)
			(void)tracked_free(q);
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
		constexpr int BATCH = 8;
		flags = lvl_alloc<int>(BATCH);
		int hflags[BATCH];
		bool fixed = false;
		long passes = 0;
		bool have_start = false;
		if (g_level_fast) {
			// one dependency-polling launch over the stored lower entries, then one relaxation pass: on a structurally
			// symmetric pattern it changes nothing and the levels are final; otherwise (dependencies implied by upper
			// entries only) the passes below continue from here -- every level found so far is a lower bound
			BHIP_CHECK(hipMemsetAsync(level, 0xff, sizeof(int) * (size_t)n, s));
			BHIP_CHECK(hipMemsetAsync(flags, 0, sizeof(int) * BATCH, s));
			hipLaunchKernelGGL(level_poll_kernel, dim3(grid), dim3(256), 0, s, pat, level, flags);
			hipLaunchKernelGGL(level_relax_kernel, dim3(grid), dim3(256), 0, s, pat, level, flags + 1);
			BHIP_CHECK(hipMemcpyAsync(hflags, flags, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
			BHIP_CHECK(hipStreamSynchronize(s));
			if (hflags[0] == 0 && g_level_fast != 2) {  // (nobody gave up waiting)
				have_start = true;
				passes = 2;
				fixed = hflags[1] == 0;
			}
		}
		if (!have_start)
			BHIP_CHECK(hipMemsetAsync(level, 0, sizeof(int) * (size_t)n, s));
		// beyond this many passes (= dependency levels) the parallel relaxation loses to one in-order pass
		const long serial_after = g_level_serial_after;
		for (long done = 0; done < (long)n + 2 && !fixed && passes < serial_after; done += BATCH) {
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
		if (!fixed) {
			// a deep dependency graph: finish with the in-order pass (exact after one sweep over the rows)
			hipLaunchKernelGGL(level_serial_kernel, dim3(1), dim3(64), 0, s, pat, level);
			BHIP_CHECK(hipGetLastError());
			BHIP_CHECK(hipStreamSynchronize(s));
			passes = -passes - 1;  // reported negative: "settled by the serial pass after that many parallel ones"
		}
		ls.build_passes = passes;

		keys = lvl_alloc<int>(n);
		iota = lvl_alloc<int>(n);
		rows = lvl_alloc<int>(n);
		hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, s, iota, n);
		size_t bytes = 0;
		BHIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, level, keys, iota, rows, n, 0, 31, s));
		BHIP_CHECK(tracked_malloc(&tmp, bytes ? bytes : 1));
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
		// per-position row descriptors and the longest row parts (single-launch passes)
		ls.meta = lvl_alloc<int4>(n);
		ls.ctl = lvl_alloc<int>(4);
		BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 4 * sizeof(int), s));
		hipLaunchKernelGGL(level_meta_kernel, dim3(grid), dim3(256), 0, s, pat, rows, ls.meta, ls.ctl + 2);
		int lens[2] = {0, 0};
		BHIP_CHECK(hipMemcpyAsync(lens, ls.ctl + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
		BHIP_CHECK(hipStreamSynchronize(s));
		ls.max_lower = lens[0];
		ls.max_upper = lens[1];
		int dev = 0, cus = 256;
		BHIP_CHECK(hipGetDevice(&dev));
		BHIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
		ls.sf_grid = cus * 4;
		ls.count = n;
		ls.nlevels = nlevels;
		ls.rows = rows;
		ls.level = level;
		rows = nullptr;
		level = nullptr;
		ls.built = true;
	} catch (...) {
		if (ls.meta)
			(void)tracked_free(ls.meta);
		if (ls.ctl)
			(void)tracked_free(ls.ctl);
		ls.meta = nullptr;
		ls.ctl = nullptr;
		for (void *q : {(void *)level, (void *)keys, (void *)iota, (void *)rows, (void *)flags, (void *)dptr, tmp})
			if (q)
				(void)tracked_free(q);
		throw;
	}
	for (void *q : {(void *)keys, (void *)iota, (void *)flags, (void *)dptr, tmp})
		if (q)
			(void)tracked_free(q);
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

// Fills x (the iterate a single-launch pass will produce) with the "pending" pattern.
void launch_syncfree_fill(double *x, long n, hipStream_t s)
{
	if (n <= 0)
		return;
	hipLaunchKernelGGL(sf_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
	                   reinterpret_cast<unsigned long long *>(x), n, BHIP_PROBE(g_sf_nowait) ? 0ull : SF_PENDING);
}

void set_syncfree_nowait(int on)
{
	g_sf_nowait = on != 0;
}

// One exact in-order pass as ONE persistent launch.  a.xout must have been filled by
// launch_syncfree_fill; a.xin (relaxation: the previous iterate) must be a different, complete vector.
// Returns false when a row part is too long for the register-held passes (caller uses
// launch_level_sweep).  The abort flag ls.ctl[1] must be checked by the caller after the stream drains.
bool launch_syncfree_sweep(const SweepArgs &a_, Part part, Post post, DSrc dsrc, const LevelSchedule &ls,
                           hipStream_t s, const LevelView *view)
{
	SweepArgs a = a_;
	const int4 *meta = ls.meta, *head = nullptr;
	if (view) {  // level-ordered copies: contiguous stream
		meta = view->meta;
		head = view->head;
		a.vals = view->vals;
		a.pat.bcolind = view->bcolind;
		const bool ilu_lower = part == PART_LOWER && post == POST_SUB && dsrc == D_NONE;
		const bool ilu_upper = part == PART_UPPER && post == POST_D_SUB && dsrc == D_VALS_DIAG;
		const bool sgs_fwd = part == PART_LOWER && post == POST_D_SUB && dsrc == D_DBLOCKS;
		const bool sgs_bwd = part == PART_UPPER && post == POST_SUB_D && dsrc == D_DBLOCKS;
		if ((ilu_lower || ilu_upper || sgs_fwd || sgs_bwd) &&
		    launch_syncfree_wide(a, ilu_upper || sgs_bwd, ls, view->ptr, view->bcolind, view->head, s, false,
		                         sgs_fwd || sgs_bwd))
			return true;
	}
	if (!ls.built)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "launch_syncfree_sweep: no level schedule");
	if (part == PART_OFFDIAG && a.xin == a.xout)
		BHIP_FAIL(BLASTED_HIP_EINVAL, "launch_syncfree_sweep: relaxation passes need a second vector");
	BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
	bool ok = false;
	switch (a.pat.bs) {
	case 1: ok = sf_dispatch_layout<1>(a, part, post, dsrc, ls, s, meta, head); break;
	case 2: ok = sf_dispatch_layout<2>(a, part, post, dsrc, ls, s, meta, head); break;
	case 3: ok = sf_dispatch_layout<3>(a, part, post, dsrc, ls, s, meta, head); break;
	case 4: ok = sf_dispatch_layout<4>(a, part, post, dsrc, ls, s, meta, head); break;
	case 5: ok = sf_dispatch_layout<5>(a, part, post, dsrc, ls, s, meta, head); break;
	case 7: ok = sf_dispatch_layout<7>(a, part, post, dsrc, ls, s, meta, head); break;
	case 8: ok = sf_dispatch_layout<8>(a, part, post, dsrc, ls, s, meta, head); break;
	default: BHIP_FAIL(BLASTED_HIP_ENOTIMPL, "block size not instantiated (1,2,3,4,5,7,8)");
	}
	BHIP_CHECK(hipGetLastError());
	return ok;
}

// Pattern part of the level-ordered storage (once per pattern).
void build_level_storage(const Pattern &pat, LevelSchedule &ls, hipStream_t s)
{
	if (!ls.built)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "build_level_storage: no level schedule");
	if (ls.storage_built)
		return;
	const int n = pat.nbrows;
	if (n == 0) {
		ls.storage_built = true;
		return;
	}
	const unsigned grid = (unsigned)((n + 255) / 256);
	int *cl = lvl_alloc<int>((size_t)n + 1), *cu = lvl_alloc<int>((size_t)n + 1);
	void *tmp = nullptr;
	try {
		BHIP_CHECK(hipMemsetAsync(cl + n, 0, sizeof(int), s));
		BHIP_CHECK(hipMemsetAsync(cu + n, 0, sizeof(int), s));
		hipLaunchKernelGGL(level_counts_kernel, dim3(grid), dim3(256), 0, s, pat, ls.rows, cl, cu);
		ls.lptr = lvl_alloc<int>((size_t)n + 1);
		ls.uptr = lvl_alloc<int>((size_t)n + 1);
		size_t bytes = 0;
		BHIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cl, ls.lptr, n + 1, s));
		BHIP_CHECK(tracked_malloc(&tmp, bytes ? bytes : 1));
		BHIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, cl, ls.lptr, n + 1, s));
		BHIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, cu, ls.uptr, n + 1, s));
		int tot[2] = {0, 0};
		BHIP_CHECK(hipMemcpyAsync(&tot[0], ls.lptr + n, sizeof(int), hipMemcpyDeviceToHost, s));
		BHIP_CHECK(hipMemcpyAsync(&tot[1], ls.uptr + n, sizeof(int), hipMemcpyDeviceToHost, s));
		BHIP_CHECK(hipStreamSynchronize(s));
		ls.nnz_lower = tot[0];
		ls.nnz_dupper = tot[1];
		if (ls.nnz_lower + ls.nnz_dupper != pat.nnzb)
			BHIP_FAIL(BLASTED_HIP_ERUNTIME, "level storage: triangle sizes do not add up (internal error)");
		ls.lcol = lvl_alloc<int>((size_t)ls.nnz_lower);
		ls.ucol = lvl_alloc<int>((size_t)ls.nnz_dupper);
		ls.lmeta = lvl_alloc<int4>(n);
		ls.umeta = lvl_alloc<int4>(n);
		ls.lhead = lvl_alloc<int4>(2 * (size_t)n);
		ls.uhead = lvl_alloc<int4>(2 * (size_t)n);
		ls.posof = lvl_alloc<int>(n);
		ls.lcolp = lvl_alloc<int>((size_t)ls.nnz_lower);
		ls.ucolp = lvl_alloc<int>((size_t)ls.nnz_dupper);
		ls.lheadp = lvl_alloc<int4>(2 * (size_t)n);
		ls.uheadp = lvl_alloc<int4>(2 * (size_t)n);
		hipLaunchKernelGGL(level_posof_kernel, dim3(grid), dim3(256), 0, s, ls.rows, n, ls.posof);
		hipLaunchKernelGGL(level_cols_kernel, dim3(grid), dim3(256), 0, s, pat, ls.rows, ls.lptr, ls.uptr, ls.lcol,
		                   ls.ucol, ls.lmeta, ls.umeta, ls.lhead, ls.uhead, ls.posof, ls.lcolp, ls.ucolp, ls.lheadp,
		                   ls.uheadp);
		BHIP_CHECK(hipGetLastError());
		BHIP_CHECK(hipStreamSynchronize(s));
		ls.storage_built = true;
	} catch (...) {
		for (void *q : {(void *)cl, (void *)cu, tmp})
			if (q)
				(void)tracked_free(q);
		throw;
	}
	for (void *q : {(void *)cl, (void *)cu, tmp})
		if (q)
			(void)tracked_free(q);
}

void launch_level_unpermute(const LevelSchedule &ls, int bs, const double *xperm, double *xnat, hipStream_t s)
{
	const long n = (long)ls.count * bs;
	if (n == 0)
		return;
	hipLaunchKernelGGL(level_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ls.rows,
	                   (long)ls.count, bs, xperm, xnat);
	BHIP_CHECK(hipGetLastError());
}

// The same compact two-triangle storage in NATURAL row order (rows = identity): the asynchronous sweeps
// then stream exactly the blocks and indices of their triangle.
void build_natural_storage(const Pattern &pat, LevelSchedule &ns, hipStream_t s)
{
	if (ns.storage_built)
		return;
	const int n = pat.nbrows;
	ns.rows = lvl_alloc<int>((size_t)(n ? n : 1));
	if (n)
		hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ns.rows, n);
	ns.count = n;
	ns.built = true;
	build_level_storage(pat, ns, s);
}

// Values of both triangles into level order (after every factorisation that is applied exactly).
void launch_level_permute_values(const Pattern &pat, const LevelSchedule &ls, const double *vals, double *lvals,
                                 double *uvals, hipStream_t s)
{
	if (!ls.storage_built)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "launch_level_permute_values: no level storage");
	if (pat.nbrows == 0)
		return;
	const unsigned grid = (unsigned)(((long)pat.nbrows + 15) / 16);
#define BHIP_PERM(B)                                                                                       \
	case B:                                                                                                \
		hipLaunchKernelGGL((level_permute_kernel<B * B>), dim3(grid), dim3(256), 0, s, pat, ls.rows, ls.lptr, \
		                   ls.uptr, vals, lvals, uvals);                                                   \
		break;
	switch (pat.bs) {
		BHIP_PERM(1)
		BHIP_PERM(2)
		BHIP_PERM(3)
		BHIP_PERM(4)
		BHIP_PERM(5)
		BHIP_PERM(7)
		BHIP_PERM(8)
	default: BHIP_FAIL(BLASTED_HIP_ENOTIMPL, "block size not instantiated (1,2,3,4,5,7,8)");
	}
#undef BHIP_PERM
	BHIP_CHECK(hipGetLastError());
}

}  // namespace bhip
