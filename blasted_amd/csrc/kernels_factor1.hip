// kernels_factor1.hip -- asynchronous scalar ILU(0) factorisation sweep (bs = 1: CSR; the reference's
// async_ilu0_factorize_kernel, src/kernels/kernels_ilu0_factorize.hpp:19-53, driven by
// src/async_ilu_factor.cpp:36-98), BASELINE.json's config 1.  Same fixed-point map as the general kernel
// (factor_sweep_kernel<1> in kernels_factor.hip), whose one lane per row walks its row through ~10 dependent,
// uncoalesced 4- and 8-byte loads per entry (1.2 TB/s algorithmic at 128^3).  Here a workgroup owns a chunk of
// F1_RCHUNK consecutive rows and everything it needs arrives in bulk:
//   phase 1  coalesced, all in flight together: row pointers, column indices, position-list pointers, the
//            (lower, upper) position pairs, the matrix values and the current factor values of the chunk -> LDS;
//   phase 2  the operands that live in OTHER rows, gathered with the staged indices: the upper factor of every
//            pair (u_kj) and, for lower entries, the diagonal u_jj of their column -> LDS;
//   phase 3  one lane per row runs the row's recurrence out of LDS alone, in storage order, reading back the
//            lower entries it has just produced (the reference's in-row Gauss-Seidel order);
//   phase 4  the chunk's new values leave as one coalesced store.
// In place (in == out, the asynchronous product mode) phase 3 updates the staged row as it goes; with
// separate buffers (synchronous test mode) every read comes from the previous iterate and the results are
// stored straight to `out`.  128^3 / 200^3 Poisson: 0.459 / 1.775 ms per sweep with the general kernel, 0.240 /
// 0.904 ms with the first form of this one.  Entries or pairs beyond the staging capacity (rows much longer than a stencil's)
// are read from global memory in phase 3.  Row lists (the level-scheduled exact factorisation) and the
// remainder evaluation stay with the general kernel.
#include "ctx.hpp"
#include "lanes.hpp"

#include <cstdlib>
#include <cstring>

namespace bhip {

namespace {

constexpr int F1_RCHUNK = 128;            // rows per workgroup
constexpr int F1_CAPE = 8 * F1_RCHUNK;    // staged entries
constexpr int F1_CAPP = 8 * F1_RCHUNK;    // staged (lower, upper) pairs

// INPLACE (in == out, the asynchronous product mode): a row's recurrence reads only entries of its own row it has
// produced in this sweep, so the chunk's current factor values are never needed -- they are not loaded (8 of the
// ~44 bytes an entry costs) and the staged matrix value is overwritten by the result (one LDS array less: four
// workgroups per CU instead of three).
template <bool INPLACE>
__global__ __launch_bounds__(256) void factor1_kernel(const FactorArgs a)
{
	__shared__ int s_rp[F1_RCHUNK + 1];
	__shared__ int s_col[F1_CAPE];
	__shared__ int s_pp[F1_CAPE + 1];
	__shared__ int s_lp[F1_CAPP];
	__shared__ int s_up[F1_CAPP];
	__shared__ double s_a[INPLACE ? 1 : F1_CAPE];  // (scaled) matrix values [separate buffers only]
	__shared__ double s_f[F1_CAPE];   // INPLACE: matrix value, then the new factor value; else the iterate read
	__shared__ double s_dv[F1_CAPE];  // lower entries: u_jj of their column
	__shared__ double s_uv[F1_CAPP];  // pairs: the upper factor u_kj

	const int tid = threadIdx.x;
	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const int r0 = (int)chunk * F1_RCHUNK;
	const int rc = (nb - r0) < F1_RCHUNK ? (nb - r0) : F1_RCHUNK;
	constexpr bool inplace = INPLACE;
	double *const s_av = INPLACE ? s_f : s_a;  // where the (scaled) matrix values are staged

	// the three index ranges depend on each other (rows -> entries -> pairs): their ends come from wave-uniform
	// scalar loads, so that all vector loads of phase 1 go out together
	const int jlo = __builtin_amdgcn_readfirstlane(a.pat.browptr[r0]);
	const int jhi = __builtin_amdgcn_readfirstlane(a.pat.browptr[r0 + rc]);
	const int nent = (jhi - jlo) < F1_CAPE ? (jhi - jlo) : F1_CAPE;
	const int plo = __builtin_amdgcn_readfirstlane(a.posptr[jlo]);
	const int phi = __builtin_amdgcn_readfirstlane(a.posptr[jlo + nent]);
	const int npair = (phi - plo) < F1_CAPP ? (phi - plo) : F1_CAPP;

	// ---- phase 1
	constexpr int NE = F1_CAPE / 256, NP = F1_CAPP / 256;
	{
		int vcol[NE], vpp[NE], vlp[NP], vup[NP];
		double va[NE], vf[NE];
		const int vrp = (tid <= rc) ? a.pat.browptr[r0 + tid] : 0;
#pragma unroll
		for (int i = 0; i < NE; i++) {
			const int q = tid + 256 * i;
			vcol[i] = (q < nent) ? a.pat.bcolind[jlo + q] : 0;
			vpp[i] = (q < nent) ? a.posptr[jlo + q] : 0;
			va[i] = (q < nent) ? a.avals[(long)jlo + q] : 0.0;
			vf[i] = (!INPLACE && q < nent) ? a.in[(long)jlo + q] : 0.0;
		}
#pragma unroll
		for (int i = 0; i < NP; i++) {
			const int q = tid + 256 * i;
			vlp[i] = (q < npair) ? a.lowerp[plo + q] : 0;
			vup[i] = (q < npair) ? a.upperp[plo + q] : 0;
		}
		if (tid <= rc)
			s_rp[tid] = vrp;
#pragma unroll
		for (int i = 0; i < NE; i++) {
			const int q = tid + 256 * i;
			if (q < nent) {
				s_col[q] = vcol[i];
				s_pp[q] = vpp[i];
				s_av[q] = va[i];
				if (!INPLACE)
					s_f[q] = vf[i];
			}
		}
		if (tid == 0)
			s_pp[nent] = phi;
#pragma unroll
		for (int i = 0; i < NP; i++) {
			const int q = tid + 256 * i;
			if (q < npair) {
				s_lp[q] = vlp[i];
				s_up[q] = vup[i];
			}
		}
	}
	__syncthreads();

	// ---- phase 2: operands from other rows, and the symmetric scaling of A
#pragma unroll
	for (int i = 0; i < NP; i++) {
		const int q = tid + 256 * i;
		if (q < npair)
			s_uv[q] = a.in[s_up[q]];
	}
#pragma unroll
	for (int i = 0; i < NE; i++) {
		const int q = tid + 256 * i;
		if (q < nent) {
			// the row of entry q: last row pointer <= jlo + q
			int lo = 0, hi = rc;
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (s_rp[mid] <= jlo + q)
					lo = mid;
				else
					hi = mid;
			}
			const int row = r0 + lo, col = s_col[q];
			if (col < row)
				s_dv[q] = a.in[a.pat.diagind[col]];
			if (a.scale) {  // (a s_i) s_j, two roundings, as src/kernels/kernels_ilu0_factorize.hpp:29-32
				s_av[q] *= a.scale[row];
				s_av[q] *= a.scale[col];
			}
		}
	}
	__syncthreads();

	// ---- phase 3: one lane per row
	if (tid < rc) {
		const int irow = r0 + tid;
		const int jbeg = s_rp[tid], jend = s_rp[tid + 1];
		for (int jpos = jbeg; jpos < jend; jpos++) {
			const int le = jpos - jlo;
			const bool staged = le < nent;
			const int col = staged ? s_col[le] : a.pat.bcolind[jpos];
			double s;
			if (staged)
				s = s_av[le];
			else {
				s = a.avals[jpos];
				if (a.scale) {
					s *= a.scale[irow];
					s *= a.scale[col];
				}
			}
			const int kb = staged ? s_pp[le] : a.posptr[jpos];
			const int ke = staged ? s_pp[le + 1] : a.posptr[jpos + 1];
			for (int k = kb; k < ke; k++) {
				const int lk = k - plo;
				const bool pst = lk < npair;
				const int lp = pst ? s_lp[lk] : a.lowerp[k];
				// l_ik sits in this row, before jpos: staged (and fresh when in place) or, beyond the capacity, in memory
				const int ll = lp - jlo;
				const double lv = (ll >= 0 && ll < nent) ? s_f[ll] : a.in[lp];
				const double uv = pst ? s_uv[lk] : a.in[a.upperp[k]];
				s -= lv * uv;
			}
			if (irow > col)
				s /= staged ? s_dv[le] : a.in[a.pat.diagind[col]];
			if (inplace && staged)
				s_f[le] = s;
			else
				a.out[jpos] = s;
		}
	}
	if (!inplace)
		return;
	__syncthreads();

	// ---- phase 4
#pragma unroll
	for (int i = 0; i < NE; i++) {
		const int q = tid + 256 * i;
		if (q < nent)
			a.out[(long)jlo + q] = s_f[q];
	}
}

// ---- round 3: the in-place sweep without scaling, on a precomputed plan ---------------------------------------------
// What bounded factor1_kernel (0.29 of the HBM peak) is the dependent chain of a workgroup at four workgroups per
// CU: two scalar range look-ups, the bulk loads, a barrier, a two-level gather (diagind[col], then u_jj), a second
// barrier, a serial phase in which a lane divides three times.  Here
//  * a per-pattern plan (f1_plan_kernel, once per pattern: 4 bytes per entry + 16 per chunk) holds, for every
//    entry, the position of its column's diagonal entry (-1 for diagonal / upper entries) and, for every chunk, its
//    entry and pair ranges -- so the ranges cost ONE scalar load, the column indices are not read at all, and
//    u_jj is a one-level gather;
//  * the thread that loaded an index issues the gather itself (no LDS round trip, no barrier in between);
//  * a lower entry WITHOUT position pairs is l_ij = a_ij / u_jj outright: the loading thread divides, 256 lanes in
//    parallel (the same operation on the same operands as in the serial phase: same bits);
//  * LDS holds 16-bit chunk-relative positions: 28.5 KB per workgroup, five workgroups per CU;
//  * the serial phase only visits entries that have pairs (a 7-point row: its diagonal entry).
// Synchronous sweeps (separate buffers) and scaled factorisations keep factor1_kernel.  The kernel never reads the chunk's
// own old factor values (a row reads what it has produced) and takes everything else from `in`, so it also runs the
// fused first sweep of a build, in = the matrix, out = the factor (FactorArgs::lrow_fresh).
constexpr int F1P_CAPE = 1024, F1P_CAPP = 1024;
static_assert(F1P_CAPE == F1_CAPE && F1P_CAPP == F1_CAPP, "the chunk descriptors are built for these capacities");

__global__ __launch_bounds__(256) void f1_plan_kernel(const Pattern pat, const int *__restrict__ posptr,
                                                      int *__restrict__ dcol, int4 *__restrict__ chunks)
{
	const int row = blockIdx.x * 256 + threadIdx.x;
	if (row >= pat.nbrows)
		return;
	const int jb = pat.browptr[row], je = pat.browptr[row + 1];
	for (int j = jb; j < je; j++) {
		const int col = pat.bcolind[j];
		dcol[j] = col < row ? pat.diagind[col] : -1;
	}
	if (row % F1_RCHUNK == 0) {
		const int rc = (pat.nbrows - row) < F1_RCHUNK ? (pat.nbrows - row) : F1_RCHUNK;
		const int jhi = pat.browptr[row + rc];
		const int nent = (jhi - jb) < F1P_CAPE ? (jhi - jb) : F1P_CAPE;
		chunks[row / F1_RCHUNK] = make_int4(jb, jhi, posptr[jb], posptr[jb + nent]);
	}
}

__global__ __launch_bounds__(256) void factor1p_kernel(const FactorArgs a)
{
	__shared__ unsigned short s_pp[F1P_CAPE + 2];  // first pair of an entry, relative to the chunk's first pair
	__shared__ unsigned short s_lp[F1P_CAPP];      // a pair's lower entry, relative to the chunk's first entry
	__shared__ double s_f[F1P_CAPE];               // matrix value, then the new factor value
	__shared__ double s_dv[F1P_CAPE];              // lower entries: u_jj of their column
	__shared__ double s_uv[F1P_CAPP];              // pairs: the upper factor u_kj
	constexpr unsigned short FAR = 0xFFFF;

	const int tid = threadIdx.x;
	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const int r0 = (int)chunk * F1_RCHUNK;
	const int rc = (nb - r0) < F1_RCHUNK ? (nb - r0) : F1_RCHUNK;
	const int4 cd = a.f1_chunks[chunk];
	const int jlo = __builtin_amdgcn_readfirstlane(cd.x), jhi = __builtin_amdgcn_readfirstlane(cd.y);
	const int plo = __builtin_amdgcn_readfirstlane(cd.z), phi = __builtin_amdgcn_readfirstlane(cd.w);
	const int nent = (jhi - jlo) < F1P_CAPE ? (jhi - jlo) : F1P_CAPE;
	const int npair = (phi - plo) < F1P_CAPP ? (phi - plo) : F1P_CAPP;

	// ---- loads: everything of the chunk, coalesced, in flight together; then the gathers, from registers
	constexpr int NE = F1P_CAPE / 256, NP = F1P_CAPP / 256;
	int rbeg = 0, rend = 0, rdg = 0;  // this thread's row of the serial phase
	if (tid < rc) {
		rbeg = a.pat.browptr[r0 + tid];
		rend = a.pat.browptr[r0 + tid + 1];
		rdg = a.pat.diagind[r0 + tid];
	}
	int vpp0[NE], vpp1[NE], vd[NE], vlp[NP], vup[NP];
	double va[NE];
#pragma unroll
	for (int i = 0; i < NE; i++) {
		const int q = tid + 256 * i;
		const bool p = q < nent;
		vpp0[i] = p ? a.posptr[jlo + q] : 0;
		vpp1[i] = p ? a.posptr[jlo + q + 1] : 0;
		vd[i] = p ? a.f1_dcol[jlo + q] : -1;
		va[i] = p ? a.avals[(long)jlo + q] : 0.0;
	}
#pragma unroll
	for (int i = 0; i < NP; i++) {
		const int q = tid + 256 * i;
		vlp[i] = (q < npair) ? a.lowerp[plo + q] : 0;
		vup[i] = (q < npair) ? a.upperp[plo + q] : 0;
	}
	double uv[NP], dv[NE];
#pragma unroll
	for (int i = 0; i < NP; i++)
		uv[i] = (tid + 256 * i < npair) ? a.in[vup[i]] : 0.0;
#pragma unroll
	for (int i = 0; i < NE; i++)
		dv[i] = vd[i] >= 0 ? a.in[vd[i]] : 1.0;
#pragma unroll
	for (int i = 0; i < NE; i++) {
		const int q = tid + 256 * i;
		if (q < nent) {
			const int rel = vpp0[i] - plo;
			s_pp[q] = rel < FAR ? (unsigned short)rel : FAR;
			if (q == nent - 1) {
				const int rel1 = vpp1[i] - plo;
				s_pp[nent] = rel1 < FAR ? (unsigned short)rel1 : FAR;
			}
			// a lower entry without pairs is final right here
			s_f[q] = (vd[i] >= 0 && vpp1[i] == vpp0[i]) ? va[i] / dv[i] : va[i];
			s_dv[q] = dv[i];
		}
	}
#pragma unroll
	for (int i = 0; i < NP; i++) {
		const int q = tid + 256 * i;
		if (q < npair) {
			const int rel = vlp[i] - jlo;
			s_lp[q] = (rel >= 0 && rel < nent) ? (unsigned short)rel : FAR;
			s_uv[q] = uv[i];
		}
	}
	__syncthreads();

	// ---- one lane per row: the entries with pairs, in storage order (the reference's in-row order)
	// (a lower entry of the row itself that is not staged is read back from what this sweep has stored: `out`, which is
	// `in` for an in-place sweep and the factor for the fused first sweep of a build, whose `in` is the matrix)
	const double *const lsrc = a.out;
	if (tid < rc) {
		for (int jpos = rbeg; jpos < rend; jpos++) {
			const int le = jpos - jlo;
			if (le < nent) {
				const int kb = s_pp[le], ke = s_pp[le + 1];
				if (ke <= npair) {
					if (kb == ke)
						continue;  // no pairs: the matrix value (diagonal / upper) or a / u_jj (lower), already there
					double s = s_f[le];
					for (int k = kb; k < ke; k++) {
						const unsigned short rel = s_lp[k];
						const double lv = rel != FAR ? s_f[rel] : lsrc[a.lowerp[plo + k]];
						s -= lv * s_uv[k];
					}
					if (jpos < rdg)
						s /= s_dv[le];
					s_f[le] = s;
					continue;
				}
			}
			// beyond the staging capacity (rows much longer than a stencil's): from memory
			const int col = a.pat.bcolind[jpos];
			double s = a.avals[jpos];
			const int kb = a.posptr[jpos], ke = a.posptr[jpos + 1];
			for (int k = kb; k < ke; k++) {
				const int lp = a.lowerp[k];
				const int ll = lp - jlo;
				const double lv = (ll >= 0 && ll < nent) ? s_f[ll] : lsrc[lp];
				s -= lv * a.in[a.upperp[k]];
			}
			if (jpos < rdg)
				s /= a.in[a.pat.diagind[col]];
			if (le < nent)
				s_f[le] = s;
			else
				a.out[jpos] = s;
		}
	}
	__syncthreads();

#pragma unroll
	for (int i = 0; i < NE; i++) {
		const int q = tid + 256 * i;
		if (q < nent)
			a.out[(long)jlo + q] = s_f[q];
	}
}

// ---- the exact scalar factorisation as one launch (round 2) ------------------------------------------------------
// The scalar twin of sffactor4_kernel (kernels_factor4.hip, where the scheme and the row plans are described): one
// LANE per row, the 256 rows of a workgroup from ONE dependency level, a row's plan (16 ints) and all its operands --
// matrix values, the u_kj of its position pairs, the u_jj of its lower entries' columns, possibly still showing the
// fill pattern -- requested up front; the wave then walks the entries in lockstep (compile-time register indices)
// and waits, entry by entry, until every lane has what it needs.  The general single-launch kernel lets every lane
// retry on its own and loses an order of magnitude to the divergence (scalar 64^3: 2.5 ms per level launches, 12.3 ms);
// rows of one level never wait for each other, so here the whole wave waits for the level before.
// Same arithmetic in the same order as the general kernels (s -= l_ik u_kj pair by pair, then s / u_jj): same bits.
constexpr int X1_MAXE = 8, X1_MAXL = 4, X1_MAXP = 8;  // = the caps of build_row_plans
constexpr unsigned long long X1_PENDING = 0xFFF8DEADBEEF0001ull;  // = SFF_PENDING (kernels_factor.hip)
constexpr int X1_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ bool x1_pending(const double v)
{
	return (unsigned long long)__double_as_longlong(v) == X1_PENDING;
}

__device__ __forceinline__ double x1_coherent(const double *p)
{
	return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
	                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// before the launch: diagonal + upper entries <- the fill pattern, except upper entries without position pairs, whose
// factor value is the (scaled) matrix entry
__global__ __launch_bounds__(256) void x1_fill_kernel(const FactorArgs a)
{
	const int row = blockIdx.x * 256 + threadIdx.x;
	if (row >= a.pat.nbrows)
		return;
	const int dg = a.pat.diagind[row], jend = a.pat.browptr[row + 1];
	unsigned long long *const fq = reinterpret_cast<unsigned long long *>(a.out);
	int p0 = a.posptr[dg];
	for (int j = dg; j < jend; j++) {
		const int p1 = a.posptr[j + 1];
		if (j > dg && p1 == p0) {
			double v = a.avals[j];
			if (a.scale) {
				v *= a.scale[row];
				v *= a.scale[a.pat.bcolind[j]];
			}
			a.out[j] = v;
		} else
			fq[j] = X1_PENDING;
		p0 = p1;
	}
}

// ME / ML / MP: entries that need work, lower entries, position pairs a row may have (register arrays): 8 / 4 / 8 in
// general, 4 / 3 / 4 for a 7-point pattern (as for sffactor4_kernel).
template <int ME, int ML, int MP>
__global__ __launch_bounds__(256) void sffactor1_kernel(const FactorArgs a, const int4 *__restrict__ desc, int *ctl)
{
	double *const f = a.out;
	const long slot = (long)blockIdx.x * 256 + threadIdx.x;  // padded position in level order
	// the row's plan (kernels_factor4.hip: x4_describe_kernel): 0 first position, 1 row, 2..9 positions of the u_kj
	// of its pairs, 10..13 diagonal positions of its lower entries' columns, 14 / 15 counts and pair codes
	const int4 d0 = desc[slot * 4 + 0], d1 = desc[slot * 4 + 1], d2 = desc[slot * 4 + 2], d3 = desc[slot * 4 + 3];
	const int jbeg = d0.x, irow = d0.y;
#define X1_UPO(TT) ((TT) == 0 ? d0.z : (TT) == 1 ? d0.w : (TT) == 2 ? d1.x : (TT) == 3 ? d1.y : (TT) == 4 ? d1.z : (TT) == 5 ? d1.w : (TT) == 6 ? d2.x : d2.y)
#define X1_DPO(Q) ((Q) == 0 ? d2.z : (Q) == 1 ? d2.w : (Q) == 2 ? d3.x : d3.y)
	const unsigned w14 = (unsigned)d3.z, w15 = (unsigned)d3.w;
	const int ne = (int)(w14 & 15u), nl = (int)((w14 >> 4) & 15u), np = (int)((w14 >> 8) & 15u);
#define X1_CODE(TT) ((TT) < 4 ? (w14 >> (12 + 5 * (TT))) : (w15 >> (5 * ((TT)-4))))
#define X1_PQ(TT) ((TT) < np ? (int)(X1_CODE(TT) & 7u) : 8)
#define X1_PLL(TT) ((int)((X1_CODE(TT) >> 3) & 3u))
	unsigned pmask = 0u;  // entries with position pairs
#pragma unroll
	for (int tt = 0; tt < MP; tt++)
		pmask |= (tt < np) ? (1u << (X1_CODE(tt) & 7u)) : 0u;
#define X1_TODO(Q) ((Q) < ne && ((Q) <= nl || ((pmask >> (Q)) & 1u) != 0u))

	double aS[ME], uv[MP], dv[ML], lres[ML];
#pragma unroll
	for (int q = 0; q < ME; q++)
		aS[q] = X1_TODO(q) ? a.avals[jbeg + q] : 0.0;
#pragma unroll
	for (int tt = 0; tt < MP; tt++)
		uv[tt] = (tt < np) ? f[X1_UPO(tt)] : 0.0;
#pragma unroll
	for (int q = 0; q < ML; q++) {
		dv[q] = (q < nl) ? f[X1_DPO(q)] : 1.0;
		lres[q] = 0.0;
	}
	if (a.scale) {
#pragma unroll
		for (int q = 0; q < ME; q++)
			if (X1_TODO(q)) {
				aS[q] *= a.scale[irow];
				aS[q] *= a.scale[a.pat.bcolind[jbeg + q]];
			}
	}

	// ---- wait until everything the rows of this wave read from other rows has been published.  ONE lane polls, for
	// the whole wave, the operand the first waiting lane misses; when that has arrived everybody re-reads coherently
	// what still shows the fill pattern.  (Every lane re-reading its operands in a loop was the first form: thousands
	// of resident waves polling like that saturate the memory system and stretch a dependency hop from 0.6 us --
	// tools/probes/pingpong_probe.hip -- to 5 us.  Waiting entry by entry cost a round trip per entry.)
	int spins = 0;
	for (;;) {
		const double *miss = nullptr;
#pragma unroll
		for (int tt = MP - 1; tt >= 0; tt--)
			if (tt < np && x1_pending(uv[tt]))
				miss = f + X1_UPO(tt);
#pragma unroll
		for (int q = ML - 1; q >= 0; q--)
			if (q < nl && x1_pending(dv[q]))
				miss = f + X1_DPO(q);
		const unsigned long long waiting = __builtin_amdgcn_ballot_w64(miss != nullptr);
		if (waiting == 0ull)
			break;
		const int lead = __builtin_ctzll(waiting);
		const unsigned long long addr = (unsigned long long)reinterpret_cast<uintptr_t>(miss);
		const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)addr, lead);
		const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(addr >> 32), lead);
		const double *const gate = reinterpret_cast<const double *>((uintptr_t)(((unsigned long long)hi << 32) | lo));
		while (x1_pending(x1_coherent(gate))) {
			spins++;
			if (spins > X1_SPIN_LIMIT ||
			    ((spins & 255) == 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
				if ((threadIdx.x & 63) == 0)
					__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
			__builtin_amdgcn_s_sleep(2);
		}
#pragma unroll
		for (int q = 0; q < ML; q++)
			if (q < nl && x1_pending(dv[q]))
				dv[q] = x1_coherent(f + X1_DPO(q));
#pragma unroll
		for (int tt = 0; tt < MP; tt++)
			if (tt < np && x1_pending(uv[tt]))
				uv[tt] = x1_coherent(f + X1_UPO(tt));
		if (++spins > X1_SPIN_LIMIT) {
			if ((threadIdx.x & 63) == 0)
				__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
	}

	// ---- the rows' recurrences, entry by entry in lockstep, on registers
#pragma unroll
	for (int q = 0; q < ME; q++) {
		if (__builtin_amdgcn_ballot_w64(q < ne) == 0ull)
			break;
		const bool valid = X1_TODO(q);
		if (__builtin_amdgcn_ballot_w64(valid) == 0ull)
			continue;
		const bool lowerq = valid && q < nl;
		double sv = aS[q];
#pragma unroll
		for (int tt = 0; tt < MP; tt++)
			if (X1_PQ(tt) == q) {
				const int ll = X1_PLL(tt);
				const double lv = ll == 0 ? lres[0] : (ll == 1 ? lres[1] : ((ll == 2 || ML < 4) ? lres[2] : lres[ML < 4 ? 2 : 3]));
				sv -= lv * uv[tt];
			}
		if (q < ML && lowerq) {
			sv = sv / dv[q < ML ? q : 0];
			lres[q < ML ? q : 0] = sv;
		}
		if (valid) {
			if (lowerq)
				f[jbeg + q] = sv;
			else
				__hip_atomic_store(reinterpret_cast<unsigned long long *>(f + jbeg + q),
				                   (unsigned long long)__double_as_longlong(sv), __ATOMIC_RELAXED,
				                   __HIP_MEMORY_SCOPE_AGENT);
		}
	}
#undef X1_CODE
#undef X1_UPO
#undef X1_DPO
#undef X1_PQ
#undef X1_PLL
#undef X1_TODO
}

int g_factor1_enabled = -1;

}  // namespace

// The exact factorisation of a scalar (CSR) matrix with stencil-like rows as one launch (see sffactor1_kernel):
// 1 = done, 0 = does not apply, -1 = a wave gave up waiting.
int launch_factor1_syncfree(const FactorArgs &a, LevelSchedule &ls, hipStream_t s)
{
	if (a.pat.bs != 1 || a.in != a.out || !ls.built || !ls.meta || !ls.ctl)
		return 0;
	if (!build_row_plans(a, ls, 256, s))
		return 0;
	BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
	hipLaunchKernelGGL(x1_fill_kernel, dim3((unsigned)(((long)a.pat.nbrows + 255) / 256)), dim3(256), 0, s, a);
	if (ls.max_lower <= 3 && ls.f4_maxpairs <= 4 && ls.f4_maxtodo <= 4)
		hipLaunchKernelGGL((sffactor1_kernel<4, 3, 4>), dim3((unsigned)ls.f4_grid), dim3(256), 0, s, a,
		                   reinterpret_cast<const int4 *>(ls.f4_desc), ls.ctl);
	else
		hipLaunchKernelGGL((sffactor1_kernel<X1_MAXE, X1_MAXL, X1_MAXP>), dim3((unsigned)ls.f4_grid), dim3(256), 0, s, a,
		                   reinterpret_cast<const int4 *>(ls.f4_desc), ls.ctl);
	BHIP_CHECK(hipGetLastError());
	int ctl[2] = {0, 0};
	BHIP_CHECK(hipMemcpyAsync(ctl, ls.ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
	BHIP_CHECK(hipStreamSynchronize(s));
	return ctl[1] == 0 ? 1 : -1;
}

// the plan of factor1p_kernel: dcol (nnz ints), chunks (factor1_plan_chunks(nbrows) int4)
long factor1_plan_chunks(int nbrows)
{
	return ((long)nbrows + F1_RCHUNK - 1) / F1_RCHUNK;
}

void build_factor1_plan(const Pattern &pat, const int *posptr, int *dcol, int4 *chunks, hipStream_t s)
{
	if (pat.nbrows == 0)
		return;
	hipLaunchKernelGGL(f1_plan_kernel, dim3((unsigned)(((long)pat.nbrows + 255) / 256)), dim3(256), 0, s, pat, posptr, dcol, chunks);
	BHIP_CHECK(hipGetLastError());
}

void set_factor1_enabled(int on)
{
	g_factor1_enabled = on;
}

// returns false when the tuned kernel does not cover the request (caller uses the general kernel)
bool launch_factor1(const FactorArgs &a, hipStream_t s)
{
	if (g_factor1_enabled < 0) {
		const char *e = std::getenv("BLASTED_HIP_FACTOR1");
		g_factor1_enabled = (e && std::strcmp(e, "0") == 0) ? 0 : 1;
	}
	if (!g_factor1_enabled || a.pat.bs != 1 || a.pat.nbrows == 0 || a.rows || !a.out)
		return false;
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + F1_RCHUNK - 1) / F1_RCHUNK);
	if ((a.in == a.out || a.lrow_fresh) && !a.scale && a.f1_dcol && a.f1_chunks)
		hipLaunchKernelGGL(factor1p_kernel, dim3(grid), dim3(256), 0, s, a);
	else if (a.in == a.out)
		hipLaunchKernelGGL(factor1_kernel<true>, dim3(grid), dim3(256), 0, s, a);
	else
		hipLaunchKernelGGL(factor1_kernel<false>, dim3(grid), dim3(256), 0, s, a);
	BHIP_CHECK(hipGetLastError());
	return true;
}

}  // namespace bhip
