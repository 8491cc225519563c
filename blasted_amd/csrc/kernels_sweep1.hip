// kernels_sweep1.hip -- the row sweeps of SCALAR (CSR) matrices with short rows, one lane per row.
//
// Reference loops: scalar_unit_lower_triangular / scalar_upper_triangular (kernels_ilu_apply.hpp:17-42),
// scalar_fgs / scalar_bgs (kernels_sgs.hpp:17-44), scalar_relax (kernels_relaxation.hpp:17-33), the CSR
// matrix_apply / gemv3 (blas/matvecs.cpp) and the scalar Jacobi application: the same operator table as
// kernels_sweep.hip, whose general kernel gives a scalar row four lanes of 8 bytes each behind an LDS index
// stage.  That form keeps ~1 KB in flight per wave (16 rows a step) and ran at 0.53 of the HBM peak at 256^3;
// what bounds it is bytes in flight per CU, not instructions.
//
// Here a lane owns a row: it reads its row's pointers (coalesced across the wave), then ALL entries and column
// indices of its row part in one straight-line predicated pass (64 rows x (12 bytes x entries) in flight per
// wave; neighbouring lanes' entries share cache lines, so the strided loads are served by the L1 line they
// opened), then the gathers, and runs the sum in storage order -- the reference's own order, no cross-lane
// reduction.  No LDS, no barrier: a workgroup is only the unit of the XCD-aware chunk numbering.  Rows longer
// than the straight-line part finish in a remainder loop; matrices whose longest row exceeds
// SWEEP1_MAX_ROW keep the general kernel (a lane walking a long row alone would serialise it).
//
// Async semantics as in kernels_sweep.hip: plain loads and one store per output; the 64 rows of a wave (and the
// NR x 256 rows of a workgroup) are read before any of them is stored.
#include "ctx.hpp"
#include "lanes.hpp"

#include <cstdlib>

namespace bhip {

namespace {

constexpr int SWEEP1_MAX_ROW = 16;

// SEQ (in-place sweeps): the workgroup is ONE wave that takes its 256 rows 64 at a time, each step after the stores
// of the step before -- the same bytes in flight per wave, but inside a chunk the sweep is in order at 64-row
// granularity instead of 256 rows read at once by four waves.
template <int PART, int POST, int DSRC, int NR, bool SEQ>
__global__ __launch_bounds__(SEQ ? 64 : 256) void sweep1_kernel(const SweepArgs a)
{
	static_assert(!SEQ || NR == 1, "the sequential form takes one row per lane and step");
	constexpr bool DIAG_FIRST = PART == PART_UPPER && (DSRC == D_VALS_DIAG || DSRC == D_RECIP_DIAG);
	constexpr int KFIX = (PART == PART_NONE) ? 0 : ((PART == PART_ALL || PART == PART_OFFDIAG) ? 8 : 4);
	constexpr int KF = KFIX > 0 ? KFIX : 1;
	constexpr int RCHUNK = 256 * NR;

	const int tid = threadIdx.x;
	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x, (unsigned)(a.xcd_shift < 0 ? 4 : a.xcd_shift));
	const long lin0 = (long)chunk * RCHUNK;
	const int rc = (int)((nb - lin0) < RCHUNK ? (nb - lin0) : RCHUNK);
	const int r0 = a.descending ? (int)(nb - lin0 - rc) : (int)lin0;  // rows [r0, r0 + rc)

	const int *const rp = a.pat.browptr + r0;
	const int *const dgp = a.pat.diagind + r0;
	const int *const __restrict__ cols = a.pat.bcolind;
	const double *const __restrict__ vals = a.vals;
	const double *const xin = a.xin;

#pragma unroll 1
	for (int seq = 0; seq < (SEQ ? 4 : 1); seq++) {
	bool ok[NR];
	int lr[NR], jbeg[NR], jend[NR], dg[NR];
#pragma unroll
	for (int q = 0; q < NR; q++) {
		const int ls = SEQ ? seq * 64 + tid : q * 256 + tid;  // position in sweep order
		ok[q] = ls < rc;
		lr[q] = ok[q] ? (a.descending ? rc - 1 - ls : ls) : 0;
		jbeg[q] = jend[q] = dg[q] = 0;
		if (ok[q]) {
			if (PART == PART_LOWER) {
				jbeg[q] = rp[lr[q]];
				jend[q] = dgp[lr[q]];
			} else if (PART == PART_UPPER) {
				dg[q] = dgp[lr[q]];
				jbeg[q] = DIAG_FIRST ? dg[q] : dg[q] + 1;  // the diagonal entry rides in front of the upper ones
				jend[q] = rp[lr[q] + 1];
			} else if (PART == PART_OFFDIAG) {
				jbeg[q] = rp[lr[q]];
				jend[q] = rp[lr[q] + 1];
				dg[q] = dgp[lr[q]];
			} else if (PART == PART_ALL) {
				jbeg[q] = rp[lr[q]];
				jend[q] = rp[lr[q] + 1];
			}
		}
	}

	double v[NR][KF], rv[NR], d[NR];
	int col[NR][KF];
#pragma unroll
	for (int q = 0; q < NR; q++) {
#pragma unroll
		for (int k = 0; k < KFIX; k++) {
			const int jj = jbeg[q] + k;
			// (relaxation: the diagonal entry of A is not part of the sum -- do not fetch it)
			const bool p = jj < jend[q] && !(PART == PART_OFFDIAG && jj == dg[q]);
			v[q][k] = p ? vals[jj] : 0.0;
			col[q][k] = (p && !(DIAG_FIRST && k == 0)) ? cols[jj] : -1;
		}
		rv[q] = 0.0;
		if (ok[q] && a.rhs) {
			rv[q] = a.rhs[r0 + lr[q]];
			if (a.rscale)
				rv[q] *= a.rscale[r0 + lr[q]];
		}
		d[q] = (DSRC == D_DBLOCKS && ok[q]) ? a.dvals[r0 + lr[q]] : 0.0;
	}

	double xv[NR][KF];
#pragma unroll
	for (int q = 0; q < NR; q++)
#pragma unroll
		for (int k = 0; k < KFIX; k++)
			xv[q][k] = col[q][k] >= 0 ? xin[col[q][k]] : 0.0;

#pragma unroll
	for (int q = 0; q < NR; q++) {
		double acc = 0.0;
#pragma unroll
		for (int k = 0; k < KFIX; k++)
			acc += v[q][k] * xv[q][k];  // xv is zero for the diagonal / absent entries
		// longer rows: the rest four entries at a time
		for (int jb = jbeg[q] + KFIX; jb < jend[q]; jb += 4) {
			double v4[4], x4[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int jj = jb + k;
				const bool p = jj < jend[q] && !(PART == PART_OFFDIAG && jj == dg[q]);
				v4[k] = p ? vals[jj] : 0.0;
				x4[k] = p ? xin[cols[jj]] : 0.0;
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
				acc += v4[k] * x4[k];
		}

		if (DSRC == D_VALS_DIAG)
			d[q] = v[q][0];
		else if (DSRC == D_RECIP_DIAG)
			d[q] = ok[q] ? 1.0 / v[q][0] : 0.0;

		double out;
		if (POST == POST_SUB)
			out = rv[q] - acc;
		else if (POST == POST_D_SUB)
			out = d[q] * (rv[q] - acc);
		else if (POST == POST_SUB_D)
			out = rv[q] - d[q] * acc;
		else {
			out = a.a * acc;
			if (a.b != 0.0)
				out += a.b * rv[q];
		}
		if (ok[q])
			a.xout[r0 + lr[q]] = out;
	}
	}
}

// ---- whole-row operators (product / gemv3, relaxation) on short rows: products staged through LDS ------------------
// One lane per row reading its own seven entries is slower than the general kernel here (above).  This form keeps
// every global access coalesced: the workgroup's threads load the chunk's entries and column indices in storage
// order (thread q takes entries q, q + 256, ...), gather x for THEIR entries straight from registers, multiply, and
// leave the products in LDS; after one barrier a lane sums its row's products in storage order (the reference's
// order), applies the operator and stores.  LDS holds one double per entry (14 KB for 256 7-point rows), the whole
// chunk -- 21 KB of entries and indices, then 14 KB of gathers -- is in flight at once, two memory round trips per
// chunk.  The 256 rows of a chunk are read before any is stored, so in place (asynchronous relaxation) the general
// kernel stays; this one takes the product and the relaxation passes into a second buffer.  Rows of at most
// SWEEP1S_MAX_ROW entries (host-side check: a chunk then fits the staging capacity).
constexpr int SWEEP1S_MAX_ROW = 8;
constexpr int SWEEP1S_CAP = 256 * SWEEP1S_MAX_ROW;

template <int PART, int POST, int DSRC>
__global__ __launch_bounds__(256) void sweep1s_kernel(const SweepArgs a)
{
	static_assert(PART == PART_ALL || PART == PART_OFFDIAG, "whole-row operators");
	__shared__ double s_prod[SWEEP1S_CAP];

	const int tid = threadIdx.x;
	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x, (unsigned)(a.xcd_shift < 0 ? 4 : a.xcd_shift));
	const long lin0 = (long)chunk * 256;
	const int rc = (int)((nb - lin0) < 256 ? (nb - lin0) : 256);
	const int r0 = a.descending ? (int)(nb - lin0 - rc) : (int)lin0;  // rows [r0, r0 + rc)
	const int jlo = __builtin_amdgcn_readfirstlane(a.pat.browptr[r0]);
	const int jhi = __builtin_amdgcn_readfirstlane(a.pat.browptr[r0 + rc]);
	const int nent = jhi - jlo;  // <= SWEEP1S_CAP (launcher)

	// this lane's row (order inside a chunk does not matter: nothing is stored before everything is read)
	const bool ok = tid < rc;
	int jbeg = 0, jend = 0, dg = -1;
	double rv = 0.0, d = 0.0;
	if (ok) {
		jbeg = a.pat.browptr[r0 + tid];
		jend = a.pat.browptr[r0 + tid + 1];
		if (PART == PART_OFFDIAG)
			dg = a.pat.diagind[r0 + tid];
		if (a.rhs) {
			rv = a.rhs[r0 + tid];
			if (a.rscale)
				rv *= a.rscale[r0 + tid];
		}
		if (DSRC == D_DBLOCKS)
			d = a.dvals[r0 + tid];
	}

	double v[SWEEP1S_MAX_ROW];
	int col[SWEEP1S_MAX_ROW];
#pragma unroll
	for (int i = 0; i < SWEEP1S_MAX_ROW; i++) {
		const int q = tid + 256 * i;
		v[i] = q < nent ? a.vals[(long)jlo + q] : 0.0;
		col[i] = q < nent ? a.pat.bcolind[jlo + q] : -1;
	}
	double xv[SWEEP1S_MAX_ROW];
#pragma unroll
	for (int i = 0; i < SWEEP1S_MAX_ROW; i++)
		xv[i] = col[i] >= 0 ? a.xin[col[i]] : 0.0;
#pragma unroll
	for (int i = 0; i < SWEEP1S_MAX_ROW; i++) {
		const int q = tid + 256 * i;
		if (q < nent)
			s_prod[q] = v[i] * xv[i];
	}
	__syncthreads();

	if (!ok)
		return;
	double acc = 0.0;
	for (int jj = jbeg; jj < jend; jj++)
		if (!(PART == PART_OFFDIAG && jj == dg))
			acc += s_prod[jj - jlo];
	double out;
	if (POST == POST_D_SUB)
		out = d * (rv - acc);
	else {
		out = a.a * acc;
		if (a.b != 0.0)
			out += a.b * rv;
	}
	a.xout[r0 + tid] = out;
}

// tuning "scalarlane=auto|0|1|2|3|4": 0 = the general kernel everywhere, 1 / 2 = this kernel with one / two rows per
// lane, 3 = as 1 and for the whole-row operators too, 4 = the one-wave sequential form; auto (-1, default) = this
// kernel (one row per lane) for sweeps that write a SECOND buffer -- synchronous / deterministic sweeps, where only
// speed differs (256^3: 0.43 against 0.53 ms per L+U pair, 0.71 against 0.57 of the HBM peak) -- and the general
// kernel for IN-PLACE sweeps.  In place the lane-per-row form is the faster sweep but the worse preconditioner: it
// has 2 048 rows per CU in flight instead of 512, so more of a sweep reads the iterate of the sweep before
// (256^3, 3+3 sweeps: distance to the exact solve 0.244 against 0.212), and inside the reference's flexible solver
// that costs more than the sweep gains -- GCR(30) on 160^3 with 3 sweeps per application: 603-607 iterations / 640-650
// ms with the general kernel, 855-882 / 834-863 ms with one lane per row, 720-722 / 731-734 ms with the sequential
// form; with 5 sweeps all three meet at 478-524 iterations / 605-622 ms (profiles/r03_scalar_kernels.txt).
int g_scalar_lane = [] {
	const char *e = std::getenv("BLASTED_HIP_SCALARLANE");
	return e ? std::atoi(e) : -1;
}();

int g_scalar_stage = 1;

template <int NR, bool SEQ>
bool dispatch1(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + 256 * NR - 1) / (256 * NR));
#define BHIP_CASE1(P, Q, D)                                                                    \
	if (part == P && post == Q && dsrc == D) {                                                 \
		hipLaunchKernelGGL((sweep1_kernel<P, Q, D, NR, SEQ>), dim3(grid), dim3(SEQ ? 64 : 256), 0, s, a); \
		return true;                                                                           \
	}
	BHIP_CASE1(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASE1(PART_UPPER, POST_SUB, D_NONE)
	BHIP_CASE1(PART_UPPER, POST_D_SUB, D_RECIP_DIAG)
	BHIP_CASE1(PART_LOWER, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE1(PART_UPPER, POST_SUB_D, D_DBLOCKS)
	BHIP_CASE1(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE1(PART_ALL, POST_AXPBY, D_NONE)
	BHIP_CASE1(PART_NONE, POST_D_SUB, D_DBLOCKS)
#undef BHIP_CASE1
	return false;
}

}  // namespace

void set_scalar_lane(int v)
{
	g_scalar_lane = v;
}

void set_scalar_stage(int v)
{
	g_scalar_stage = v;
}

bool launch_sweep1(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	if (a.pat.bs != 1 || g_scalar_lane == 0 || a.pat.max_row_len > SWEEP1_MAX_ROW)
		return false;
	// whole rows (SpMV / gemv3, relaxation): a lane's seven entries are 84 bytes apart from its neighbour's, seven
	// loads a wave each spread over 28 cache lines -- measured SLOWER than the general kernel from 200^3 on (256^3:
	// product 0.50 against 0.38 ms, relaxation step 1.05 against 0.86 ms; 128^3 equal), so these keep the general
	// kernel unless asked for ("scalarlane=3")
	if ((part == PART_ALL || part == PART_OFFDIAG) && g_scalar_lane != 3) {
		// the staged form: product always, relaxation passes into a second buffer ("scalarstage=0": never)
		if (!g_scalar_stage || a.pat.max_row_len > SWEEP1S_MAX_ROW || (part == PART_OFFDIAG && a.xin == a.xout))
			return false;
		const unsigned grid = (unsigned)(((long)a.pat.nbrows + 255) / 256);
		if (part == PART_ALL && post == POST_AXPBY && dsrc == D_NONE)
			hipLaunchKernelGGL((sweep1s_kernel<PART_ALL, POST_AXPBY, D_NONE>), dim3(grid), dim3(256), 0, s, a);
		else if (part == PART_OFFDIAG && post == POST_D_SUB && dsrc == D_DBLOCKS)
			hipLaunchKernelGGL((sweep1s_kernel<PART_OFFDIAG, POST_D_SUB, D_DBLOCKS>), dim3(grid), dim3(256), 0, s, a);
		else
			return false;
		BHIP_CHECK(hipGetLastError());
		return true;
	}
	if (g_scalar_lane < 0 && a.xin == a.xout && part != PART_NONE)
		return false;
	const bool done = g_scalar_lane == 2 ? dispatch1<2, false>(a, part, post, dsrc, s)
	                  : (g_scalar_lane == 4 ? dispatch1<1, true>(a, part, post, dsrc, s) : dispatch1<1, false>(a, part, post, dsrc, s));
	if (done)
		BHIP_CHECK(hipGetLastError());
	return done;
}

}  // namespace bhip
