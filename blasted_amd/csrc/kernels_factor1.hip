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
			if (a.scale)
				s_av[q] *= a.scale[row] * a.scale[col];
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
				if (a.scale)
					s *= a.scale[irow] * a.scale[col];
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

int g_factor1_enabled = -1;

}  // namespace

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
	if (a.in == a.out)
		hipLaunchKernelGGL(factor1_kernel<true>, dim3(grid), dim3(256), 0, s, a);
	else
		hipLaunchKernelGGL(factor1_kernel<false>, dim3(grid), dim3(256), 0, s, a);
	BHIP_CHECK(hipGetLastError());
	return true;
}

}  // namespace bhip
