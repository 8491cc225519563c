// kernels_factor4.hip -- tuned asynchronous block-ILU(0) factorisation sweep for bs = 4, column-major
// blocks (async_block_ilu0_factorize, src/kernels/kernels_ilu0_factorize.hpp:71-98; sweep driver
// src/async_blockilu_factor.cpp:186-204).  Same fixed-point map as factor_sweep_kernel in
// kernels_factor.hip; what differs is how the 4x4x4 block products are done and how indices arrive:
//
//  * one wave owns 4 block-rows at a time and all block products run on the matrix core:
//    v_mfma_f64_4x4x4_4b_f64 computes four independent 4x4x4 products per instruction.  Its operand
//    layout (probed on gfx950, tools/probes/mfma_f64_4x4x4_probe.hip): for block slot b
//        A(i,k) in lane 16k + 4b + i,   B(k,j) in lane 16k + 4b + j,   C/D(i,j) in lane 16i + 4b + j.
//    A block is therefore loaded straight into operand layout by giving every lane the right element
//    offset inside the 128-byte block -- "offA" (element (r = L%4, c = L/16)) or "offD" (element
//    (r = L/16, c = L%4)) -- and no lane ever exchanges data for a product.
//  * upper / diagonal blocks:  S = A - sum L U      with A-operand = L (offA), B-operand = U (offD);
//    S comes out in D layout and is stored with offD.
//  * lower blocks need S as the A operand of the final product S * inverse(U_jj).  Computing the
//    TRANSPOSED sum  S^T = A^T - sum U^T L^T  (A-operand = U loaded with offD, B-operand = L loaded
//    with offA: the same two loads, roles swapped) leaves S exactly in A-operand layout.  Both kinds of
//    block can sit in one wave: the role swap is a per-lane select.
//  * inverse(U_jj) is formed in B-operand layout by the adjugate (Eigen's closed form for n <= 4), the
//    only place lanes exchange values (17 double shuffles per lower block, four blocks at a time).
//  * browptr, bcolind, posptr and the (lower, upper) position pairs of a 64-row chunk are staged in
//    LDS once, coalesced; no value load waits on an index load from HBM.
// Every entry of the factor is produced in registers and stored once per sweep.
#include "ctx.hpp"
#include "lanes.hpp"
#include "stage.hpp"

#include <cstdlib>
#include <cstring>

namespace bhip {

namespace {

constexpr int F4_RCHUNK = 64;             // rows per workgroup
constexpr int F4_CAPB = 16 * F4_RCHUNK;   // staged block positions (column index + posptr)
constexpr int F4_CAPP = 16 * F4_RCHUNK;   // staged (lower, upper) pairs

__device__ __forceinline__ double mfma444(const double a, const double b, const double c)
{
	return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// v holds U(r = k, c = j) in lane 16k + 4b + j (B-operand layout).  Returns inverse(U)(k, j) in the same
// lane: adjugate / determinant.
__device__ __forceinline__ double inverse_b_layout(const double v, const int k, const int b4, const int j)
{
	// cofactor C(j,k): delete row j and column k
	double M[3][3];
#pragma unroll
	for (int x = 0; x < 3; x++)
#pragma unroll
		for (int y = 0; y < 3; y++) {
			const int ri = x + (x >= j ? 1 : 0);
			const int ci = y + (y >= k ? 1 : 0);
			M[x][y] = __shfl(v, 16 * ri + b4 + ci, 64);
		}
	const double minor = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) -
	                     M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
	                     M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
	const double cof = ((j + k) & 1) ? -minor : minor;  // C(j,k) -> inverse(k,j) = C(j,k)/det
	// det = sum_q U(0,q) C(0,q);  C(0,q) is held by the lane computing inverse(q,0): lane 16q + 4b
	double det = 0.0;
#pragma unroll
	for (int q = 0; q < 4; q++)
		det += __shfl(v, b4 + q, 64) * __shfl(cof, 16 * q + b4, 64);
	return cof * (1.0 / det);
}

__global__ __launch_bounds__(256) void factor4_kernel(const FactorArgs a)
{
	__shared__ int s_rp[F4_RCHUNK + 1];
	__shared__ int s_col[F4_CAPB];
	__shared__ int s_pp[F4_CAPB + 1];
	__shared__ int s_lp[F4_CAPP];
	__shared__ int s_up[F4_CAPP];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int k = lane >> 4, b = (lane >> 2) & 3, m = lane & 3;
	const int b4 = 4 * b;
	const int offA = k * 4 + m;  // element (r = m, c = k)
	const int offD = m * 4 + k;  // element (r = k, c = m)

	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const int r0 = (int)chunk * F4_RCHUNK;
	const int rc = (nb - r0) < F4_RCHUNK ? (nb - r0) : F4_RCHUNK;

	int jlo, plo;
	stage_factor_indices<F4_RCHUNK, F4_CAPB, F4_CAPP>(a.pat, a.posptr, a.lowerp, a.upperp, r0, rc, tid, s_rp, s_col,
	                                                   s_pp, s_lp, s_up, jlo, plo);

	for (int step = 0; step < F4_RCHUNK / 16; step++) {
		const int ls = step * 16 + wave * 4 + b;  // this block slot's row inside the chunk
		const bool rowok = ls < rc;
		const int irow = r0 + ls;
		const int jbeg = rowok ? s_rp[ls] : 0;
		const int len = rowok ? s_rp[ls + 1] - jbeg : 0;

		for (int p = 0; __any(p < len); p++) {
			const bool valid = p < len;
			const int jpos = jbeg + p;
			const int bidx = jpos - jlo;
			int col = 0, kb = 0, ke = 0;
			if (valid) {
				if (bidx < F4_CAPB) {
					col = s_col[bidx];
					kb = s_pp[bidx];
					ke = s_pp[bidx + 1];
				} else {
					col = a.pat.bcolind[jpos];
					kb = a.posptr[jpos];
					ke = a.posptr[jpos + 1];
				}
			}
			const bool lower = valid && irow > col;
			// lower blocks work on S^T (A-operand layout), the others on S (D layout)
			const int offS = lower ? offA : offD;
			double sval = valid ? a.avals[(long)jpos * 16 + offS] : 0.0;
			double dval = 0.0;  // U_jj in B-operand layout, lower blocks only
			if (lower)
				dval = a.in[(long)a.pat.diagind[col] * 16 + offD];
			if (a.scale && valid) {
				// (r,c) of this lane's entry: D layout (k, m); transposed layout (m, k)
				const int r = lower ? m : k, c = lower ? k : m;
				sval *= a.scale[(long)irow * 4 + r] * a.scale[(long)col * 4 + c];
			}

			double acc = 0.0;
			const int cnt = ke - kb;
			for (int kk = 0; __any(kk < cnt); kk++) {
				double lval = 0.0, uval = 0.0;
				if (kk < cnt) {
					const int pidx = kb + kk - plo;
					int lp, up;
					if (pidx < F4_CAPP) {
						lp = s_lp[pidx];
						up = s_up[pidx];
					} else {
						lp = a.lowerp[kb + kk];
						up = a.upperp[kb + kk];
					}
					lval = a.in[(long)lp * 16 + offA];
					uval = a.in[(long)up * 16 + offD];
				}
				// upper/diag: sum += L U ; lower: sum^T += U^T L^T (same loads, roles swapped)
				acc = mfma444(lower ? uval : lval, lower ? lval : uval, acc);
			}
			double res = sval - acc;

			if (__any(lower)) {
				const double inv = inverse_b_layout(lower ? dval : ((k == m) ? 1.0 : 0.0), k, b4, m);
				const double prod = mfma444(lower ? res : 0.0, lower ? inv : 0.0, 0.0);
				if (lower)
					res = prod;  // S * inverse(U_jj), D layout
			}

			if (valid) {
				double *const dst = a.out + (long)jpos * 16 + offD;
				*dst = res;
			}
		}
	}
}

int g_factor4_enabled = -1;

}  // namespace

void set_factor4_enabled(int on)
{
	g_factor4_enabled = on;
}

// returns false when the tuned kernel does not cover the request (caller uses the generic kernel)
bool launch_factor4(const FactorArgs &a, hipStream_t s)
{
	if (g_factor4_enabled < 0) {
		const char *e = std::getenv("BLASTED_HIP_FACTOR4");
		g_factor4_enabled = (e && std::strcmp(e, "0") == 0) ? 0 : 1;
	}
	if (!g_factor4_enabled || a.pat.bs != 4 || a.pat.rowmajor || a.pat.nbrows == 0)
		return false;
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + F4_RCHUNK - 1) / F4_RCHUNK);
	hipLaunchKernelGGL(factor4_kernel, dim3(grid), dim3(256), 0, s, a);
	BHIP_CHECK(hipGetLastError());
	return true;
}

}  // namespace bhip
