// kernels_factor8.hip -- asynchronous block-ILU(0) factorisation sweep for bs = 8, column-major blocks
// (async_block_ilu0_factorize, src/kernels/kernels_ilu0_factorize.hpp:71-98), BASELINE.json's config 5:
// "block-ILU with MFMA for the 8x8 block GEMMs".  At bs = 8 the block product is a dense 8x8x8
// contraction and goes to the matrix core.
//
// One wave owns one 8x8 block at a time: its 64 lanes are the 64 entries of the result, arranged as the
// four 4x4 output tiles (ti,tj) of v_mfma_f64_4x4x4_4b_f64 (block slot b = 2 ti + tj; operand layout
// probed on gfx950, see kernels_factor4.hip):
//     lane 16k + 4b + m   holds   D(4ti + k, 4tj + m),
//     A-operand for inner tile tk:  X(4ti + m, 4tk + k),     B-operand:  Y(4tk + k, 4tj + m).
// An 8x8x8 product is two MFMA instructions (tk = 0, 1) accumulating into the same four tiles, and every
// operand is loaded from its 512-byte block directly in operand layout (per-lane element offsets): the
// sum  S = A - sum L U  needs no lane exchange at all.  For a lower block, S is re-read in A-operand
// layout with two shuffles and multiplied by inverse(U_jj) on the matrix core too.
//
// inverse(U_jj): the reference inverts U_jj inside the row kernel, once per lower block and sweep
// (kernels_ilu0_factorize.hpp:91).  Here the diagonal blocks of the sweep's input iterate are inverted
// once per sweep by a pre-pass (invert_blocks_kernel, Gauss-Jordan with partial pivoting) into a side
// array that the lower blocks read in B-operand layout.  For synchronous (Jacobi) sweeps this is the
// same arithmetic on the same iterate; for in-place asynchronous sweeps the inverse may be one update
// older than the freshest diagonal, which chaotic iteration permits.
//
// All control flow is wave-uniform (a wave walks one block-row); indices of a 32-row chunk are staged in
// LDS; each entry of the factor is produced in registers and stored once.
#include "ctx.hpp"
#include "lanes.hpp"
#include "stage.hpp"

#include <cstdlib>
#include <cstring>

namespace bhip {

namespace {

constexpr int F8_RCHUNK = 32;
constexpr int F8_CAPB = 16 * F8_RCHUNK;
constexpr int F8_CAPP = 16 * F8_RCHUNK;

__device__ __forceinline__ double mfma444(const double a, const double b, const double c)
{
	return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

constexpr int F8_MW = 4, F8_ML = 3, F8_MP = 4;  // the fast path's window: blocks with work, lower ones, pairs

// RM (round 4): ROW-major blocks = the column-major image of the transposed blocks; the kernel then works on X' = X^T:
// S' = A' - sum U'_kj L'_ik (the two blocks of a pair swap their operand roles) and L'_ij = inverse(U'_jj) S' (a product
// from the left: the inverse is read in A-operand layout from the side array -- the pre-pass inverts in the blocks' own
// layout -- and S' is re-read in B-operand layout with two shuffles).
template <bool INPLACE_FAST, bool RM>
__global__ __launch_bounds__(256) void factor8_kernel(const FactorArgs a, const double *dinv)
{
	__shared__ int s_rp[F8_RCHUNK + 1];
	__shared__ int s_col[F8_CAPB];
	__shared__ int s_pp[F8_CAPB + 1];
	__shared__ int s_lp[F8_CAPP];
	__shared__ int s_up[F8_CAPP];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int k = lane >> 4, b = (lane >> 2) & 3, m = lane & 3;
	const int ti = b >> 1, tj = b & 1;
	// element offsets inside a column-major 8x8 block (offset = c*8 + r)
	const int offD = (4 * tj + m) * 8 + 4 * ti + k;   // (r = 4ti+k, c = 4tj+m)
	const int offA0 = (0 + k) * 8 + 4 * ti + m;        // (r = 4ti+m, c = k)      tk = 0
	const int offA1 = (4 + k) * 8 + 4 * ti + m;        // (r = 4ti+m, c = 4+k)    tk = 1
	const int offB0 = (4 * tj + m) * 8 + 0 + k;        // (r = k,     c = 4tj+m)  tk = 0
	const int offB1 = (4 * tj + m) * 8 + 4 + k;        // (r = 4+k,   c = 4tj+m)  tk = 1
	const int srcA0 = 16 * m + 4 * (2 * ti + 0) + k;   // lane holding S(4ti+m, k)   in D layout
	const int srcA1 = 16 * m + 4 * (2 * ti + 1) + k;   // lane holding S(4ti+m, 4+k)
	const int srcB0 = 16 * k + 4 * (0 + tj) + m;       // lane holding S(k,   4tj+m) in D layout (row-major form)
	const int srcB1 = 16 * k + 4 * (2 + tj) + m;       // lane holding S(4+k, 4tj+m)
	// what the second factor of a product S x (or x S') is read with, and the shuffles that re-read a result as the
	// first (column-major) or the second (row-major) factor
	const int offX0 = RM ? offA0 : offB0, offX1 = RM ? offA1 : offB1;
	const int srcS0 = RM ? srcB0 : srcA0, srcS1 = RM ? srcB1 : srcA1;
	// the scaling factors of this lane's entry: D layout holds X(4ti+k, 4tj+m), row-major X'(4ti+k, 4tj+m) = X(4tj+m, 4ti+k)
	const int scr = RM ? 4 * tj + m : 4 * ti + k, scc = RM ? 4 * ti + k : 4 * tj + m;

	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const int r0 = (int)chunk * F8_RCHUNK;
	const int rc = (nb - r0) < F8_RCHUNK ? (nb - r0) : F8_RCHUNK;

	int jlo, plo;
	stage_factor_indices<F8_RCHUNK, F8_CAPB, F8_CAPP>(a.pat, a.posptr, a.lowerp, a.upperp, r0, rc, tid, s_rp, s_col,
	                                                   s_pp, s_lp, s_up, jlo, plo);

	for (int ls = wave; ls < rc; ls += 4) {
		const int irow = r0 + ls;
		const int jbeg = __builtin_amdgcn_readfirstlane(s_rp[ls]);
		const int jend = __builtin_amdgcn_readfirstlane(s_rp[ls + 1]);

		// ---- round 3: stencil-like rows of an in-place sweep, everything requested up front.  The loop below takes a
		// row block by block: a block's loads go out after the store of the block before (in place, `in` and `out`
		// alias), and a pair's l_ik -- a lower block of THIS row, stored a moment ago -- is read back from memory
		// behind that store.  A row whose work lies in its first F8_MW blocks (at most F8_ML lower ones, at most
		// F8_MP pairs; the blocks behind are pair-less upper blocks that the sweep leaves alone) instead requests
		// its matrix blocks, the inverses for its lower blocks and the u_kj of all its pairs together (8 KB in
		// flight for an interior 7-point row), then runs the blocks in storage order out of registers, handing each
		// finished lower block on in A-operand layout (the two shuffles a re-read would have saved are the price).
		// Same operations on the same operands in the same order as the loop below.
		if (INPLACE_FAST && jend - jlo <= F8_CAPB) {
			bool valid[F8_MW], work[F8_MW], low[F8_MW];
			int col[F8_MW], kb[F8_MW], ke[F8_MW];
			bool fast = true;
#pragma unroll
			for (int e = 0; e < F8_MW; e++) {
				const int jpos = jbeg + e;
				valid[e] = jpos < jend;
				const int bidx = valid[e] ? jpos - jlo : 0;
				col[e] = __builtin_amdgcn_readfirstlane(s_col[bidx]);
				kb[e] = __builtin_amdgcn_readfirstlane(s_pp[bidx]);
				ke[e] = __builtin_amdgcn_readfirstlane(s_pp[bidx + 1]);
				work[e] = valid[e] && !(a.skip_fixed && col[e] > irow && ke[e] == kb[e]);
				low[e] = work[e] && col[e] < irow;
				if (low[e] && e >= F8_ML)
					fast = false;
			}
			// the blocks behind the window must all be left alone
			for (int jpos = jbeg + F8_MW; jpos < jend; jpos++) {
				const int bidx = jpos - jlo;
				const int c = __builtin_amdgcn_readfirstlane(s_col[bidx]);
				const int k0 = __builtin_amdgcn_readfirstlane(s_pp[bidx]), k1 = __builtin_amdgcn_readfirstlane(s_pp[bidx + 1]);
				if (!(a.skip_fixed && c > irow && k1 == k0))
					fast = false;
			}
			const int pb = __builtin_amdgcn_readfirstlane(s_pp[jbeg - jlo]);
			const int wend = (jbeg + F8_MW < jend ? jbeg + F8_MW : jend) - jlo;
			const int np = __builtin_amdgcn_readfirstlane(s_pp[wend]) - pb;
			if (np > F8_MP || pb - plo + np > F8_CAPP)
				fast = false;
			if (fast) {
				double av[F8_MW], dv[F8_ML][2], uo[F8_MP][2], lres[F8_ML][2];
				int slot[F8_MP];
#pragma unroll
				for (int e = 0; e < F8_MW; e++)
					av[e] = work[e] ? a.avals[(long)(jbeg + e) * 64 + offD] : 0.0;
#pragma unroll
				for (int e = 0; e < F8_ML; e++) {
					dv[e][0] = low[e] ? dinv[(long)col[e] * 64 + offX0] : 0.0;
					dv[e][1] = low[e] ? dinv[(long)col[e] * 64 + offX1] : 0.0;
					lres[e][0] = lres[e][1] = 0.0;
				}
#pragma unroll
				for (int q = 0; q < F8_MP; q++) {
					const int pidx = q < np ? pb - plo + q : 0;
					const int up = __builtin_amdgcn_readfirstlane(s_up[pidx]);
					slot[q] = __builtin_amdgcn_readfirstlane(s_lp[pidx]) - jbeg;
					uo[q][0] = q < np ? a.in[(long)up * 64 + offX0] : 0.0;
					uo[q][1] = q < np ? a.in[(long)up * 64 + offX1] : 0.0;
				}
				if (a.scale) {
#pragma unroll
					for (int e = 0; e < F8_MW; e++)
						if (work[e])
							av[e] *= a.scale[(long)irow * 8 + scr] * a.scale[(long)col[e] * 8 + scc];
				}
#pragma unroll
				for (int e = 0; e < F8_MW; e++) {
					if (!work[e])
						continue;
					double acc = 0.0;
#pragma unroll
					for (int q = 0; q < F8_MP; q++) {
						if (q < np && pb + q >= kb[e] && pb + q < ke[e]) {
							double l0 = lres[0][0], l1 = lres[0][1];
#pragma unroll
							for (int t = 1; t < F8_ML; t++) {
								l0 = slot[q] == t ? lres[t][0] : l0;
								l1 = slot[q] == t ? lres[t][1] : l1;
							}
							acc = RM ? mfma444(uo[q][0], l0, acc) : mfma444(l0, uo[q][0], acc);
							acc = RM ? mfma444(uo[q][1], l1, acc) : mfma444(l1, uo[q][1], acc);
						}
					}
					double res = av[e] - acc;
					if (e < F8_ML && low[e]) {
						const double sa0 = __shfl(res, srcS0, 64), sa1 = __shfl(res, srcS1, 64);
						double prod = RM ? mfma444(dv[e][0], sa0, 0.0) : mfma444(sa0, dv[e][0], 0.0);
						prod = RM ? mfma444(dv[e][1], sa1, prod) : mfma444(sa1, dv[e][1], prod);
						res = prod;
						lres[e][0] = __shfl(res, srcS0, 64);
						lres[e][1] = __shfl(res, srcS1, 64);
					}
					a.out[(long)(jbeg + e) * 64 + offD] = res;
				}
				continue;
			}
		}

		for (int jpos = jbeg; jpos < jend; jpos++) {
			const int bidx = jpos - jlo;
			int col, kb, ke;
			if (bidx < F8_CAPB) {
				col = s_col[bidx];
				kb = s_pp[bidx];
				ke = s_pp[bidx + 1];
			} else {
				col = a.pat.bcolind[jpos];
				kb = a.posptr[jpos];
				ke = a.posptr[jpos + 1];
			}
			col = __builtin_amdgcn_readfirstlane(col);
			kb = __builtin_amdgcn_readfirstlane(kb);
			ke = __builtin_amdgcn_readfirstlane(ke);
			const bool lower = irow > col;
			if (a.skip_fixed && col > irow && ke == kb)
				continue;  // an upper block without pairs: the sweep before has stored its value, a_ij

			double sval = a.avals[(long)jpos * 64 + offD];
			double d0 = 0.0, d1 = 0.0;
			if (lower) {
				d0 = dinv[(long)col * 64 + offX0];
				d1 = dinv[(long)col * 64 + offX1];
			}
			if (a.scale)
				sval *= a.scale[(long)irow * 8 + scr] * a.scale[(long)col * 8 + scc];

			double acc = 0.0;
			for (int kk = kb; kk < ke; kk++) {
				const int pidx = kk - plo;
				int lp, up;
				if (pidx < F8_CAPP) {
					lp = s_lp[pidx];
					up = s_up[pidx];
				} else {
					lp = a.lowerp[kk];
					up = a.upperp[kk];
				}
				const double *const lblk = (a.lrow_fresh ? a.out : a.in) + (long)lp * 64, *const ublk = a.in + (long)up * 64;
				const double *const first = RM ? ublk : lblk, *const second = RM ? lblk : ublk;
				const double l0 = first[offA0], l1 = first[offA1];
				const double u0 = second[offB0], u1 = second[offB1];
				acc = mfma444(l0, u0, acc);
				acc = mfma444(l1, u1, acc);
			}
			double res = sval - acc;

			if (lower) {
				const double sa0 = __shfl(res, srcS0, 64), sa1 = __shfl(res, srcS1, 64);
				double prod = RM ? mfma444(d0, sa0, 0.0) : mfma444(sa0, d0, 0.0);
				prod = RM ? mfma444(d1, sa1, prod) : mfma444(sa1, d1, prod);
				res = prod;
			}

			double *const dst = a.out + (long)jpos * 64 + offD;
			*dst = res;
		}
	}
}

int g_factor8_enabled = -1;

}  // namespace

void set_factor8_enabled(int on)
{
	g_factor8_enabled = on;
}

// returns false when the tuned kernel does not cover the request (caller uses the generic kernel).
// dinv_scratch: nbrows*64 doubles, receives the inverted diagonal blocks of a.in.
bool launch_factor8(const FactorArgs &a, double *dinv_scratch, hipStream_t s)
{
	if (g_factor8_enabled < 0) {
		const char *e = std::getenv("BLASTED_HIP_FACTOR8");
		g_factor8_enabled = (e && std::strcmp(e, "0") == 0) ? 0 : 1;
	}
	if (!g_factor8_enabled || !dinv_scratch || a.pat.bs != 8 || a.pat.nbrows == 0)
		return false;
	launch_invert_diag_blocks(a.pat, a.in, 1, dinv_scratch, 0, s);
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + F8_RCHUNK - 1) / F8_RCHUNK);
	// (tuning "factor8=2": in-place sweeps with the block-by-block loop only, the round-2 kernel)
	const bool fast = a.in == a.out && g_factor8_enabled != 2;
	if (a.pat.rowmajor) {
		if (fast)
			hipLaunchKernelGGL((factor8_kernel<true, true>), dim3(grid), dim3(256), 0, s, a, (const double *)dinv_scratch);
		else
			hipLaunchKernelGGL((factor8_kernel<false, true>), dim3(grid), dim3(256), 0, s, a, (const double *)dinv_scratch);
	} else if (fast)
		hipLaunchKernelGGL((factor8_kernel<true, false>), dim3(grid), dim3(256), 0, s, a, (const double *)dinv_scratch);
	else
		hipLaunchKernelGGL((factor8_kernel<false, false>), dim3(grid), dim3(256), 0, s, a, (const double *)dinv_scratch);
	BHIP_CHECK(hipGetLastError());
	return true;
}

}  // namespace bhip
