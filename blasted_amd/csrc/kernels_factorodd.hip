// kernels_factorodd.hip -- the asynchronous block-ILU(0) factorisation sweep for column-major 5x5 and
// 7x7 blocks (async_block_ilu0_factorize, src/kernels/kernels_ilu0_factorize.hpp:71-98), in the lane
// layout of kernels_sweepodd.hip.  The general kernel gives a padded 8x8 lane tile -- a whole wave -- to
// one block-row, so one load instruction moves a single 200- (392-)byte block; here 16 (32) lanes own a
// block-row, L = (bs*bs+1)/2 of them hold a block as 16 bytes per lane (8-byte aligned), and a wave
// works on 4 (2) rows at once.  The bs x bs x bs products L_ik U_kj and S U_jj^-1 go through two
// wave-private LDS tiles: every lane writes its two entries of both factors and reads the row of the
// left and the column of the right factor its two results need (LDS executes a wave's instructions in
// order, so no barrier).  U_jj^-1 comes from the per-sweep pre-pass over the diagonal blocks
// (invert_blocks_tpb_kernel), as for the general kernel at bs >= 5.  Every entry of the factor is stored
// once per sweep and a row group reads back its own stores in storage order, as the general kernel does.
//
// Round 4: a wave of this kernel is a CHAIN of dependent memory round trips (column and pair-list bounds -> position
// pair -> operand blocks -> product -> inverse -> store) and the sweep's time follows the chain, not the bytes or the
// request count (profiles/r04_factorodd_prefetch_ab.txt; an LDS cache for the row's own lower blocks that removed half
// of the operand requests but cost resident waves was 32-70 % slower: r04_factorodd_ldscache_ab.txt).  So whatever is
// known early is requested early: a block's column and the end of its pair list one block ahead, the position pair of
// step k + 1 before the operand blocks of step k, the inverse a lower block is multiplied with at the block's start.
// Seven waves per SIMD (72 registers) instead of eight: at 64 the requests in flight spill.  Config 4's three-sweep
// build 13.7 -> 12.0 ms, bs = 7 patterns 3.2 -> 2.4 ms.  (Tried beyond that and not kept: the operand blocks of step
// k + 1 before the product of step k -- more registers than it saves round trips -- and the first position pair of the
// next block, which requests pairs for blocks that have none.)
#include "ctx.hpp"
#include "lanes.hpp"

#include <cstdlib>
#include <cstring>

namespace bhip {

namespace {

typedef double fd2_t __attribute__((ext_vector_type(2)));
typedef fd2_t fd2u_t __attribute__((aligned(8)));

// PROBE (measurements only, tuning "factorprobe=1..5", WRONG results): 1 = the block products skip the LDS tiles (each
// lane multiplies its own entries), 2 = the operand blocks of the pairs are not loaded (the matrix block stands in),
// 3 = no pairs at all and no store, 4 = no pairs, no matrix block, no inverse (indices + store), 5 = indices only
// RM: ROW-major blocks -- the column-major image of the transposed blocks, so every product is taken the other way
// round (U'_kj L'_ik, inverse(U'_jj) S'; the inverse of the transposed block is the transposed inverse, which is what the
// pre-pass leaves in row-major storage) and the scaling factors swap their roles.
template <int BS, int PROBE = 0, bool RM = false>
__global__ __launch_bounds__(256, 7) void factorodd_kernel(const FactorArgs a, const double *__restrict__ dinv)
{
	static_assert(BS == 5 || BS == 7, "odd block sizes 5, 7");
	constexpr int BS2 = BS * BS, L = (BS2 + 1) / 2;
	constexpr int G = BS == 5 ? 16 : 32, RPW = 64 / G, RPB = 4 * RPW;

	__shared__ double s_l[4][RPW][BS2 + 1];
	__shared__ double s_u[4][RPW][BS2 + 1];

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / G, t = lane % G;
	const bool actA = t < L - 1, actB = t < L;
	const int eA = actA ? 2 * t : BS2 - 2, eB = actA ? 2 * t + 1 : BS2 - 1;
	const long boff = actA ? 2 * t : BS2 - 2;  // in doubles
	const int rA = eA % BS, cA = eA / BS, rB = eB % BS, cB = eB / BS;
	double *const tl = &s_l[wave][g][0];
	double *const tu = &s_u[wave][g][0];

	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const long rowlin = (long)chunk * RPB + wave * RPW + g;
	const bool rowok = rowlin < (a.rows ? a.nrows : a.pat.nbrows);
	const int irow = rowok ? (a.rows ? a.rows[rowlin] : (int)rowlin) : 0;
	int jbeg = 0, jend = 0;
	if (rowok) {
		jbeg = a.pat.browptr[irow];
		jend = a.pat.browptr[irow + 1];
	}

	// out(r,c) = sum_m x(r,m) y(m,c) for this lane's two entries; x, y given as this lane's entry pairs
	auto gemm = [&](const fd2_t x, const fd2_t y, double &oA, double &oB) {
		if (PROBE == 1) {
			oA = x.x * y.x;
			oB = x.y * y.y;
			return;
		}
		if (actA) {
			tl[eA] = x.x;
			tu[eA] = y.x;
		}
		if (actB) {
			tl[eB] = x.y;
			tu[eB] = y.y;
		}
		__builtin_amdgcn_wave_barrier();
		double sA = 0.0, sB = 0.0;
#pragma unroll
		for (int m = 0; m < BS; m++) {
			sA += tl[rA + BS * m] * tu[m + BS * cA];
			sB += tl[rB + BS * m] * tu[m + BS * cB];
		}
		__builtin_amdgcn_wave_barrier();
		oA = sA;
		oB = sB;
	};

	int coln = 0, kbegn = 0, kendn = 0;  // the next block's column and pair list, requested a block ahead
	if (jbeg < jend) {
		coln = a.pat.bcolind[jbeg];
		kbegn = a.posptr[jbeg];
		kendn = a.posptr[jbeg + 1];
	}
	for (int jpos = jbeg; jpos < jend; jpos++) {
		const int col = coln, kbeg = kbegn, kend = kendn;
		if (jpos + 1 < jend) {
			coln = a.pat.bcolind[jpos + 1];
			kbegn = kend;
			kendn = a.posptr[jpos + 2];
		}
		if (a.skip_fixed && col > irow && kend == kbeg)
			continue;  // an upper block without pairs: the sweep before has stored its value, a_ij
		fd2_t s;
		s.x = s.y = 0.0;
		if (actB && PROBE < 4)
			s = *reinterpret_cast<const fd2u_t *>(a.avals + (long)jpos * BS2 + boff);
		if (PROBE >= 4)
			s.x = s.y = (double)(col + kend);
		if (a.scale && actB) {
			// (row-major: this lane's entry (rA, cA) of the column-major image is entry (cA, rA) of the block)
			s.x *= a.scale[(long)irow * BS + (RM ? cA : rA)] * a.scale[(long)col * BS + (RM ? rA : cA)];
			s.y *= a.scale[(long)irow * BS + (RM ? cB : rB)] * a.scale[(long)col * BS + (RM ? rB : cB)];
		}
		fd2_t dv;
		dv.x = dv.y = 0.0;
		if (irow > col && PROBE < 4 && actB)  // the inverse a lower block is multiplied with at the end: requested now
			dv = *reinterpret_cast<const fd2u_t *>(dinv + (long)col * BS2 + boff);
		const int kstop = PROBE >= 3 ? kbeg : kend;
		int lnext = 0, unext = 0;
		if (kbeg < kstop) {
			lnext = a.lowerp[kbeg];
			unext = a.upperp[kbeg];
		}
		for (int k = kbeg; k < kstop; k++) {
			const int lpos = lnext, upos = unext;
			if (k + 1 < kstop) {  // the next step's position pair travels with this step's operand blocks
				lnext = a.lowerp[k + 1];
				unext = a.upperp[k + 1];
			}
			fd2_t lv, uv;
			lv.x = lv.y = uv.x = uv.y = 0.0;
			if (PROBE == 2) {
				lv = s;
				uv = s;
			} else if (actB) {
				lv = *reinterpret_cast<const fd2u_t *>((a.lrow_fresh ? a.out : a.in) + (long)lpos * BS2 + boff);
				uv = *reinterpret_cast<const fd2u_t *>(a.in + (long)upos * BS2 + boff);
			}
			double pA, pB;
			if (RM)
				gemm(uv, lv, pA, pB);
			else
				gemm(lv, uv, pA, pB);
			s.x -= pA;
			s.y -= pB;
		}
		if (irow > col && PROBE < 4) {
			double pA, pB;
			if (RM)
				gemm(dv, s, pA, pB);
			else
				gemm(s, dv, pA, pB);
			s.x = pA;
			s.y = pB;
		}
		double *const dst = a.out + (long)jpos * BS2 + boff;
		if (PROBE == 3 || PROBE == 5) {
			if (s.x == 1.2345e300 && actA)  // never: keeps the work alive without the store
				*dst = s.y;
		} else if (actA)
			*reinterpret_cast<fd2u_t *>(dst) = s;
		else if (actB)
			dst[1] = s.y;  // last lane: only the block's last entry is its own
	}
}

int g_factorodd_enabled = -1;
int g_factor_probe = 0;

}  // namespace

void set_factor_probe(int v)
{
	g_factor_probe = v;
}

static void launch_factorodd_rows(const FactorArgs &a, const double *dinv, hipStream_t s);

void set_factorodd_enabled(int on)
{
	g_factorodd_enabled = on;
}

// returns false when the tuned kernel does not cover the request (caller uses the general kernel).
// dinv_scratch: nbrows*bs*bs doubles, receives the inverted diagonal blocks of a.in.
bool launch_factorodd(const FactorArgs &a, double *dinv_scratch, hipStream_t s)
{
	if (g_factorodd_enabled < 0) {
		const char *e = std::getenv("BLASTED_HIP_FACTORODD");
		g_factorodd_enabled = (e && std::strcmp(e, "0") == 0) ? 0 : 1;
	}
	const int bs = a.pat.bs;
	if (!g_factorodd_enabled || !dinv_scratch || (bs != 5 && bs != 7) || a.pat.nbrows == 0 || a.rows)
		return false;
	launch_invert_diag_blocks(a.pat, a.in, 1, dinv_scratch, 0, s);
	launch_factorodd_rows(a, dinv_scratch, s);
	BHIP_CHECK(hipGetLastError());
	return true;
}

// the kernel alone over all rows or over a.rows; dinv must hold the inverses the listed rows need
static void launch_factorodd_rows(const FactorArgs &a, const double *dinv, hipStream_t s)
{
	const long n = a.rows ? a.nrows : a.pat.nbrows;
	if (n <= 0)
		return;
	if (a.pat.bs == 5) {
		const unsigned grid = (unsigned)((n + 15) / 16);
#ifdef BHIP_PROBES
		if (g_factor_probe == 1)
			hipLaunchKernelGGL((factorodd_kernel<5, 1>), dim3(grid), dim3(256), 0, s, a, dinv);
		else if (g_factor_probe == 2)
			hipLaunchKernelGGL((factorodd_kernel<5, 2>), dim3(grid), dim3(256), 0, s, a, dinv);
		else if (g_factor_probe == 3)
			hipLaunchKernelGGL((factorodd_kernel<5, 3>), dim3(grid), dim3(256), 0, s, a, dinv);
		else if (g_factor_probe == 4)
			hipLaunchKernelGGL((factorodd_kernel<5, 4>), dim3(grid), dim3(256), 0, s, a, dinv);
		else if (g_factor_probe == 5)
			hipLaunchKernelGGL((factorodd_kernel<5, 5>), dim3(grid), dim3(256), 0, s, a, dinv);
		else
#endif
		if (a.pat.rowmajor)
			hipLaunchKernelGGL((factorodd_kernel<5, 0, true>), dim3(grid), dim3(256), 0, s, a, dinv);
		else
			hipLaunchKernelGGL(factorodd_kernel<5>, dim3(grid), dim3(256), 0, s, a, dinv);
	} else {
		const unsigned grid = (unsigned)((n + 7) / 8);
		if (a.pat.rowmajor)
			hipLaunchKernelGGL((factorodd_kernel<7, 0, true>), dim3(grid), dim3(256), 0, s, a, dinv);
		else
			hipLaunchKernelGGL(factorodd_kernel<7>, dim3(grid), dim3(256), 0, s, a, dinv);
	}
}

}  // namespace bhip
