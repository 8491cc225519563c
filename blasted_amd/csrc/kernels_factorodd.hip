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
template <int BS, int PROBE = 0>
__global__ __launch_bounds__(256, 8) void factorodd_kernel(const FactorArgs a, const double *__restrict__ dinv)
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

	for (int jpos = jbeg; jpos < jend; jpos++) {
		const int col = a.pat.bcolind[jpos];
		const int kbeg = a.posptr[jpos], kend = a.posptr[jpos + 1];
		if (a.skip_fixed && col > irow && kend == kbeg)
			continue;  // an upper block without pairs: the sweep before has stored its value, a_ij
		fd2_t s;
		s.x = s.y = 0.0;
		if (actB && PROBE < 4)
			s = *reinterpret_cast<const fd2u_t *>(a.avals + (long)jpos * BS2 + boff);
		if (PROBE >= 4)
			s.x = s.y = (double)(col + kend);
		if (a.scale && actB) {
			s.x *= a.scale[(long)irow * BS + rA] * a.scale[(long)col * BS + cA];
			s.y *= a.scale[(long)irow * BS + rB] * a.scale[(long)col * BS + cB];
		}
		for (int k = kbeg; k < (PROBE >= 3 ? kbeg : kend); k++) {
			fd2_t lv, uv;
			lv.x = lv.y = uv.x = uv.y = 0.0;
			if (PROBE == 2) {
				lv = s;
				uv = s;
			} else if (actB) {
				lv = *reinterpret_cast<const fd2u_t *>((a.lrow_fresh ? a.out : a.in) + (long)a.lowerp[k] * BS2 + boff);
				uv = *reinterpret_cast<const fd2u_t *>(a.in + (long)a.upperp[k] * BS2 + boff);
			}
			double pA, pB;
			gemm(lv, uv, pA, pB);
			s.x -= pA;
			s.y -= pB;
		}
		if (irow > col && PROBE < 4) {
			fd2_t dv;
			dv.x = dv.y = 0.0;
			if (actB)
				dv = *reinterpret_cast<const fd2u_t *>(dinv + (long)col * BS2 + boff);
			double pA, pB;
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
	if (!g_factorodd_enabled || !dinv_scratch || (bs != 5 && bs != 7) || a.pat.rowmajor || a.pat.nbrows == 0 ||
	    a.rows)
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
			hipLaunchKernelGGL(factorodd_kernel<5>, dim3(grid), dim3(256), 0, s, a, dinv);
	} else {
		const unsigned grid = (unsigned)((n + 7) / 8);
		hipLaunchKernelGGL(factorodd_kernel<7>, dim3(grid), dim3(256), 0, s, a, dinv);
	}
}

}  // namespace bhip
