// kernels_factor.hip -- asynchronous (block-)ILU(0) factorisation sweeps and the small dense block
// work around them.
//
//   async_block_ilu0_factorize   kernels_ilu0_factorize.hpp:71-98    factor_sweep_kernel (bs>1)
//   async_ilu0_factorize_kernel  kernels_ilu0_factorize.hpp:19-53    factor_sweep_kernel (bs==1)
//   diagonal-block inversion     async_blockilu_factor.cpp:143-146   invert_blocks_kernel
//   BJacobi/Jacobi compute       solverops_jacobi.cpp:43-45,141-147  invert_blocks_kernel
//   fact_init_{original,sgs}     async_blockilu_factor.cpp:63-93,206-254, async_ilu_factor.cpp:109-151
//   getScalingVector             rawsrmatrixutils.cpp:343-350
//   block/scalar_ilu0_nonlinear_res  async_blockilu_factor.cpp:256-297, async_ilu_factor.cpp:179-217
//   diagonal_dominance           matrix_properties.cpp:10-77
//
// Mapping: SUB = BSP*BSP lanes (BSP = bs rounded up to a power of two) own one block-row and walk
// its stored blocks; lane (r,c) holds entry (r,c) of the current block.  The bs x bs x bs products
// L_ik U_kj and S U_jj^-1 and the on-the-fly inverse of U_jj are done with wavefront shuffles between
// the lanes of the group: no LDS allocation, no partial result ever leaves registers, and every entry
// of iluvals is stored exactly once per sweep (kernels_ilu0_factorize.hpp:34-40).
#include "ctx.hpp"
#include "lanes.hpp"

namespace bhip {

template <int BS>
struct FGeo {
	static constexpr int BSP = BS <= 1 ? 1 : (BS <= 2 ? 2 : (BS <= 4 ? 4 : 8));
	static constexpr int SUB = BSP * BSP;
	static constexpr int RPW = 64 / SUB;
	static constexpr int RPB = 4 * RPW;
};

// Lane (r,c) of a group holds a(r,c) of a BS x BS matrix (lanes with r>=BS or c>=BS hold anything).
// Returns inverse(r,c) in lane (r,c).  n<=4: adjugate / determinant (the closed form Eigen's
// fixed-size inverse() uses); larger: Gauss-Jordan with partial pivoting on the identity-padded
// BSP x BSP matrix.
template <int BS, int BSP>
__device__ __forceinline__ double group_inverse(const double a, const int gbase, const int r, const int c)
{
	if (BS == 1)
		return 1.0 / a;
	if (BS == 2) {
		const double a00 = __shfl(a, gbase + 0, 64), a10 = __shfl(a, gbase + 1, 64);
		const double a01 = __shfl(a, gbase + BSP, 64), a11 = __shfl(a, gbase + BSP + 1, 64);
		const double invdet = 1.0 / (a00 * a11 - a01 * a10);
		const double adj = (r == 0 && c == 0) ? a11 : (r == 1 && c == 1) ? a00 : (r == 0 ? -a01 : -a10);
		return adj * invdet;
	}
	if (BS == 3 || BS == 4) {
		// this lane computes cofactor C(c,r): delete row c and column r
		const int rr = r < BS ? r : 0, cc = c < BS ? c : 0;
		double M[3][3];
#pragma unroll
		for (int x = 0; x < BS - 1; x++)
#pragma unroll
			for (int y = 0; y < BS - 1; y++) {
				const int ri = x + (x >= cc ? 1 : 0);  // x-th row of {0..BS-1} \ {cc}
				const int ci = y + (y >= rr ? 1 : 0);  // y-th column of {0..BS-1} \ {rr}
				M[x][y] = __shfl(a, gbase + ri + ci * BSP, 64);
			}
		double minor;
		if (BS == 3)
			minor = M[0][0] * M[1][1] - M[0][1] * M[1][0];
		else
			minor = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) -
			        M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
			        M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
		const double cof = ((rr + cc) & 1) ? -minor : minor;  // = C(cc, rr)
		// det = sum_q a(0,q) C(0,q); C(0,q) lives in lane (r=q, c=0)
		double det = 0.0;
#pragma unroll
		for (int q = 0; q < BS; q++)
			det += __shfl(a, gbase + 0 + q * BSP, 64) * __shfl(cof, gbase + q, 64);
		return cof * (1.0 / det);
	}
	// Gauss-Jordan, partial pivoting
	double m = (r < BS && c < BS) ? a : (r == c ? 1.0 : 0.0);
	double inv = (r == c) ? 1.0 : 0.0;
#pragma unroll 1
	for (int k = 0; k < BS; k++) {
		int p = k;
		double best = fabs(__shfl(m, gbase + k + k * BSP, 64));
		for (int q = k + 1; q < BS; q++) {
			const double v = fabs(__shfl(m, gbase + q + k * BSP, 64));
			if (v > best) {
				best = v;
				p = q;
			}
		}
		// swap rows k and p
		const int srcrow = (r == k) ? p : (r == p ? k : r);
		m = __shfl(m, gbase + srcrow + c * BSP, 64);
		inv = __shfl(inv, gbase + srcrow + c * BSP, 64);
		const double piv = 1.0 / __shfl(m, gbase + k + k * BSP, 64);
		const double mk = __shfl(m, gbase + k + c * BSP, 64) * piv;     // row k, my column
		const double ik = __shfl(inv, gbase + k + c * BSP, 64) * piv;
		const double f = __shfl(m, gbase + r + k * BSP, 64);            // my row, column k
		if (r == k) {
			m = mk;
			inv = ik;
		} else {
			m -= f * mk;
			inv -= f * ik;
		}
	}
	return inv;
}

// out(r,c) = sum_m x(r,m) y(m,c) with x, y distributed one entry per lane
template <int BS, int BSP>
__device__ __forceinline__ double group_gemm(const double x, const double y, const int gbase,
                                             const int r, const int c)
{
	double s = 0.0;
#pragma unroll
	for (int m = 0; m < BS; m++)
		s += __shfl(x, gbase + r + m * BSP, 64) * __shfl(y, gbase + m + c * BSP, 64);
	return s;
}

template <int BS, bool RM, bool RESID>
__global__ __launch_bounds__(256) void factor_sweep_kernel(const FactorArgs a, double *resid_partial)
{
	using Ge = FGeo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, BS2 = BS * BS;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / SUB, u = lane % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(SUB - 1);

	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const long rowlin = (long)chunk * Ge::RPB + wave * Ge::RPW + g;
	const bool rowok = rowlin < (a.rows ? a.nrows : a.pat.nbrows);
	const int irow = rowok ? (a.rows ? a.rows[rowlin] : (int)rowlin) : 0;
	int jbeg = 0, jend = 0;
	if (rowok) {
		jbeg = a.pat.browptr[irow];
		jend = a.pat.browptr[irow + 1];
	}
	double resid = 0.0;

	for (int jpos = jbeg; jpos < jend; jpos++) {
		const int col = a.pat.bcolind[jpos];
		if (!RESID && a.skip_fixed && col > irow && a.posptr[jpos + 1] == a.posptr[jpos])
			continue;  // an upper entry without pairs: the sweep before has stored its value, a_ij
		double s = active ? a.avals[(long)jpos * BS2 + e] : 0.0;
		if (a.scale && active) {
			if (BS == 1) {
				s *= a.scale[irow];
				s *= a.scale[col];
			} else
				s *= a.scale[(long)irow * BS + r] * a.scale[(long)col * BS + c];
		}
		const int kbeg = a.posptr[jpos], kend = a.posptr[jpos + 1];
		for (int k = kbeg; k < kend; k++) {
			const double lv = active ? (a.lrow_fresh ? a.out : a.in)[(long)a.lowerp[k] * BS2 + e] : 0.0;
			const double uv = active ? a.in[(long)a.upperp[k] * BS2 + e] : 0.0;
			if (BS == 1)
				s -= lv * uv;
			else
				s -= group_gemm<BS, BSP>(lv, uv, gbase, r, c);
		}
		if (RESID) {
			// A - LU on the pattern, async_blockilu_factor.cpp:278-288
			const double cur = active ? a.in[(long)jpos * BS2 + e] : 0.0;
			if (irow > col) {
				const double dv = active ? a.in[(long)a.pat.diagind[col] * BS2 + e] : 0.0;
				s -= (BS == 1) ? cur * dv : group_gemm<BS, BSP>(cur, dv, gbase, r, c);
			} else
				s -= cur;
			if (active)
				resid += fabs(s);
		} else if (irow > col) {
			double res;
			if (BS >= 5 && a.dinv_scratch) {
				// bs >= 5: the diagonal blocks of the sweep's input iterate were inverted by the per-sweep
				// pre-pass (launch_factor_sweep); kernels_factor8.hip explains why this is the same map
				const double inv = active ? a.dinv_scratch[(long)col * BS2 + e] : 0.0;
				res = group_gemm<BS, BSP>(s, inv, gbase, r, c);
			} else {
				const double dv = active ? a.in[(long)a.pat.diagind[col] * BS2 + e] : 0.0;
				if (BS == 1)
					res = s / dv;
				else {
					const double inv = a.diag_inverted ? dv : group_inverse<BS, BSP>(dv, gbase, r, c);
					res = group_gemm<BS, BSP>(s, inv, gbase, r, c);
				}
			}
			if (active)
				a.out[(long)jpos * BS2 + e] = res;
		} else if (BS > 1 && a.diag_inverted && irow == col) {
			// exact factorisation: this diagonal block is final -- store its inverse, the form every later
			// reader (lower blocks of later rows, the triangular solves) wants
			const double inv = group_inverse<BS, BSP>(s, gbase, r, c);
			if (active)
				a.out[(long)jpos * BS2 + e] = inv;
		} else if (active) {
			a.out[(long)jpos * BS2 + e] = s;
		}
	}

	if (RESID) {
		// one partial per workgroup, summed on the host side in fixed order
		__shared__ double wsum[4];
		for (int off = 32; off > 0; off >>= 1)
			resid += __shfl_xor(resid, off, 64);
		if (lane == 0)
			wsum[wave] = resid;
		__syncthreads();
		if (threadIdx.x == 0)
			resid_partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
	}
}

// dst block = inverse(src block); blocks addressed by row or through diagind
template <int BS, bool RM>
__global__ __launch_bounds__(256) void invert_blocks_kernel(const Pattern pat, const double *src,
                                                            const int src_by_diag, double *dst,
                                                            const int dst_by_diag)
{
	using Ge = FGeo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, BS2 = BS * BS;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / SUB, u = lane % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(SUB - 1);
	const long rowlin = (long)blockIdx.x * Ge::RPB + wave * Ge::RPW + g;
	const bool rowok = rowlin < pat.nbrows;
	const int i = rowok ? (int)rowlin : 0;
	const long dgpos = rowok ? pat.diagind[i] : 0;
	const long sblk = src_by_diag ? dgpos : i, dblk = dst_by_diag ? dgpos : i;
	const double av = (rowok && active) ? src[sblk * BS2 + e] : ((r == c) ? 1.0 : 0.0);
	const double inv = group_inverse<BS, BSP>(av, gbase, r, c);
	if (rowok && active)
		dst[dblk * BS2 + e] = inv;
}

// Same operation for bs >= 5 (Gauss-Jordan with partial pivoting, as Eigen's PartialPivLU-based
// inverse() does above n = 4), organised the other way round: ONE THREAD inverts one block entirely in
// registers -- no cross-lane traffic at all -- and the 64 blocks of a wave travel between HBM and the
// threads through an LDS transpose, so that global loads and stores stay coalesced (a wave instruction
// moves one whole block).  All register-array indices are compile-time constants (row swaps and the
// final column un-permutation are conditional moves).
// (Round 2, bs = 8, where this pass is a quarter of a factorisation sweep -- 0.74 ms per 10^6 blocks, 1 GB at
// 1.4 TB/s, four waves per CU because of its 215 VGPRs and 33 KB of LDS: two wave-cooperative replacements were
// built, verified and removed.  A wave inverting one block with lane (r,c) = entry (r,c), Gauss-Jordan without
// row moves, three ds_bpermute exchanges and a DPP pivot search per step, ~50 VGPRs, coalesced 512-byte loads,
// as a pre-pass: 1.0 ms.  The same routine inside the factor kernel, run by the wave that has just finished a
// row's diagonal block, so that no pre-pass is needed after the first sweep: the sweep kernel went from 2.07 to
// 2.43 ms.  Eight dependent steps of cross-lane exchanges are a longer critical path than 64 independent
// register-resident eliminations; the thread-per-block form stays.)
template <int BS, bool RM>
__global__ __launch_bounds__(64) void invert_blocks_tpb_kernel(const Pattern pat, const double *src,
                                                               const int src_by_diag, double *dst,
                                                               const int dst_by_diag, const int *rows,
                                                               const int nlist)
{
	constexpr int BS2 = BS * BS, LD = BS2 + 1;  // +1: conflict-free strided LDS reads
	// bs 7 / 8: the transpose goes through LDS in two halves of 32 blocks (16.6 KB instead of 33 KB per 64
	// threads: the 160-215 VGPRs then bound the occupancy -- 8 to 12 waves per CU -- instead of LDS at 4 to 6)
	constexpr int NH = BS >= 7 ? 2 : 1, HB = 64 / NH;
	__shared__ double tile[HB * LD];
	const int t = threadIdx.x;
	const long total = rows ? nlist : pat.nbrows;  // optional row list (level-scheduled factorisation)
	const long row0 = (long)blockIdx.x * 64;
	const bool rowok = row0 + t < total;
	const long myrow = rowok ? (rows ? rows[row0 + t] : row0 + t) : 0;
	const long dgpos = rowok ? pat.diagind[myrow] : 0;
	const long sblk = src_by_diag ? dgpos : myrow, dblk = dst_by_diag ? dgpos : myrow;
	const int nrows = (total - row0) < 64 ? (int)(total - row0) : 64;

	double m[BS][BS];
#pragma unroll
	for (int r = 0; r < BS; r++)
#pragma unroll
		for (int c = 0; c < BS; c++)
			m[r][c] = (r == c) ? 1.0 : 0.0;
	// HBM -> LDS, one block per wave instruction; every thread picks its own block up from the tile
#pragma unroll
	for (int h = 0; h < NH; h++) {
		const int iend = nrows < (h + 1) * HB ? nrows : (h + 1) * HB;
		for (int i = h * HB; i < iend; i++) {
			const long blk = __shfl(sblk, i, 64);
			if (t < BS2)
				tile[(i - h * HB) * LD + t] = src[blk * BS2 + t];
		}
		__syncthreads();
		if (rowok && t / HB == h) {
#pragma unroll
			for (int r = 0; r < BS; r++)
#pragma unroll
				for (int c = 0; c < BS; c++)
					m[r][c] = tile[(t - h * HB) * LD + (RM ? r * BS + c : c * BS + r)];
		}
		if (NH > 1)
			__syncthreads();
	}

	// in-place Gauss-Jordan with partial pivoting; piv[p] = row swapped with p at step p
	int piv[BS];
#pragma unroll
	for (int p = 0; p < BS; p++) {
		int pr = p;
		double best = fabs(m[p][p]);
#pragma unroll
		for (int r = p + 1; r < BS; r++) {
			const double v = fabs(m[r][p]);
			if (v > best) {
				best = v;
				pr = r;
			}
		}
		piv[p] = pr;
#pragma unroll
		for (int r = p + 1; r < BS; r++) {
			const bool sw = (pr == r);
#pragma unroll
			for (int c = 0; c < BS; c++) {
				const double a = m[p][c], b = m[r][c];
				m[p][c] = sw ? b : a;
				m[r][c] = sw ? a : b;
			}
		}
		const double pinv = 1.0 / m[p][p];
		m[p][p] = 1.0;
#pragma unroll
		for (int c = 0; c < BS; c++)
			m[p][c] *= pinv;
#pragma unroll
		for (int r = 0; r < BS; r++) {
			if (r == p)
				continue;
			const double f = m[r][p];
			m[r][p] = 0.0;
#pragma unroll
			for (int c = 0; c < BS; c++)
				m[r][c] -= f * m[p][c];
		}
	}
	// undo the row swaps as column swaps, last first
#pragma unroll
	for (int p = BS - 1; p >= 0; p--) {
#pragma unroll
		for (int q = p + 1; q < BS; q++) {
			const bool sw = (piv[p] == q);
#pragma unroll
			for (int r = 0; r < BS; r++) {
				const double a = m[r][p], b = m[r][q];
				m[r][p] = sw ? b : a;
				m[r][q] = sw ? a : b;
			}
		}
	}

	__syncthreads();
#pragma unroll
	for (int h = 0; h < NH; h++) {
		if (t / HB == h) {
#pragma unroll
			for (int r = 0; r < BS; r++)
#pragma unroll
				for (int c = 0; c < BS; c++)
					tile[(t - h * HB) * LD + (RM ? r * BS + c : c * BS + r)] = m[r][c];
		}
		__syncthreads();
		const int iend = nrows < (h + 1) * HB ? nrows : (h + 1) * HB;
		for (int i = h * HB; i < iend; i++) {
			const long blk = __shfl(dblk, i, 64);
			if (t < BS2)
				dst[blk * BS2 + t] = tile[(i - h * HB) * LD + t];
		}
		if (NH > 1)
			__syncthreads();
	}
}

// Round 3: the same Gauss-Jordan for 5 <= bs <= 8 with EIGHT LANES per block -- lane r of an aligned 8-lane group holds
// row r (bs doubles), eight blocks per wave.  Between round 2's two forms: the thread-per-block kernel above needs no
// cross-lane traffic but 160-215 registers and an LDS transpose (four to twelve waves per CU; 0.45 ms per 10^6 8x8
// blocks, 0.53 ms per 2*10^6 5x5 blocks: 1.5-2.3 TB/s), the 64-lanes-per-block form put one block's eight dependent
// steps on a whole wave (1.0 ms).  Here a step is: pivot search = a three-step (value, index) butterfly inside the
// group (DPP), the pivot row broadcast from its lane (two DPP moves per word), one FMA per held entry; rows are
// exchanged by a permute only when a pivot is off the diagonal (a wave-uniform test).  42-64 registers (8 waves per
// SIMD), loads and stores of bs contiguous doubles per group and column.  Measured (tools/invert_ab.py,
// profiles/r03_invert_ab.txt): 10^6 8x8 blocks 0.454 -> 0.287 ms, 2*10^6 5x5 blocks 0.532 -> 0.271 ms, same inverses
// to the last bit of the comparison with torch.linalg.inv (4e-16); three build sweeps of config 5 / config 4, whose
// every sweep starts with this pass, 8.70 -> 8.07 ms / 16.78 -> 15.68 ms.  The operations on every entry are the thread-per-block
// kernel's, in its order (same pivot choice: the first row of largest magnitude; NaN never wins a pivot search).
template <int P>
__device__ __forceinline__ double group8_bcast(const double v, const int lane)
{
	constexpr int Q = P & 3, CTRL = Q | (Q << 2) | (Q << 4) | (Q << 6);
	const double t = dpp_mov<CTRL>(v);                                  // lane Q of the own quad
	const double o = (P >> 2) ? dpp_mov<0x104>(t) : dpp_mov<0x114>(t);  // row_shl:4 / row_shr:4: the other quad's
	return (((lane >> 2) & 1) == (P >> 2)) ? t : o;
}

// (p is a compile-time constant wherever this is called -- inside a fully unrolled loop -- so the switch folds)
__device__ __forceinline__ double group8_bcast_p(const double v, const int p, const int lane)
{
	switch (p) {
	case 0: return group8_bcast<0>(v, lane);
	case 1: return group8_bcast<1>(v, lane);
	case 2: return group8_bcast<2>(v, lane);
	case 3: return group8_bcast<3>(v, lane);
	case 4: return group8_bcast<4>(v, lane);
	case 5: return group8_bcast<5>(v, lane);
	case 6: return group8_bcast<6>(v, lane);
	default: return group8_bcast<7>(v, lane);
	}
}

__device__ __forceinline__ int dpp_xor_int(const int v, const int which, const int lane)
{
	if (which == 1)
		return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
	if (which == 2)
		return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
	const int up = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0xf, false);
	const int dn = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
	return (lane & 4) ? dn : up;
}

__device__ __forceinline__ double dpp_xor_double(const double v, const int which, const int lane)
{
	if (which == 1)
		return dpp_mov<0xB1>(v);
	if (which == 2)
		return dpp_mov<0x4E>(v);
	return xor4_value(v, lane);
}

template <int BS, bool RM>
__global__ __launch_bounds__(256) void invert_blocks_rowlane_kernel(const Pattern pat, const double *src,
                                                                    const int src_by_diag, double *dst,
                                                                    const int dst_by_diag)
{
	static_assert(BS >= 5 && BS <= 8, "eight lanes per block: 5 <= bs <= 8");
	constexpr int BS2 = BS * BS;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane >> 3, r = lane & 7;
	const long lin = ((long)blockIdx.x * 4 + wave) * 8 + g;
	const bool ok = lin < pat.nbrows;
	const int i = ok ? (int)lin : 0;
	const long dgpos = ok ? pat.diagind[i] : 0;
	const long sblk = src_by_diag ? dgpos : i, dblk = dst_by_diag ? dgpos : i;
	const bool act = ok && r < BS;

	double m[BS];
#pragma unroll
	for (int c = 0; c < BS; c++)
		m[c] = act ? src[sblk * BS2 + (RM ? r * BS + c : c * BS + r)] : ((r == c) ? 1.0 : 0.0);

	int piv[BS];
#pragma unroll
	for (int p = 0; p < BS; p++) {
		// the first row r >= p of largest |m[r][p]| (a NaN is taken only where the thread-per-block kernel takes it: at p)
		double v = (r >= p && r < BS) ? fabs(m[p]) : -1.0;
		if (v != v)
			v = (r == p) ? __builtin_huge_val() : -1.0;
		int idx = r;
#pragma unroll
		for (int w = 1; w <= 4; w <<= 1) {
			const double ov = dpp_xor_double(v, w, lane);
			const int oi = dpp_xor_int(idx, w, lane);
			const bool take = (ov > v) || (ov == v && oi < idx);
			v = take ? ov : v;
			idx = take ? oi : idx;
		}
		const int pr = idx;
		piv[p] = pr;
		if (__builtin_amdgcn_ballot_w64(pr != p) != 0ull) {
			const int partner = (r == p) ? pr : ((r == pr) ? p : r);
			const int srcl = (lane & ~7) | partner;
#pragma unroll
			for (int c = 0; c < BS; c++)
				m[c] = __shfl(m[c], srcl, 64);
		}
		if (r == p) {
			const double pinv = 1.0 / m[p];
			m[p] = 1.0;
#pragma unroll
			for (int c = 0; c < BS; c++)
				m[c] *= pinv;
		}
		double prow[BS];
#pragma unroll
		for (int c = 0; c < BS; c++)
			prow[c] = group8_bcast_p(m[c], p, lane);
		if (r != p) {
			const double f = m[p];
			m[p] = 0.0;
#pragma unroll
			for (int c = 0; c < BS; c++)
				m[c] -= f * prow[c];
		}
	}
	// undo the row swaps as column swaps, last first
#pragma unroll
	for (int p = BS - 1; p >= 0; p--) {
#pragma unroll
		for (int q = p + 1; q < BS; q++) {
			const bool sw = (piv[p] == q);
			const double a = m[p], b = m[q];
			m[p] = sw ? b : a;
			m[q] = sw ? a : b;
		}
	}
	if (act) {
#pragma unroll
		for (int c = 0; c < BS; c++)
			dst[dblk * BS2 + (RM ? r * BS + c : c * BS + r)] = m[c];
	}
}

int g_invert_rowlane = 1;  // tuning "invertrow=0|1": the eight-lanes-per-block inversion for 5 <= bs <= 8

// INIT_F_ORIGINAL with scaling / INIT_F_SGS first pass: ilu = scaled A
template <int BS, bool RM>
__global__ void scaled_copy_kernel(const Pattern pat, const double *avals, const double *scale,
                                   double *ilu)
{
	constexpr int BS2 = BS * BS;
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long nv = (long)pat.nnzb * BS2;
	if (idx >= nv)
		return;
	const long jpos = idx / BS2;
	const int e = (int)(idx % BS2);
	const int r = RM ? e / BS : e % BS, c = RM ? e % BS : e / BS;
	// row of jpos: binary search in browptr
	int lo = 0, hi = pat.nbrows;
	while (hi - lo > 1) {
		const int mid = (lo + hi) >> 1;
		if (pat.browptr[mid] <= jpos)
			lo = mid;
		else
			hi = mid;
	}
	const int irow = lo, col = pat.bcolind[jpos];
	double v = avals[idx];
	if (scale) {
		if (BS == 1) {  // (a s_i) s_j
			v *= scale[irow];
			v *= scale[col];
		} else
			v *= scale[(long)irow * BS + r] * scale[(long)col * BS + c];
	}
	ilu[idx] = v;
}

// INIT_F_SGS second pass: strictly lower blocks right-multiplied by D_col^-1
// (dinv: inverses of the (scaled) diagonal blocks, indexed by block-row)
template <int BS, bool RM>
__global__ __launch_bounds__(256) void sgs_init_lower_kernel(const Pattern pat, const double *dinv,
                                                             double *ilu)
{
	using Ge = FGeo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, BS2 = BS * BS;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / SUB, u = lane % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(SUB - 1);
	const long rowlin = (long)blockIdx.x * Ge::RPB + wave * Ge::RPW + g;
	const bool rowok = rowlin < pat.nbrows;
	const int i = rowok ? (int)rowlin : 0;
	int jbeg = 0, jend = 0;
	if (rowok) {
		jbeg = pat.browptr[i];
		jend = pat.diagind[i];
	}
	for (int j = jbeg; j < jend; j++) {
		const int col = pat.bcolind[j];
		const double lv = active ? ilu[(long)j * BS2 + e] : 0.0;
		const double dv = active ? dinv[(long)col * BS2 + e] : 0.0;
		const double res = (BS == 1) ? lv * dv : group_gemm<BS, BSP>(lv, dv, gbase, r, c);
		if (active)
			ilu[(long)j * BS2 + e] = res;
	}
}

__global__ void scaling_vector_kernel(const Pattern pat, const double *vals, double *scale)
{
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const int bs = pat.bs;
	if (idx >= (long)pat.nbrows * bs)
		return;
	const int i = (int)(idx / bs), j = (int)(idx % bs);
	scale[idx] = 1.0 / sqrt(vals[(long)pat.diagind[i] * bs * bs + j * bs + j]);
}

__global__ void mul_inplace_kernel(double *z, const double *s, long n)
{
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx < n)
		z[idx] *= s[idx];
}

// per-row diagonal dominance terms (matrix_properties.cpp:19-66): writes 4 doubles per workgroup
template <int BS, bool RM>
__global__ __launch_bounds__(256) void diag_dominance_kernel(const Pattern pat, const double *fv,
                                                             double *partial)
{
	constexpr int BS2 = BS * BS;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	double lavg = 0, lmin = 1e30, uavg = 0, umin = 1e30;
	if (i < pat.nbrows) {
		const int dg = pat.diagind[i];
		for (int rr = 0; rr < BS; rr++) {
			double du = 0, dl = 0;
			for (int cc = 0; cc < BS; cc++)
				if (cc != rr)
					du += fabs(fv[(long)dg * BS2 + (RM ? rr * BS + cc : cc * BS + rr)]);
			for (int jj = dg + 1; jj < pat.browptr[i + 1]; jj++)
				for (int cc = 0; cc < BS; cc++)
					du += fabs(fv[(long)jj * BS2 + (RM ? rr * BS + cc : cc * BS + rr)]);
			for (int jj = pat.browptr[i]; jj < dg; jj++)
				for (int cc = 0; cc < BS; cc++)
					dl += fabs(fv[(long)jj * BS2 + (RM ? rr * BS + cc : cc * BS + rr)]);
			dl = 1.0 - dl;
			du = 1.0 - du / fabs(fv[(long)dg * BS2 + rr * BS + rr]);
			lavg += dl;
			uavg += du;
			lmin = fmin(lmin, dl);
			umin = fmin(umin, du);
		}
	}
	__shared__ double sh[4][256];
	sh[0][threadIdx.x] = lavg;
	sh[1][threadIdx.x] = lmin;
	sh[2][threadIdx.x] = uavg;
	sh[3][threadIdx.x] = umin;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if (threadIdx.x < off) {
			sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
			sh[1][threadIdx.x] = fmin(sh[1][threadIdx.x], sh[1][threadIdx.x + off]);
			sh[2][threadIdx.x] += sh[2][threadIdx.x + off];
			sh[3][threadIdx.x] = fmin(sh[3][threadIdx.x], sh[3][threadIdx.x + off]);
		}
		__syncthreads();
	}
	if (threadIdx.x == 0)
		for (int q = 0; q < 4; q++)
			partial[(long)blockIdx.x * 4 + q] = sh[q][0];
}

// ------------------------------------------------------------------------------------ dispatch

#define BHIP_BS_SWITCH(BSVAL, RMVAL, CALL)                                       \
	switch (BSVAL) {                                                             \
	case 1: { constexpr int BS = 1; constexpr bool RM = false; CALL; } break;    \
	case 2: if (RMVAL) { constexpr int BS = 2; constexpr bool RM = true; CALL; } \
	        else { constexpr int BS = 2; constexpr bool RM = false; CALL; } break; \
	case 3: if (RMVAL) { constexpr int BS = 3; constexpr bool RM = true; CALL; } \
	        else { constexpr int BS = 3; constexpr bool RM = false; CALL; } break; \
	case 4: if (RMVAL) { constexpr int BS = 4; constexpr bool RM = true; CALL; } \
	        else { constexpr int BS = 4; constexpr bool RM = false; CALL; } break; \
	case 5: if (RMVAL) { constexpr int BS = 5; constexpr bool RM = true; CALL; } \
	        else { constexpr int BS = 5; constexpr bool RM = false; CALL; } break; \
	case 7: if (RMVAL) { constexpr int BS = 7; constexpr bool RM = true; CALL; } \
	        else { constexpr int BS = 7; constexpr bool RM = false; CALL; } break; \
	case 8: if (RMVAL) { constexpr int BS = 8; constexpr bool RM = true; CALL; } \
	        else { constexpr int BS = 8; constexpr bool RM = false; CALL; } break; \
	default: BHIP_FAIL(BLASTED_HIP_ENOTIMPL, "block size not instantiated (1,2,3,4,5,7,8)"); \
	}

void launch_factor_sweep(const FactorArgs &a, hipStream_t s)
{
	if (a.pat.nbrows == 0)
		return;
	if (launch_factor1(a, s))
		return;
	if (launch_factor4(a, s))
		return;
	if (launch_factor8(a, a.dinv_scratch, s))
		return;
	if (launch_factorodd(a, a.dinv_scratch, s))
		return;
	if (a.pat.bs >= 5 && a.dinv_scratch)
		launch_invert_diag_blocks(a.pat, a.in, 1, a.dinv_scratch, 0, s);
	BHIP_BS_SWITCH(a.pat.bs, a.pat.rowmajor, {
		const unsigned grid = (unsigned)(((long)a.pat.nbrows + FGeo<BS>::RPB - 1) / FGeo<BS>::RPB);
		hipLaunchKernelGGL((factor_sweep_kernel<BS, RM, false>), dim3(grid), dim3(256), 0, s, a,
		                   (double *)nullptr);
	})
	BHIP_CHECK(hipGetLastError());
}

// ---- the exact factorisation as ONE launch ---------------------------------------------------------------
// Round 2 measured that a level of the per-level form below costs one wave-lifetime of mostly idle chip, 766
// times at 256^3, and that storage order has nothing to do with it (tools/probes/level_ordered_factor.py).  This
// is the form the exact solves already have (kernels_level.hip): one launch, rows taken in level order by
// increasing workgroup id, and a row WAITS for the rows it depends on instead of a launch boundary, so that the
// levels overlap.  What a row reads from other rows are blocks of their diagonal + upper part (the u_kj of its
// position pairs and the inverted diagonal block of each lower entry's column); those blocks are pre-filled
// with the "pending" NaN pattern and published entry by entry with 8-byte agent-scope stores, so every entry is
// its own ready flag: a reader looks through the caches first (an entry is written once after the fill: anything
// but "pending" is final wherever it is read from) and re-reads coherently only what is still pending.  No
// fences, no flag array.  A lane group never blocks -- the rows of one wave may depend on each other across a
// level boundary -- it retries its current entry in the wave's next round; spins are bounded, a wave that runs
// out raises the abort flag and the host redoes the factorisation level by level.
constexpr unsigned long long SFF_PENDING = 0xFFF8DEADBEEF0001ull;  // = SF_PENDING of kernels_level.hip
constexpr int SFF_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ bool sff_pending(const double v)
{
	return (unsigned long long)__double_as_longlong(v) == SFF_PENDING;
}

// value of *p, through the caches first and coherently if that still shows the fill pattern
__device__ __forceinline__ double sff_read(const double *p)
{
	double v = *p;
	if (sff_pending(v))
		v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
		                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	return v;
}

__device__ __forceinline__ void sff_publish(double *p, const double v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
	                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the diagonal + upper blocks of every row <- "pending" (16 lanes per row)
template <int BS>
__global__ __launch_bounds__(256) void sff_fill_kernel(const Pattern pat, double *f)
{
	const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
	if (row >= pat.nbrows)
		return;
	const long beg = (long)pat.diagind[row] * (BS * BS), end = (long)pat.browptr[row + 1] * (BS * BS);
	unsigned long long *const q = reinterpret_cast<unsigned long long *>(f);
	for (long k = beg + (threadIdx.x & 15); k < end; k += 16)
		q[k] = SFF_PENDING;
}

template <int BS, bool RM>
__global__ __launch_bounds__(256) void sff_factor_kernel(const FactorArgs a, const int4 *__restrict__ meta,
                                                         const int count, int *ctl)
{
	using Ge = FGeo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, BS2 = BS * BS;
	constexpr unsigned long long GMASK = SUB == 64 ? ~0ull : ((1ull << SUB) - 1ull);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / SUB, u = lane % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(SUB - 1);

	const long pos = (long)blockIdx.x * Ge::RPB + wave * Ge::RPW + g;  // position in level order
	const bool rowok = pos < count;
	const int4 md = rowok ? meta[pos] : make_int4(0, 0, 0, 0);  // {row, browptr, diagind, browptr of the next row}
	const int irow = md.x;
	int jpos = rowok ? md.y : 0;
	const int jend = rowok ? md.w : 0;
	double *const f = a.out;  // in place: a.in == a.out

	int spins = 0;
	for (;;) {
		const bool busy = jpos < jend;
		if (__builtin_amdgcn_ballot_w64(busy) == 0ull)
			return;
		if (busy) {
			// one attempt at entry jpos of this group's row; `ready` drops when an operand is still pending
			const int col = a.pat.bcolind[jpos];
			double s = active ? a.avals[(long)jpos * BS2 + e] : 0.0;
			if (a.scale && active) {
				if (BS == 1) {  // (a s_i) s_j, two roundings, as the reference's scalar kernel and every other form here
					s *= a.scale[irow];
					s *= a.scale[col];
				} else
					s *= a.scale[(long)irow * BS + r] * a.scale[(long)col * BS + c];
			}
			bool ready = true;
			const int kbeg = a.posptr[jpos], kend = a.posptr[jpos + 1];
			for (int k = kbeg; k < kend; k++) {
				// l_ik: a lower block of this row, stored by this group earlier in this launch
				const double lv = active ? f[(long)a.lowerp[k] * BS2 + e] : 0.0;
				const double uv = active ? sff_read(f + (long)a.upperp[k] * BS2 + e) : 0.0;
				ready = ready && !sff_pending(uv);
				if (BS == 1)
					s -= lv * uv;
				else
					s -= group_gemm<BS, BSP>(lv, uv, gbase, r, c);
			}
			double dv = 0.0;
			if (irow > col) {
				dv = active ? sff_read(f + (long)a.pat.diagind[col] * BS2 + e) : 0.0;
				ready = ready && !sff_pending(dv);
			}
			// the whole lane group commits or retries together
			const unsigned long long rb = __builtin_amdgcn_ballot_w64(ready);
			const bool gready = ((rb >> gbase) & GMASK) == GMASK;
			if (gready) {
				if (irow > col) {
					// diagonal blocks are stored inverted (bs > 1); the scalar factor keeps u_jj itself
					const double res = (BS == 1) ? s / dv : group_gemm<BS, BSP>(s, dv, gbase, r, c);
					if (active)
						f[(long)jpos * BS2 + e] = res;
				} else if (irow == col && BS > 1) {
					const double inv = group_inverse<BS, BSP>(s, gbase, r, c);
					if (active)
						sff_publish(f + (long)jpos * BS2 + e, inv);
				} else if (active) {
					sff_publish(f + (long)jpos * BS2 + e, s);
				}
				jpos++;
			}
		}
		spins++;
		if (spins > SFF_SPIN_LIMIT ||
		    ((spins & 255) == 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
			if (lane == 0)
				__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
	}
}

// ---- the exact factorisation as one launch on ROW PLANS, any block size with at least 16 lanes per row (3 ... 8)
// The lane layout and the arithmetic of factor_sweep_kernel (lane (r,c) of a group holds entry (r,c) of every block;
// products and inverses through the group's shuffles: the same bits), the waiting scheme of sffactor4_kernel
// (kernels_factor4.hip, where it is described): a row's plan is one 64-byte record in padded level order, all its
// operands are requested up front, a wave waits ONCE for what its rows read from other rows with one lane polling
// one element, and the recurrence then runs on registers.  The kernel above walks a row entry by entry through
// three to four dependent index and operand round trips each: 18 us per level at 100^3 bs=8.
// Stencil-like rows only (the caps of build_row_plans); the rows of a workgroup come from one level.
constexpr int XP_MAXE = 8, XP_MAXL = 4, XP_MAXP = 8;

template <int BS, bool RM>
__global__ __launch_bounds__(256) void sffplan_fill_kernel(const FactorArgs a)
{
	constexpr int BS2 = BS * BS;
	const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
	const int t = threadIdx.x & 15;
	if (row >= a.pat.nbrows)
		return;
	const int dg = a.pat.diagind[row], jend = a.pat.browptr[row + 1];
	unsigned long long *const fq = reinterpret_cast<unsigned long long *>(a.out);
	int p0 = a.posptr[dg];
	for (int j = dg; j < jend; j++) {
		const int p1 = a.posptr[j + 1];
		if (j > dg && p1 == p0) {
			// an upper block without position pairs: its factor value is the (scaled) matrix block
			const int col = a.scale ? a.pat.bcolind[j] : 0;
			for (int e = t; e < BS2; e += 16) {
				double v = a.avals[(long)j * BS2 + e];
				if (a.scale) {
					const int r = RM ? e / BS : e % BS, c = RM ? e % BS : e / BS;
					v *= a.scale[(long)row * BS + r] * a.scale[(long)col * BS + c];
				}
				a.out[(long)j * BS2 + e] = v;
			}
		} else
			for (int e = t; e < BS2; e += 16)
				fq[(long)j * BS2 + e] = SFF_PENDING;
		p0 = p1;
	}
}

// ME / ML / MP: entries that need work, lower entries, position pairs a row may have (register arrays): 8 / 4 / 8 in
// general, 4 / 3 / 4 for a 7-point pattern (as for sffactor4_kernel).
// (occupancy bounds of the 7-point instantiation: bs = 5 fits eight waves per SIMD with one register less, 64)
template <int BS, bool RM, int ME, int ML, int MP>
__global__ __launch_bounds__(256, (ME == 4 && BS == 5) ? 8 : 1) void sffplan_kernel(const FactorArgs a, const int *__restrict__ desc, int *ctl)
{
	using Ge = FGeo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, BS2 = BS * BS;
	static_assert(SUB >= 16, "a row's plan is read by 16 lanes of its group");
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / SUB, u = lane % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(SUB - 1);
	double *const f = a.out;

	const long slot = (long)blockIdx.x * Ge::RPB + wave * Ge::RPW + g;  // padded position in level order
	const int dw = (u < 16) ? desc[slot * 16 + u] : 0;
	const int jbeg = __shfl(dw, gbase + 0, 64), irow = __shfl(dw, gbase + 1, 64);
	const unsigned w14 = (unsigned)__shfl(dw, gbase + 14, 64), w15 = (unsigned)__shfl(dw, gbase + 15, 64);
	const int ne = (int)(w14 & 15u), nl = (int)((w14 >> 4) & 15u), np = (int)((w14 >> 8) & 15u);
#define XP_CODE(TT) ((TT) < 4 ? (w14 >> (12 + 5 * (TT))) : (w15 >> (5 * ((TT)-4))))
#define XP_PQ(TT) ((TT) < np ? (int)(XP_CODE(TT) & 7u) : 8)
#define XP_PLL(TT) ((int)((XP_CODE(TT) >> 3) & 3u))
	unsigned pmask = 0u;  // entries with position pairs; an upper entry without any has been stored by the fill pass
#pragma unroll
	for (int tt = 0; tt < MP; tt++)
		pmask |= (tt < np) ? (1u << (XP_CODE(tt) & 7u)) : 0u;
#define XP_TODO(Q) ((Q) < ne && ((Q) <= nl || ((pmask >> (Q)) & 1u) != 0u))

	// ---- operands; blocks of other rows may still show the fill pattern
	double aS[ME], uv[MP], dv[ML], lres[ML];
	int upo[MP], dpo[ML];
#pragma unroll
	for (int q = 0; q < ME; q++)
		aS[q] = (XP_TODO(q) && active) ? a.avals[(long)(jbeg + q) * BS2 + e] : 0.0;
#pragma unroll
	for (int tt = 0; tt < MP; tt++) {
		upo[tt] = __shfl(dw, gbase + 2 + tt, 64);
		uv[tt] = (tt < np && active) ? f[(long)upo[tt] * BS2 + e] : 0.0;
	}
#pragma unroll
	for (int q = 0; q < ML; q++) {
		dpo[q] = __shfl(dw, gbase + 10 + q, 64);
		dv[q] = (q < nl && active) ? f[(long)dpo[q] * BS2 + e] : 0.0;
		lres[q] = 0.0;
	}
	if (a.scale) {
#pragma unroll
		for (int q = 0; q < ME; q++)
			if (XP_TODO(q) && active) {
				const int col = a.pat.bcolind[jbeg + q];
				aS[q] *= a.scale[(long)irow * BS + r] * a.scale[(long)col * BS + c];
			}
	}

	// ---- one wait for everything; one lane polls one element for the whole wave (see x4_rows)
	int spins = 0;
	for (;;) {
		const double *miss = nullptr;
#pragma unroll
		for (int tt = MP - 1; tt >= 0; tt--)
			if (tt < np && sff_pending(uv[tt]))
				miss = f + (long)upo[tt] * BS2 + e;
#pragma unroll
		for (int q = ML - 1; q >= 0; q--)
			if (q < nl && sff_pending(dv[q]))
				miss = f + (long)dpo[q] * BS2 + e;
		const unsigned long long waiting = __builtin_amdgcn_ballot_w64(miss != nullptr);
		if (waiting == 0ull)
			break;
		const int lead = __builtin_ctzll(waiting);
		const unsigned long long addr = (unsigned long long)reinterpret_cast<uintptr_t>(miss);
		const unsigned alo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)addr, lead);
		const unsigned ahi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(addr >> 32), lead);
		const double *const gate = reinterpret_cast<const double *>((uintptr_t)(((unsigned long long)ahi << 32) | alo));
		for (;;) {
			const double gv = __longlong_as_double((long long)__hip_atomic_load(
			    reinterpret_cast<const unsigned long long *>(gate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
			if (!sff_pending(gv))
				break;
			spins++;
			if (spins > SFF_SPIN_LIMIT ||
			    ((spins & 255) == 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
				if (lane == 0)
					__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
			__builtin_amdgcn_s_sleep(2);
		}
#pragma unroll
		for (int q = 0; q < ML; q++)
			if (q < nl && sff_pending(dv[q]))
				dv[q] = __longlong_as_double((long long)__hip_atomic_load(
				    reinterpret_cast<const unsigned long long *>(f + (long)dpo[q] * BS2 + e), __ATOMIC_RELAXED,
				    __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
		for (int tt = 0; tt < MP; tt++)
			if (tt < np && sff_pending(uv[tt]))
				uv[tt] = __longlong_as_double((long long)__hip_atomic_load(
				    reinterpret_cast<const unsigned long long *>(f + (long)upo[tt] * BS2 + e), __ATOMIC_RELAXED,
				    __HIP_MEMORY_SCOPE_AGENT));
		if (++spins > SFF_SPIN_LIMIT) {
			if (lane == 0)
				__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
	}

	// ---- the rows' recurrences on registers, entry by entry, the groups of a wave in lockstep (the group
	// products exchange lanes: no divergence around them)
#pragma unroll
	for (int q = 0; q < ME; q++) {
		if (__builtin_amdgcn_ballot_w64(q < ne) == 0ull)
			break;
		const bool valid = XP_TODO(q);
		if (__builtin_amdgcn_ballot_w64(valid) == 0ull)
			continue;
		double sv = aS[q];
#pragma unroll
		for (int tt = 0; tt < MP; tt++) {
			const bool in = valid && XP_PQ(tt) == q;
			if (__builtin_amdgcn_ballot_w64(in) == 0ull)
				continue;
			const int ll = XP_PLL(tt);
			const double lv = ll == 0 ? lres[0] : (ll == 1 ? lres[1] : ((ll == 2 || ML < 4) ? lres[2] : lres[ML < 4 ? 2 : 3]));
			sv -= group_gemm<BS, BSP>(in ? lv : 0.0, in ? uv[tt] : 0.0, gbase, r, c);
		}
		const bool lowerq = valid && q < nl, diagq = valid && q == nl;
		if (q < ML && __builtin_amdgcn_ballot_w64(lowerq) != 0ull) {
			// S * inverse(U_jj): diagonal blocks are stored inverted as soon as they are final
			const double prod = group_gemm<BS, BSP>(lowerq ? sv : 0.0, lowerq ? dv[q < ML ? q : 0] : 0.0, gbase, r, c);
			if (lowerq) {
				sv = prod;
				lres[q < ML ? q : 0] = prod;
			}
		}
		if (q <= ML && __builtin_amdgcn_ballot_w64(diagq) != 0ull) {
			const double inv = group_inverse<BS, BSP>(diagq ? sv : ((r == c) ? 1.0 : 0.0), gbase, r, c);
			if (diagq)
				sv = inv;
		}
		if (valid && active) {
			double *const dst = f + (long)(jbeg + q) * BS2 + e;
			if (lowerq)
				*dst = sv;  // read by this row (from registers) and by the triangular solves later
			else
				sff_publish(dst, sv);
		}
	}
#undef XP_CODE
#undef XP_PQ
#undef XP_PLL
#undef XP_TODO
}

// tuning "factorsf=0|1|2|3": one launch per level | one launch where it pays | always one launch | always one
// launch of the general kernel (no matrix-core kernel at bs = 4); the factor is the same bits in every form
static int g_factor_syncfree = 1;
static int g_factor_plan = 1;  // "factorsf=p0|p1": plan kernel for block sizes other than 1 and 4 off | on
static int g_factor_fake_abort = 0;  // "factorsf=a1": tests -- behave as if a wave had given up waiting
void set_invert_rowlane(int on)
{
	g_invert_rowlane = on != 0;
}

void set_factor_syncfree(int on)
{
	if (on >= 20)
		g_factor_fake_abort = on - 20;
	else if (on >= 10)
		g_factor_plan = on - 10;
	else
		g_factor_syncfree = on;
}

// diagonal + upper blocks of the factor <- the pending pattern (before a single-launch exact factorisation)
void launch_factor_pending_fill(const Pattern &pat, double *f, hipStream_t s)
{
	BHIP_BS_SWITCH(pat.bs, pat.rowmajor, {
		(void)RM;
		const unsigned fgrid = (unsigned)(((long)pat.nbrows + 15) / 16);
		hipLaunchKernelGGL((sff_fill_kernel<BS>), dim3(fgrid), dim3(256), 0, s, pat, f);
	})
}

// The exact factorisation as one launch: 1 = done, 0 = does not apply, -1 = a wave gave up waiting (the caller
// then runs launch_factor_levels, which does not depend on what this attempt left behind).
int launch_factor_syncfree(const FactorArgs &a, LevelSchedule &ls, hipStream_t s)
{
	if (!g_factor_syncfree || !ls.built || !ls.meta || !ls.ctl || a.in != a.out || a.pat.nbrows == 0)
		return 0;
	if (a.pat.bs > 1 && !a.diag_inverted)
		return 0;  // (the remainder diagnostics want the un-inverted factor: per-level form)
	if (g_factor_fake_abort) {
		// what an attempt that gave up leaves behind at worst: the fill pattern all over the diagonal + upper part
		launch_factor_pending_fill(a.pat, a.out, s);
		return -1;
	}
	if (g_factor_syncfree != 3 && a.pat.bs == 1) {
		// stencil-like scalar rows: one lane per row, the rows of a workgroup from one level
		const int r1 = launch_factor1_syncfree(a, ls, s);
		if (r1 != 0)
			return r1;
	}
	if (g_factor_syncfree != 3 && a.pat.bs != 1 && a.pat.bs != 4 && a.pat.bs != 2 && g_factor_plan) {
		// stencil-like rows: the plan kernel in the general lane layout
		if (build_row_plans(a, ls, a.pat.bs == 3 ? 16 : 4, s)) {
			BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
			BHIP_BS_SWITCH(a.pat.bs, a.pat.rowmajor, {
				if constexpr (FGeo<BS>::SUB >= 16) {
					// (the plans were laid out for bs == 3 ? 16 : 4 rows per workgroup above; bs 1, 2, 4 never get here)
					static_assert(BS < 3 || BS == 4 || FGeo<BS>::RPB == (BS == 3 ? 16 : 4), "rows per workgroup");
					hipLaunchKernelGGL((sffplan_fill_kernel<BS, RM>), dim3((unsigned)(((long)a.pat.nbrows + 15) / 16)),
					                   dim3(256), 0, s, a);
					if (ls.max_lower <= 3 && ls.f4_maxpairs <= 4 && ls.f4_maxtodo <= 4)
						hipLaunchKernelGGL((sffplan_kernel<BS, RM, 4, 3, 4>), dim3((unsigned)ls.f4_grid), dim3(256), 0, s, a,
						                   ls.f4_desc, ls.ctl);
					else
						hipLaunchKernelGGL((sffplan_kernel<BS, RM, XP_MAXE, XP_MAXL, XP_MAXP>), dim3((unsigned)ls.f4_grid),
						                   dim3(256), 0, s, a, ls.f4_desc, ls.ctl);
				}
			})
			BHIP_CHECK(hipGetLastError());
			int ctl[2] = {0, 0};
			BHIP_CHECK(hipMemcpyAsync(ctl, ls.ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
			BHIP_CHECK(hipStreamSynchronize(s));
			return ctl[1] == 0 ? 1 : -1;
		}
	}
	if (g_factor_syncfree != 3 && a.pat.bs == 4) {
		// stencil-like rows at bs = 4: the matrix-core kernel that prepares a row before it waits
		const int r4 = launch_factor4_syncfree(a, ls, s);
		if (r4 != 0)
			return r4;
	}
	// Where it pays (round 2, ms per exact factorisation, one launch per level -> one launch):
	//   bs=8 100^3 7.96 -> 5.58, bs=7 128^3 11.75 -> 9.54, unstructured bs=5 (1006 levels) 32.6 -> 15.5,
	//   bs=4 256^3 20.0 -> 18.5, bs=4 128^3 5.85 -> 5.99, bs=3 128^3 5.74 -> 6.30, bs=2 128^3 5.60 -> 19.5,
	//   scalar 64^3 2.54 -> 12.3.
	// It pays where the per-level form suffers most from its launch boundaries (deep, narrow level structures; one
	// row per wave) and loses where many rows share a wave.  Not because of divergence between lane groups that got
	// out of phase: a form of the round that is executed in lockstep by the whole wave (wave-maximum loop bounds,
	// selects instead of branches, every group computing the product with the inverse and -- when any group is at
	// its diagonal -- the inverse) was built and was slower everywhere (bs=8 5.6 -> 7.0 ms, unstructured 15.9 ->
	// 17.4, bs=4 256^3 18.5 -> 20.5).  The rows of a wave sit in one level and wait for the previous one together;
	// this kernel asks for a row's operands entry by entry, so a waiting row has nothing else in flight -- unlike
	// the exact solves, which request everything independent up front.  That form (a row's indices and A blocks
	// staged at the start, polls only for the dependent operands) is what the headline bs=4 case would need.
	// The default takes the single launch where a wave is one row (bs >= 5); "factorsf=2" forces it everywhere (tests).
	if (g_factor_syncfree < 2 && a.pat.bs < 5)
		return 0;
	BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
	BHIP_BS_SWITCH(a.pat.bs, a.pat.rowmajor, {
		const unsigned fgrid = (unsigned)(((long)a.pat.nbrows + 15) / 16);
		hipLaunchKernelGGL((sff_fill_kernel<BS>), dim3(fgrid), dim3(256), 0, s, a.pat, a.out);
		const unsigned grid = (unsigned)(((long)ls.count + FGeo<BS>::RPB - 1) / FGeo<BS>::RPB);
		hipLaunchKernelGGL((sff_factor_kernel<BS, RM>), dim3(grid), dim3(256), 0, s, a, ls.meta, ls.count, ls.ctl);
	})
	BHIP_CHECK(hipGetLastError());
	int ctl[2] = {0, 0};
	BHIP_CHECK(hipMemcpyAsync(ctl, ls.ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
	BHIP_CHECK(hipStreamSynchronize(s));
	return ctl[1] == 0 ? 1 : -1;
}

// Exact ILU(0) in one pass (the reference's sequential factorisation, threadedfactor = false: one
// in-order sweep, src/async_blockilu_factor.cpp:186-204 with one thread): one launch per dependency
// level over that level's rows, in place.  A row's entries need final values of rows in earlier levels
// only (its lower neighbours' upper parts and diagonals), and the group that owns a row walks its
// entries in storage order, reading back what it has just stored -- exactly the serial recurrence.
int launch_factor_levels(FactorArgs a, const LevelSchedule &ls, hipStream_t s)
{
	if (!ls.built)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "launch_factor_levels: no level schedule");
	if (a.in != a.out)
		BHIP_FAIL(BLASTED_HIP_EINVAL, "launch_factor_levels: in place only");
	// (Tried: the whole factorisation as ONE dependency-polling launch like the exact solves -- rows wait
	// for their lower neighbours' "finished" flags, factor entries read with agent-scope atomic loads.  It
	// is correct but several times slower than one launch per level (256^3 bs=4: 167 ms against 29.5 ms):
	// a row walks its entries sequentially through ~20 dependent, now uncached, loads.)
	// (Tried for bs = 5 / 7: the tuned pair-layout kernel per level plus a launch that inverts the level's
	// diagonal blocks.  Two launches per level cost more than the on-the-fly inverses save: 72.8 ms
	// instead of 52.7 ms on the unstructured bs=5 case, 21.8 against 22.1 ms at bs=7.)
	// (Tried: a row-at-once kernel that fetches everything final before the level -- indices, A blocks, pair
	// lists, upper operands, inverted diagonals -- up front and runs the row's recurrence from registers and
	// LDS, to shorten the ~20 serialised loads of a row.  Bit-identical, and no faster: 21.9 against 20.2 ms,
	// per-level times 5.3 / 28.8 / 53 us (min / mean / max) against 4.2 / 28 / 49.  The levels are bound by
	// throughput (1.3 ns per row = 2.1 TB/s of 8-byte-per-lane loads), not by the dependent chain; the lever
	// is the MFMA kernel's 16-byte layout on row lists, not latency.)
	// (Tried: the matrix-core kernel on row lists -- factor4_kernel's operand layouts and MFMA products, one
	// row per block slot, the row's finished lower blocks kept in LDS: 19.96 ms against 20.2 ms.  Neither the
	// dependent chain nor the block arithmetic bounds a level; its rows are scattered over the natural-order
	// storage (a wavefront i+j+k = const), every operand is a lone 128-byte access, and ~2.2 TB/s is what that
	// pattern gets.  -- Round 2 measured that conclusion away: on the matrix symmetrically permuted into its own
	// level order, where every level's rows, blocks and gather targets are contiguous, this same loop takes 19.0 ms
	// instead of 20.1 (tools/probes/level_ordered_factor.py).  A level costs one wave-lifetime of mostly idle chip,
	// 766 times; the lever is overlapping levels inside one launch, as the exact solves do.)
	// (Tried: the 766 launches of the 256^3 problem as one instantiated hipGraph.  On a private stream the
	// level loop takes 22.0 ms instead of ~23.7 ms -- the kernels themselves are 21.4 ms, the rest of the
	// 29.6 ms call is the initial copy of the values and the final inversion of the diagonal blocks -- and
	// launched into the null stream, which is what torch and the tests hand over, it gains nothing.)
	a.dinv_scratch = nullptr;  // diagonal blocks of earlier levels: inverted on the fly, or stored inverted (diag_inverted)
	for (int l = 0; l < ls.nlevels; l++) {
		a.rows = ls.rows + ls.ptr[l];
		a.nrows = ls.ptr[l + 1] - ls.ptr[l];
		BHIP_BS_SWITCH(a.pat.bs, a.pat.rowmajor, {
			const unsigned grid = (unsigned)(((long)a.nrows + FGeo<BS>::RPB - 1) / FGeo<BS>::RPB);
			hipLaunchKernelGGL((factor_sweep_kernel<BS, RM, false>), dim3(grid), dim3(256), 0, s, a,
			                   (double *)nullptr);
		})
	}
	BHIP_CHECK(hipGetLastError());
	return ls.nlevels;
}

double run_nonlinear_res(const FactorArgs &a, double *dev_scratch, hipStream_t s)
{
	if (a.pat.nbrows == 0)
		return 0.0;
	unsigned grid = 0;
	BHIP_BS_SWITCH(a.pat.bs, a.pat.rowmajor, {
		grid = (unsigned)(((long)a.pat.nbrows + FGeo<BS>::RPB - 1) / FGeo<BS>::RPB);
		hipLaunchKernelGGL((factor_sweep_kernel<BS, RM, true>), dim3(grid), dim3(256), 0, s, a,
		                   dev_scratch);
	})
	BHIP_CHECK(hipGetLastError());
	std::vector<double> h(grid);
	BHIP_CHECK(hipMemcpyAsync(h.data(), dev_scratch, sizeof(double) * grid, hipMemcpyDeviceToHost, s));
	BHIP_CHECK(hipStreamSynchronize(s));
	double sum = 0;
	for (unsigned i = 0; i < grid; i++)
		sum += h[i];
	return sum;
}

void launch_invert_diag_blocks(const Pattern &pat, const double *src, long src_by_diag, double *dst,
                               long dst_by_diag, hipStream_t s)
{
	if (pat.nbrows == 0)
		return;
	BHIP_BS_SWITCH(pat.bs, pat.rowmajor, {
		if (BS >= 5 && g_invert_rowlane) {
			const unsigned grid = (unsigned)(((long)pat.nbrows + 31) / 32);
			hipLaunchKernelGGL((invert_blocks_rowlane_kernel<(BS >= 5 ? BS : 5), RM>), dim3(grid), dim3(256), 0, s,
			                   pat, src, (int)src_by_diag, dst, (int)dst_by_diag);
		} else if (BS >= 5) {
			const unsigned grid = (unsigned)(((long)pat.nbrows + 63) / 64);
			hipLaunchKernelGGL((invert_blocks_tpb_kernel<(BS >= 5 ? BS : 5), RM>), dim3(grid), dim3(64), 0, s,
			                   pat, src, (int)src_by_diag, dst, (int)dst_by_diag, (const int *)nullptr, 0);
		} else {
			const unsigned grid = (unsigned)(((long)pat.nbrows + FGeo<BS>::RPB - 1) / FGeo<BS>::RPB);
			hipLaunchKernelGGL((invert_blocks_kernel<BS, RM>), dim3(grid), dim3(256), 0, s, pat, src,
			                   (int)src_by_diag, dst, (int)dst_by_diag);
		}
	})
	BHIP_CHECK(hipGetLastError());
}

void launch_scaling_vector(const Pattern &pat, const double *vals, double *scale, hipStream_t s)
{
	const long n = (long)pat.nbrows * pat.bs;
	if (n == 0)
		return;
	hipLaunchKernelGGL(scaling_vector_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pat,
	                   vals, scale);
	BHIP_CHECK(hipGetLastError());
}

// partial[b] = sum over workgroup b's slice of (x[i] - y[i])^2
__global__ __launch_bounds__(256) void diff_norm2_kernel(const double *x, const double *y, long n, double *partial)
{
	__shared__ double wsum[4];
	double acc = 0.0;
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
		const double d = x[i] - y[i];
		acc += d * d;
	}
	for (int off = 32; off > 0; off >>= 1)
		acc += __shfl_xor(acc, off, 64);
	if ((threadIdx.x & 63) == 0)
		wsum[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0)
		partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ||x - y||_2 (BJacobiSRPreconditioner::apply_relax's convergence check, src/solverops_jacobi.cpp:91-101);
// dev_scratch: at least 1024 doubles; the partial sums are added on the host in fixed order
double run_diff_norm(const double *x, const double *y, long n, double *dev_scratch, hipStream_t s)
{
	if (n == 0)
		return 0.0;
	const long want = (n + 255) / 256;
	const unsigned grid = (unsigned)(want < 1024 ? want : 1024);
	hipLaunchKernelGGL(diff_norm2_kernel, dim3(grid), dim3(256), 0, s, x, y, n, dev_scratch);
	BHIP_CHECK(hipGetLastError());
	double h[1024];
	BHIP_CHECK(hipMemcpyAsync(h, dev_scratch, sizeof(double) * grid, hipMemcpyDeviceToHost, s));
	BHIP_CHECK(hipStreamSynchronize(s));
	double sum = 0.0;
	for (unsigned i = 0; i < grid; i++)
		sum += h[i];
	return std::sqrt(sum);
}

void launch_scale_vec(double *z, const double *scale, long n, hipStream_t s)
{
	if (n == 0)
		return;
	hipLaunchKernelGGL(mul_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, scale, n);
	BHIP_CHECK(hipGetLastError());
}

// iluvals initial guess.  dblk_scratch: nbrows*bs*bs doubles, used by INIT_F_SGS only.
void launch_fact_init(const Pattern &pat, const double *avals, const double *scale, int init_type,
                      double *iluvals, double *dblk_scratch, hipStream_t s)
{
	const long nv = (long)pat.nnzb * pat.bs * pat.bs;
	if (nv == 0)
		return;
	const unsigned egrid = (unsigned)((nv + 255) / 256);
	auto copy_scaled = [&]() {
		if (!scale) {
			BHIP_CHECK(hipMemcpyAsync(iluvals, avals, sizeof(double) * nv, hipMemcpyDeviceToDevice, s));
			return;
		}
		BHIP_BS_SWITCH(pat.bs, pat.rowmajor, {
			hipLaunchKernelGGL((scaled_copy_kernel<BS, RM>), dim3(egrid), dim3(256), 0, s, pat, avals,
			                   scale, iluvals);
		})
	};
	switch (init_type) {
	case BLASTED_HIP_INIT_F_ZERO:
		BHIP_CHECK(hipMemsetAsync(iluvals, 0, sizeof(double) * nv, s));
		if (pat.bs > 1)
			break;
		// scalar: the reference falls through into INIT_F_ORIGINAL (async_ilu_factor.cpp:48-54)
		[[fallthrough]];
	case BLASTED_HIP_INIT_F_ORIGINAL: copy_scaled(); break;
	case BLASTED_HIP_INIT_F_SGS:
		copy_scaled();
		// D^-1 of the (scaled) diagonal, then L <- L D_col^-1
		launch_invert_diag_blocks(pat, iluvals, 1, dblk_scratch, 0, s);
		BHIP_BS_SWITCH(pat.bs, pat.rowmajor, {
			const unsigned grid = (unsigned)(((long)pat.nbrows + FGeo<BS>::RPB - 1) / FGeo<BS>::RPB);
			hipLaunchKernelGGL((sgs_init_lower_kernel<BS, RM>), dim3(grid), dim3(256), 0, s, pat,
			                   dblk_scratch, iluvals);
		})
		break;
	default: break;  // INIT_F_NONE: keep the contents
	}
	BHIP_CHECK(hipGetLastError());
}

void run_diag_dominance(const Pattern &pat, const double *fvals, double *dev_scratch, double *out4,
                        hipStream_t s)
{
	const unsigned grid = (unsigned)((pat.nbrows + 255) / 256);
	BHIP_BS_SWITCH(pat.bs, pat.rowmajor, {
		hipLaunchKernelGGL((diag_dominance_kernel<BS, RM>), dim3(grid), dim3(256), 0, s, pat, fvals,
		                   dev_scratch);
	})
	BHIP_CHECK(hipGetLastError());
	std::vector<double> h((size_t)grid * 4);
	BHIP_CHECK(hipMemcpyAsync(h.data(), dev_scratch, sizeof(double) * grid * 4, hipMemcpyDeviceToHost, s));
	BHIP_CHECK(hipStreamSynchronize(s));
	double lavg = 0, lmin = 1e30, uavg = 0, umin = 1e30;
	for (unsigned b = 0; b < grid; b++) {
		lavg += h[b * 4 + 0];
		lmin = h[b * 4 + 1] < lmin ? h[b * 4 + 1] : lmin;
		uavg += h[b * 4 + 2];
		umin = h[b * 4 + 3] < umin ? h[b * 4 + 3] : umin;
	}
	const double den = (double)pat.nbrows * pat.bs;
	out4[0] = lavg / den;
	out4[1] = lmin;
	out4[2] = uavg / den;
	out4[3] = umin;
}

}  // namespace bhip
