// kernels_sweep.hip -- the row-sweep kernel family: every operator on the hot path is one
// "x_i <- post( rhs_i, D_i, sum_{j in part(i)} A_ij x_j )" pass over all block-rows.
//
//   reference routine (one OpenMP loop over rows each)              part      post     D
//   block_unit_lower_triangular   kernels_ilu_apply.hpp:54-67       LOWER     SUB      -
//   block_upper_triangular        kernels_ilu_apply.hpp:79-94       UPPER     D_SUB    factor diag (inverted)
//   scalar_upper_triangular       kernels_ilu_apply.hpp:30-42       UPPER     D_SUB    1/factor diag
//   block_fgs / scalar_fgs        kernels_sgs.hpp:47-60,17-29       LOWER     D_SUB    dblocks
//   block_bgs / scalar_bgs        kernels_sgs.hpp:62-76,31-44       UPPER     SUB_D    dblocks
//   block_relax_kernel/scalar_relax kernels_relaxation.hpp:17-54    OFFDIAG   D_SUB    dblocks
//   BLAS_BSR::matrix_apply/gemv3  blas/matvecs.cpp:26-75            ALL       AXPBY    -
//   BJacobi apply                 solverops_jacobi.cpp:51-63        NONE      D_SUB    dblocks
//
// Mapping (wave64): a group of G lanes owns one block-row; inside the group SUB = BSP*BSP lanes
// (BSP = bs rounded up to a power of two) own one stored block, lane (r,c) holding entry (r,c), so a
// block is read by one coalesced load and NB = G/SUB blocks of the row are in flight per iteration.
// The bs x bs mat-vec is one FMA per lane followed by an xor-butterfly over the column bits (and the
// block-slot bits); D is applied with a second butterfly.  Groups of one wave own consecutive rows,
// so rhs loads and the result store of a wave are contiguous.  Workgroups are renumbered so that each
// XCD sweeps one contiguous range of rows (its L2 then serves the neighbour gathers, and inside an
// XCD later workgroups see earlier ones' updates: Gauss-Seidel-like inside, Jacobi-like across).
//
// Async semantics: with xin == xout the sweep is chaotic relaxation -- iterate values written by other
// waves of the same launch may or may not be observed (plain loads; stale L1/L2 lines are older
// iterates, which is what the reference's `omp for ... nowait` permits too).  Each output double is
// produced in registers and stored once: no partial sum is ever published
// (kernels_ilu0_factorize.hpp:34-40).
#include "ctx.hpp"

namespace bhip {

template <int BS>
struct Geo {
	static constexpr int BSP = BS <= 1 ? 1 : (BS <= 2 ? 2 : (BS <= 4 ? 4 : 8));
	static constexpr int SUB = BSP * BSP;
	// lanes per block-row
	static constexpr int G = BS == 1 ? 4 : (BS == 2 ? 8 : SUB);
	static constexpr int NB = G / SUB;
	static constexpr int RPW = 64 / G;        // rows per wave
	static constexpr int RPB = 4 * RPW;       // rows per 256-thread workgroup
};

// Contiguous range of workgroup ids per XCD (workgroups are dealt round-robin to the 8 XCDs).
__device__ __forceinline__ unsigned xcd_chunk(unsigned bid, unsigned nwg)
{
	const unsigned xcd = bid & 7u, local = bid >> 3;
	const unsigned base = nwg >> 3, rem = nwg & 7u;
	return xcd * base + (xcd < rem ? xcd : rem) + local;
}

template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ double butterfly(double v)
{
#pragma unroll
	for (int off = OFF_LO; off < OFF_HI; off <<= 1)
		v += __shfl_xor(v, off, 64);
	return v;
}

template <int BS, bool RM, int PART, int POST, int DSRC>
__global__ __launch_bounds__(256) void sweep_kernel(const SweepArgs a)
{
	using Ge = Geo<BS>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, G = Ge::G, NB = Ge::NB, BS2 = BS * BS;

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane / G, t = lane % G;
	const int sub = t / SUB, u = t % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;

	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const long rowlin = (long)chunk * Ge::RPB + wave * Ge::RPW + g;
	const bool rowok = rowlin < a.pat.nbrows;
	const int i = rowok ? (a.descending ? a.pat.nbrows - 1 - (int)rowlin : (int)rowlin) : 0;

	int jbeg = 0, jend = 0, dg = 0;
	if (rowok) {
		dg = a.pat.diagind[i];
		if (PART == PART_LOWER) {
			jbeg = a.pat.browptr[i];
			jend = dg;
		} else if (PART == PART_UPPER) {
			jbeg = dg + 1;
			jend = a.pat.browptr[i + 1];
		} else if (PART == PART_OFFDIAG || PART == PART_ALL) {
			jbeg = a.pat.browptr[i];
			jend = a.pat.browptr[i + 1];
		}
	}

	// D entry and rhs are independent of the gather loop: issue their loads first
	double d = 0.0;
	if (DSRC != D_NONE && rowok && active) {
		if (DSRC == D_VALS_DIAG)
			d = a.dvals[(long)dg * BS2 + e];
		else if (DSRC == D_DBLOCKS)
			d = a.dvals[(long)i * BS2 + e];
		else
			d = 1.0 / a.dvals[dg];
	}
	double rv = 0.0;
	if (rowok && r < BS && a.rhs) {
		rv = a.rhs[(long)i * BS + r];
		if (a.rscale)
			rv *= a.rscale[(long)i * BS + r];
	}

	double acc = 0.0;
	if (PART != PART_NONE) {
		for (int jj = jbeg + sub; jj < jend; jj += NB) {
			if (PART == PART_OFFDIAG && jj == dg)
				continue;
			const int col = a.pat.bcolind[jj];
			if (active)
				acc += a.vals[(long)jj * BS2 + e] * a.xin[(long)col * BS + c];
		}
		acc = butterfly<BSP, G>(acc);  // over the column bits and the block-slot bits
	}

	const int gbase = lane & ~(G - 1);
	double out;
	if (POST == POST_SUB) {
		out = rv - acc;
	} else if (POST == POST_D_SUB) {
		const double v = rv - acc;                    // lanes (r,*) hold v[r]
		const double vc = __shfl(v, gbase + c, 64);   // lane (c,0) of block slot 0 holds v[c]
		out = butterfly<BSP, SUB>(active ? d * vc : 0.0);
	} else if (POST == POST_SUB_D) {
		const double wc = __shfl(acc, gbase + c, 64);
		out = rv - butterfly<BSP, SUB>(active ? d * wc : 0.0);
	} else {
		out = a.a * acc;
		if (a.b != 0.0)
			out += a.b * rv;
	}

	if (rowok && sub == 0 && c == 0 && r < BS) {
		if (a.changed && !(a.xout[(long)i * BS + r] == out))
			*a.changed = 1;
		a.xout[(long)i * BS + r] = out;
	}
}

template <int BS, bool RM>
static void dispatch_ops(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + Geo<BS>::RPB - 1) / Geo<BS>::RPB);
	if (grid == 0)
		return;
#define BHIP_CASE(P, Q, D)                                                                       \
	if (part == P && post == Q && dsrc == D) {                                                   \
		hipLaunchKernelGGL((sweep_kernel<BS, RM, P, Q, D>), dim3(grid), dim3(256), 0, s, a);     \
		return;                                                                                  \
	}
	BHIP_CASE(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASE(PART_UPPER, POST_D_SUB, D_VALS_DIAG)
	BHIP_CASE(PART_UPPER, POST_D_SUB, D_RECIP_DIAG)
	BHIP_CASE(PART_LOWER, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE(PART_UPPER, POST_SUB_D, D_DBLOCKS)
	BHIP_CASE(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE(PART_ALL, POST_AXPBY, D_NONE)
	BHIP_CASE(PART_NONE, POST_D_SUB, D_DBLOCKS)
#undef BHIP_CASE
	BHIP_FAIL(BLASTED_HIP_EINVAL, "launch_sweep: operator combination not instantiated");
}

template <int BS>
static void dispatch_layout(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	if (BS > 1 && a.pat.rowmajor)
		dispatch_ops<BS, true>(a, part, post, dsrc, s);
	else
		dispatch_ops<BS, false>(a, part, post, dsrc, s);
}

bool sweep_supported(int bs)
{
	return bs >= 1 && bs <= 8 && bs != 6;
}

void launch_sweep(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	if (launch_sweepw(a, part, post, dsrc, s))
		return;
	switch (a.pat.bs) {
	case 1: dispatch_layout<1>(a, part, post, dsrc, s); break;
	case 2: dispatch_layout<2>(a, part, post, dsrc, s); break;
	case 3: dispatch_layout<3>(a, part, post, dsrc, s); break;
	case 4: dispatch_layout<4>(a, part, post, dsrc, s); break;
	case 5: dispatch_layout<5>(a, part, post, dsrc, s); break;
	case 7: dispatch_layout<7>(a, part, post, dsrc, s); break;
	case 8: dispatch_layout<8>(a, part, post, dsrc, s); break;
	default: BHIP_FAIL(BLASTED_HIP_ENOTIMPL, "block size not instantiated (1,2,3,4,5,7,8)");
	}
	BHIP_CHECK(hipGetLastError());
}

}  // namespace bhip
