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
// This file is the general family: every instantiated block size (1, 2, 3, 4, 5, 7, 8) in both block
// layouts; column-major bs = 4 and 8 are normally taken by the wide kernel of kernels_sweepw.hip.
//
// Mapping (wave64): a group of G lanes owns one block-row; inside the group SUB = BSP*BSP lanes
// (BSP = bs rounded up to a power of two) own one stored block, lane (r,c) = (u % BSP, u / BSP) holding
// entry (r,c), so a block is one coalesced load and NB = G/SUB blocks of the row ride in one load
// instruction (scalar CSR: 4 entries).  A workgroup owns RCHUNK consecutive rows of the sweep and stages
// their browptr / diagind / bcolind range in LDS once, coalesced, so that no value load waits on an
// index load from HBM; all loads of a row step (KFIX predicated block passes, straight-line) are issued
// before the first use, rows with more blocks finish in a remainder loop.  The bs x bs mat-vec is one
// FMA per lane plus an all-reduce over the column bits and the block-slot bits on the VALU (lanes.hpp);
// D is applied with one broadcast and a second all-reduce.  Workgroups are renumbered so that the XCDs
// take turns on super-chunks of consecutive chunks (lanes.hpp, xcd_chunk).
//
// Async semantics: with xin == xout the sweep is chaotic relaxation -- iterate values written by other
// waves of the same launch may or may not be observed (plain loads; stale L1/L2 lines are older
// iterates, which is what the reference's `omp for ... nowait` permits too).  Each output double is
// produced in registers and stored once: no partial sum is ever published
// (kernels_ilu0_factorize.hpp:34-40).
#include "ctx.hpp"
#include "lanes.hpp"
#include "stage.hpp"
#include "sweep_geo.hpp"

namespace bhip {

template <int BS, bool RM, int PART, int POST, int DSRC, int UNR, bool BIG = false, bool Z1 = false>
__global__ __launch_bounds__(256) void sweep_kernel(const SweepArgs a)
{
	static_assert(!Z1 || (BS == 1 && PART == PART_LOWER && POST == POST_SUB), "fused z1: scalar lower sweep only");
	static_assert((Geo<BS, BIG>::RCHUNK / Geo<BS, BIG>::RSTEP) % UNR == 0, "row steps per chunk must be a multiple of UNR");
	using Ge = Geo<BS, BIG>;
	constexpr int BSP = Ge::BSP, SUB = Ge::SUB, G = Ge::G, NB = Ge::NB, BS2 = BS * BS;
	constexpr int RPW = Ge::RPW, RSTEP = Ge::RSTEP, RCHUNK = Ge::RCHUNK, CAP = Ge::CAP;
	// straight-line passes: enough for a 7-point row's lower part or diagonal + upper part, twice as
	// many for operators that visit the whole row
	constexpr int KFIX = (NB >= 4 ? 1 : 4) * ((PART == PART_ALL || PART == PART_OFFDIAG) ? 2 : 1);

	__shared__ int s_rp[RCHUNK + 1];
	__shared__ int s_dg[RCHUNK];
	__shared__ int s_col[CAP];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const int slot = t / SUB, u = t % SUB;
	const int r = u % BSP, c = u / BSP;
	const bool active = (r < BS) && (c < BS);
	const int e = RM ? r * BS + c : c * BS + r;
	const int gbase = lane & ~(G - 1);

	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x, (unsigned)(a.xcd_shift < 0 ? 4 : a.xcd_shift));
	const long lin0 = (long)chunk * RCHUNK;
	const int rc = (int)((nb - lin0) < RCHUNK ? (nb - lin0) : RCHUNK);
	const int r0 = a.descending ? (int)(nb - lin0 - rc) : (int)lin0;  // rows [r0, r0 + rc)

	int jlo, jhi;
	stage_chunk_indices<PART, RCHUNK, CAP>(a.pat, r0, rc, tid, s_rp, s_dg, s_col, jlo, jhi);

	// wave-uniform 64-bit bases + 32-bit per-lane byte offsets (limits checked by sweep_offsets_fit)
	const char *const vbase = reinterpret_cast<const char *>(a.vals + (long)jlo * BS2);
	const char *const xbase = reinterpret_cast<const char *>(a.xin);
	const char *const rbase = reinterpret_cast<const char *>(a.rhs + (long)r0 * BS);
	const char *const sbase = reinterpret_cast<const char *>(a.rscale + (long)r0 * BS);
	const char *const dbase = reinterpret_cast<const char *>(a.dvals + (long)r0 * BS2);
	char *const obase = reinterpret_cast<char *>(a.xout + (long)r0 * BS);

	for (int step = 0; step < RCHUNK / RSTEP; step += UNR) {
		// UNR row steps are issued together: all their loads are in flight before the first use
		bool ok[UNR];
		int lr[UNR], dg[UNR], jbeg[UNR], jend[UNR];
#pragma unroll
		for (int q = 0; q < UNR; q++) {
			const int ls = (step + q) * RSTEP + wave * RPW + g;  // position in sweep order
			ok[q] = ls < rc;
			lr[q] = ok[q] ? (a.descending ? rc - 1 - ls : ls) : 0;
			const int rp0 = s_rp[lr[q]], rp1 = s_rp[lr[q] + 1];
			dg[q] = s_dg[lr[q]];
			jbeg[q] = 0;
			jend[q] = 0;
			if (ok[q]) {
				if (PART == PART_LOWER) {
					jbeg[q] = rp0;
					jend[q] = dg[q];
				} else if (PART == PART_UPPER) {
					// the diagonal item rides in front of the upper ones when D is the factor's own diagonal
					jbeg[q] = (DSRC == D_VALS_DIAG || DSRC == D_RECIP_DIAG) ? dg[q] : dg[q] + 1;
					jend[q] = rp1;
				} else if (PART == PART_OFFDIAG || PART == PART_ALL) {
					jbeg[q] = rp0;
					jend[q] = rp1;
				}
			}
		}

		double bv[UNR][KFIX], xv[UNR][KFIX], d[UNR], rv[UNR], dz1[UNR];
#pragma unroll
		for (int q = 0; q < UNR; q++) {
#pragma unroll
			for (int k = 0; k < KFIX; k++) {
				const int jj = jbeg[q] + slot + k * NB;
				bv[q][k] = 0.0;
				xv[q][k] = 0.0;
				// (relaxation: the diagonal block of A is not part of the sum -- do not fetch it)
				if (PART != PART_NONE && jj < jend[q] && active && !(PART == PART_OFFDIAG && jj == dg[q])) {
					bv[q][k] = *reinterpret_cast<const double *>(
					    vbase + ((unsigned)(jj - jlo) * (unsigned)(BS2 * 8) + 8u * (unsigned)e));
					const bool isdiag = (jj == dg[q]);
					if (!((PART == PART_UPPER && (DSRC == D_VALS_DIAG || DSRC == D_RECIP_DIAG) && isdiag) ||
					      (PART == PART_OFFDIAG && isdiag))) {
						const int cidx = jj - jlo;
						const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
						xv[q][k] = *reinterpret_cast<const double *>(
						    xbase + ((unsigned)col * (unsigned)(BS * 8) + 8u * (unsigned)c));
					}
				}
			}
			d[q] = 0.0;
			if (DSRC == D_DBLOCKS && ok[q] && active && slot == 0)
				d[q] = *reinterpret_cast<const double *>(dbase + ((unsigned)lr[q] * (unsigned)(BS2 * 8) + 8u * (unsigned)e));
			// (Z1, scalar lower sweep: the row's u_ii from the contiguous copy of the factor's diagonal, for the fused
			// first upper sweep)
			double du = 1.0;
			if (Z1 && ok[q] && slot == 0)
				du = *reinterpret_cast<const double *>(dbase + ((unsigned)lr[q] * 8u));
			dz1[q] = du;
			rv[q] = 0.0;
			if (ok[q] && r < BS && a.rhs) {
				rv[q] = *reinterpret_cast<const double *>(rbase + ((unsigned)lr[q] * (unsigned)(BS * 8) + 8u * (unsigned)r));
				if (a.rscale)
					rv[q] *= *reinterpret_cast<const double *>(sbase + ((unsigned)lr[q] * (unsigned)(BS * 8) + 8u * (unsigned)r));
			}
		}

#pragma unroll
		for (int q = 0; q < UNR; q++) {
			double acc = 0.0;
			if (PART != PART_NONE) {
#pragma unroll
				for (int k = 0; k < KFIX; k++) {
					if (PART == PART_UPPER && (DSRC == D_VALS_DIAG || DSRC == D_RECIP_DIAG) && k == 0) {
						const bool isd = (slot == 0);  // item 0 of the row: its diagonal block / entry
						if (DSRC == D_VALS_DIAG)
							d[q] = isd ? bv[q][0] : d[q];
						else
							d[q] = (isd && ok[q]) ? 1.0 / bv[q][0] : d[q];
						acc += isd ? 0.0 : bv[q][0] * xv[q][0];
					} else
						acc += bv[q][k] * xv[q][k];  // xv is zero for skipped / absent items
				}
				// longer rows: the rest in groups of KGRP predicated straight-line passes -- one memory round
				// trip per group instead of one per pass
				constexpr int KGRP = 4;
				for (int jb = jbeg[q] + slot + KFIX * NB; jb < jend[q]; jb += KGRP * NB) {
					double v4[KGRP], x4[KGRP];
#pragma unroll
					for (int k = 0; k < KGRP; k++) {
						const int jj = jb + k * NB;
						v4[k] = 0.0;
						x4[k] = 0.0;
						if (jj < jend[q] && active && !(PART == PART_OFFDIAG && jj == dg[q])) {
							const int cidx = jj - jlo;
							const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
							v4[k] = *reinterpret_cast<const double *>(
							    vbase + ((unsigned)(jj - jlo) * (unsigned)(BS2 * 8) + 8u * (unsigned)e));
							x4[k] = *reinterpret_cast<const double *>(
							    xbase + ((unsigned)col * (unsigned)(BS * 8) + 8u * (unsigned)c));
						}
					}
#pragma unroll
					for (int k = 0; k < KGRP; k++)
						acc += v4[k] * x4[k];
				}
				acc = allreduce_bits<Ge::LOBIT, Ge::HIBIT>(acc);  // lanes (r,*,*) now hold row r of the sum
			}

			double out;
			if (POST == POST_SUB) {
				out = rv[q] - acc;
			} else if (POST == POST_D_SUB || POST == POST_SUB_D) {
				const double w = (POST == POST_D_SUB) ? rv[q] - acc : acc;
				const double wc = __shfl(w, gbase + c, 64);  // lane (c,0) of block slot 0 holds component c
				// D lives in block slot 0 only (zero elsewhere): the all-reduce is its column sum
				const double p = allreduce_bits<Ge::LOBIT, Ge::HIBIT>((active && slot == 0) ? d[q] * wc : 0.0);
				out = (POST == POST_D_SUB) ? p : rv[q] - p;
			} else {
				out = a.a * acc;
				if (a.b != 0.0)
					out += a.b * rv[q];
			}

			if (ok[q] && slot == 0 && c == 0 && r < BS) {
				double *const dst = reinterpret_cast<double *>(obase + ((unsigned)lr[q] * (unsigned)(BS * 8) + 8u * (unsigned)r));
				*dst = out;
				// the arithmetic of the upper sweep with D_RECIP_DIAG whose gathered iterate is all zeros: d = 1 / u_ii,
				// z = d * (y - 0)
				if (Z1)
					a.z1out[(long)r0 + lr[q]] = (1.0 / dz1[q]) * out;
			}
		}
	}
}

static int g_sweep_unroll = 0;  // 0 = default (2 where it applies), 1 = never unroll (measurements), 2 = scalar rows too
void set_sweep_unroll(int u)
{
	g_sweep_unroll = u;
}

template <int BS, bool RM>
static void dispatch_ops(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + Geo<BS>::RCHUNK - 1) / Geo<BS>::RCHUNK);
	if (grid == 0)
		return;
	// scalar rows, many of them: 256 rows per workgroup (sweep_geo.hpp)
	const bool big = BS == 1 && a.pat.nbrows >= (1 << 20);
	const unsigned gridbig = (unsigned)(((long)a.pat.nbrows + Geo<BS, true>::RCHUNK - 1) / Geo<BS, true>::RCHUNK);
#define BHIP_CASE(P, Q, D)                                                                       \
	if (part == P && post == Q && dsrc == D) {                                                   \
		/* one block-row per wave (bs >= 5): two row steps in flight for the triangular sweeps */ \
		constexpr int U = (Geo<BS>::G == 64 && (P == PART_LOWER || P == PART_UPPER)) ? 2 : 1;    \
		if (BS == 1 && big && g_sweep_unroll >= 2)                                               \
			hipLaunchKernelGGL((sweep_kernel<BS, RM, P, Q, D, (BS == 1 ? 4 : 1), BS == 1>), dim3(gridbig), dim3(256), 0, s, a); \
		else if (BS == 1 && g_sweep_unroll >= 2)                                                 \
			hipLaunchKernelGGL((sweep_kernel<BS, RM, P, Q, D, (BS == 1 ? 2 : 1)>), dim3(grid), dim3(256), 0, s, a); \
		else if (BS == 1 && big && g_sweep_unroll != 1)                                          \
			hipLaunchKernelGGL((sweep_kernel<BS, RM, P, Q, D, U, BS == 1>), dim3(gridbig), dim3(256), 0, s, a); \
		else if (g_sweep_unroll == 1)                                                            \
			hipLaunchKernelGGL((sweep_kernel<BS, RM, P, Q, D, 1>), dim3(grid), dim3(256), 0, s, a); \
		else                                                                                     \
			hipLaunchKernelGGL((sweep_kernel<BS, RM, P, Q, D, U>), dim3(grid), dim3(256), 0, s, a); \
		return;                                                                                  \
	}
	BHIP_CASE(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASE(PART_UPPER, POST_SUB, D_NONE)  // out = rhs - U x, strictly upper part (exact relaxation passes)
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

// 32-bit byte offsets inside a chunk: the vector and the blocks of one 256-row chunk must stay below
// 4 GiB.  Checked once per pattern in set_pattern (capi.hip).
bool sweep_offsets_fit(const Pattern &pat)
{
	const long blkbytes = (long)pat.bs * pat.bs * 8;
	return (long)pat.nbrows * pat.bs * 8 < (1L << 32) && 256L * pat.max_row_len * blkbytes < (1L << 32);
}

void launch_sweep(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	if (a.pat.nbrows == 0)
		return;  // an empty subdomain
	if (a.z1out) {
		// the scalar lower sweep with the first upper sweep fused into its stores (capi.hip, small applications): a.dvals
		// is the factor's diagonal, one entry per row
		if (a.pat.bs != 1 || part != PART_LOWER || post != POST_SUB || dsrc != D_NONE || a.pat.nbrows >= (1 << 20) || !a.dvals)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "launch_sweep: fused second output for this operator");
		const unsigned grid = (unsigned)(((long)a.pat.nbrows + Geo<1>::RCHUNK - 1) / Geo<1>::RCHUNK);
		hipLaunchKernelGGL((sweep_kernel<1, false, PART_LOWER, POST_SUB, D_NONE, 1, false, true>), dim3(grid), dim3(256), 0, s, a);
		BHIP_CHECK(hipGetLastError());
		return;
	}
	if (launch_sweepw(a, part, post, dsrc, s))
		return;
	if (launch_sweepodd(a, part, post, dsrc, s))
		return;
	if (launch_sweepwr(a, part, post, dsrc, s))
		return;
	if (launch_sweep1(a, part, post, dsrc, s))
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
