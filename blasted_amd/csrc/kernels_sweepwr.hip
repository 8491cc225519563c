// kernels_sweepwr.hip -- the tuned row-sweep kernel for ROW-major blocks of size 4 and 8 (the reference
// instantiates bs = 4 RowMajor, src/solverops_ilu0.cpp:385-395; the general kernel of kernels_sweep.hip
// moves 8 bytes per lane).  Same operators, staging and chunking as kernels_sweepw.hip; what the layout
// changes: lane q of a block holds entries (2q, 2q+1) = row r = q / (bs/2), columns 2h, 2h+1 with
// h = q % (bs/2).  So the x segment is a 16-byte pair (x_2h, x_2h+1), a lane accumulates one number
// (its row's partial sum over two columns), the all-reduce runs over the h bits only, and rhs / result
// are one double per lane; the diagonal-block product needs w_2h, w_2h+1 from the lanes of rows 2h, 2h+1.
#include "ctx.hpp"
#include "lanes.hpp"
#include "stage.hpp"

#include <cstdlib>
#include <cstring>

namespace bhip {

namespace {

typedef double rd2_t __attribute__((ext_vector_type(2)));

template <int BS>
struct RGeo {
	static constexpr int HB = BS / 2;       // lanes per block row
	static constexpr int LPB = BS * HB;     // lanes per block = lanes per block-row (one slot)
	static constexpr int G = LPB;           // 8 (bs=4), 32 (bs=8)
	static constexpr int RPW = 64 / G;
	static constexpr int RSTEP = 4 * RPW;
	static constexpr int HBITS = HB == 2 ? 1 : 2;
	static constexpr int BLKBYTES = BS * BS * 8;
	static constexpr int ROWBYTES = BS * 8;
};

// LS ("late store", round 4; what kernels_sweepw.hip has had since round 3 for column-major blocks): in-place bs = 4
// triangular sweeps collect a workgroup's results in registers and store its 128 rows once, after the last step --
// lane 2j + m of a row group keeps components (2m, 2m+1) of the row the group computed in step j (two lane permutes
// per step), one 16-byte store per lane at the end instead of four 8-byte ones per step.  Other workgroups see a
// chunk's rows up to three steps later: the more repeatable operator (see kernels_sweepw.hip for what that is worth
// inside a flexible solver).  One row step in flight.
template <int BS, int PART, int POST, int DSRC, int RCHUNK, bool LS = false>
__global__ __launch_bounds__(256) void sweepwr_kernel(const SweepArgs a)
{
	static_assert(!LS || (BS == 4 && RCHUNK == 128 && (PART == PART_LOWER || PART == PART_UPPER)), "late store: bs 4 triangular sweeps");
	using Ge = RGeo<BS>;
	constexpr int HB = Ge::HB, G = Ge::G, RPW = Ge::RPW, RSTEP = Ge::RSTEP;
	constexpr int CAP = 8 * RCHUNK;
	constexpr int KFIX = 4 * ((PART == PART_ALL || PART == PART_OFFDIAG) ? 2 : 1);
	constexpr bool DIAG_RIDES = PART == PART_UPPER && DSRC == D_VALS_DIAG;
	static_assert(BS == 4 || BS == 8, "wide kernel: bs 4 or 8");
	static_assert(RCHUNK % RSTEP == 0, "chunk must be a multiple of the row step");

	__shared__ int s_rp[RCHUNK + 1];
	__shared__ int s_dg[RCHUNK];
	__shared__ int s_col[CAP];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, q = lane % G;
	const int r = q / HB, h = q % HB;  // row of this lane's two entries, column pair (2h, 2h+1)
	const int gbase = lane & ~(G - 1);

	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x, (unsigned)(a.xcd_shift < 0 ? 4 : a.xcd_shift));
	const long lin0 = (long)chunk * RCHUNK;
	const int rc = (int)((nb - lin0) < RCHUNK ? (nb - lin0) : RCHUNK);
	const int r0 = a.descending ? (int)(nb - lin0 - rc) : (int)lin0;

	int jlo, jhi;
	stage_chunk_indices<PART, RCHUNK, CAP>(a.pat, r0, rc, tid, s_rp, s_dg, s_col, jlo, jhi);

	const char *const vbase = reinterpret_cast<const char *>(a.vals + (long)jlo * (BS * BS));
	const char *const xbase = reinterpret_cast<const char *>(a.xin);
	const char *const rbase = reinterpret_cast<const char *>(a.rhs + (long)r0 * BS);
	const char *const sbase = reinterpret_cast<const char *>(a.rscale + (long)r0 * BS);
	const char *const dbase = reinterpret_cast<const char *>(a.dvals + (long)r0 * (BS * BS));
	char *const obase = reinterpret_cast<char *>(a.xout + (long)r0 * BS);

	rd2_t late;
	late.x = late.y = 0.0;
	for (int step = 0; step < RCHUNK / RSTEP; step++) {
		const int ls = step * RSTEP + wave * RPW + g;
		const bool ok = ls < rc;
		const int lr = ok ? (a.descending ? rc - 1 - ls : ls) : 0;
		const int rp0 = s_rp[lr], rp1 = s_rp[lr + 1], dg = s_dg[lr];
		int jbeg = 0, jend = 0;
		if (ok) {
			if (PART == PART_LOWER) {
				jbeg = rp0;
				jend = dg;
			} else if (PART == PART_UPPER) {
				jbeg = DIAG_RIDES ? dg : dg + 1;
				jend = rp1;
			} else if (PART == PART_OFFDIAG || PART == PART_ALL) {
				jbeg = rp0;
				jend = rp1;
			}
		}

		rd2_t bv[KFIX], xv[KFIX];
#pragma unroll
		for (int k = 0; k < KFIX; k++) {
			const int jj = jbeg + k;
			bv[k].x = bv[k].y = 0.0;
			xv[k].x = xv[k].y = 0.0;
			if (PART != PART_NONE && jj < jend && !(PART == PART_OFFDIAG && jj == dg)) {
				bv[k] = __builtin_nontemporal_load(reinterpret_cast<const rd2_t *>(
				    vbase + ((unsigned)(jj - jlo) * (unsigned)Ge::BLKBYTES + 16u * (unsigned)q)));
				if (!(DIAG_RIDES && jj == dg)) {
					const int cidx = jj - jlo;
					const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
					xv[k] = *reinterpret_cast<const rd2_t *>(
					    xbase + ((unsigned)col * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
				}
			}
		}
		rd2_t dv;
		dv.x = dv.y = 0.0;
		if (DSRC == D_DBLOCKS && ok)
			dv = *reinterpret_cast<const rd2_t *>(dbase + ((unsigned)lr * (unsigned)Ge::BLKBYTES + 16u * (unsigned)q));
		double rv = 0.0;
		if (ok && a.rhs) {
			rv = *reinterpret_cast<const double *>(rbase + ((unsigned)lr * (unsigned)Ge::ROWBYTES + 8u * (unsigned)r));
			if (a.rscale)
				rv *= *reinterpret_cast<const double *>(sbase + ((unsigned)lr * (unsigned)Ge::ROWBYTES + 8u * (unsigned)r));
		}

		double acc = 0.0;
		if (PART != PART_NONE) {
#pragma unroll
			for (int k = 0; k < KFIX; k++) {
				if (DIAG_RIDES && k == 0)
					dv = bv[0];  // item 0 of the row is its (inverted) diagonal block; xv[0] is zero
				else
					acc += bv[k].x * xv[k].x + bv[k].y * xv[k].y;
			}
			for (int jj = jbeg + KFIX; jj < jend; jj++) {
				if (PART == PART_OFFDIAG && jj == dg)
					continue;
				const rd2_t v2 = __builtin_nontemporal_load(reinterpret_cast<const rd2_t *>(
				    vbase + ((unsigned)(jj - jlo) * (unsigned)Ge::BLKBYTES + 16u * (unsigned)q)));
				const int cidx = jj - jlo;
				const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
				const rd2_t x2 = *reinterpret_cast<const rd2_t *>(
				    xbase + ((unsigned)col * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
				acc += v2.x * x2.x + v2.y * x2.y;
			}
			acc = allreduce_bits<0, Ge::HBITS>(acc);  // over the column-pair bits: component r of the sum
		}

		double out;
		if (POST == POST_SUB) {
			out = rv - acc;
		} else if (POST == POST_D_SUB || POST == POST_SUB_D) {
			const double w = (POST == POST_D_SUB) ? rv - acc : acc;  // component r, in every lane of row r
			const double w0 = __shfl(w, gbase + (2 * h) * HB, 64);
			const double w1 = __shfl(w, gbase + (2 * h + 1) * HB, 64);
			const double pr = allreduce_bits<0, Ge::HBITS>(dv.x * w0 + dv.y * w1);
			out = (POST == POST_D_SUB) ? pr : rv - pr;
		} else {
			out = a.a * acc;
			if (a.b != 0.0)
				out += a.b * rv;
		}
		if (LS) {
			// components (2m, 2m+1), m = q % 2, of this step's row: held by the lanes of rows 2m and 2m+1 of the group
			const double v0 = __shfl(out, gbase + 4 * (q & 1), 64), v1 = __shfl(out, gbase + 4 * (q & 1) + 2, 64);
			if ((q >> 1) == step) {
				late.x = v0;
				late.y = v1;
			}
		} else if (ok && h == 0)
			*reinterpret_cast<double *>(obase + ((unsigned)lr * (unsigned)Ge::ROWBYTES + 8u * (unsigned)r)) = out;
	}
	if (LS) {
		const int ls = (q >> 1) * RSTEP + wave * RPW + g;  // the row this lane's piece belongs to: step q / 2 of its group
		if (ls < rc) {
			const int lr = a.descending ? rc - 1 - ls : ls;
			*reinterpret_cast<rd2_t *>(obase + ((unsigned)lr * (unsigned)Ge::ROWBYTES + 16u * (unsigned)(q & 1))) = late;
		}
	}
}

int g_sweepwr_enabled = [] {
	const char *e = std::getenv("BLASTED_HIP_SWEEPWR");
	return (e && std::strcmp(e, "0") == 0) ? 0 : 1;
}();

template <int BS, int PART, int POST, int DSRC>
void launch_r(const SweepArgs &a, hipStream_t s)
{
	constexpr int RCHUNK = 128;
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + RCHUNK - 1) / RCHUNK);
	constexpr bool LSOK = BS == 4 && (PART == PART_LOWER || PART == PART_UPPER);
	if (LSOK && a.latestore && !a.interleave && a.xin == a.xout &&
	    (reinterpret_cast<uintptr_t>(a.xout) & 15u) == 0)
		hipLaunchKernelGGL((sweepwr_kernel<BS, PART, POST, DSRC, RCHUNK, LSOK>), dim3(grid), dim3(256), 0, s, a);
	else
		hipLaunchKernelGGL((sweepwr_kernel<BS, PART, POST, DSRC, RCHUNK>), dim3(grid), dim3(256), 0, s, a);
}

template <int BS>
bool launch_r_bs(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
#define BHIP_CASER(P, Q, D)                           \
	if (part == P && post == Q && dsrc == D) {        \
		launch_r<BS, P, Q, D>(a, s);                  \
		return true;                                  \
	}
	BHIP_CASER(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASER(PART_UPPER, POST_D_SUB, D_VALS_DIAG)
	BHIP_CASER(PART_LOWER, POST_D_SUB, D_DBLOCKS)
	BHIP_CASER(PART_UPPER, POST_SUB_D, D_DBLOCKS)
	BHIP_CASER(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS)
	BHIP_CASER(PART_ALL, POST_AXPBY, D_NONE)
	BHIP_CASER(PART_NONE, POST_D_SUB, D_DBLOCKS)
#undef BHIP_CASER
	return false;
}

}  // namespace

void set_sweepwr_enabled(int on)
{
	g_sweepwr_enabled = on;
}

// returns false when the tuned kernel does not cover the request (caller uses the generic family)
bool launch_sweepwr(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	const int bs = a.pat.bs;
	if (!g_sweepwr_enabled || (bs != 4 && bs != 8) || !a.pat.rowmajor || a.pat.nbrows == 0)
		return false;
	auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
	if (misaligned(a.vals) || misaligned(a.dvals) || misaligned(a.xin))
		return false;
	const bool ok = (bs == 4) ? launch_r_bs<4>(a, part, post, dsrc, s) : launch_r_bs<8>(a, part, post, dsrc, s);
	if (ok)
		BHIP_CHECK(hipGetLastError());
	return ok;
}

}  // namespace bhip
