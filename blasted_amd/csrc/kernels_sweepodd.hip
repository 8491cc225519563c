// kernels_sweepodd.hip -- the tuned row-sweep kernel for column-major blocks of odd size 3, 5, 7.
// bs = 5 is the reference's other stock block size (src/solverops_ilu0.cpp:385-395, the 3-D
// compressible-flow case its author targets) and BASELINE.json's config 4; 3 and 7 are the sizes of its
// SpMV fixtures.  Same operators and arithmetic as the generic family in kernels_sweep.hip (table
// there); the generic kernel pads a block to a power-of-two lane tile, so one wave load instruction
// moves a single 5x5 (7x7) block or four 3x3 blocks -- this kernel moves 4 (2, 8) blocks, 16 bytes per lane:
//
//  * G = 8 / 16 / 32 lanes own one block-row, L = (bs*bs+1)/2 = 5 / 13 / 25 of them one stored block
//    per pass: lane q < L-1 reads entries (2q, 2q+1) as 16 bytes, the last lane reads the block's last
//    two entries and uses the second.  A block starts at a multiple of 8*bs*bs bytes, so half of these
//    loads are 8- but not 16-byte aligned (gfx950 global loads take that);
//  * the x segment is gathered as 16 bytes per lane too: the two entries of a lane lie in column c or
//    in columns c, c+1 of the block, so (x_c, x_{c+1}) covers both (last column: (x_{bs-2}, x_{bs-1}));
//  * entry e = bs*c + r contributes to component r, which follows no power-of-two lane pattern: the
//    partial products of a row go through a (bs*bs)-double LDS tile and bs lanes per row sum their
//    component's column (wave-private tile: LDS executes a wave's instructions in order, no barrier);
//    the diagonal-block product of the upper solve / Gauss-Seidel goes the same way;
//  * row chunk, LDS-staged browptr / diagind / bcolind range, straight-line predicated block passes,
//    32-bit chunk-relative offsets and XCD-contiguous chunk numbering as in kernels_sweepw.hip.
#include "ctx.hpp"
#include "lanes.hpp"
#include "stage.hpp"

#include <cstdlib>
#include <cstring>

namespace bhip {

namespace {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef d2_t d2u_t __attribute__((aligned(8)));  // 16-byte access at 8-byte alignment

__device__ __forceinline__ d2_t load16u(const char *p)
{
	return *reinterpret_cast<const d2u_t *>(p);
}

__device__ __forceinline__ d2_t load16u_nt(const char *p)
{
	return __builtin_nontemporal_load(reinterpret_cast<const d2u_t *>(p));
}

// NT: the block stream is read with non-temporal loads (tuning "sweepodd=nt0" / "sweepodd=nt1").
// OCC: waves per SIMD the register allocation must leave room for (second launch bound).  Left alone the
// compiler spends 78-86 VGPRs on the triangular sweeps (5 waves per SIMD); asked for 6 it needs 62-70 without a
// spill, i.e. 7-8 resident waves -- and resident waves are bytes in flight, which is what bounds this kernel.
// (Round 3, built, verified -- 955 parity / level / fuzz tests green -- measured and removed: the late store of
// kernels_sweepw.hip for this kernel, G / bs steps sharing one result register through a lane permute and the
// workgroup's rows stored once after the last step.  8-10 more registers = one wave per SIMD less: the unstructured
// sweeps 0.9 % and Poisson 128^3 bs=5 1.6 % SLOWER; two applications differ by 6.4e-3 instead of 1.9e-2 and GCR(30)
// needs 310 instead of 337 iterations at 3 sweeps, 193 against 191 at 5 (100^3 bs=5): too little either way.
// Round 4, measured and removed: the rows of a chunk taken in the order of their length, so that the rows of a wave
// need the same number of load rounds (config 4: a wave's four rows average 7 blocks where its longest has 11-12) --
// 3-4 % slower, profiles/r04_sweepodd_balanced_ab.txt; six or eight straight-line block passes instead of four for the
// triangular sweeps (70-79 / 86-95 registers): +0.6 % / -7 % on config 4, same file.)
// RM (round 4): ROW-major blocks.  The lanes hold the same 16-byte pieces of a block; entry e of the stored image is
// then A(e / bs, e % bs): it multiplies x_{e % bs} and belongs to component e / bs, i.e. the roles of the two indices
// swap -- the partial products of a component are CONTIGUOUS in the tile, and the x pair a lane gathers is
// (x_{e % bs}, x_{e % bs + 1}) except where its second entry wraps into the next row of the block and wants x_0, which
// the group's first lane holds (one lane permute per pass).
template <int BS, int PART, int POST, int DSRC, int RCHUNK, bool NT = true, int OCC = 1, bool RM = false>
__global__ __launch_bounds__(256, OCC) void sweepodd_kernel(const SweepArgs a)
{
	static_assert(BS == 3 || BS == 5 || BS == 7, "odd block sizes 3, 5, 7");
	constexpr int BS2 = BS * BS, L = (BS2 + 1) / 2;     // lanes that hold a block
	constexpr int G = BS == 3 ? 8 : (BS == 5 ? 16 : 32);  // lanes per block-row
	constexpr int RPW = 64 / G, RSTEP = 4 * RPW, CAP = (BS == 3 ? 8 : 16) * RCHUNK;
	constexpr int BLKBYTES = BS2 * 8, ROWBYTES = BS * 8;
	static_assert(RCHUNK % RSTEP == 0, "chunk must be a multiple of the row step");
	constexpr int KFIX = (PART == PART_ALL || PART == PART_OFFDIAG) ? 8 : 4;
	constexpr bool DIAG_RIDES = PART == PART_UPPER && DSRC == D_VALS_DIAG;
	constexpr bool USES_D = POST == POST_D_SUB || POST == POST_SUB_D;

	__shared__ int s_rp[RCHUNK + 1];
	__shared__ int s_dg[RCHUNK];
	__shared__ int s_col[CAP];
	__shared__ double s_acc[4][RPW][BS2 + 1];  // [wave][row of the step][entry]
	__shared__ double s_w[4][RPW][BS + 1];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const bool actA = t < L - 1, actB = t < L;
	const int eA = actA ? 2 * t : BS2 - 2, eB = actA ? 2 * t + 1 : BS2 - 1;
	const unsigned boff = actA ? 16u * (unsigned)t : 8u * (unsigned)(BS2 - 2);  // last lane: the last two entries
	// the index of x (and of the vector D multiplies) that this lane's two entries go with: their column, row-major their
	// position inside the row of the block
	const int cA = RM ? eA % BS : ((eA / BS) < BS ? eA / BS : BS - 1), cB = RM ? eB % BS : ((eB / BS) < BS ? eB / BS : BS - 1);
	const int cx = cA < BS - 2 ? cA : BS - 2;  // gathered pair (x_cx, x_cx+1)
	const bool hiA = cA != cx, hiB = cB != cx;
	const bool wrapB = RM && cB == 0 && cA == BS - 1;  // second entry = first of the block's next row: x_0
	const int lane0 = (threadIdx.x & 63) & ~(G - 1);   // the group's first lane (entries 0, 1: its pair starts at x_0)

	const int nb = a.pat.nbrows;
	// (default here: the XCDs take turns on 64 chunks = 8192 rows, not 16 -- on the unstructured configuration an XCD
	// then owns whole windows of the numbering and its L2 serves more of the gathers: lower / upper sweep 0.706 / 0.826 ->
	// 0.674 / 0.804 ms, no change on stencil matrices, none in GCR iterations: profiles/r03_xcdsuper.txt)
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x, (unsigned)(a.xcd_shift < 0 ? 6 : a.xcd_shift));
	const long lin0 = (long)chunk * RCHUNK;
	const int rc = (int)((nb - lin0) < RCHUNK ? (nb - lin0) : RCHUNK);
	const int r0 = a.descending ? (int)(nb - lin0 - rc) : (int)lin0;

	int jlo, jhi;
	stage_chunk_indices<PART, RCHUNK, CAP>(a.pat, r0, rc, tid, s_rp, s_dg, s_col, jlo, jhi);

	const char *const vbase = reinterpret_cast<const char *>(a.vals + (long)jlo * BS2);
	const char *const xbase = reinterpret_cast<const char *>(a.xin);
	const char *const rbase = reinterpret_cast<const char *>(a.rhs + (long)r0 * BS);
	const char *const sbase = reinterpret_cast<const char *>(a.rscale + (long)r0 * BS);
	const char *const dbase = reinterpret_cast<const char *>(a.dvals + (long)r0 * BS2);
	char *const obase = reinterpret_cast<char *>(a.xout + (long)r0 * BS);
	double *const tile = &s_acc[wave][g][0];
	double *const wvec = &s_w[wave][g][0];

	// One row step = the loads of a step (index look-up, KFIX straight-line block passes with their x pairs,
	// D block, right-hand side), the multiply-accumulate, and the reduction through the wave-private LDS tile.
	// (Tried in round 2: software-pipelined steps -- the loads of step s+1 issued into the freed load registers
	// before the LDS reduction and the store of step s.  Bit-identical for synchronous sweeps and no faster on
	// any of bs 3 / 5 / 7 (+-0.3 %), while the state carried across iterations cost 14 VGPRs = one wave per
	// SIMD and 7 % on the unstructured case: what bounds this kernel is bytes in flight per CU, i.e. resident
	// waves, not the LDS round trips.)
	struct Step {
		d2_t bv[KFIX], xv[KFIX];
		d2_t dv;
		double rv;
		int jbeg, jend, dg, lr;
		bool ok;
	};
	auto issue = [&](const int step, Step &q) {
		const int ls = step * RSTEP + wave * RPW + g;  // position in sweep order
		q.ok = ls < rc;
		q.lr = q.ok ? (a.descending ? rc - 1 - ls : ls) : 0;
		const int rp0 = s_rp[q.lr], rp1 = s_rp[q.lr + 1];
		q.dg = s_dg[q.lr];
		q.jbeg = q.jend = 0;
		if (q.ok) {
			if (PART == PART_LOWER) {
				q.jbeg = rp0;
				q.jend = q.dg;
			} else if (PART == PART_UPPER) {
				q.jbeg = DIAG_RIDES ? q.dg : q.dg + 1;  // first item = the (inverted) diagonal block
				q.jend = rp1;
			} else if (PART == PART_OFFDIAG || PART == PART_ALL) {
				q.jbeg = rp0;
				q.jend = rp1;
			}
		}
#pragma unroll
		for (int k = 0; k < KFIX; k++) {
			const int jj = q.jbeg + k;
			q.bv[k].x = q.bv[k].y = 0.0;
			q.xv[k].x = q.xv[k].y = 0.0;
			// (relaxation: the diagonal block of A is not part of the sum -- do not fetch it)
			if (PART != PART_NONE && jj < q.jend && actB && !(PART == PART_OFFDIAG && jj == q.dg)) {
				q.bv[k] = NT ? load16u_nt(vbase + ((unsigned)(jj - jlo) * (unsigned)BLKBYTES + boff))
				             : load16u(vbase + ((unsigned)(jj - jlo) * (unsigned)BLKBYTES + boff));
				const bool isdiag = (jj == q.dg);
				if (!((DIAG_RIDES && isdiag) || (PART == PART_OFFDIAG && isdiag))) {
					const int cidx = jj - jlo;
					int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
					if (BHIP_PROBE(a.probe) == 1)
						col = r0 + q.lr;  // (timing experiment: no gather, the row's own segment)
					q.xv[k] = load16u(xbase + ((unsigned)col * (unsigned)ROWBYTES + 8u * (unsigned)cx));
				}
			}
		}
		q.dv.x = q.dv.y = 0.0;
		if (DSRC == D_DBLOCKS && q.ok && actB)
			q.dv = load16u(dbase + ((unsigned)q.lr * (unsigned)BLKBYTES + boff));
		q.rv = 0.0;
		if (q.ok && t < BS && a.rhs) {
			q.rv = *reinterpret_cast<const double *>(rbase + ((unsigned)q.lr * (unsigned)ROWBYTES + 8u * (unsigned)t));
			if (a.rscale)
				q.rv *= *reinterpret_cast<const double *>(sbase + ((unsigned)q.lr * (unsigned)ROWBYTES + 8u * (unsigned)t));
		}
	};

	constexpr int NSTEPS = RCHUNK / RSTEP;
	for (int step = 0; step < NSTEPS; step++) {
		Step q;
		issue(step, q);
		// ---- products of this step (consumes the load registers)
		double accA = 0.0, accB = 0.0;
		d2_t dv = q.dv;
		if (PART != PART_NONE) {
#pragma unroll
			for (int k = 0; k < KFIX; k++) {
				if (DIAG_RIDES && k == 0) {
					dv = q.bv[0];  // item 0 of the row is its diagonal block: xv[0] was not loaded (zero)
				} else {
					const double x0 = RM ? __shfl(q.xv[k].x, lane0, 64) : 0.0;
					accA += q.bv[k].x * (hiA ? q.xv[k].y : q.xv[k].x);
					accB += q.bv[k].y * (wrapB ? x0 : (hiB ? q.xv[k].y : q.xv[k].x));
				}
			}
			// rows longer than KFIX blocks (unstructured meshes: ~7 lower and ~8 diagonal+upper blocks at 14
			// neighbours): the rest goes in groups of KGRP predicated straight-line passes, one memory round
			// trip per group -- one block per iteration made every extra block a dependent round trip
			constexpr int KGRP = 4;
			for (int jb = q.jbeg + KFIX; jb < q.jend; jb += KGRP) {
				d2_t v4[KGRP], x4[KGRP];
#pragma unroll
				for (int k = 0; k < KGRP; k++) {
					const int jj = jb + k;
					v4[k].x = v4[k].y = 0.0;
					x4[k].x = x4[k].y = 0.0;
					if (jj < q.jend && actB && !(PART == PART_OFFDIAG && jj == q.dg)) {
						v4[k] = NT ? load16u_nt(vbase + ((unsigned)(jj - jlo) * (unsigned)BLKBYTES + boff))
						           : load16u(vbase + ((unsigned)(jj - jlo) * (unsigned)BLKBYTES + boff));
						const int cidx = jj - jlo;
						int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
						if (BHIP_PROBE(a.probe) == 1)
							col = r0 + q.lr;
						x4[k] = load16u(xbase + ((unsigned)col * (unsigned)ROWBYTES + 8u * (unsigned)cx));
					}
				}
#pragma unroll
				for (int k = 0; k < KGRP; k++) {
					const double x0 = RM ? __shfl(x4[k].x, lane0, 64) : 0.0;
					accA += v4[k].x * (hiA ? x4[k].y : x4[k].x);
					accB += v4[k].y * (wrapB ? x0 : (hiB ? x4[k].y : x4[k].x));
				}
			}
		}
		const double rv = q.rv;
		const bool ok = q.ok;
		const int lr = q.lr;

		// ---- 25 partial products -> 5 components through the wave-private tile
		double sum = 0.0;  // component t of sum_j A_ij x_j, lanes t < BS
		if (PART != PART_NONE) {
			if (actA)
				tile[eA] = accA;
			if (actB)
				tile[eB] = accB;
			__builtin_amdgcn_wave_barrier();
			if (t < BS) {
#pragma unroll
				for (int cc = 0; cc < BS; cc++)
					sum += RM ? tile[t * BS + cc] : tile[cc * BS + t];
			}
			__builtin_amdgcn_wave_barrier();
		}

		double out;
		if (POST == POST_SUB) {
			out = rv - sum;
		} else if (USES_D) {
			const double w = (POST == POST_D_SUB) ? rv - sum : sum;
			if (t < BS)
				wvec[t] = w;
			__builtin_amdgcn_wave_barrier();
			const double wA = wvec[cA], wB = wvec[cB];
			if (actA)
				tile[eA] = dv.x * wA;
			if (actB)
				tile[eB] = dv.y * wB;
			__builtin_amdgcn_wave_barrier();
			double pr = 0.0;
			if (t < BS) {
#pragma unroll
				for (int cc = 0; cc < BS; cc++)
					pr += RM ? tile[t * BS + cc] : tile[cc * BS + t];
			}
			__builtin_amdgcn_wave_barrier();
			out = (POST == POST_D_SUB) ? pr : rv - pr;
		} else {
			out = a.a * sum;
			if (a.b != 0.0)
				out += a.b * rv;
		}

		if (ok && t < BS)
			*reinterpret_cast<double *>(obase + ((unsigned)lr * (unsigned)ROWBYTES + 8u * (unsigned)t)) = out;
	}
}

int g_sweepodd_enabled = [] {
	const char *e = std::getenv("BLASTED_HIP_SWEEPODD");
	return (e && std::strcmp(e, "0") == 0) ? 0 : 1;
}();

// tuning "sweepodd=nt0" (default) / "sweepodd=nt1": block stream with plain / non-temporal loads.  A block of
// 72 / 200 / 392 bytes shares its first and last 128-byte line with its neighbours, which the next load
// instruction of the same wave wants from the caches: same-process A/B, lower + upper sweep in ms, nt1 -> nt0:
// unstructured bs=5 1.627 -> 1.557, Poisson 128^3 bs=3 0.416 -> 0.319, bs=5 0.848 -> 0.784, bs=7 1.438 -> 1.436.
int g_sweepodd_nt = 0;
int g_sweepodd_occ = 1;  // tuning "sweepodd=occ1" (default) / "sweepodd=occ0": occupancy-bounded register allocation

template <int PART, int POST, int DSRC>
void launch5(const SweepArgs &a, hipStream_t s)
{
	constexpr int RCHUNK = 128;
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + RCHUNK - 1) / RCHUNK);
	// (whole-row parts: eight straight-line passes; five waves is what they had before the grouped remainder
	// passes cost them 21 registers -- SpMV at bs=5 0.69 -> 0.78 ms without the bound)
	constexpr int OCCT = (PART == PART_LOWER || PART == PART_UPPER) ? 6 : ((PART == PART_ALL || PART == PART_OFFDIAG) ? 5 : 1);
#define BHIP_ODD(B)                                                                                                    \
	if (a.pat.rowmajor)                                                                                                \
		hipLaunchKernelGGL((sweepodd_kernel<B, PART, POST, DSRC, RCHUNK, false, OCCT, true>), dim3(grid), dim3(256), 0, s, a); \
	else if (g_sweepodd_nt)                                                                                            \
		hipLaunchKernelGGL((sweepodd_kernel<B, PART, POST, DSRC, RCHUNK, true, 1>), dim3(grid), dim3(256), 0, s, a);    \
	else if (g_sweepodd_occ)                                                                                           \
		hipLaunchKernelGGL((sweepodd_kernel<B, PART, POST, DSRC, RCHUNK, false, OCCT>), dim3(grid), dim3(256), 0, s, a); \
	else                                                                                                               \
		hipLaunchKernelGGL((sweepodd_kernel<B, PART, POST, DSRC, RCHUNK, false, 1>), dim3(grid), dim3(256), 0, s, a);
	switch (a.pat.bs) {
	case 3: BHIP_ODD(3) break;
	case 5: BHIP_ODD(5) break;
	default: BHIP_ODD(7) break;
	}
#undef BHIP_ODD
}

}  // namespace

void set_sweepodd_enabled(int on)
{
	if (on == 2 || on == 3)  // "sweepodd=nt1" / "sweepodd=nt0"
		g_sweepodd_nt = on == 2 ? 1 : 0;
	else if (on == 4 || on == 5)  // "sweepodd=occ1" / "sweepodd=occ0"
		g_sweepodd_occ = on == 4 ? 1 : 0;
	else
		g_sweepodd_enabled = on;
}

// returns false when the tuned kernel does not cover the request (caller uses the generic family)
bool launch_sweepodd(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	const int bs = a.pat.bs;
	if (!g_sweepodd_enabled || (bs != 3 && bs != 5 && bs != 7) || a.pat.nbrows == 0)
		return false;
	// 8-byte aligned arrays are all this kernel needs
	bool ok = true;
#define BHIP_CASE5(P, Q, D)                           \
	if (part == P && post == Q && dsrc == D)          \
		launch5<P, Q, D>(a, s);                       \
	else
	BHIP_CASE5(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASE5(PART_UPPER, POST_D_SUB, D_VALS_DIAG)
	BHIP_CASE5(PART_LOWER, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE5(PART_UPPER, POST_SUB_D, D_DBLOCKS)
	BHIP_CASE5(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE5(PART_ALL, POST_AXPBY, D_NONE)
	BHIP_CASE5(PART_NONE, POST_D_SUB, D_DBLOCKS)
	ok = false;
#undef BHIP_CASE5
	if (ok)
		BHIP_CHECK(hipGetLastError());
	return ok;
}

}  // namespace bhip
