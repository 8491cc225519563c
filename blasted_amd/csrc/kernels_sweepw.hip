// kernels_sweepw.hip -- the tuned ("wide") row-sweep kernel for column-major blocks of size 4 and 8:
// bs = 4 is the PETSc BAIJ layout of BASELINE.json's headline configuration
// (src/blasted_petsc.cpp:256), bs = 8 its config 5.  Same operators and the same arithmetic as the
// generic family in kernels_sweep.hip (see the table there); what differs is the data movement:
//
//  * a workgroup owns RCHUNK consecutive block-rows of the sweep; browptr / diagind and the bcolind
//    range of the whole chunk are read from HBM once, coalesced, into LDS, so no value load waits on
//    an index load from memory (the dependent chain per row is LDS -> {block, x segment} -> result);
//  * a block is read as 16 bytes per lane (global_load_dwordx4) by LPB = bs*bs/2 lanes: lane q holds
//    entries (2q, 2q+1) = rows 2h, 2h+1 of column c, with h = q % (bs/2), c = q / (bs/2); one block
//    slot per row: 8 rows per wave at bs=4, 2 at bs=8;
//  * all loads of a row step -- KFIX predicated block passes, straight-line, times UNR steps -- are
//    issued before the first use; rows with more blocks finish in a remainder loop;
//  * the x segment of a block is gathered as one double per lane (its column's entry); the mat-vec is
//    two FMAs per lane plus an all-reduce over the column bits and the block-slot bit done on the VALU:
//    DPP row rotations inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across rows (the LDS
//    pipe only serves the staged indices);
//  * the inverted diagonal block needed by the upper solve sits directly in front of the row's upper
//    blocks in memory and is fetched by the same load instruction as block slot 0;
//  * rhs is read and the result written as 16 bytes per lane (bs*8 contiguous bytes per row);
//  * wave-uniform 64-bit bases + 32-bit per-lane byte offsets (halves address registers/arithmetic);
//  * XCD-aware chunk numbering: the XCDs take turns on super-chunks of 16 chunks (lanes.hpp, xcd_chunk).
#include "ctx.hpp"
#include "lanes.hpp"
#include "stage.hpp"

#include <cstdlib>
#include <cstring>

#ifndef BHIP_DIAG
#define BHIP_DIAG 0  // tools/probes/build_diag.sh: timing-only variants (wrong results) that drop one cost at a time
#endif

namespace bhip {

typedef double double2_t __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ double2_t load_block16(const double *p)
{
	if (NT)
		return __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(p));
	return *reinterpret_cast<const double2_t *>(p);
}

// v of lane w of the caller's quad, in every lane of the quad (w is a compile-time constant after unrolling)
__device__ __forceinline__ double quad_bcast(const double v, const int w)
{
	switch (w) {
	case 0: return dpp_mov<0x00>(v);
	case 1: return dpp_mov<0x55>(v);
	case 2: return dpp_mov<0xAA>(v);
	default: return dpp_mov<0xFF>(v);
	}
}

template <int BS, int NBV>
struct WGeo {
	static constexpr int HB = BS / 2;         // lanes per block column
	static constexpr int LPB = BS * HB;       // lanes per block (16 bytes each)
	static constexpr int NB = NBV;            // block slots per row (2, or 1 for the triangular sweeps at bs=4)
	static constexpr int G = LPB * NB;        // lanes per block-row: 16 or 8 (bs=4), 64 (bs=8)
	static constexpr int RPW = 64 / G;        // rows per wave and step
	static constexpr int RSTEP = 4 * RPW;     // rows per workgroup and step
	static constexpr int HBITS = HB == 2 ? 1 : 2;
	static constexpr int GBITS = G == 8 ? 3 : (G == 16 ? 4 : (G == 32 ? 5 : 6));
	static constexpr int BLKBYTES = BS * BS * 8;
	static constexpr int ROWBYTES = BS * 8;
};

// SC: the iterate is gathered and stored with relaxed AGENT-scope accesses (sc1: coherent across the eight
// per-XCD L2s) instead of plain ones -- the experiment SURVEY section 7 asks for ("without agent scope a sweep
// degrades to per-XCD Jacobi"); the matrix stream stays non-temporal.  Tuning string "...,c1".
__device__ __forceinline__ double sc_load(const char *p)
{
	return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
	                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void sc_store(char *p, const double v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
	                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LR ("long rows"): what happens to the blocks of a row part beyond the KFIX straight-line passes.  false:
// one block per loop iteration -- costs no registers, and never runs on stencil matrices whose parts fit the
// straight-line passes (the headline case: 48-54 VGPRs, 8 waves per SIMD).  true: groups of four predicated
// straight-line passes, one memory round trip per group instead of one per block, for patterns with longer
// rows (unstructured meshes); the group's registers take the kernel to 74-78 VGPRs, 6 waves per SIMD, which
// is why it is a separate instantiation chosen by the pattern's longest row.
// (Round 3, built, verified and removed: "Gauss-Seidel inside the wave" -- after the parallel phase the wave walked its
// eight rows in sweep order and corrected each row's sum for its in-wave predecessor: the neighbour's NEW value by a
// lane permute, minus the gathered old one, times the coupling block still in registers.  Correct, 52-64 registers,
// better first sweeps (distance to the exact solves after 3+3 sweeps 0.092 instead of 0.161 at 256^3) but the same
// asymptotic contraction (0.439 against 0.445: the slow error modes sit in the couplings between waves, not inside
// them), and the seven dependent permute-and-reduce rounds cost 21 % on the lower and 81 % on the upper sweep:
// profiles/r03_sweep_order_quality.txt.  The interleaved row order below does better for 12 %.)
// IW ("interleaved, with the wave's registers", round 3; bs = 4, one slot per row, one step in flight, triangular
// sweeps): the interleaved row order -- lane group g of wave w takes rows 4g, 4g+1, 4g+2, 4g+3 of the wave's 32-row
// window in its four steps -- with two of its memory round trips taken out:
//  * the window's right-hand side (32 rows x 32 bytes = 1 KiB) is ONE coalesced 16-byte-per-lane load at the start
//    of the chunk, handed to the step that needs it by a lane permute inside the 8-lane group, instead of a 32-byte
//    piece per row and step, 128 bytes apart;
//  * the coupling the interleaved order exists for -- a group's next row needs the row the group has JUST finished
//    (its predecessor in a lower sweep, its successor in an upper one) -- is served from the group's registers: in an
//    in-place sweep the gathered x segment of the block whose column is that row is replaced by the new result, so
//    the freshness of that value no longer depends on the store having reached the caches.
// Results are still stored step by step: collecting a window's results and storing it once, coalesced (built and
// measured), makes the sweep as fast as the natural order (+2.5 % instead of +10 %) but delays what OTHER workgroups
// see by up to three steps -- the row one grid line back belongs to the workgroup two chunks earlier, which in the
// step-by-step form has usually stored it in time -- and the contraction per sweep falls back from 0.30 to 0.41
// (natural order: 0.445): 52 ms instead of 43 ms to 1e-6 at 256^3.  Synchronous (double-buffered) sweeps use the
// window but never the forwarding: they stay Jacobi sweeps, bit-identical in every row order.
// 256^3 bs=4, lower + upper sweep: natural order 3.14 ms, this 3.45 ms, the round-1 form through memory 3.56 ms.
template <int BS, int PART, int POST, int DSRC, int RCHUNK, bool NT, int UNR, int NBV, bool SC = false, bool LR = false,
          bool IW = false, bool LS = false>
__global__ __launch_bounds__(256) void sweepw_kernel(const SweepArgs a)
{
	// LS ("late store", round 3; default -- with two row steps in flight -- for the in-place bs = 4 triangular sweeps
	// in natural row order): the results of a lane group's steps are collected in registers and a workgroup's 128
	// rows are stored once, after its last step (16 bytes per lane, 256-byte runs), instead of 32 bytes per row and
	// step.  Other workgroups see a chunk's rows up to three steps later: the sweep is MORE Jacobi-like -- distance to the
	// exact solves after 3+3 / 10+10 sweeps 0.24 / 2.6e-3 instead of 0.16 / 5.5e-4 at 256^3 -- and a BETTER
	// preconditioner, because what it does depends less on which wave won which race.  Measured on Poisson (x) 4x4 block,
	// 160^3 (profiles/r03_sweep_order_quality.txt): difference between two applications to the same vector / GCR(30)
	// iterations at 3 and 5 sweeps / time of a 256^3 lower + upper sweep --
	//     stores step by step, one step in flight (rounds 1-2)   2.2-2.5e-2   943-981 / 516-520   3.03 ms
	//     late store, one step in flight  ("latestore=1")        1.7e-2       798-802 / 512-517   3.03 ms
	//     late store, two steps in flight ("latestore=2", DEFAULT) 8.8e-3     684     / 498       2.98 ms
	//     late store, four steps in flight ("latestore=4")       8.9e-4       656     / 491       3.24 ms
	//     interleaved order ("interleave=1")                     2.3e-2       1173-1196 / 510     3.32-3.45 ms
	//     synchronous sweeps (mode DETERMINISTIC)                0            655     / -         (2 of 3 sweeps read the matrix)
	// i.e. among in-place sweeps on this GPU the more repeatable operator is the better one, down to the synchronous
	// sweeps' iteration count, and two steps in flight with a late store is also the fastest form.
	// bs = 8 (two 16-byte pieces per lane, one row step in flight), 100^3: 0.691 against 0.719 ms per sweep pair, two
	// applications differ by 5.7e-3 instead of 1.9e-2, GCR(30) 313 iterations instead of 363-365 at 3 sweeps.
	// (The late store for LONG rows -- the LR instantiation, unstructured matrices -- was built and measured too: GCR(30)
	// on an unstructured Laplacian (x) 4x4 block, 1.33 M rows, 33-34 iterations without / 35 with at 3 sweeps, 67-68 / 77
	// at 1 sweep, the upper sweep 0.458 / 0.467 ms: no gain, not kept; profiles/r03_solve_compare_unstructured.txt.)
	// (The late store for the in-place RELAXATION passes at bs = 4 -- whole rows, PART_OFFDIAG -- was built and measured at
	// the end of round 3: 85 / 140 registers with one / two steps in flight, config 3's pass 2.61-2.67 ms in every
	// form on the same box, i.e. no difference; not kept.)
	static_assert(!LS || (((BS == 4 && (UNR == 1 || UNR == 2 || UNR == 4)) || (BS == 8 && UNR == 1)) && NBV == 1 && !SC && !IW &&
	                      (PART == PART_LOWER || PART == PART_UPPER)), "late store");
	static_assert(!IW || (BS == 4 && UNR == 1 && NBV == 1 && !SC && (PART == PART_LOWER || PART == PART_UPPER)),
	              "register-carried interleave: bs 4, triangular sweeps, one slot per row, one step in flight");

	using Ge = WGeo<BS, NBV>;
	constexpr int HB = Ge::HB, LPB = Ge::LPB, NB = Ge::NB, G = Ge::G, RPW = Ge::RPW, RSTEP = Ge::RSTEP;
	constexpr int CAP = 8 * RCHUNK;  // staged column indices
	static_assert(BS == 4 || BS == 8, "wide kernel: bs 4 or 8");
	static_assert(RCHUNK % (RSTEP * UNR) == 0, "chunk must be a multiple of the unrolled step");

	__shared__ int s_rp[RCHUNK + 1];
	__shared__ int s_dg[RCHUNK];
	__shared__ int s_col[CAP];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const int slot = t / LPB, q = t % LPB;
	const int c = q / HB, h = q % HB;  // column of this lane's two entries, row pair (2h, 2h+1)

	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x, (unsigned)(a.xcd_shift < 0 ? 4 : a.xcd_shift));
	// rows of this chunk in index order: [r0, r0 + rc)
	const long lin0 = (long)chunk * RCHUNK;
	const int rc = (int)((nb - lin0) < RCHUNK ? (nb - lin0) : RCHUNK);
	const int descending = (BHIP_DIAG >= 4) ? (BHIP_DIAG == 5) : a.descending;  // 4: ascending, 5: descending whatever the sweep
	const int r0 = descending ? (int)(nb - lin0 - rc) : (int)lin0;

	int jlo, jhi;
#if BHIP_DIAG >= 3
	{
		const int per = (PART == PART_LOWER) ? 3 : 4;
		const int total = a.pat.browptr[nb];
		for (int k = tid; k <= rc; k += 256) {
			int v = (r0 + k) * per;
			v = v < total - per ? v : total - per;
			s_rp[k] = v;
			if (k < rc)
				s_dg[k] = (PART == PART_LOWER) ? v + per : v;
		}
		jlo = r0 * per;
		jlo = jlo < total - per ? jlo : total - per;
		jhi = jlo;
		__syncthreads();
	}
#elif BHIP_DIAG == 2
	stage_chunk_indices<PART, RCHUNK, 0>(a.pat, r0, rc, tid, s_rp, s_dg, s_col, jlo, jhi);
#else
	stage_chunk_indices<PART, RCHUNK, CAP>(a.pat, r0, rc, tid, s_rp, s_dg, s_col, jlo, jhi);
#endif

	// Chunk-relative addressing: wave-uniform 64-bit bases (SGPRs) + 32-bit per-lane byte offsets.  The
	// host checks that a chunk's blocks and the whole vector stay below 4 GiB (launch_sweepw).
	const char *const vbase = reinterpret_cast<const char *>(a.vals + (long)jlo * (BS * BS));
	const char *const xbase = reinterpret_cast<const char *>(a.xin);
	const char *const rbase = reinterpret_cast<const char *>(a.rhs + (long)r0 * BS);
	const char *const sbase = reinterpret_cast<const char *>(a.rscale + (long)r0 * BS);
	const char *const dbase = reinterpret_cast<const char *>(a.dvals + (long)r0 * (BS * BS));
	char *const obase = reinterpret_cast<char *>(a.xout + (long)r0 * BS);

	// straight-line passes: 2*NB items cover a 7-point row's lower or diagonal+upper part; operators
	// that visit the whole row (SpMV, relaxation) get twice as many
	constexpr int KFIX = (4 / NB) * ((PART == PART_ALL || PART == PART_OFFDIAG) ? 2 : 1);

	// IW: this lane's 16-byte piece of the wave's window -- row (lane / 2) of the window, half (lane % 2)
	constexpr int NSTEPS_W = RCHUNK / RSTEP;
	static_assert(!IW || NSTEPS_W * HB == G, "a lane group's pieces of the window must be its own lanes");
	double2_t win_r;
	win_r.x = win_r.y = 0.0;
	// LS: a lane keeps one 16-byte piece per slot -- piece (q % HB) of the row its group computes in step
	// slot * (G / HB) + q / HB (bs = 4: 4 steps, one slot; bs = 8: 16 steps, two slots)
	constexpr int LS_SPAN = G / HB;                                      // steps covered by one slot
	constexpr int LS_SLOTS = LS ? (RCHUNK / RSTEP + LS_SPAN - 1) / LS_SPAN : 1;
	double2_t late_o[LS_SLOTS];
#pragma unroll
	for (int j = 0; j < LS_SLOTS; j++)
		late_o[j].x = late_o[j].y = 0.0;
	const bool inplace = a.xin == a.xout;
	double prev0 = 0.0, prev1 = 0.0;  // the group's result of the step before (rows 2h, 2h+1)
	if (IW) {
		const int wls = (wave * RPW + g) * NSTEPS_W + (q >> 1);  // sweep position of this lane's piece
		const bool win_ok = wls < rc;
		const int win_lr = win_ok ? (descending ? rc - 1 - wls : wls) : 0;
		if (win_ok && a.rhs)
			win_r = *reinterpret_cast<const double2_t *>(rbase + ((unsigned)win_lr * (unsigned)Ge::ROWBYTES + 16u * (unsigned)(q & 1)));
	}

#pragma unroll 1
	for (int step0 = 0; step0 < RCHUNK / RSTEP; step0 += UNR) {
		int lrow[UNR], jbeg[UNR], jend[UNR], dgp[UNR];
		bool ok[UNR];
#pragma unroll
		for (int u = 0; u < UNR; u++) {
			// position in sweep order.  Interleaved: the rows one step computes side by side are NSTEPS apart,
			// so a row's predecessor in the sweep belongs to the step before (already stored) instead of to
			// the same step (stale): the in-place sweep is Gauss-Seidel-like along the chunk, not Jacobi-like.
			constexpr int NSTEPS = RCHUNK / RSTEP;
			const int slotpos = wave * RPW + g;
			const int ls = (IW || a.interleave) ? slotpos * NSTEPS + (step0 + u) : (step0 + u) * RSTEP + slotpos;
			ok[u] = ls < rc;
			const int lr = ok[u] ? (descending ? rc - 1 - ls : ls) : 0;
			lrow[u] = lr;
			const int rp0 = s_rp[lr], rp1 = s_rp[lr + 1];
			dgp[u] = s_dg[lr];
			jbeg[u] = jend[u] = 0;
			if (ok[u]) {
				if (PART == PART_LOWER) {
					jbeg[u] = rp0;
					jend[u] = dgp[u];
				} else if (PART == PART_UPPER) {
					jbeg[u] = (DSRC == D_VALS_DIAG) ? dgp[u] : dgp[u] + 1;  // first item = diagonal block
					jend[u] = rp1;
				} else if (PART == PART_OFFDIAG || PART == PART_ALL) {
					jbeg[u] = rp0;
					jend[u] = rp1;
				}
			}
		}

		double2_t bv[UNR][KFIX];
		double xv[UNR][KFIX];
		double2_t dv[UNR], r2[UNR], s2[UNR];
		// bs=4, one slot per row: the x segments of four passes (4 x 32 bytes) are fetched by ONE 16-byte
		// load per lane -- lane w of quad Q takes half Q of the segment of pass 4*kg + w -- and handed out
		// by quad broadcasts in the arithmetic below, instead of one 8-byte gather per pass (same-process
		// A/B at 256^3: lower sweep 1.517 -> 1.487 ms, upper sweep unchanged)
		constexpr bool XG = (BS == 4 && NB == 1 && BHIP_DIAG == 0);
		constexpr int NXG = XG ? KFIX / 4 : 1;
		double2_t xg[UNR][NXG];
		constexpr bool xg_on = XG;
		int colmine = -1;  // IW: the column of the pass this lane gathers for (pass q % 4)
		if (xg_on) {
#pragma unroll
			for (int u = 0; u < UNR; u++) {
#pragma unroll
				for (int kg = 0; kg < NXG; kg++) {
					const int jj = jbeg[u] + 4 * kg + (q & 3);
					xg[u][kg].x = 0.0;
					xg[u][kg].y = 0.0;
					const bool skip = (jj == dgp[u]) && ((PART == PART_UPPER && DSRC == D_VALS_DIAG) || PART == PART_OFFDIAG);
					if (PART != PART_NONE && jj < jend[u] && !skip) {
						const int cidx = jj - jlo;
						const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
						if (IW && kg == 0)
							colmine = col;
						const char *const xp = xbase + ((unsigned)col * (unsigned)Ge::ROWBYTES + 16u * (unsigned)(q >> 2));
						if (SC) {
							xg[u][kg].x = sc_load(xp);
							xg[u][kg].y = sc_load(xp + 8);
						} else
							xg[u][kg] = *reinterpret_cast<const double2_t *>(xp);
					}
				}
			}
		}
#pragma unroll
		for (int u = 0; u < UNR; u++) {
#pragma unroll
			for (int k = 0; k < KFIX; k++) {
				const int jj = jbeg[u] + slot + k * NB;
				bv[u][k].x = 0.0;
				bv[u][k].y = 0.0;
				xv[u][k] = 0.0;
				// (relaxation: the diagonal block of A is not part of the sum -- do not fetch it)
				if (PART != PART_NONE && jj < jend[u] && !(PART == PART_OFFDIAG && jj == dgp[u])) {
					bv[u][k] = load_block16<NT>(reinterpret_cast<const double *>(
					    vbase + ((unsigned)(jj - jlo) * (unsigned)Ge::BLKBYTES + 16u * (unsigned)q)));
					const bool isdiag = (jj == dgp[u]);
					if (!xg_on && !((PART == PART_UPPER && DSRC == D_VALS_DIAG && isdiag) ||
					             (PART == PART_OFFDIAG && isdiag))) {
#if BHIP_DIAG == 1 || BHIP_DIAG >= 3
						xv[u][k] = 1.0;
#elif BHIP_DIAG == 2
						xv[u][k] = *reinterpret_cast<const double *>(
						    xbase + ((unsigned)(r0 + lrow[u]) * (unsigned)Ge::ROWBYTES + 8u * (unsigned)c));
#else
						const int cidx = jj - jlo;
						const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
						const char *const xp = xbase + ((unsigned)col * (unsigned)Ge::ROWBYTES + 8u * (unsigned)c);
						xv[u][k] = SC ? sc_load(xp) : *reinterpret_cast<const double *>(xp);
#endif
					}
				}
			}
			dv[u].x = 0.0;
			dv[u].y = 0.0;
			if (DSRC == D_DBLOCKS && ok[u] && slot == 0)
				dv[u] = load_block16<false>(reinterpret_cast<const double *>(
				    dbase + ((unsigned)lrow[u] * (unsigned)Ge::BLKBYTES + 16u * (unsigned)q)));
			r2[u].x = r2[u].y = 0.0;
			s2[u].x = s2[u].y = 1.0;
			if (IW) {
				// this step's row of the window: pieces 2 * step, 2 * step + 1 of the lane group
				const int src = (lane & ~(G - 1)) | (2 * (step0 + u)) | (q & 1);
				r2[u].x = __shfl(win_r.x, src, 64);
				r2[u].y = __shfl(win_r.y, src, 64);
				if (ok[u] && a.rhs && a.rscale)
					s2[u] = *reinterpret_cast<const double2_t *>(
					    sbase + ((unsigned)lrow[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
			} else if (ok[u] && a.rhs) {
				r2[u] = *reinterpret_cast<const double2_t *>(
				    rbase + ((unsigned)lrow[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
				if (a.rscale)
					s2[u] = *reinterpret_cast<const double2_t *>(
					    sbase + ((unsigned)lrow[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
			}
		}

#pragma unroll
		for (int u = 0; u < UNR; u++) {
			double d0 = dv[u].x, d1 = dv[u].y;  // entries (2q, 2q+1) of D, in block slot 0
			double acc0 = 0.0, acc1 = 0.0;
			if (PART != PART_NONE) {
				if (xg_on) {
#pragma unroll
					for (int k = 0; k < KFIX; k++) {
						// component c of the segment of pass k: held by lane k%4 of this lane's quad (c/2 = quad)
						const double gx = quad_bcast(xg[u][k / 4].x, k & 3), gy = quad_bcast(xg[u][k / 4].y, k & 3);
						xv[u][k] = (c & 1) ? gy : gx;
					}
					if (IW && inplace && step0 + u > 0) {
						// the row this group finished in the step before: its new segment, from the group's registers
						const int prow = r0 + lrow[u] + (descending ? 1 : -1);
						const int mine = (colmine == prow) ? 1 : 0;
						const double a00 = dpp_mov<0x00>(prev0), a01 = dpp_mov<0x00>(prev1);  // quad_perm [0,0,0,0]
						const double a10 = dpp_mov<0x55>(prev0), a11 = dpp_mov<0x55>(prev1);  // quad_perm [1,1,1,1]
						const bool b1 = (q & 2) != 0, b2 = (q & 4) != 0;
						const double f0 = b2 ? a10 : a00, f1 = b2 ? a11 : a01;
						const double fresh = b1 ? f1 : f0;  // component c of the previous row's result
#pragma unroll
						for (int k = 0; k < KFIX && k < 4; k++) {
							// pass k was gathered by lane k of every quad: is its column the previous row?
							const int hit = k == 0 ? __builtin_amdgcn_update_dpp(0, mine, 0x00, 0xf, 0xf, false)
							              : k == 1 ? __builtin_amdgcn_update_dpp(0, mine, 0x55, 0xf, 0xf, false)
							              : k == 2 ? __builtin_amdgcn_update_dpp(0, mine, 0xAA, 0xf, 0xf, false)
							                       : __builtin_amdgcn_update_dpp(0, mine, 0xFF, 0xf, 0xf, false);
							xv[u][k] = hit ? fresh : xv[u][k];
						}
					}
				}
#pragma unroll
				for (int k = 0; k < KFIX; k++) {
					if (PART == PART_UPPER && DSRC == D_VALS_DIAG && k == 0) {
						// item 0 of the row is its (inverted) diagonal block: keep it as D
						const bool isd = (slot == 0);
						d0 = isd ? bv[u][0].x : d0;
						d1 = isd ? bv[u][0].y : d1;
						acc0 += isd ? 0.0 : bv[u][0].x * xv[u][0];
						acc1 += isd ? 0.0 : bv[u][0].y * xv[u][0];
					} else {
						acc0 += bv[u][k].x * xv[u][k];  // xv is zero for skipped / absent items
						acc1 += bv[u][k].y * xv[u][k];
					}
				}
				if (!LR) {
					for (int jj = jbeg[u] + slot + KFIX * NB; jj < jend[u]; jj += NB) {
						if (PART == PART_OFFDIAG && jj == dgp[u])
							continue;
						const double2_t v2 = load_block16<NT>(reinterpret_cast<const double *>(
						    vbase + ((unsigned)(jj - jlo) * (unsigned)Ge::BLKBYTES + 16u * (unsigned)q)));
						const int cidx = jj - jlo;
						const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
						const char *const xp = xbase + ((unsigned)col * (unsigned)Ge::ROWBYTES + 8u * (unsigned)c);
						const double xc = SC ? sc_load(xp) : *reinterpret_cast<const double *>(xp);
						acc0 += v2.x * xc;
						acc1 += v2.y * xc;
					}
				} else {
					constexpr int KGRP = 4;
					for (int jb = jbeg[u] + slot + KFIX * NB; jb < jend[u]; jb += KGRP * NB) {
						double2_t v4[KGRP];
						double x4[KGRP];
#pragma unroll
						for (int k = 0; k < KGRP; k++) {
							const int jj = jb + k * NB;
							v4[k].x = v4[k].y = 0.0;
							x4[k] = 0.0;
							if (jj < jend[u] && !(PART == PART_OFFDIAG && jj == dgp[u])) {
								v4[k] = load_block16<NT>(reinterpret_cast<const double *>(
								    vbase + ((unsigned)(jj - jlo) * (unsigned)Ge::BLKBYTES + 16u * (unsigned)q)));
								const int cidx = jj - jlo;
								const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
								const char *const xp = xbase + ((unsigned)col * (unsigned)Ge::ROWBYTES + 8u * (unsigned)c);
								x4[k] = SC ? sc_load(xp) : *reinterpret_cast<const double *>(xp);
							}
						}
#pragma unroll
						for (int k = 0; k < KGRP; k++) {
							acc0 += v4[k].x * x4[k];
							acc1 += v4[k].y * x4[k];
						}
					}
				}
				// over the column bits and the block-slot bit; the row-pair bits h stay
				acc0 = allreduce_bits<Ge::HBITS, Ge::GBITS>(acc0);
				acc1 = allreduce_bits<Ge::HBITS, Ge::GBITS>(acc1);
			}
			const double rv0 = r2[u].x * s2[u].x, rv1 = r2[u].y * s2[u].y;

			double o0, o1;
			// the post-operation on the row's sum (acc0, acc1 = rows 2h, 2h+1 of sum_j A_ij x_j, in every lane)
			auto post_op = [&](const double s0, const double s1) {
				if (POST == POST_SUB) {
					o0 = rv0 - s0;
					o1 = rv1 - s1;
				} else if (POST == POST_D_SUB || POST == POST_SUB_D) {
					const double w0 = (POST == POST_D_SUB) ? rv0 - s0 : s0;  // rows 2h, 2h+1 of the vector D multiplies
					const double w1 = (POST == POST_D_SUB) ? rv1 - s1 : s1;
					// this lane needs component c of that vector: rows 2(c/2), 2(c/2)+1 are held by the lanes
					// whose row-pair index h equals c/2; every aligned group of HB lanes contains all h
					double wc;
					if (BS == 4) {
						const double a00 = dpp_mov<0x00>(w0), a01 = dpp_mov<0x00>(w1);  // quad_perm [0,0,0,0]
						const double a10 = dpp_mov<0x55>(w0), a11 = dpp_mov<0x55>(w1);  // quad_perm [1,1,1,1]
						const bool b1 = (q & 2) != 0, b2 = (q & 4) != 0;
						const double t0 = b2 ? a10 : a00, t1 = b2 ? a11 : a01;
						wc = b1 ? t1 : t0;
					} else {
						const int src = (lane & ~(HB - 1)) | (c >> 1);
						const double t0 = __shfl(w0, src, 64), t1 = __shfl(w1, src, 64);
						wc = (c & 1) ? t1 : t0;
					}
					// D lives in block slot 0 only (zero elsewhere): the all-reduce is its column sum
					const double p0 = allreduce_bits<Ge::HBITS, Ge::GBITS>(d0 * wc);
					const double p1 = allreduce_bits<Ge::HBITS, Ge::GBITS>(d1 * wc);
					if (POST == POST_D_SUB) {
						o0 = p0;
						o1 = p1;
					} else {
						o0 = rv0 - p0;
						o1 = rv1 - p1;
					}
				} else {
					o0 = a.a * s0;
					o1 = a.a * s1;
					if (a.b != 0.0) {
						o0 += a.b * rv0;
						o1 += a.b * rv1;
					}
				}
			};
			post_op(acc0, acc1);

			if (IW) {
				prev0 = o0;
				prev1 = o1;
			}
			if (LS) {
				const int st = step0 + u;
#pragma unroll
				for (int j = 0; j < LS_SLOTS; j++) {
					if (st / LS_SPAN == j && (q / HB) == st % LS_SPAN) {  // (o0, o1) of a lane are piece h = q % HB of its row
						late_o[j].x = o0;
						late_o[j].y = o1;
					}
				}
			} else if (ok[u] && slot == 0 && q < HB) {
				double2_t o2;
				o2.x = o0;
				o2.y = o1;
				char *const dst = obase + ((unsigned)lrow[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)q);
				if (SC) {
					sc_store(dst, o0);
					sc_store(dst + 8, o1);
				} else if (IW && BHIP_PROBE(a.probe) == 2) {
					// (timing experiment, wrong results: only one row in 64 is stored)
					if (lrow[u] % 64 == 0)
						*reinterpret_cast<double2_t *>(dst) = o2;
				} else if (IW && BHIP_PROBE(a.probe) == 3)
					__builtin_nontemporal_store(o2, reinterpret_cast<double2_t *>(dst));
				else
					*reinterpret_cast<double2_t *>(dst) = o2;
			}
		}
	}
	if (LS) {
#pragma unroll
		for (int j = 0; j < LS_SLOTS; j++) {
			const int ls = (j * LS_SPAN + q / HB) * RSTEP + wave * RPW + g;
			if (slot == 0 && ls < rc) {
				const int lr = descending ? rc - 1 - ls : ls;
				*reinterpret_cast<double2_t *>(obase + ((unsigned)lr * (unsigned)Ge::ROWBYTES + 16u * (unsigned)(q % HB))) = late_o[j];
			}
		}
	}
}

struct Variant {
	int rchunk = 128, nt = 1, unr = 1, enabled = 1;
	int sc = 0;   // 1: agent-scope gathers and stores of the iterate (bs=4 triangular sweeps, r128,nt1,u1)
	int nb1 = 1;  // one block slot per row (bs=4: 8 lanes per row, 8 rows per wave; bs=8: 32 lanes, 2 rows); 0 = two slots, 3 = triangular sweeps only
};

static Variant parse_variant(const char *e)
{
	Variant v;
	// BLASTED_HIP_SWEEPW = "generic" | "r<128|256>,nt<0|1>,u<1|2>"   (tuning / A-B measurements)
	if (!e)
		return v;
	if (std::strcmp(e, "generic") == 0) {
		v.enabled = 0;
		return v;
	}
	int r = 128, nt = 1, unr = 1, nb1 = 1, sc = 0;
	const int got = std::sscanf(e, "r%d,nt%d,u%d,s%d,c%d", &r, &nt, &unr, &nb1, &sc);
	if (got >= 3) {
		v.rchunk = r;
		v.nt = nt;
		v.unr = unr;
		v.sc = (got == 5 && sc == 1) ? 1 : 0;
		if (got >= 4)
			v.nb1 = nb1 == 1 ? 1 : (nb1 == 3 ? 3 : 0);  // ",s1" (default): one block slot per row; ",s2": two; ",s3": one for the triangular sweeps only
	}
	return v;
}

static Variant &current_variant()
{
	static Variant v = parse_variant(std::getenv("BLASTED_HIP_SWEEPW"));
	return v;
}

// tuning hook behind blasted_hip_set_tuning(): same syntax as the BLASTED_HIP_SWEEPW variable
void set_sweepw_variant(const char *spec)
{
	current_variant() = parse_variant(spec);
}

template <int BS, int PART, int POST, int DSRC>
static bool launch_variant(const SweepArgs &a, const Variant &v_, hipStream_t s)
{
	// Synchronous (double-buffered) bs = 4 triangular sweeps take two row steps in flight: a row of such a sweep depends
	// on nothing the launch writes, so the result is the same bits and the sweep 1.5-2 % faster (round 2's u2 A/B)
	Variant v = v_;
	// (stencil-like rows only: longer row parts keep the one-step instantiation with the grouped remainder passes)
	if (BS == 4 && (PART == PART_LOWER || PART == PART_UPPER) && a.xin != a.xout && v.rchunk == 128 && v.unr == 1 &&
	    (a.pat.max_row_len + 1) / 2 + 1 <= 5)
		v.unr = 2;
#define BHIP_V(RV, NTV, UV)                                                                            \
	if (v.rchunk == RV && v.nt == NTV && v.unr == UV) {                                                \
		/* whole-row operators carry 4 straight-line passes, and bs=8 twice the registers per pass:     \
		   keep them at one step per pass of the loop so that 8 waves per SIMD stay resident */          \
		constexpr int UEFF = (PART == PART_ALL || PART == PART_OFFDIAG || BS == 8) ? 1 : UV;            \
		const unsigned grid = (unsigned)(((long)a.pat.nbrows + RV - 1) / RV);                          \
		/* one slot per row (bs=4: 8 lanes per row, 8 rows per wave; bs=8: 32 lanes, 2 rows; one step in   \
		   flight) fills every load pass whatever the row length (a 7-point row has 3 lower blocks: with  \
		   two slots its second pass is half empty) -- bs=4 at 256^3: ILU pair -6 %, SGS pair -11 %, SpMV \
		   -4 %, relaxation pass -5 %; bs=8 at 100^3: lower sweep 4.6 -> 6.3 TB/s, SpMV 5.1 -> 5.9 */      \
		/* one slot, triangular sweeps at bs=4: "u2" = two row steps in flight.  -1.5..2 % time on both  \
		   sweeps at 256^3 in every build of placement_variance.py -- but more rows in flight read more    \
		   stale neighbours: the error after 3+3 / 10+10 sweeps rises from 0.162 / 5.6e-4 to 0.190 /       \
		   9.8e-4, i.e. the contraction per sweep from 0.444 to 0.471 = 8 % more sweeps for the same       \
		   accuracy.  The default is therefore u1. */                                                      \
		constexpr int UN1 = (BS == 4 && (PART == PART_LOWER || PART == PART_UPPER)) ? UV : 1;            \
		constexpr bool SCOK = BS == 4 && RV == 128 && NTV == 1 && UV == 1 &&                            \
		                      (PART == PART_LOWER || PART == PART_UPPER);                               \
		if (v.sc && SCOK)                                                                              \
			hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, true, 1, 1, SCOK>), dim3(grid),  \
			                   dim3(256), 0, s, a);                                                    \
		else if (v.nb1 && (PART == PART_LOWER || PART == PART_UPPER || v.nb1 == 1)) {                  \
			/* rows whose part can exceed the straight-line passes: the grouped remainder (LR) */          \
			constexpr int KSTRAIGHT = (4 / 1) * ((PART == PART_ALL || PART == PART_OFFDIAG) ? 2 : 1);      \
			const bool whole = PART == PART_ALL || PART == PART_OFFDIAG;                                   \
			const int longest_part = whole ? a.pat.max_row_len : (a.pat.max_row_len + 1) / 2 + 1;          \
			/* interleaved row order through the wave's registers (IW, see the kernel); "interleave=2" keeps the  \
			   round-1 form that goes through memory */                                                     \
			constexpr bool IWOK = BS == 4 && RV == 128 && UV == 1 && (PART == PART_LOWER || PART == PART_UPPER); \
			constexpr bool LS8OK = BS == 8 && RV == 128 && UV == 1 && (PART == PART_LOWER || PART == PART_UPPER); \
			if (longest_part > KSTRAIGHT + 1 && UN1 == 1)                                                  \
				hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), 1, 1, false, true>), \
				                   dim3(grid), dim3(256), 0, s, a);                                        \
			else if (IWOK && a.latestore == 2 && !a.interleave && a.xin == a.xout)                         \
				hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), (IWOK ? 2 : 1), 1, false, false, false, IWOK>), \
				                   dim3(grid), dim3(256), 0, s, a);                                        \
			else if (IWOK && a.latestore == 4 && !a.interleave && a.xin == a.xout)                         \
				hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), (IWOK ? 4 : 1), 1, false, false, false, IWOK>), \
				                   dim3(grid), dim3(256), 0, s, a);                                        \
			else if (IWOK && a.latestore && !a.interleave && a.xin == a.xout)                              \
				hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), 1, 1, false, false, false, IWOK>), \
				                   dim3(grid), dim3(256), 0, s, a);                                        \
			else if (LS8OK && a.latestore && !a.interleave && a.xin == a.xout)                             \
				hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), 1, 1, false, false, false, LS8OK>), \
				                   dim3(grid), dim3(256), 0, s, a);                                        \
			else if (IWOK && a.interleave == 1)                                                            \
				hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), 1, 1, false, false, IWOK>), \
				                   dim3(grid), dim3(256), 0, s, a);                                        \
			else                                                                                           \
				hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), UN1, 1>), dim3(grid), \
				                   dim3(256), 0, s, a);                                                    \
		}                                                                                              \
		else                                                                                           \
			hipLaunchKernelGGL((sweepw_kernel<BS, PART, POST, DSRC, RV, (NTV != 0), UEFF, 2>), dim3(grid), \
			                   dim3(256), 0, s, a);                                                    \
		return true;                                                                                   \
	}
	BHIP_V(128, 1, 2)
	BHIP_V(128, 0, 2)
	BHIP_V(128, 1, 1)
	BHIP_V(128, 0, 1)
	BHIP_V(256, 1, 2)
	BHIP_V(256, 0, 1)
#undef BHIP_V
	return false;
}

template <int BS>
static bool launch_bs(const SweepArgs &a, Part part, Post post, DSrc dsrc, const Variant &v, hipStream_t s)
{
	bool ok = false;
#define BHIP_CASEW(P, Q, D)                                   \
	if (part == P && post == Q && dsrc == D)                  \
		ok = launch_variant<BS, P, Q, D>(a, v, s);
	BHIP_CASEW(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASEW(PART_UPPER, POST_SUB, D_NONE)
	BHIP_CASEW(PART_UPPER, POST_D_SUB, D_VALS_DIAG)
	BHIP_CASEW(PART_LOWER, POST_D_SUB, D_DBLOCKS)
	BHIP_CASEW(PART_UPPER, POST_SUB_D, D_DBLOCKS)
	BHIP_CASEW(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS)
	BHIP_CASEW(PART_ALL, POST_AXPBY, D_NONE)
	BHIP_CASEW(PART_NONE, POST_D_SUB, D_DBLOCKS)
#undef BHIP_CASEW
	return ok;
}

// returns false when the tuned kernel does not cover the request (caller uses the generic family)
bool launch_sweepw(const SweepArgs &a_, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	SweepArgs a = a_;
	// the interleaved row order is for in-place TRIANGULAR sweeps (where a fresher predecessor is worth its price,
	// profiles/r03_sweep_order_quality.txt); products, synchronous sweeps and relaxation passes keep the natural order
	// ("interleave=3", measurements: every in-place pass, relaxation included, in the through-memory form)
	if (a.xin != a.xout || (!(part == PART_LOWER || part == PART_UPPER) && a.interleave != 3))
		a.interleave = 0;
	if (a.interleave == 3)
		a.interleave = 2;
	const Variant &v = current_variant();
	const int bs = a.pat.bs;
	if (!v.enabled || (bs != 4 && bs != 8) || a.pat.rowmajor || a.pat.nbrows == 0)
		return false;
	// 16-byte loads need 16-byte aligned arrays (hipMalloc gives 256; borrowed pointers are checked)
	auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
	if (misaligned(a.vals) || misaligned(a.dvals) || misaligned(a.rhs) || misaligned(a.rscale) ||
	    misaligned(a.xout))
		return false;
	const bool ok = (bs == 4) ? launch_bs<4>(a, part, post, dsrc, v, s) : launch_bs<8>(a, part, post, dsrc, v, s);
	if (ok)
		BHIP_CHECK(hipGetLastError());
	return ok;
}

}  // namespace bhip
