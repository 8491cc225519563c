// kernels_levelw.hip -- the streaming form of the one-launch exact triangular solve for column-major
// blocks of size 4 and 8 (the lane layout of kernels_sweepw.hip) and 3, 5, 7 (that of
// kernels_sweepodd.hip), on the LEVEL-ORDERED copies of the factor's triangles (kernels_level.hip: build_level_storage / launch_level_permute_values).
//
// In level order consecutive positions are consecutive in memory, so the pass streams like the wide sweep
// kernel -- blocks read as 16 bytes per lane, reduction on the VALU -- with these differences:
//  * iterate entries a row depends on are POLLED (relaxed agent-scope atomic loads) until they stop
//    being the "pending" pattern the output vector was filled with, and results are published with
//    8-byte atomic stores; the block loads of a step are issued together with its first polls, so a
//    step costs one memory round trip when its dependencies are complete;
//  * a workgroup owns just one row step and workgroups take their positions in grid order (no XCD
//    renumbering): a row waits only for earlier
//    positions, i.e. for workgroups with smaller ids, which every XCD starts first (see
//    kernels_level.hip for the forward-progress argument and the bounded-spin abort).
// Rows of one step may depend on each other across a level boundary, so a row is committed as soon as
// ITS lanes have everything and the wave loops until all its rows are done.
// (Round 2, tried here after it had paid in the single-launch factorisation kernels: one lane polling one missing
// entry for the whole wave before everybody re-reads.  A row of a solve misses at most four 8-byte entries, the gate
// is one more dependent round trip, and the exact apply got slower: 256^3 bs=4 5.76 -> 5.96 ms, 100^3 bs=8 1.39 ->
// 1.51, 64^3 0.69 -> 0.75.  Not kept.  Nor an instantiation of the bs = 4 lower solve with three straight-line
// passes instead of four for 7-point patterns -- 58-64 registers, eight waves per SIMD instead of seven: 5.62 against
// 5.62 ms; nor eight waves by a register bound (spills: 5.9 ms).)
#include "ctx.hpp"
#include "lanes.hpp"

namespace bhip {

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr unsigned long long SFW_PENDING = 0xFFF8DEADBEEF0001ull;  // = SF_PENDING of kernels_level.hip
constexpr int SFW_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ double sfw_load(const char *p)
{
	const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
	                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return __longlong_as_double((long long)v);
}

__device__ __forceinline__ void sfw_store(char *p, double v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
	                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool sfw_pending(double v)
{
	return (unsigned long long)__double_as_longlong(v) == SFW_PENDING;
}

template <int BS, int NBV>
struct LWGeo {
	static constexpr int HB = BS / 2;
	static constexpr int LPB = BS * HB;
	static constexpr int NB = NBV;  // block slots per row: the lower solve (3 blocks on a 7-point row) is faster with
	                                // one (2.46 against 2.64 ms at 256^3), the upper solve (1 + 3) with two (2.92 / 3.03)
	static constexpr int G = LPB * NB;
	static constexpr int RPW = 64 / G;
	static constexpr int RSTEP = 4 * RPW;
	static constexpr int HBITS = HB == 2 ? 1 : 2;
	static constexpr int GBITS = G == 8 ? 3 : (G == 16 ? 4 : (G == 32 ? 5 : 6));
	static constexpr int BLKBYTES = BS * BS * 8;
	static constexpr int ROWBYTES = BS * 8;
};

// UPPER = false: y_i = rhs_i - sum_{lower} L_ij y_j over the strictly-lower copy (ptr = lptr);
// UPPER = true : z_i = D_i (rhs_i - sum_{upper} U_ij z_j) over the diagonal+upper copy, whose first
//                block of every row is the (inverted) diagonal block.
// A workgroup is UNR row steps, all requested at once.  Dependent chain of a row: {block pointers, row
// number, its first four column indices (head)} -> {blocks, right-hand side, polled iterate entries} ->
// result; no LDS, no workgroup barrier.
// PERM: the iterate being produced (a.xout) is LEVEL-ORDERED -- entry block p belongs to position p -- and
// `cols` / `head` hold positions.  A row's own result is then contiguous with its neighbours' in the
// sweep, and the entries it gathers lie in the few levels before it: natural-order vectors cost a
// scattered 32-byte DRAM burst per gather, which was what bounded this pass.  The lower solve reads its
// right-hand side r by row number and writes y level-ordered; the upper solve reads that y by position
// and writes z both level-ordered (for its own gathers) and, through a.xnat, in natural order.  Gathers
// look through the caches first: an entry is written once after the fill, so anything but "pending" is
// final wherever it is read from; a (possibly stale) "pending" is re-read coherently.
// SGS = false: the ILU solves above.  SGS = true: the exact symmetric Gauss-Seidel passes on the copies of
// the MATRIX with D^-1 from a.dvals (dblocks, by row): forward y_i = D_i^-1 (rhs_i - sum_lower A_ij y_j),
// backward z_i = rhs_i - D_i^-1 sum_upper A_ij z_j (the copy's first block, A's own diagonal, is skipped).
template <int BS, bool UPPER, int UNR, bool PERM, int NBV, bool SGS>
__global__ __launch_bounds__(256) void sfw_kernel(const SweepArgs a, const int *__restrict__ ptr,
                                                  const int *__restrict__ cols, const int4 *__restrict__ head,
                                                  const int *__restrict__ rows, const int count, int *ctl)
{
	using Ge = LWGeo<BS, NBV>;
	constexpr int HB = Ge::HB, LPB = Ge::LPB, NB = Ge::NB, G = Ge::G, RPW = Ge::RPW, RSTEP = Ge::RSTEP;
	constexpr int KFIX = 4 / NB;  // the four head entries
	constexpr unsigned long long GMASK = G == 64 ? ~0ull : ((1ull << G) - 1ull);
	static_assert(BS == 4 || BS == 8, "wide kernel: bs 4 or 8");

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const int slot = t / LPB, q = t % LPB;
	const int c = q / HB, h = q % HB;
	const int gbase = lane & ~(G - 1);
	const bool desc = a.descending != 0;

	int row[UNR], pp[UNR], jbeg[UNR], jend[UNR];
	int4 hd[UNR];
	bool ok[UNR];
#pragma unroll
	for (int u = 0; u < UNR; u++) {
		const long pos = ((long)blockIdx.x * UNR + u) * RSTEP + wave * RPW + g;  // position in sweep order
		ok[u] = pos < count;
		const int p = ok[u] ? (int)(desc ? count - 1 - pos : pos) : 0;
		row[u] = rows[p];
		pp[u] = p;
		jbeg[u] = ok[u] ? ptr[p] : 0;
		jend[u] = ok[u] ? ptr[p + 1] : 0;
		hd[u] = head[2 * p];  // the first four of the eight head entries
	}

	const char *const vbase = reinterpret_cast<const char *>(a.vals);
	const char *const xbase = reinterpret_cast<const char *>(a.xout);
	const char *const rbase = reinterpret_cast<const char *>(a.rhs);
	const char *const sbase = reinterpret_cast<const char *>(a.rscale);
	char *const obase = reinterpret_cast<char *>(a.xout);

	// blocks, right-hand side and the first polls of all UNR steps: one round trip
	double2_t bv[UNR][KFIX];
	double xv[UNR][KFIX];
	unsigned xo[UNR][KFIX];
	unsigned dep[UNR];
	double2_t r2[UNR], dvs[UNR];
#pragma unroll
	for (int u = 0; u < UNR; u++) {
		dep[u] = 0u;
#pragma unroll
		for (int k = 0; k < KFIX; k++) {
			const int jj = jbeg[u] + slot + k * NB;
			bv[u][k].x = 0.0;
			bv[u][k].y = 0.0;
			xv[u][k] = 0.0;
			xo[u][k] = 0u;
			if (jj < jend[u] && !(SGS && UPPER && jj == jbeg[u])) {
				bv[u][k] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(
				    vbase + ((long)jj * Ge::BLKBYTES + 16 * q)));
				if (!(UPPER && jj == jbeg[u])) {  // the diagonal block multiplies no iterate entry
					const int idx = slot + k * NB;  // item number inside the row part: its column is in the head
					const int col = idx == 0 ? hd[u].x : (idx == 1 ? hd[u].y : (idx == 2 ? hd[u].z : hd[u].w));
					xo[u][k] = (unsigned)col * (unsigned)Ge::ROWBYTES + 8u * (unsigned)c;
					xv[u][k] = PERM ? *reinterpret_cast<const double *>(xbase + xo[u][k]) : sfw_load(xbase + xo[u][k]);
					if (sfw_pending(xv[u][k]))
						dep[u] |= 1u << k;
				}
			}
		}
		dvs[u].x = dvs[u].y = 0.0;
		if (SGS && ok[u] && slot == 0)
			dvs[u] = *reinterpret_cast<const double2_t *>(reinterpret_cast<const char *>(a.dvals) +
			                                              ((long)row[u] * Ge::BLKBYTES + 16 * q));
		r2[u].x = r2[u].y = 0.0;
		if (ok[u]) {
			const unsigned rsel = (PERM && UPPER) ? (unsigned)pp[u] : (unsigned)row[u];
			r2[u] = *reinterpret_cast<const double2_t *>(
			    rbase + (rsel * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
			if (a.rscale) {
				const double2_t s2 = *reinterpret_cast<const double2_t *>(
				    sbase + ((unsigned)row[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
				r2[u].x *= s2.x;
				r2[u].y *= s2.y;
			}
		}
	}
	// rows longer than KFIX * NB blocks: the remainder is fetched (and polled) when the row commits

	bool done[UNR];
#pragma unroll
	for (int u = 0; u < UNR; u++)
		done[u] = !ok[u];
	int spins = 0;
	for (;;) {
		bool alldone = true;
#pragma unroll
		for (int u = 0; u < UNR; u++) {
			if (__builtin_amdgcn_ballot_w64(!done[u]) == 0ull)
				continue;
#pragma unroll
			for (int k = 0; k < KFIX; k++) {
				if (dep[u] & (1u << k)) {
					const double v = sfw_load(xbase + xo[u][k]);
					if (!sfw_pending(v)) {
						xv[u][k] = v;
						dep[u] &= ~(1u << k);
					}
				}
			}
			const unsigned long long rb = __builtin_amdgcn_ballot_w64(dep[u] == 0u);
			bool gready = ((rb >> gbase) & GMASK) == GMASK;
			if (__builtin_amdgcn_ballot_w64(gready && !done[u]) != 0ull) {
				double acc0 = 0.0, acc1 = 0.0, d0 = SGS ? dvs[u].x : 0.0, d1 = SGS ? dvs[u].y : 0.0;
#pragma unroll
				for (int k = 0; k < KFIX; k++) {
					if (!SGS && UPPER && k == 0) {
						const bool isd = (slot == 0);  // item 0 of the row: its inverted diagonal block
						d0 = isd ? bv[u][0].x : 0.0;
						d1 = isd ? bv[u][0].y : 0.0;
						acc0 += isd ? 0.0 : bv[u][0].x * (gready ? xv[u][0] : 0.0);
						acc1 += isd ? 0.0 : bv[u][0].y * (gready ? xv[u][0] : 0.0);
					} else {
						acc0 += bv[u][k].x * (gready ? xv[u][k] : 0.0);
						acc1 += bv[u][k].y * (gready ? xv[u][k] : 0.0);
					}
				}
				// remainder of long rows: a committing group polls these entries here; they belong to earlier
				// positions, and a group whose tail is not complete yet retries the whole row in a later round
				bool tail_ok = true;
				if (gready && !done[u]) {
					for (int jj = jbeg[u] + slot + KFIX * NB; jj < jend[u]; jj += NB) {
						const double2_t v2 = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(
						    vbase + ((long)jj * Ge::BLKBYTES + 16 * q)));
						const int col = cols[jj];
						const double xc = sfw_load(xbase + ((unsigned)col * (unsigned)Ge::ROWBYTES + 8u * (unsigned)c));
						if (sfw_pending(xc))
							tail_ok = false;
						acc0 += v2.x * xc;
						acc1 += v2.y * xc;
					}
				}
				const unsigned long long tb = __builtin_amdgcn_ballot_w64(tail_ok);
				gready = gready && (((tb >> gbase) & GMASK) == GMASK);
				acc0 = allreduce_bits<Ge::HBITS, Ge::GBITS>(acc0);
				acc1 = allreduce_bits<Ge::HBITS, Ge::GBITS>(acc1);
				double o0, o1;
				if (!UPPER && !SGS) {
					o0 = r2[u].x - acc0;
					o1 = r2[u].y - acc1;
				} else {
					// the vector the diagonal block multiplies: rhs - sum (ILU upper, SGS forward), sum (SGS backward)
					const double w0 = (SGS && UPPER) ? acc0 : r2[u].x - acc0, w1 = (SGS && UPPER) ? acc1 : r2[u].y - acc1;
					double wc;
					if (BS == 4) {
						const double a00 = dpp_mov<0x00>(w0), a01 = dpp_mov<0x00>(w1);  // quad_perm [0,0,0,0]
						const double a10 = dpp_mov<0x55>(w0), a11 = dpp_mov<0x55>(w1);  // quad_perm [1,1,1,1]
						const bool b1 = (q & 2) != 0, b2 = (q & 4) != 0;
						const double s0 = b2 ? a10 : a00, s1 = b2 ? a11 : a01;
						wc = b1 ? s1 : s0;
					} else {
						const int src = (lane & ~(HB - 1)) | (c >> 1);
						const double t0 = __shfl(w0, src, 64), t1 = __shfl(w1, src, 64);
						wc = (c & 1) ? t1 : t0;
					}
					o0 = allreduce_bits<Ge::HBITS, Ge::GBITS>(d0 * wc);
					o1 = allreduce_bits<Ge::HBITS, Ge::GBITS>(d1 * wc);
					if (SGS && UPPER) {
						o0 = r2[u].x - o0;
						o1 = r2[u].y - o1;
					}
				}
				if (!done[u] && gready && slot == 0 && q < HB) {
					const unsigned osel = PERM ? (unsigned)pp[u] : (unsigned)row[u];
					char *const dst = obase + (osel * (unsigned)Ge::ROWBYTES + 16u * (unsigned)q);
					sfw_store(dst, o0);
					sfw_store(dst + 8, o1);
					if (PERM && a.xnat) {
						double2_t o2;
						o2.x = o0;
						o2.y = o1;
						*reinterpret_cast<double2_t *>(reinterpret_cast<char *>(a.xnat) +
						                               ((unsigned)row[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)q)) = o2;
					}
				}
				done[u] = done[u] || gready;
			}
			if (!done[u])
				alldone = false;
		}
		if (__builtin_amdgcn_ballot_w64(!alldone) == 0ull)
			return;
		spins++;
		if (spins > SFW_SPIN_LIMIT ||
		    ((spins & 255) == 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
			if (lane == 0)
				__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		__builtin_amdgcn_s_sleep(1);
	}
}

// ---- round 3: what bounds this pass, measured (tools/exact_solve_ab.py, profiles/r03_exact_solve_experiments.txt) ----
// With the iterate pre-filled with zeros instead of the pending pattern (tuning "levelnowait=1": nobody waits, wrong
// results) the two passes of the 256^3 bs=4 application take 4.4-4.75 ms of the 5.55-5.7 ms: the dependencies cost
// about 1.0 ms (0.65 us per level and triangle), the rest is the stream itself -- which moves more than the
// triangular sweeps' count: a 32-byte head per row, the pending fill of each output, the right-hand side gathered by
// row number and z scattered back to natural order in 32-byte pieces.  At 128^3 it is the other way round (0.70 of
// 1.71 ms is stream, 1.3 us per level).  Two restructurings were built on this kernel, verified bit-identical to it,
// measured and removed: (1) a PERSISTENT grid (as many workgroups as are resident, wave w taking the row steps
// w, w + W, ...) with the next step's blocks requested before the current step's polls -- loads return in order, so
// every poll after the first waits for the prefetch: 7.0 ms; (2) the same with only the next step's INDEX loads in
// flight -- no launch and no index round trip per step, but the loop's carried state costs 18 registers (86 / 78
// instead of 68 / 56: 5-6 resident waves instead of 7-8): 6.2 ms, also with nobody waiting (5.2 against 4.75).
// Start-up is not what a step costs; resident waves x bytes per wave are.  (3) One block slot per row in the upper
// pass (8 rows per wave instead of 4): 4.40 against 4.55 ms with nobody waiting, 5.59 against 5.57 ms with the
// dependencies -- not kept (it would also change the summation order).

// ---- odd block sizes 3, 5, 7 (the lane layout of kernels_sweepodd.hip) -----------------------------------
// G = 8 / 16 / 32 lanes per row, L = (bs*bs+1)/2 of them hold a block as 16 bytes per lane, the stride-bs
// reduction goes through a wave-private LDS tile.  One row step per wave, up to eight blocks of a row
// part requested at once (their column indices come from the head); longer rows finish at commit time.
typedef double2_t d2u_t __attribute__((aligned(8)));  // 16-byte access at 8-byte alignment

template <int BS, bool UPPER, bool PERM>
__global__ __launch_bounds__(256) void sfodd_kernel(const SweepArgs a, const int *__restrict__ ptr,
                                                    const int *__restrict__ cols, const int4 *__restrict__ head,
                                                    const int *__restrict__ rows, const int count, int *ctl)
{
	static_assert(BS == 3 || BS == 5 || BS == 7, "odd block sizes 3, 5, 7");
	constexpr int BS2 = BS * BS, L = (BS2 + 1) / 2;
	constexpr int G = BS == 3 ? 8 : (BS == 5 ? 16 : 32);
	constexpr int RPW = 64 / G, RSTEP = 4 * RPW, KFIX = 8;
	constexpr int BLKBYTES = BS2 * 8, ROWBYTES = BS * 8;
	constexpr unsigned long long GMASK = (1ull << G) - 1ull;

	__shared__ double s_acc[4][RPW][BS2 + 1];
	__shared__ double s_w[4][RPW][BS + 1];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const int gbase = lane & ~(G - 1);
	const bool actA = t < L - 1, actB = t < L;
	const int eA = actA ? 2 * t : BS2 - 2, eB = actA ? 2 * t + 1 : BS2 - 1;
	const long boff = actA ? 16 * t : 8 * (BS2 - 2);
	const int cA = (eA / BS) < BS ? eA / BS : BS - 1, cB = (eB / BS) < BS ? eB / BS : BS - 1;
	const int cx = cA < BS - 2 ? cA : BS - 2;
	const bool hiA = cA != cx, hiB = cB != cx;
	const bool desc = a.descending != 0;

	const long pos = (long)blockIdx.x * RSTEP + wave * RPW + g;  // position in sweep order
	const bool ok = pos < count;
	const int p = ok ? (int)(desc ? count - 1 - pos : pos) : 0;
	const int row = rows[p];
	const int jbeg = ok ? ptr[p] : 0, jend = ok ? ptr[p + 1] : 0;
	const int4 h0 = head[2 * p], h1 = head[2 * p + 1];

	const char *const vbase = reinterpret_cast<const char *>(a.vals);
	const char *const xbase = reinterpret_cast<const char *>(a.xout);
	char *const obase = reinterpret_cast<char *>(a.xout);
	double *const tile = &s_acc[wave][g][0];
	double *const wvec = &s_w[wave][g][0];

	d2u_t bv[KFIX];
	double xa[KFIX], xb[KFIX];  // x_cx, x_cx+1 of each block's column segment
	unsigned xo[KFIX];
	unsigned dep = 0u;
#pragma unroll
	for (int k = 0; k < KFIX; k++) {
		const int jj = jbeg + k;
		bv[k].x = bv[k].y = 0.0;
		xa[k] = xb[k] = 0.0;
		xo[k] = 0u;
		if (jj < jend && actB) {
			bv[k] = __builtin_nontemporal_load(reinterpret_cast<const d2u_t *>(vbase + ((long)jj * BLKBYTES + boff)));
			if (!(UPPER && k == 0)) {  // item 0 of the upper copy is the diagonal block
				const int col = k == 0 ? h0.x : (k == 1 ? h0.y : (k == 2 ? h0.z : (k == 3 ? h0.w : (k == 4 ? h1.x : (k == 5 ? h1.y : (k == 6 ? h1.z : h1.w))))));
				xo[k] = (unsigned)col * (unsigned)ROWBYTES + 8u * (unsigned)cx;
				if (PERM) {  // level-ordered iterate: first look through the caches (see sfw_kernel)
					const d2u_t x2 = *reinterpret_cast<const d2u_t *>(xbase + xo[k]);
					xa[k] = x2.x;
					xb[k] = x2.y;
				} else {
					xa[k] = sfw_load(xbase + xo[k]);
					xb[k] = sfw_load(xbase + xo[k] + 8);
				}
				if (sfw_pending(xa[k]) || sfw_pending(xb[k]))
					dep |= 1u << k;
			}
		}
	}
	double rv = 0.0;
	if (ok && t < BS) {
		const long rsel = (PERM && UPPER) ? (long)p : (long)row;  // the level-ordered y of the lower solve
		rv = a.rhs[rsel * BS + t];
		if (a.rscale)
			rv *= a.rscale[(long)row * BS + t];
	}

	bool done = !ok;
	int spins = 0;
	for (;;) {
#pragma unroll
		for (int k = 0; k < KFIX; k++) {
			if (dep & (1u << k)) {
				const double va = sfw_load(xbase + xo[k]), vb = sfw_load(xbase + xo[k] + 8);
				if (!sfw_pending(va) && !sfw_pending(vb)) {
					xa[k] = va;
					xb[k] = vb;
					dep &= ~(1u << k);
				}
			}
		}
		const unsigned long long rb = __builtin_amdgcn_ballot_w64(dep == 0u);
		bool gready = ((rb >> gbase) & GMASK) == GMASK;
		if (__builtin_amdgcn_ballot_w64(gready && !done) != 0ull) {
			double accA = 0.0, accB = 0.0;
			d2u_t dv;
			dv.x = dv.y = 0.0;
#pragma unroll
			for (int k = 0; k < KFIX; k++) {
				if (UPPER && k == 0) {
					dv = bv[0];
				} else if (gready) {
					accA += bv[k].x * (hiA ? xb[k] : xa[k]);
					accB += bv[k].y * (hiB ? xb[k] : xa[k]);
				}
			}
			bool tail_ok = true;
			if (gready && !done && actB) {
				for (int jj = jbeg + KFIX; jj < jend; jj++) {
					const d2u_t v2 =
					    __builtin_nontemporal_load(reinterpret_cast<const d2u_t *>(vbase + ((long)jj * BLKBYTES + boff)));
					const unsigned o = (unsigned)cols[jj] * (unsigned)ROWBYTES + 8u * (unsigned)cx;
					const double va = sfw_load(xbase + o), vb = sfw_load(xbase + o + 8);
					if (sfw_pending(va) || sfw_pending(vb))
						tail_ok = false;
					accA += v2.x * (hiA ? vb : va);
					accB += v2.y * (hiB ? vb : va);
				}
			}
			const unsigned long long tb = __builtin_amdgcn_ballot_w64(tail_ok);
			gready = gready && (((tb >> gbase) & GMASK) == GMASK);
			if (actA)
				tile[eA] = accA;
			if (actB)
				tile[eB] = accB;
			__builtin_amdgcn_wave_barrier();
			double sum = 0.0;
			if (t < BS) {
#pragma unroll
				for (int cc = 0; cc < BS; cc++)
					sum += tile[cc * BS + t];
			}
			__builtin_amdgcn_wave_barrier();
			double out = rv - sum;
			if (UPPER) {
				if (t < BS)
					wvec[t] = out;
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
						pr += tile[cc * BS + t];
				}
				__builtin_amdgcn_wave_barrier();
				out = pr;
			}
			if (!done && gready && t < BS) {
				sfw_store(obase + ((unsigned)(PERM ? p : row) * (unsigned)ROWBYTES + 8u * (unsigned)t), out);
				if (PERM && a.xnat)
					a.xnat[(long)row * BS + t] = out;
			}
			done = done || gready;
		}
		if (__builtin_amdgcn_ballot_w64(!done) == 0ull)
			return;
		spins++;
		if (spins > SFW_SPIN_LIMIT ||
		    ((spins & 255) == 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
			if (lane == 0)
				__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		__builtin_amdgcn_s_sleep(1);
	}
}

int g_levelw_enabled = 1;
int g_levelw_variant = 0;  // 0: one row step per wave (default), 1: two
}  // namespace

void set_levelw_enabled(int on)
{
	g_levelw_enabled = on != 0;
	g_levelw_variant = on >= 2 ? on - 1 : 0;  // "levelwide=2" / "=3": tuning variants
}

// One exact triangular pass over the level-ordered copy described by (ptr, cols, a.vals).  a.xout must
// have been filled with the pending pattern.  Returns false when this kernel does not cover the case
// (caller uses the general single-launch kernel).  The abort flag ls.ctl[1] is checked by the caller.
bool syncfree_wide_supported(const Pattern &pat)
{
	return g_levelw_enabled && pat.bs >= 3 && pat.bs <= 8 && pat.bs != 6 && !pat.rowmajor;
}

bool launch_syncfree_wide(const SweepArgs &a, bool upper, const LevelSchedule &ls, const int *ptr, const int *cols,
                          const int4 *head, hipStream_t s, bool permuted, bool sgs)
{
	const int bs = a.pat.bs;
	if (!g_levelw_enabled || a.pat.rowmajor || ls.count == 0)
		return false;
	if (sgs && bs != 4 && bs != 8)
		return false;  // the Gauss-Seidel passes exist in the bs 4 / 8 kernel only
	if (bs == 3 || bs == 5 || bs == 7) {
		BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
#define BHIP_LO(B, UP)                                                                                          \
	{                                                                                                           \
		constexpr int RS = 4 * (64 / (B == 3 ? 8 : (B == 5 ? 16 : 32)));                                        \
		const unsigned grid = (unsigned)(((long)ls.count + RS - 1) / RS);                                       \
		if (permuted)                                                                                           \
			hipLaunchKernelGGL((sfodd_kernel<B, UP, true>), dim3(grid), dim3(256), 0, s, a, ptr, cols, head,    \
			                   ls.rows, ls.count, ls.ctl);                                                      \
		else                                                                                                    \
			hipLaunchKernelGGL((sfodd_kernel<B, UP, false>), dim3(grid), dim3(256), 0, s, a, ptr, cols, head,   \
			                   ls.rows, ls.count, ls.ctl);                                                      \
	}
		if (bs == 3) {
			if (upper) BHIP_LO(3, true) else BHIP_LO(3, false)
		} else if (bs == 5) {
			if (upper) BHIP_LO(5, true) else BHIP_LO(5, false)
		} else {
			if (upper) BHIP_LO(7, true) else BHIP_LO(7, false)
		}
#undef BHIP_LO
		BHIP_CHECK(hipGetLastError());
		return true;
	}
	if (bs != 4 && bs != 8)
		return false;
	auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
	if (misaligned(a.vals) || misaligned(a.rhs) || misaligned(a.rscale) || misaligned(a.xout) ||
	    (sgs && misaligned(a.dvals)))
		return false;
	// A workgroup is ONE row step (16 rows at bs=4, 4 at bs=8), all of it requested at once: rows of a level
	// are independent, so nothing may be serialised behind another row's wait -- with 128-row chunks the
	// last row of a level finished 8 (32) round trips late and every level paid for it (21.7 ms instead of
	// 7.2 ms per exact apply at 256^3).  One step per wave keeps 8 waves per SIMD resident.
	BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
#define BHIP_LW(B, UP, U)                                                                                  \
	{                                                                                                      \
		constexpr int NBV = (UP || B == 8) ? 2 : 1;                                                        \
		constexpr int RC = LWGeo<B, NBV>::RSTEP * U;                                                       \
		const unsigned grid = (unsigned)(((long)ls.count + RC - 1) / RC);                                  \
		if (sgs && permuted)                                                                               \
			hipLaunchKernelGGL((sfw_kernel<B, UP, U, true, NBV, true>), dim3(grid), dim3(256), 0, s, a, ptr, cols, \
			                   head, ls.rows, ls.count, ls.ctl);                                           \
		else if (sgs)                                                                                      \
			hipLaunchKernelGGL((sfw_kernel<B, UP, U, false, NBV, true>), dim3(grid), dim3(256), 0, s, a, ptr, cols, \
			                   head, ls.rows, ls.count, ls.ctl);                                           \
		else if (permuted)                                                                                 \
			hipLaunchKernelGGL((sfw_kernel<B, UP, U, true, NBV, false>), dim3(grid), dim3(256), 0, s, a, ptr, cols, \
			                   head, ls.rows, ls.count, ls.ctl);                                           \
		else                                                                                               \
			hipLaunchKernelGGL((sfw_kernel<B, UP, U, false, NBV, false>), dim3(grid), dim3(256), 0, s, a, ptr, cols, \
			                   head, ls.rows, ls.count, ls.ctl);                                           \
	}
	const int v = g_levelw_variant;
	if (bs == 4) {
		if (upper) {
			if (v == 1) BHIP_LW(4, true, 2) else BHIP_LW(4, true, 1)
		} else {
			if (v == 1) BHIP_LW(4, false, 2) else BHIP_LW(4, false, 1)
		}
	} else {
		if (upper) {
			if (v == 1) BHIP_LW(8, true, 2) else BHIP_LW(8, true, 1)
		} else {
			if (v == 1) BHIP_LW(8, false, 2) else BHIP_LW(8, false, 1)
		}
	}
#undef BHIP_LW
	BHIP_CHECK(hipGetLastError());
	return true;
}

}  // namespace bhip
