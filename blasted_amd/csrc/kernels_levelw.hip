// kernels_levelw.hip -- the streaming form of the one-launch exact triangular solve for column-major
// blocks of size 4 and 8 (the layouts of kernels_sweepw.hip), on the LEVEL-ORDERED copies of the
// factor's triangles (kernels_level.hip: build_level_storage / launch_level_permute_values).
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

template <int BS>
struct LWGeo {
	static constexpr int HB = BS / 2;
	static constexpr int LPB = BS * HB;
	static constexpr int NB = 2;
	static constexpr int G = LPB * NB;
	static constexpr int RPW = 64 / G;
	static constexpr int RSTEP = 4 * RPW;
	static constexpr int HBITS = HB == 2 ? 1 : 2;
	static constexpr int GBITS = G == 16 ? 4 : 6;
	static constexpr int BLKBYTES = BS * BS * 8;
	static constexpr int ROWBYTES = BS * 8;
};

// UPPER = false: y_i = rhs_i - sum_{lower} L_ij y_j over the strictly-lower copy (ptr = lptr);
// UPPER = true : z_i = D_i (rhs_i - sum_{upper} U_ij z_j) over the diagonal+upper copy, whose first
//                block of every row is the (inverted) diagonal block.
// A workgroup is UNR row steps, all requested at once.  Dependent chain of a row: {block pointers, row
// number, its first four column indices (head)} -> {blocks, right-hand side, polled iterate entries} ->
// result; no LDS, no workgroup barrier.
template <int BS, bool UPPER, int UNR>
__global__ __launch_bounds__(256) void sfw_kernel(const SweepArgs a, const int *__restrict__ ptr,
                                                  const int *__restrict__ cols, const int4 *__restrict__ head,
                                                  const int *__restrict__ rows, const int count, int *ctl)
{
	using Ge = LWGeo<BS>;
	constexpr int HB = Ge::HB, LPB = Ge::LPB, NB = Ge::NB, G = Ge::G, RPW = Ge::RPW, RSTEP = Ge::RSTEP;
	constexpr int KFIX = 2;
	constexpr unsigned long long GMASK = G == 64 ? ~0ull : ((1ull << G) - 1ull);
	static_assert(BS == 4 || BS == 8, "wide kernel: bs 4 or 8");

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const int slot = t / LPB, q = t % LPB;
	const int c = q / HB, h = q % HB;
	const int gbase = lane & ~(G - 1);
	const bool desc = a.descending != 0;

	int row[UNR], jbeg[UNR], jend[UNR];
	int4 hd[UNR];
	bool ok[UNR];
#pragma unroll
	for (int u = 0; u < UNR; u++) {
		const long pos = ((long)blockIdx.x * UNR + u) * RSTEP + wave * RPW + g;  // position in sweep order
		ok[u] = pos < count;
		const int p = ok[u] ? (int)(desc ? count - 1 - pos : pos) : 0;
		row[u] = rows[p];
		jbeg[u] = ok[u] ? ptr[p] : 0;
		jend[u] = ok[u] ? ptr[p + 1] : 0;
		hd[u] = head[p];
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
	double2_t r2[UNR];
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
			if (jj < jend[u]) {
				bv[u][k] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(
				    vbase + ((long)jj * Ge::BLKBYTES + 16 * q)));
				if (!(UPPER && jj == jbeg[u])) {  // the diagonal block multiplies no iterate entry
					const int col = (k == 0) ? (slot == 0 ? hd[u].x : hd[u].y) : (slot == 0 ? hd[u].z : hd[u].w);
					xo[u][k] = (unsigned)col * (unsigned)Ge::ROWBYTES + 8u * (unsigned)c;
					xv[u][k] = sfw_load(xbase + xo[u][k]);
					if (sfw_pending(xv[u][k]))
						dep[u] |= 1u << k;
				}
			}
		}
		r2[u].x = r2[u].y = 0.0;
		if (ok[u]) {
			r2[u] = *reinterpret_cast<const double2_t *>(
			    rbase + ((unsigned)row[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)h));
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
				double acc0 = 0.0, acc1 = 0.0, d0 = 0.0, d1 = 0.0;
#pragma unroll
				for (int k = 0; k < KFIX; k++) {
					if (UPPER && k == 0) {
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
				if (!UPPER) {
					o0 = r2[u].x - acc0;
					o1 = r2[u].y - acc1;
				} else {
					const double w0 = r2[u].x - acc0, w1 = r2[u].y - acc1;
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
				}
				if (!done[u] && gready && slot == 0 && q < HB) {
					char *const dst = obase + ((unsigned)row[u] * (unsigned)Ge::ROWBYTES + 16u * (unsigned)q);
					sfw_store(dst, o0);
					sfw_store(dst + 8, o1);
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
bool launch_syncfree_wide(const SweepArgs &a, bool upper, const LevelSchedule &ls, const int *ptr, const int *cols,
                          const int4 *head, hipStream_t s)
{
	const int bs = a.pat.bs;
	if (!g_levelw_enabled || (bs != 4 && bs != 8) || a.pat.rowmajor || ls.count == 0)
		return false;
	auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
	if (misaligned(a.vals) || misaligned(a.rhs) || misaligned(a.rscale) || misaligned(a.xout))
		return false;
	// A workgroup is ONE row step (16 rows at bs=4, 4 at bs=8), all of it requested at once: rows of a level
	// are independent, so nothing may be serialised behind another row's wait -- with 128-row chunks the
	// last row of a level finished 8 (32) round trips late and every level paid for it (21.7 ms instead of
	// 7.2 ms per exact apply at 256^3).  One step per wave keeps 8 waves per SIMD resident.
	BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
#define BHIP_LW(B, UP, U)                                                                                  \
	{                                                                                                      \
		constexpr int RC = LWGeo<B>::RSTEP * U;                                                            \
		const unsigned grid = (unsigned)(((long)ls.count + RC - 1) / RC);                                  \
		hipLaunchKernelGGL((sfw_kernel<B, UP, U>), dim3(grid), dim3(256), 0, s, a, ptr, cols, head, ls.rows, \
		                   ls.count, ls.ctl);                                                              \
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
