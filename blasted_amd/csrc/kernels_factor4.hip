// kernels_factor4.hip -- tuned asynchronous block-ILU(0) factorisation sweep for bs = 4, column-major
// blocks (async_block_ilu0_factorize, src/kernels/kernels_ilu0_factorize.hpp:71-98; sweep driver
// src/async_blockilu_factor.cpp:186-204).  Same fixed-point map as factor_sweep_kernel in
// kernels_factor.hip; what differs is how the 4x4x4 block products are done and how indices arrive:
//
//  * one wave owns 4 block-rows at a time and all block products run on the matrix core:
//    v_mfma_f64_4x4x4_4b_f64 computes four independent 4x4x4 products per instruction.  Its operand
//    layout (probed on gfx950, tools/probes/mfma_f64_4x4x4_probe.hip): for block slot b
//        A(i,k) in lane 16k + 4b + i,   B(k,j) in lane 16k + 4b + j,   C/D(i,j) in lane 16i + 4b + j.
//    A block is therefore loaded straight into operand layout by giving every lane the right element
//    offset inside the 128-byte block -- "offA" (element (r = L%4, c = L/16)) or "offD" (element
//    (r = L/16, c = L%4)) -- and no lane ever exchanges data for a product.
//  * upper / diagonal blocks:  S = A - sum L U      with A-operand = L (offA), B-operand = U (offD);
//    S comes out in D layout and is stored with offD.
//  * lower blocks need S as the A operand of the final product S * inverse(U_jj).  Computing the
//    TRANSPOSED sum  S^T = A^T - sum U^T L^T  (A-operand = U loaded with offD, B-operand = L loaded
//    with offA: the same two loads, roles swapped) leaves S exactly in A-operand layout.  Both kinds of
//    block can sit in one wave: the role swap is a per-lane select.
//  * inverse(U_jj) is formed in B-operand layout by the adjugate (Eigen's closed form for n <= 4), the
//    only place lanes exchange values (17 double shuffles per lower block, four blocks at a time).
//  * browptr, bcolind, posptr and the (lower, upper) position pairs of a 64-row chunk are staged in
//    LDS once, coalesced; no value load waits on an index load from HBM.
// Every entry of the factor is produced in registers and stored once per sweep.
#include "ctx.hpp"
#include "lanes.hpp"
#include "stage.hpp"

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

namespace bhip {

namespace {

constexpr int F4_RCHUNK = 64;             // rows per workgroup
constexpr int F4_CAPB = 16 * F4_RCHUNK;   // staged block positions (column index + posptr)
constexpr int F4_CAPP = 16 * F4_RCHUNK;   // staged (lower, upper) pairs

__device__ __forceinline__ double mfma444(const double a, const double b, const double c)
{
	return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// v holds U(r = k, c = j) in lane 16k + 4b + j (B-operand layout).  Returns inverse(U)(k, j) in the same
// lane: adjugate / determinant.
__device__ __forceinline__ double inverse_b_layout(const double v, const int k, const int b4, const int j)
{
	// cofactor C(j,k): delete row j and column k
	double M[3][3];
#pragma unroll
	for (int x = 0; x < 3; x++)
#pragma unroll
		for (int y = 0; y < 3; y++) {
			const int ri = x + (x >= j ? 1 : 0);
			const int ci = y + (y >= k ? 1 : 0);
			M[x][y] = __shfl(v, 16 * ri + b4 + ci, 64);
		}
	const double minor = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) -
	                     M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
	                     M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
	const double cof = ((j + k) & 1) ? -minor : minor;  // C(j,k) -> inverse(k,j) = C(j,k)/det
	// det = sum_q U(0,q) C(0,q);  C(0,q) is held by the lane computing inverse(q,0): lane 16q + 4b
	double det = 0.0;
#pragma unroll
	for (int q = 0; q < 4; q++)
		det += __shfl(v, b4 + q, 64) * __shfl(cof, 16 * q + b4, 64);
	return cof * (1.0 / det);
}

// Round 3, tried and removed: an up-front row path as in kernels_factor8.hip (a wave whose four rows are stencil-like
// requests their matrix blocks, the u_jj of their lower blocks and the u_kj of their pairs together -- 5 KB in flight
// instead of ~1 KB -- and hands finished lower blocks on in registers instead of reading them back behind their
// store).  Verified against the exact factor, 88 registers (five waves per SIMD instead of eight), and SLOWER:
// 128^3 0.853 -> 0.922 ms, 256^3 6.86 -> 7.41 ms per sweep.  At bs=8 the same change is worth 27 %; here the sweep
// already moves 36.6 GB in 6.9 ms = 5.3 TB/s of read+write traffic, near what a copy reaches on these boxes
// (profiles/r01l_stream_ceiling.txt: 5.8-6.3 TB/s), so there is no latency left to hide and the lost waves cost.
// Round 4, after request-ahead had paid in kernels_factorodd.hip (12-25 %): here, at the same eight waves and 62
// registers, the operand blocks of pair k + 1 requested before the product of pair k cost 1.5 %, and the position of a
// lower block's u_jj requested one block ahead 10 % (profiles/r04_factor4_pipe_ab.txt).  Not kept.
//
// RM (round 4): ROW-major blocks (the reference instantiates bs = 4 RowMajor for its own drivers,
// src/solverops_ilu0.cpp:390).  A row-major block is the column-major image of its transpose, so the kernel works on
// X' = X^T throughout: U'_ij = A'_ij - sum U'_kj L'_ik (the two operand loads swap their element offsets, the product
// its operands) and L'_ij = inverse(U'_jj) S' -- a product from the LEFT: S' comes out in D layout, which is also the
// B-operand layout, and inverse(U'_jj) = inverse(U_jj)^T in A-operand layout is inverse(U_jj) in B-operand layout,
// i.e. what inverse_b_layout returns when u_jj is loaded with the transposed offsets.  No role swap for lower blocks.
template <bool RM>
__global__ __launch_bounds__(256) void factor4_kernel(const FactorArgs a)
{
	__shared__ int s_rp[F4_RCHUNK + 1];
	__shared__ int s_col[F4_CAPB];
	__shared__ int s_pp[F4_CAPB + 1];
	__shared__ int s_lp[F4_CAPP];
	__shared__ int s_up[F4_CAPP];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int k = lane >> 4, b = (lane >> 2) & 3, m = lane & 3;
	const int b4 = 4 * b;
	const int offA = k * 4 + m;  // element (r = m, c = k)
	const int offD = m * 4 + k;  // element (r = k, c = m)

	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk(blockIdx.x, gridDim.x);
	const int r0 = (int)chunk * F4_RCHUNK;
	const int rc = (nb - r0) < F4_RCHUNK ? (nb - r0) : F4_RCHUNK;

	int jlo, plo;
	stage_factor_indices<F4_RCHUNK, F4_CAPB, F4_CAPP>(a.pat, a.posptr, a.lowerp, a.upperp, r0, rc, tid, s_rp, s_col,
	                                                   s_pp, s_lp, s_up, jlo, plo);

	for (int step = 0; step < F4_RCHUNK / 16; step++) {
		const int ls = step * 16 + wave * 4 + b;  // this block slot's row inside the chunk
		const bool rowok = ls < rc;
		const int irow = r0 + ls;
		const int jbeg = rowok ? s_rp[ls] : 0;
		const int len = rowok ? s_rp[ls + 1] - jbeg : 0;

		for (int p = 0; __any(p < len); p++) {
			const bool inrow = p < len;
			const int jpos = jbeg + p;
			const int bidx = jpos - jlo;
			int col = 0, kb = 0, ke = 0;
			if (inrow) {
				if (bidx < F4_CAPB) {
					col = s_col[bidx];
					kb = s_pp[bidx];
					ke = s_pp[bidx + 1];
				} else {
					col = a.pat.bcolind[jpos];
					kb = a.posptr[jpos];
					ke = a.posptr[jpos + 1];
				}
			}
			// (sweeps after the first, in place: an upper block without pairs already holds its value, a_ij)
			const bool valid = inrow && !(a.skip_fixed && col > irow && ke == kb);
			const bool lower = valid && irow > col;
			// lower blocks work on S^T (A-operand layout), the others on S (D layout); row-major: everything on S' (D layout)
			const int offS = (lower && !RM) ? offA : offD;
			double sval = valid ? a.avals[(long)jpos * 16 + offS] : 0.0;
			double dval = 0.0;  // U_jj in B-operand layout, lower blocks only
			if (lower)
				dval = a.in[(long)a.pat.diagind[col] * 16 + (RM ? offA : offD)];
			if (a.scale && valid) {
				// (r,c) of this lane's entry: D layout (k, m); transposed layout (m, k); row-major D layout holds X'(k, m) = X(m, k)
				const int r = (lower || RM) ? m : k, c = (lower || RM) ? k : m;
				sval *= a.scale[(long)irow * 4 + r] * a.scale[(long)col * 4 + c];
			}

			double acc = 0.0;
			const int cnt = ke - kb;
			for (int kk = 0; __any(kk < cnt); kk++) {
				double lval = 0.0, uval = 0.0;
				if (kk < cnt) {
					const int pidx = kb + kk - plo;
					int lp, up;
					if (pidx < F4_CAPP) {
						lp = s_lp[pidx];
						up = s_up[pidx];
					} else {
						lp = a.lowerp[kb + kk];
						up = a.upperp[kb + kk];
					}
					// (lrow_fresh -- the fused first sweep of a build, in = the matrix, out = the factor: the row's own lower
					// block l_ik is read back from `out`, where other lanes of THIS wave stored it a few instructions
					// ago with plain stores.  That relies on in-wave memory ordering: a wave's vector-memory accesses to
					// one address are performed in program order -- the stores above are issued before this load and the
					// hardware returns the stored data (same wave, same address, same cache path; the compiler cannot
					// reorder them: `out` is not restrict-qualified and the accesses may alias).  No other wave writes
					// this row.  tests/test_gpu_parity.py::test_fused_initialisation_builds_the_same_factor runs the
					// fused build on factor storage POISONED with another matrix's factor: a stale read would surface.)
					lval = (a.lrow_fresh ? a.out : a.in)[(long)lp * 16 + (RM ? offD : offA)];
					uval = a.in[(long)up * 16 + (RM ? offA : offD)];
				}
				// upper/diag: sum += L U ; lower: sum^T += U^T L^T (same loads, roles swapped); row-major: sum' += U' L'
				if (RM)
					acc = mfma444(uval, lval, acc);
				else
					acc = mfma444(lower ? uval : lval, lower ? lval : uval, acc);
			}
			double res = sval - acc;

			if (__any(lower)) {
				const double inv = inverse_b_layout(lower ? dval : ((k == m) ? 1.0 : 0.0), k, b4, m);
				const double prod = RM ? mfma444(lower ? inv : 0.0, lower ? res : 0.0, 0.0)
				                       : mfma444(lower ? res : 0.0, lower ? inv : 0.0, 0.0);
				if (lower)
					res = prod;  // S * inverse(U_jj), D layout (row-major: inverse(U'_jj) * S')
			}

			if (valid) {
				double *const dst = a.out + (long)jpos * 16 + offD;
				*dst = res;
			}
		}
	}
}

// ---- the exact factorisation as one launch, bs = 4 (round 2) ------------------------------------------------
// The general single-launch kernel (sff_factor_kernel, kernels_factor.hip) asks for a row's operands entry by
// entry, through four dependent index loads per row.  Here a row is prepared BEFORE it waits:
//   - its plan -- block positions of everything it reads, the shape of its pair lists -- is one 64-byte record,
//     written once per pattern in (padded) level order (x4_describe_kernel);
//   - all its operand blocks -- the A blocks and the (possibly still pending) blocks of other rows -- are requested
//     up front into registers, in matrix-core operand layout.
// What remains between "my predecessors have published" and "I have published" is a coherent re-read of what was
// pending, the products and the inverse on registers, and the stores.
// One wave = four block slots = four rows of ONE level (workgroups are laid out per level, so that the rows of a
// wave never depend on each other and the wave can walk its entries in lockstep, q = 0, 1, ... with compile-time
// register indices).  For stencil-like rows only: at most X4_MAXE entries, X4_MAXL of them lower, X4_MAXP position
// pairs per row; other patterns take the general kernel.
// Measured (MI355X, ms per exact factorisation, one launch per level -> general single launch -> this):
//   256^3 bs=4 (766 levels) 20.1 -> 18.5 -> 8.6;  128^3 bs=4 (382 levels) 5.9 -> 5.9 -> 1.95.
// How it got there, at 256^3: first form (indices through the matrix's own arrays, level of a workgroup by binary
// search) 17.8; per-workgroup table 14.6; row plans 13.3; shape in scalar registers where the four rows agree
// 13.0; upper blocks without pairs stored by the fill pass instead 11.5; one wait per row with a single polling
// lane per wave (x4_rows) 10.1; register arrays sized for a 7-point row where the pattern is one 9.1, at six waves 8.6.  With nobody waiting (wrong factor, experiment) the launches took 10.4 when the real
// ones took 11.5: the rest is the rate at which four waves per SIMD (124 registers, 80 of them operand blocks) turn
// rows over.  Tried and dropped: a
// resident grid whose waves walk the units with the next plan requested ahead (15.5 against 14.1 for the form it
// was tried on); five waves per SIMD by register bound (spills: 31 ms); re-reading the pending operands of LATER
// entries while waiting for the current one (no change).
constexpr int X4_MAXE = 8, X4_MAXL = 4, X4_MAXP = 8;
constexpr unsigned long long X4_PENDING = 0xFFF8DEADBEEF0001ull;  // = SFF_PENDING (kernels_factor.hip)
constexpr int X4_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ bool x4_pending(const double v)
{
	return (unsigned long long)__double_as_longlong(v) == X4_PENDING;
}

__device__ __forceinline__ double x4_coherent(const double *p)
{
	return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
	                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void x4_publish(double *p, const double v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
	                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The plan of a row (16 ints, one per lane of its block slot), in padded level order (every level starts a new
// workgroup of 16 rows): what the kernel below would otherwise collect through four dependent index loads (level
// order -> row pointers -> column indices / position-list pointers -> pairs / diagonal positions).
//   0 first block position of the row   1 block-row   2..9 position of the upper block of pair tt
//   10..13 position of the diagonal block of the column of lower entry q
//   14: ne | nl << 4 | np << 8 | code(0..3) << (12 + 5 tt)     15: code(4..7) << 5 (tt - 4)
//   code(tt) = (entry the pair belongs to) | (lower entry of this row it multiplies) << 3
constexpr int X4_DESC = 16;

__global__ __launch_bounds__(256) void x4_describe_kernel(const FactorArgs a, const int4 *__restrict__ meta,
                                                          const int2 *__restrict__ wgpos, const int nwg,
                                                          const int rpwg, int *__restrict__ desc)
{
	const long gid = (long)blockIdx.x * 256 + threadIdx.x;
	const long slot = gid >> 4;  // padded position
	const int t = (int)(gid & 15);
	if (slot >= (long)nwg * rpwg)
		return;
	const int2 wp = wgpos[slot / rpwg];
	const int pos = wp.x + (int)(slot % rpwg);
	int w = 0;
	if (pos < wp.y) {
		const int4 md = meta[pos];
		const int jbeg = md.y, ne = md.w - md.y, nl = md.z - md.y;
		const int pbeg = a.posptr[jbeg], np = a.posptr[md.w] - pbeg;
		if (t == 0)
			w = jbeg;
		else if (t == 1)
			w = md.x;
		else if (t < 10)
			w = (t - 2 < np) ? a.upperp[pbeg + t - 2] : 0;
		else if (t < 14)
			w = (t - 10 < nl) ? a.pat.diagind[a.pat.bcolind[jbeg + t - 10]] : 0;
		else {
			unsigned bits = (t == 14) ? (unsigned)(ne | (nl << 4) | (np << 8)) : 0u;
			const int tt0 = (t == 14) ? 0 : 4, sh0 = (t == 14) ? 12 : 0;
			for (int x = 0; x < 4; x++) {
				const int tt = tt0 + x;
				if (tt < np) {
					int q = 0;
					while (q + 1 < ne && a.posptr[jbeg + q + 1] - pbeg <= tt)
						q++;
					const int ll = a.lowerp[pbeg + tt] - jbeg;
					bits |= (unsigned)(q | (ll << 3)) << (sh0 + 5 * x);
				}
			}
			w = (int)bits;
		}
	}
	desc[gid] = w;
}

// Before the launch: the diagonal + upper part of every row <- the fill pattern, except upper blocks WITHOUT
// position pairs: their factor value is the (scaled) matrix block, which is stored here, in natural order at
// streaming rate, instead of going through the dependency kernel (3 of a 7-point row's 7 blocks).
__global__ __launch_bounds__(256) void x4_fill_kernel(const FactorArgs a)
{
	const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
	const int t = threadIdx.x & 15, base = threadIdx.x & 48;  // 16 lanes per row (rows of at most X4_MAXE blocks)
	const bool rowok = row < a.pat.nbrows;
	const int dg = rowok ? a.pat.diagind[row] : 0;
	const int nu = rowok ? a.pat.browptr[row + 1] - dg : 0;  // diagonal + upper blocks
	const int pp = (t <= nu && rowok) ? a.posptr[dg + t] : 0;
	const int colr = (a.scale && t < nu) ? a.pat.bcolind[dg + t] : 0;
	bool copy[X4_MAXE];
	double v[X4_MAXE];
#pragma unroll
	for (int x = 0; x < X4_MAXE; x++) {
		const int p0 = __shfl(pp, base + x, 64), p1 = __shfl(pp, base + x + 1, 64);
		copy[x] = x > 0 && x < nu && p0 == p1;
		v[x] = copy[x] ? a.avals[(long)(dg + x) * 16 + t] : 0.0;
	}
	unsigned long long *const fq = reinterpret_cast<unsigned long long *>(a.out);
#pragma unroll
	for (int x = 0; x < X4_MAXE; x++) {
		const int cx = __shfl(colr, base + x, 64);
		if (copy[x]) {
			if (a.scale)
				v[x] *= a.scale[(long)row * 4 + (a.pat.rowmajor ? t >> 2 : t & 3)] *
				        a.scale[(long)cx * 4 + (a.pat.rowmajor ? t & 3 : t >> 2)];
			a.out[(long)(dg + x) * 16 + t] = v[x];
		} else if (x < nu)
			fq[(long)(dg + x) * 16 + t] = X4_PENDING;
	}
}

// The four rows of a wave.  UNI: all of them (but for empty slots at the end of a level) have the same shape --
// the same plan words 14 and 15, which is what the interior rows of a stencil look like -- and the shape arrives in
// scalar registers: which entry is lower, which pair belongs to which entry, where a row ends are then scalar
// branches instead of 64-bit lane masks and selects (the instruction issue of exactly that bookkeeping is what
// bounded the first form of this kernel: 11 ms at 256^3 with nobody waiting).  Returns false if the wave gave up.
// ME / ML / MP: how many entries that need work, lower entries and position pairs a row may have (the register
// arrays): 8 / 4 / 8 in general; 4 / 3 / 4 -- a 7-point row -- leaves room for a wave more per SIMD.
template <bool UNI, int ME, int ML, int MP>
__device__ __forceinline__ bool x4_rows(const FactorArgs &a, const int dw, const unsigned w14, const unsigned w15,
                                        const bool ok, const int lane, int *ctl)
{
	const int k = lane >> 4, b4 = ((lane >> 2) & 3) * 4, m = lane & 3;
	const int t = 4 * k + m;     // index of this lane inside its block slot
	const int offA = a.pat.rowmajor ? m * 4 + k : k * 4 + m;  // element (r = m, c = k)
	const int offD = a.pat.rowmajor ? k * 4 + m : m * 4 + k;  // element (r = k, c = m)
	double *const f = a.out;
#define X4_SLOTLANE(Q) (16 * ((Q) >> 2) + b4 + ((Q) & 3))
	const int jbeg = __shfl(dw, X4_SLOTLANE(0), 64);
	const int ne = (int)(w14 & 15u), nl = (int)((w14 >> 4) & 15u), np = (int)((w14 >> 8) & 15u);
	// pair tt: the entry it belongs to (8 = none), the lower entry of this row it multiplies
#define X4_CODE(TT) ((TT) < 4 ? (w14 >> (12 + 5 * (TT))) : (w15 >> (5 * ((TT)-4))))
#define X4_PQ(TT) ((TT) < np ? (int)(X4_CODE(TT) & 7u) : 8)
#define X4_PLL(TT) ((int)((X4_CODE(TT) >> 3) & 3u))
#define X4_ANY(C) (UNI ? (C) : (__builtin_amdgcn_ballot_w64(C) != 0ull))
	// entries with position pairs; an upper entry without any has been stored by x4_fill_kernel already
	unsigned pmask = 0u;
#pragma unroll
	for (int tt = 0; tt < MP; tt++)
		pmask |= (tt < np) ? (1u << (X4_CODE(tt) & 7u)) : 0u;
#define X4_TODO(Q) ((Q) < ne && ((Q) <= nl || ((pmask >> (Q)) & 1u) != 0u))

	// ---- operands, in operand layout; blocks of other rows may still show the fill pattern
	// (lane exchanges are kept out of divergent code: a lane that is switched off hands out nothing)
	double aS[ME], uvD[MP], dvB[ML];
#pragma unroll
	for (int q = 0; q < ME; q++)
		aS[q] = (X4_TODO(q) && ok) ? a.avals[(long)(jbeg + q) * 16 + (q < nl ? offA : offD)] : 0.0;
#pragma unroll
	for (int tt = 0; tt < MP; tt++) {
		const int up = __shfl(dw, X4_SLOTLANE(2 + tt), 64);
		uvD[tt] = (tt < np && ok) ? f[(long)up * 16 + offD] : 0.0;
	}
#pragma unroll
	for (int q = 0; q < ML; q++) {
		const int dp = __shfl(dw, X4_SLOTLANE(10 + q), 64);
		dvB[q] = (q < nl && ok) ? f[(long)dp * 16 + offD] : 0.0;
	}
	if (a.scale) {
		const int irow = __shfl(dw, X4_SLOTLANE(1), 64);
		const int colr = (t < ne && ok) ? a.pat.bcolind[jbeg + t] : 0;
#pragma unroll
		for (int q = 0; q < ME; q++) {
			const int cq = __shfl(colr, X4_SLOTLANE(q), 64);
			// (r,c) of this lane's element: D layout (k,m), transposed (m,k)
			const int r = q < nl ? m : k, c = q < nl ? k : m;
			if (X4_TODO(q) && ok)
				aS[q] *= a.scale[(long)irow * 4 + r] * a.scale[(long)cq * 4 + c];
		}
	}

	// ---- the rows' recurrence, entry by entry, the four slots in lockstep
	double lresA[ML];  // finished lower blocks of the row, as element (r = m, c = k): the A-operand layout
#pragma unroll
	for (int q = 0; q < ML; q++)
		lresA[q] = 0.0;
	// ---- wait until everything the rows of this wave read from other rows has been published.  ONE lane polls, for
	// the whole wave, the element the first waiting lane misses; when that has arrived everybody re-reads coherently
	// what still shows the fill pattern.  (The first form waited entry by entry with every lane re-reading its own
	// operands in a loop: a round trip per entry, and thousands of resident waves polling like that load the memory
	// system enough to stretch a dependency hop from the 0.6 us of tools/probes/pingpong_probe.hip to several.)
	int spins = 0;
	for (;;) {
		const double *miss = nullptr;
		int up[MP], dp[ML];
#pragma unroll
		for (int tt = MP - 1; tt >= 0; tt--) {
			up[tt] = __shfl(dw, X4_SLOTLANE(2 + tt), 64);
			if (tt < np && x4_pending(uvD[tt]))
				miss = f + (long)up[tt] * 16 + offD;
		}
#pragma unroll
		for (int q = ML - 1; q >= 0; q--) {
			dp[q] = __shfl(dw, X4_SLOTLANE(10 + q), 64);
			if (q < nl && x4_pending(dvB[q]))
				miss = f + (long)dp[q] * 16 + offD;
		}
		const unsigned long long waiting = __builtin_amdgcn_ballot_w64(miss != nullptr);
		if (waiting == 0ull)
			break;
		const int lead = __builtin_ctzll(waiting);
		const unsigned long long addr = (unsigned long long)reinterpret_cast<uintptr_t>(miss);
		const unsigned alo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)addr, lead);
		const unsigned ahi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(addr >> 32), lead);
		const double *const gate = reinterpret_cast<const double *>((uintptr_t)(((unsigned long long)ahi << 32) | alo));
		while (x4_pending(x4_coherent(gate))) {
			spins++;
			if (spins > X4_SPIN_LIMIT ||
			    ((spins & 255) == 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
				if (lane == 0)
					__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return false;
			}
			__builtin_amdgcn_s_sleep(2);
		}
#pragma unroll
		for (int q = 0; q < ML; q++)
			if (q < nl && x4_pending(dvB[q]))
				dvB[q] = x4_coherent(f + (long)dp[q] * 16 + offD);
#pragma unroll
		for (int tt = 0; tt < MP; tt++)
			if (tt < np && x4_pending(uvD[tt]))
				uvD[tt] = x4_coherent(f + (long)up[tt] * 16 + offD);
		if (++spins > X4_SPIN_LIMIT) {
			if (lane == 0)
				__hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return false;
		}
	}

	// ---- the rows' recurrences, entry by entry, the four slots in lockstep, on registers
#pragma unroll
	for (int q = 0; q < ME; q++) {
		if (!X4_ANY(q < ne))
			break;
		const bool valid = X4_TODO(q);
		if (!X4_ANY(valid))
			continue;
		const bool lowerq = q < nl, diagq = valid && q == nl;
		// S = A - sum L U (upper / diagonal entries, D layout) or S^T = A^T - sum U^T L^T (lower entries, A layout).
		// Pair by pair, each product summed from zero and then subtracted: the order of the general kernels
		// (s -= group_gemm(l, u), kernels_factor.hip), so that the factor is the same bits.
		double res = aS[q];
#pragma unroll
		for (int tt = 0; tt < MP; tt++) {
			const bool in = X4_PQ(tt) == q;
			if (!X4_ANY(in))
				continue;
			// l_ik: lower entry number pll of this row, finished above, as element (r = m, c = k)
			const int ll = X4_PLL(tt);
			const double lA = ll == 0 ? lresA[0] : (ll == 1 ? lresA[1] : ((ll == 2 || ML < 4) ? lresA[2] : lresA[ML < 4 ? 2 : 3]));
			const double lv = (UNI || in) ? lA : 0.0, uv = (UNI || in) ? uvD[tt] : 0.0;
			res -= mfma444(lowerq ? uv : lv, lowerq ? lv : uv, 0.0);
		}
		if (q < ML && X4_ANY(valid && lowerq)) {
			// S * inverse(U_jj): diagonal blocks are stored inverted as soon as they are final
			const bool lw = valid && lowerq;
			const double prod = mfma444((UNI || lw) ? res : 0.0, (UNI || lw) ? dvB[q < ML ? q : 0] : 0.0, 0.0);
			if (lw)
				res = prod;
			const double tr = __shfl(res, 16 * m + b4 + k, 64);  // the block transposed inside its slot
			lresA[q < ML ? q : 0] = lw ? tr : 0.0;
		}
		if (q <= ML && X4_ANY(diagq)) {
			const double inv = inverse_b_layout((diagq && ok) ? res : ((k == m) ? 1.0 : 0.0), k, b4, m);
			if (diagq)
				res = inv;
		}
		if (valid && ok) {
			double *const dst = f + (long)(jbeg + q) * 16 + offD;
			if (lowerq)
				*dst = res;  // read by this row (from registers) and by the triangular solves later
			else
				x4_publish(dst, res);
		}
	}
	return true;
#undef X4_SLOTLANE
#undef X4_CODE
#undef X4_PQ
#undef X4_PLL
#undef X4_ANY
#undef X4_TODO
}

// One wave = one unit of four rows of one level; workgroups in level order (that the row with the lowest number
// among the unfinished ones can always move rests on workgroups being started in the order of their numbers, as
// for the other single-launch kernels).
// (the occupancy bound of the small instantiation only keeps two stray accumulator registers out of the count:
// 78 + 2 registers are five waves per SIMD, 78 are six)
template <int ME, int ML, int MP>
__global__ __launch_bounds__(256, ME == 4 ? 6 : 1) void sffactor4_kernel(const FactorArgs a, const int *__restrict__ desc, int *ctl)
{
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int b = (lane >> 2) & 3, t = 4 * (lane >> 4) + (lane & 3);
	// the rows' plans: one word per lane of the slot (a wave reads 256 consecutive bytes)
	const int dw = desc[((long)blockIdx.x * 16 + (tid >> 6) * 4 + b) * X4_DESC + t];
	const unsigned w14 = (unsigned)__shfl(dw, 48 + 4 * b + 2, 64), w15 = (unsigned)__shfl(dw, 48 + 4 * b + 3, 64);
	const unsigned s14 = (unsigned)__builtin_amdgcn_readfirstlane((int)w14);
	const unsigned s15 = (unsigned)__builtin_amdgcn_readfirstlane((int)w15);
	if (s14 == 0u)
		return;  // (slots fill up in order: an empty first slot is an empty wave)
	const bool same = (w14 == s14 && w15 == s15) || w14 == 0u;
	if (__builtin_amdgcn_ballot_w64(!same) == 0ull)
		(void)x4_rows<true, ME, ML, MP>(a, dw, s14, s15, w14 != 0u, lane, ctl);
	else
		(void)x4_rows<false, ME, ML, MP>(a, dw, w14, w15, w14 != 0u, lane, ctl);
}

// max over the rows of the number of position pairs of a row
// out[1]: max over the rows of 1 + the index, inside the row, of the last entry that needs work (the lower ones, the
// diagonal, upper ones with position pairs)
__global__ __launch_bounds__(256) void max_pairs_kernel(const Pattern pat, const int *__restrict__ posptr, int *out)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= pat.nbrows)
		return;
	const int rp0 = pat.browptr[i], rp1 = pat.browptr[i + 1], dg = pat.diagind[i];
	atomicMax(out, posptr[rp1] - posptr[rp0]);
	int last = dg - rp0 + 1;
	for (int j = dg + 1; j < rp1; j++)
		if (posptr[j + 1] != posptr[j])
			last = j - rp0 + 1;
	atomicMax(out + 1, last);
}

int g_factor4_enabled = -1;
int g_x4_small = 1;  // "factor4=s0|s1": the small-array instantiation of the single-launch exact factorisation

}  // namespace

// Once per pattern: the longest pair list of a row, the per-level workgroup layout (rpwg rows per workgroup, every
// level starting a new one) and the rows' plans in that padded level order.  false: the pattern's rows do not fit
// the caps of the plan kernels (X4_MAXE entries, X4_MAXL lower ones, X4_MAXP pairs).
bool build_row_plans(const FactorArgs &a, LevelSchedule &ls, int rpwg, hipStream_t s)
{
	if (a.pat.max_row_len > X4_MAXE || ls.max_lower > X4_MAXL)
		return false;
	if (ls.f4_maxpairs < 0) {
		// Nothing of the cache (f4_maxpairs, f4_maxtodo, f4_grid, f4_desc) is committed before the plans exist on the
		// device: a failure on the way (out of memory for the ~1 GB plan array at 256^3, a failed launch) leaves
		// f4_maxpairs < 0 and frees what was allocated, so the next factorisation tries again instead of launching a
		// plan kernel with a null plan pointer.
		int *d = nullptr;
		int2 *wgpos = nullptr;
		int *desc = nullptr;
		try {
			BHIP_CHECK(tracked_malloc(&d, 2 * sizeof(int)));
			BHIP_CHECK(hipMemsetAsync(d, 0, 2 * sizeof(int), s));
			hipLaunchKernelGGL(max_pairs_kernel, dim3((unsigned)((a.pat.nbrows + 255) / 256)), dim3(256), 0, s, a.pat,
			                   a.posptr, d);
			int h[2] = {0, 0};
			BHIP_CHECK(hipMemcpyAsync(h, d, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
			BHIP_CHECK(hipStreamSynchronize(s));
			(void)tracked_free(d);
			d = nullptr;
			const int maxpairs = h[0], maxtodo = h[1];
			if (maxpairs > X4_MAXP) {
				ls.f4_maxtodo = maxtodo;
				ls.f4_maxpairs = maxpairs;
				return false;
			}
			std::vector<int2> wg;
			for (int l = 0; l < ls.nlevels; l++)
				for (int q = ls.ptr[(size_t)l]; q < ls.ptr[(size_t)l + 1]; q += rpwg)
					wg.push_back(make_int2(q, std::min(q + rpwg, ls.ptr[(size_t)l + 1])));
			if ((long)wg.size() * rpwg > 2L * a.pat.nbrows + 4096) {
				// narrow levels (a banded / one-dimensional ordering): padding every level to a workgroup would
				// multiply the plan array -- such a pattern keeps the kernels that take rows as they come
				ls.f4_maxtodo = maxtodo;
				ls.f4_maxpairs = 1 << 30;
				return false;
			}
			const int grid = (int)wg.size();
			BHIP_CHECK(tracked_malloc(&wgpos, sizeof(int2) * wg.size()));
			BHIP_CHECK(hipMemcpyAsync(wgpos, wg.data(), sizeof(int2) * wg.size(), hipMemcpyHostToDevice, s));
			const long nd = (long)grid * rpwg * X4_DESC;
			BHIP_CHECK(tracked_malloc(&desc, sizeof(int) * (size_t)nd));
			hipLaunchKernelGGL(x4_describe_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, a, ls.meta, wgpos,
			                   grid, rpwg, desc);
			BHIP_CHECK(hipGetLastError());
			BHIP_CHECK(hipStreamSynchronize(s));
			(void)tracked_free(wgpos);
			wgpos = nullptr;
			ls.f4_desc = desc;
			ls.f4_grid = grid;
			ls.f4_maxtodo = maxtodo;
			ls.f4_maxpairs = maxpairs;  // last: this is what marks the cache valid
		} catch (...) {
			if (d)
				(void)tracked_free(d);
			if (wgpos)
				(void)tracked_free(wgpos);
			if (desc)
				(void)tracked_free(desc);
			throw;
		}
	}
	return ls.f4_maxpairs <= X4_MAXP && ls.f4_desc != nullptr;
}

// The exact factorisation of a bs = 4 matrix (either block layout) with stencil-like rows as one launch (see
// sffactor4_kernel): 1 = done, 0 = does not apply, -1 = a wave gave up waiting.
int launch_factor4_syncfree(const FactorArgs &a, LevelSchedule &ls, hipStream_t s)
{
	if (a.pat.bs != 4 || !a.diag_inverted || a.in != a.out || !ls.built || !ls.meta || !ls.ctl)
		return 0;
	if (!build_row_plans(a, ls, 16, s))
		return 0;
	BHIP_CHECK(hipMemsetAsync(ls.ctl, 0, 2 * sizeof(int), s));
	hipLaunchKernelGGL(x4_fill_kernel, dim3((unsigned)(((long)a.pat.nbrows + 15) / 16)), dim3(256), 0, s, a);
	// a 7-point-like pattern (at most 3 lower entries, 4 pairs, and nothing to do beyond a row's fourth entry) takes
	// the instantiation with the small register arrays: 78 instead of 124 registers, a wave or two more per SIMD
	if (g_x4_small && ls.max_lower <= 3 && ls.f4_maxpairs <= 4 && ls.f4_maxtodo <= 4)
		hipLaunchKernelGGL((sffactor4_kernel<4, 3, 4>), dim3((unsigned)ls.f4_grid), dim3(256), 0, s, a, ls.f4_desc, ls.ctl);
	else
		hipLaunchKernelGGL((sffactor4_kernel<X4_MAXE, X4_MAXL, X4_MAXP>), dim3((unsigned)ls.f4_grid), dim3(256), 0, s, a,
		                   ls.f4_desc, ls.ctl);
	BHIP_CHECK(hipGetLastError());
	int ctl[2] = {0, 0};
	BHIP_CHECK(hipMemcpyAsync(ctl, ls.ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
	BHIP_CHECK(hipStreamSynchronize(s));
	return ctl[1] == 0 ? 1 : -1;
}

void set_factor4_enabled(int on)
{
	if (on >= 10)
		g_x4_small = on - 10;
	else
		g_factor4_enabled = on;
}

// returns false when the tuned kernel does not cover the request (caller uses the generic kernel)
bool launch_factor4(const FactorArgs &a, hipStream_t s)
{
	if (g_factor4_enabled < 0) {
		const char *e = std::getenv("BLASTED_HIP_FACTOR4");
		g_factor4_enabled = (e && std::strcmp(e, "0") == 0) ? 0 : 1;
	}
	if (!g_factor4_enabled || a.pat.bs != 4 || a.pat.nbrows == 0)
		return false;
	const unsigned grid = (unsigned)(((long)a.pat.nbrows + F4_RCHUNK - 1) / F4_RCHUNK);
	if (a.pat.rowmajor)
		hipLaunchKernelGGL(factor4_kernel<true>, dim3(grid), dim3(256), 0, s, a);
	else
		hipLaunchKernelGGL(factor4_kernel<false>, dim3(grid), dim3(256), 0, s, a);
	BHIP_CHECK(hipGetLastError());
	return true;
}

}  // namespace bhip
