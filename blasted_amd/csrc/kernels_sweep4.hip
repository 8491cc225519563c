// kernels_sweep4.hip -- the tuned row-sweep kernel for bs = 4, column-major blocks (the PETSc BAIJ
// layout, src/blasted_petsc.cpp:256, and BASELINE.json's headline configuration).  Same operators and
// same arithmetic as the generic family in kernels_sweep.hip (see the table there); what differs is the
// data movement:
//
//  * a workgroup owns RCHUNK consecutive block-rows of the sweep; browptr / diagind and the bcolind
//    range of the whole chunk are read from HBM once, coalesced, into LDS, so no value load waits on
//    an index load from memory (the dependent chain per row is LDS -> {block, x segment} -> result);
//  * a 128-byte block is read by 8 lanes as 16 bytes each (global_load_dwordx4): lane q holds entries
//    (2q, 2q+1) = rows 2(q&1), 2(q&1)+1 of column q>>1;  NB blocks of a row are in flight per load
//    instruction (NB = 4: the three lower blocks of a 7-point row, or its diagonal + three upper
//    blocks, are one instruction);
//  * the x segment of a block is gathered as one double per lane (its column's entry), the 4x4
//    mat-vec is two FMAs per lane plus an xor-butterfly over the column bits and the block slots;
//  * the inverted diagonal block needed by the upper solve sits directly in front of the row's upper
//    blocks in memory and is fetched by the same load instruction as block slot 0;
//  * rhs is read and the result written as 16 bytes per lane (32 contiguous bytes per row).
#include "ctx.hpp"

#include <cstdlib>
#include <cstring>

namespace bhip {

typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned xcd_chunk4(unsigned bid, unsigned nwg)
{
	const unsigned xcd = bid & 7u, local = bid >> 3;
	const unsigned base = nwg >> 3, rem = nwg & 7u;
	return xcd * base + (xcd < rem ? xcd : rem) + local;
}

template <bool NT>
__device__ __forceinline__ double2_t load_block16(const double *p)
{
	if (NT)
		return __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(p));
	return *reinterpret_cast<const double2_t *>(p);
}

// ---- cross-lane sums on the VALU (DPP), keeping the LDS pipe free for the index reads.
// Lane numbering inside a 16-lane DPP row: bit 0 = h (row pair), bits 1-2 = c (column), bit 3 = block
// slot parity.  row_ror 8, 4, 2 applied in this order is an all-reduce over bits 3, 2, 1 (after each
// step the value is periodic in the bit just summed, so the wrap-around of the rotation lands on an
// equal value).  The 16-lane rows of a 32-lane group are combined with v_permlane16_swap.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(const double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row16_allreduce_c_slot(double v)
{
	v += dpp_mov<0x128>(v);  // row_ror:8
	v += dpp_mov<0x124>(v);  // row_ror:4
	v += dpp_mov<0x122>(v);  // row_ror:2
	return v;
}

__device__ __forceinline__ double pair_rows_sum(const double v)
{
	typedef unsigned v2u __attribute__((ext_vector_type(2)));
	const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
	const v2u a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
	const v2u b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
	return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}

template <int G>
__device__ __forceinline__ double group_allreduce(double v)
{
	v = row16_allreduce_c_slot(v);
	if (G == 32)
		v = pair_rows_sum(v);
	return v;
}

template <int PART, int POST, int DSRC, int NB, int RCHUNK, bool NT, int UNR>
__global__ __launch_bounds__(256) void sweep4_kernel(const SweepArgs a)
{
	constexpr int G = 8 * NB;            // lanes per block-row
	constexpr int RPW = 64 / G;          // rows per wave and step
	constexpr int RSTEP = 4 * RPW;       // rows per workgroup and step
	constexpr int CAP = 8 * RCHUNK;      // staged column indices
	static_assert(RCHUNK % (RSTEP * UNR) == 0, "chunk must be a multiple of the unrolled step");

	__shared__ int s_rp[RCHUNK + 1];
	__shared__ int s_dg[RCHUNK];
	__shared__ int s_col[CAP];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int g = lane / G, t = lane % G;
	const int slot = t >> 3, q = t & 7;
	const int c = q >> 1, h = q & 1;     // column of this lane's two entries, row pair (2h, 2h+1)

	const int nb = a.pat.nbrows;
	const unsigned chunk = xcd_chunk4(blockIdx.x, gridDim.x);
	// rows of this chunk in index order: [r0, r0 + rc)
	const long lin0 = (long)chunk * RCHUNK;
	int rc = (int)((nb - lin0) < RCHUNK ? (nb - lin0) : RCHUNK);
	const int r0 = a.descending ? (int)(nb - lin0 - rc) : (int)lin0;

	for (int k = tid; k <= rc; k += 256)
		s_rp[k] = a.pat.browptr[r0 + k];
	for (int k = tid; k < rc; k += 256)
		s_dg[k] = a.pat.diagind[r0 + k];
	__syncthreads();
	// column indices the chunk needs
	int jlo, jhi;
	if (PART == PART_LOWER) {
		jlo = s_rp[0];
		jhi = s_dg[rc - 1];
	} else if (PART == PART_UPPER) {
		jlo = s_dg[0];
		jhi = s_rp[rc];
	} else {
		jlo = s_rp[0];
		jhi = s_rp[rc];
	}
	jlo = __builtin_amdgcn_readfirstlane(jlo);  // wave-uniform by construction: let the compiler know
	jhi = __builtin_amdgcn_readfirstlane(jhi);
	if (PART != PART_NONE) {
		const int ncol = (jhi - jlo) < CAP ? (jhi - jlo) : CAP;
		for (int k = tid; k < ncol; k += 256)
			s_col[k] = a.pat.bcolind[jlo + k];
	}
	__syncthreads();

	// Chunk-relative addressing: wave-uniform 64-bit bases (SGPRs) + 32-bit per-lane byte offsets, which
	// halves the address registers and arithmetic of the hot loop.  (xin keeps the full base: column
	// indices are not chunk-local; n*8 < 4 GiB is checked on the host.)
	const char *const vbase = reinterpret_cast<const char *>(a.vals + (long)jlo * 16);
	const char *const xbase = reinterpret_cast<const char *>(a.xin);
	const char *const rbase = reinterpret_cast<const char *>(a.rhs + (long)r0 * 4);
	const char *const sbase = reinterpret_cast<const char *>(a.rscale + (long)r0 * 4);
	const char *const dbase = reinterpret_cast<const char *>(a.dvals + (long)r0 * 16);
	char *const obase = reinterpret_cast<char *>(a.xout + (long)r0 * 4);

	// Each pass of this loop handles UNR row steps.  All loads of the UNR steps (KFIX predicated block
	// passes per row, straight-line) are issued before the first use, so one wave keeps UNR*KFIX 16-byte
	// block loads plus as many x gathers in flight; rows with more than NB*KFIX blocks finish in a
	// remainder loop.
	// straight-line passes: 2*NB items cover a 7-point row's lower or diagonal+upper part; operators
	// that visit the whole row (SpMV, relaxation) get twice as many
	constexpr int KFIX = ((NB >= 4) ? 1 : 2) * ((PART == PART_ALL || PART == PART_OFFDIAG) ? 2 : 1);
	for (int step0 = 0; step0 < RCHUNK / RSTEP; step0 += UNR) {
		int lrow[UNR], jbeg[UNR], jend[UNR], dgp[UNR];
		bool ok[UNR];
#pragma unroll
		for (int u = 0; u < UNR; u++) {
			const int ls = (step0 + u) * RSTEP + wave * RPW + g;  // position in sweep order
			ok[u] = ls < rc;
			const int lr = ok[u] ? (a.descending ? rc - 1 - ls : ls) : 0;
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
#pragma unroll
		for (int u = 0; u < UNR; u++) {
#pragma unroll
			for (int k = 0; k < KFIX; k++) {
				const int jj = jbeg[u] + slot + k * NB;
				bv[u][k].x = 0.0;
				bv[u][k].y = 0.0;
				xv[u][k] = 0.0;
				if (PART != PART_NONE && jj < jend[u]) {
					bv[u][k] = load_block16<NT>(reinterpret_cast<const double *>(
					    vbase + ((unsigned)(jj - jlo) * 128u + 16u * (unsigned)q)));
					const bool isdiag = (jj == dgp[u]);
					if (!((PART == PART_UPPER && DSRC == D_VALS_DIAG && isdiag) ||
					      (PART == PART_OFFDIAG && isdiag))) {
						const int cidx = jj - jlo;
						const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
						xv[u][k] = *reinterpret_cast<const double *>(xbase + ((unsigned)col * 32u + 8u * (unsigned)c));
					}
				}
			}
			dv[u].x = 0.0;
			dv[u].y = 0.0;
			if (DSRC == D_DBLOCKS && ok[u] && slot == 0)
				dv[u] = load_block16<false>(reinterpret_cast<const double *>(
				    dbase + ((unsigned)lrow[u] * 128u + 16u * (unsigned)q)));
			r2[u].x = r2[u].y = 0.0;
			s2[u].x = s2[u].y = 1.0;
			if (ok[u] && a.rhs) {
				r2[u] = *reinterpret_cast<const double2_t *>(rbase + ((unsigned)lrow[u] * 32u + 16u * (unsigned)h));
				if (a.rscale)
					s2[u] = *reinterpret_cast<const double2_t *>(sbase + ((unsigned)lrow[u] * 32u + 16u * (unsigned)h));
			}
		}

#pragma unroll
		for (int u = 0; u < UNR; u++) {
			double d0 = dv[u].x, d1 = dv[u].y;  // entries (2q, 2q+1) of D, in block slot 0
			double acc0 = 0.0, acc1 = 0.0;
			if (PART != PART_NONE) {
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
				for (int jj = jbeg[u] + slot + KFIX * NB; jj < jend[u]; jj += NB) {
					if (PART == PART_OFFDIAG && jj == dgp[u])
						continue;
					const double2_t v2 = load_block16<NT>(reinterpret_cast<const double *>(
					    vbase + ((unsigned)(jj - jlo) * 128u + 16u * (unsigned)q)));
					const int cidx = jj - jlo;
					const int col = (cidx < CAP) ? s_col[cidx] : a.pat.bcolind[jj];
					const double xc = *reinterpret_cast<const double *>(xbase + ((unsigned)col * 32u + 8u * (unsigned)c));
					acc0 += v2.x * xc;
					acc1 += v2.y * xc;
				}
				acc0 = group_allreduce<G>(acc0);
				acc1 = group_allreduce<G>(acc1);
			}
			const double rv0 = r2[u].x * s2[u].x, rv1 = r2[u].y * s2[u].y;

			double o0, o1;
			if (POST == POST_SUB) {
				o0 = rv0 - acc0;
				o1 = rv1 - acc1;
			} else if (POST == POST_D_SUB || POST == POST_SUB_D) {
				const double w0 = (POST == POST_D_SUB) ? rv0 - acc0 : acc0;  // rows 2h, 2h+1 of the vector D multiplies
				const double w1 = (POST == POST_D_SUB) ? rv1 - acc1 : acc1;
				// component c = 2*b2 + b1 of that vector (b1, b2 = bits 1, 2 of the lane): rows 2*b2 and
				// 2*b2+1 are held by the lanes of this quad whose bit 0 equals b2
				const double a00 = dpp_mov<0x00>(w0), a01 = dpp_mov<0x00>(w1);  // quad_perm [0,0,0,0]
				const double a10 = dpp_mov<0x55>(w0), a11 = dpp_mov<0x55>(w1);  // quad_perm [1,1,1,1]
				const bool b1 = (q & 2) != 0, b2 = (q & 4) != 0;
				const double s0 = b2 ? a10 : a00, s1 = b2 ? a11 : a01;
				const double wc = b1 ? s1 : s0;
				// D lives in block slot 0 only (zero elsewhere): the row-wide all-reduce is its column sum
				const double p0 = row16_allreduce_c_slot(d0 * wc);
				const double p1 = row16_allreduce_c_slot(d1 * wc);
				if (POST == POST_D_SUB) {
					o0 = p0;
					o1 = p1;
				} else {
					o0 = rv0 - p0;
					o1 = rv1 - p1;
				}
			} else {
				o0 = a.a * acc0;
				o1 = a.a * acc1;
				if (a.b != 0.0) {
					o0 += a.b * rv0;
					o1 += a.b * rv1;
				}
			}

			if (ok[u] && slot == 0 && q < 2) {
				double2_t o2;
				o2.x = o0;
				o2.y = o1;
				double2_t *const dst = reinterpret_cast<double2_t *>(obase + ((unsigned)lrow[u] * 32u + 16u * (unsigned)q));
				if (a.changed) {
					const double2_t old = *dst;
					if (!(old.x == o0) || !(old.y == o1))
						*a.changed = 1;
				}
				*dst = o2;
			}
		}
	}
}

struct Variant {
	int nb = 2, rchunk = 128, nt = 1, unr = 2, enabled = 1;
};

static Variant parse_variant(const char *e)
{
	Variant v;
	// BLASTED_HIP_SWEEP4 = "generic" | "nb<2|4>,r<128|256>,nt<0|1>,u<1|2|4>"   (tuning / A-B measurements)
	if (!e)
		return v;
	if (std::strcmp(e, "generic") == 0) {
		v.enabled = 0;
		return v;
	}
	int nb = 2, r = 128, nt = 1, unr = 2;
	if (std::sscanf(e, "nb%d,r%d,nt%d,u%d", &nb, &r, &nt, &unr) == 4) {
		v.nb = nb;
		v.rchunk = r;
		v.nt = nt;
		v.unr = unr;
	}
	return v;
}

static Variant &current_variant()
{
	static Variant v = parse_variant(std::getenv("BLASTED_HIP_SWEEP4"));
	return v;
}

// tuning hook behind blasted_hip_set_tuning(): same syntax as the BLASTED_HIP_SWEEP4 variable
void set_sweep4_variant(const char *spec)
{
	current_variant() = parse_variant(spec);
}

template <int PART, int POST, int DSRC>
static bool launch_variant(const SweepArgs &a, const Variant &v, hipStream_t s)
{
#define BHIP_V(NBV, RV, NTV, UV)                                                                      \
	if (v.nb == NBV && v.rchunk == RV && v.nt == NTV && v.unr == UV) {                                \
		const unsigned grid = (unsigned)(((long)a.pat.nbrows + RV - 1) / RV);                         \
		constexpr int UEFF = (PART == PART_ALL || PART == PART_OFFDIAG) ? 1 : UV; /* 4 passes: keep 8 waves */ \
		hipLaunchKernelGGL((sweep4_kernel<PART, POST, DSRC, NBV, RV, (NTV != 0), UEFF>), dim3(grid),   \
		                   dim3(256), 0, s, a);                                                       \
		return true;                                                                                  \
	}
	BHIP_V(2, 256, 0, 1)
	BHIP_V(2, 256, 1, 1)
	BHIP_V(2, 128, 0, 1)
	BHIP_V(2, 128, 1, 1)
	BHIP_V(2, 64, 1, 1)
	BHIP_V(2, 64, 0, 1)
	BHIP_V(2, 256, 0, 2)
	BHIP_V(2, 128, 0, 2)
	BHIP_V(2, 128, 1, 2)
	BHIP_V(4, 256, 0, 1)
	BHIP_V(4, 128, 0, 1)
#undef BHIP_V
	return false;
}

// returns false when the tuned kernel does not cover the request (caller uses the generic family)
bool launch_sweep4(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s)
{
	const Variant &v = current_variant();
	if (!v.enabled || a.pat.bs != 4 || a.pat.rowmajor || a.pat.nbrows == 0)
		return false;
	// 32-bit byte offsets: the vector must be below 4 GiB and a 256-row chunk of blocks below 4 GiB
	// (33 M blocks; a matrix with such rows takes the generic family)
	if ((long)a.pat.nbrows * 32 >= (1L << 32) || (long)a.pat.nnzb >= (1L << 25) * 7)
		return false;
	// 16-byte loads need 16-byte aligned arrays (hipMalloc gives 256; borrowed pointers are checked)
	auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
	if (misaligned(a.vals) || misaligned(a.dvals) || misaligned(a.rhs) || misaligned(a.rscale) ||
	    misaligned(a.xout))
		return false;
	bool ok = false;
#define BHIP_CASE4(P, Q, D)                                   \
	if (part == P && post == Q && dsrc == D)                  \
		ok = launch_variant<P, Q, D>(a, v, s);
	BHIP_CASE4(PART_LOWER, POST_SUB, D_NONE)
	BHIP_CASE4(PART_UPPER, POST_D_SUB, D_VALS_DIAG)
	BHIP_CASE4(PART_LOWER, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE4(PART_UPPER, POST_SUB_D, D_DBLOCKS)
	BHIP_CASE4(PART_OFFDIAG, POST_D_SUB, D_DBLOCKS)
	BHIP_CASE4(PART_ALL, POST_AXPBY, D_NONE)
	BHIP_CASE4(PART_NONE, POST_D_SUB, D_DBLOCKS)
#undef BHIP_CASE4
	if (ok)
		BHIP_CHECK(hipGetLastError());
	return ok;
}

}  // namespace bhip
