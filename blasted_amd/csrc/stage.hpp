// stage.hpp -- the prologue shared by the row-sweep kernels: a workgroup's pointer rows and the column
// indices of its chunk go from HBM to LDS once, coalesced.
#pragma once
#include "ctx.hpp"

namespace bhip {

// Rows [r0, r0 + rc) of the chunk: s_rp[0..rc] = browptr, s_dg[0..rc) = diagind, s_col[0..min(CAP, jhi-jlo))
// = bcolind[jlo..], where [jlo, jhi) is the index range the PART of these rows spans.  Ends with the
// barrier that publishes them.
//  * jlo / jhi come from wave-uniform (scalar) loads, so the pointer rows and the column indices are
//    fetched side by side behind ONE barrier instead of two dependent vector-load phases;
//  * the loads are predicated and straight-line, all in flight together: written as `for (k = tid; k < n;
//    k += 256)` the compiler's remainder loop makes one memory round trip per 256 indices.
// (256^3 bs=4: lower sweep -4 %.)
template <int PART, int RCHUNK, int CAP>
__device__ __forceinline__ void stage_chunk_indices(const Pattern &pat, const int r0, const int rc, const int tid,
                                                    int *const s_rp, int *const s_dg, int *const s_col,
                                                    int &jlo, int &jhi)
{
	if (PART == PART_LOWER) {
		jlo = pat.browptr[r0];
		jhi = pat.diagind[r0 + rc - 1];
	} else if (PART == PART_UPPER) {
		jlo = pat.diagind[r0];
		jhi = pat.browptr[r0 + rc];
	} else {
		jlo = pat.browptr[r0];
		jhi = pat.browptr[r0 + rc];
	}
	jlo = __builtin_amdgcn_readfirstlane(jlo);  // wave-uniform by construction: let the compiler know
	jhi = __builtin_amdgcn_readfirstlane(jhi);
	constexpr int NRP = (RCHUNK + 256) / 256, NCOL = (CAP + 255) / 256;
	int vrp[NRP], vdg[NRP], vcol[NCOL];
#pragma unroll
	for (int i = 0; i < NRP; i++) {
		const int k = tid + 256 * i;
		vrp[i] = (k <= rc) ? pat.browptr[r0 + k] : 0;
		vdg[i] = (k < rc) ? pat.diagind[r0 + k] : 0;
	}
	const int ncol = (PART == PART_NONE) ? 0 : ((jhi - jlo) < CAP ? (jhi - jlo) : CAP);
#pragma unroll
	for (int i = 0; i < NCOL; i++) {
		const int k = tid + 256 * i;
		vcol[i] = (k < ncol) ? pat.bcolind[jlo + k] : 0;
	}
#pragma unroll
	for (int i = 0; i < NRP; i++) {
		const int k = tid + 256 * i;
		if (k <= rc)
			s_rp[k] = vrp[i];
		if (k < rc)
			s_dg[k] = vdg[i];
	}
#pragma unroll
	for (int i = 0; i < NCOL; i++) {
		const int k = tid + 256 * i;
		if (k < ncol)
			s_col[k] = vcol[i];
	}
	__syncthreads();
}

// The factorisation kernels' prologue: rows [r0, r0 + rc) of the chunk -> s_rp[0..rc] = browptr,
// s_col / s_pp = bcolind / posptr of its first nblk = min(CAPB, blocks of the chunk) blocks (s_pp one more),
// s_lp / s_up = the first npair = min(CAPP, pairs of those blocks) position pairs.  The three ranges
// depend on each other (rows -> blocks -> pairs); their ends are chased with wave-uniform scalar loads, so
// that all vector loads go out together behind one barrier instead of three dependent phases.
template <int RCHUNK, int CAPB, int CAPP>
__device__ __forceinline__ void stage_factor_indices(const Pattern &pat, const int *__restrict__ posptr,
                                                     const int *__restrict__ lowerp, const int *__restrict__ upperp,
                                                     const int r0, const int rc, const int tid, int *const s_rp,
                                                     int *const s_col, int *const s_pp, int *const s_lp,
                                                     int *const s_up, int &jlo, int &plo)
{
	static_assert(RCHUNK < 256, "one pointer row per thread");
	jlo = __builtin_amdgcn_readfirstlane(pat.browptr[r0]);
	const int jhi = __builtin_amdgcn_readfirstlane(pat.browptr[r0 + rc]);
	const int nblk = (jhi - jlo) < CAPB ? (jhi - jlo) : CAPB;
	plo = __builtin_amdgcn_readfirstlane(posptr[jlo]);
	const int phi = __builtin_amdgcn_readfirstlane(posptr[jlo + nblk]);
	const int npair = (phi - plo) < CAPP ? (phi - plo) : CAPP;
	constexpr int NB_ = (CAPB + 256) / 256, NP_ = (CAPP + 255) / 256;
	int vcol[NB_], vpp[NB_], vlp[NP_], vup[NP_];
	const int vrp = (tid <= rc) ? pat.browptr[r0 + tid] : 0;
#pragma unroll
	for (int i = 0; i < NB_; i++) {
		const int q = tid + 256 * i;
		vcol[i] = (q < nblk) ? pat.bcolind[jlo + q] : 0;
		vpp[i] = (q <= nblk) ? posptr[jlo + q] : 0;
	}
#pragma unroll
	for (int i = 0; i < NP_; i++) {
		const int q = tid + 256 * i;
		vlp[i] = (q < npair) ? lowerp[plo + q] : 0;
		vup[i] = (q < npair) ? upperp[plo + q] : 0;
	}
	if (tid <= rc)
		s_rp[tid] = vrp;
#pragma unroll
	for (int i = 0; i < NB_; i++) {
		const int q = tid + 256 * i;
		if (q < nblk)
			s_col[q] = vcol[i];
		if (q <= nblk)
			s_pp[q] = vpp[i];
	}
#pragma unroll
	for (int i = 0; i < NP_; i++) {
		const int q = tid + 256 * i;
		if (q < npair) {
			s_lp[q] = vlp[i];
			s_up[q] = vup[i];
		}
	}
	__syncthreads();
}

}  // namespace bhip
