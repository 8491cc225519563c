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

}  // namespace bhip
