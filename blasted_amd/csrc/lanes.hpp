// lanes.hpp -- wave64 cross-lane helpers shared by the sweep kernels (device code only).
#pragma once

#include <hip/hip_runtime.h>

namespace bhip {

// Which chunk of rows a workgroup takes.  Workgroups are dealt round-robin to the 8 XCDs; the XCDs take
// turns on super-chunks of XCD_SUPER consecutive chunks: workgroup (xcd, local) gets chunk
// (local / S) * 8S + xcd * S + local % S.  Inside a super-chunk (2048 rows at 128 rows per chunk) an XCD's
// L2 serves the near-neighbour gathers and later workgroups of the XCD observe earlier ones' updates,
// while all XCDs stay inside one moving window of the matrix, like a single in-order sweep.  The first
// version gave each XCD one contiguous eighth of the rows: same preconditioner quality (distance to the
// exact solve after 3+3 sweeps 0.158 against 0.162), but eight streams an eighth of the buffers apart,
// whose speed depended on where the driver had put the buffers -- in about half of the builds of the
// 256^3 problem the upper sweep ran 7 % and the lower sweep 2-9 % slower than with this mapping, never
// faster (profiles/r01l_xcd_mapping.txt).  Chunks beyond the last full group of 8S keep their own index.
constexpr unsigned XCD_SUPER = 16;
__device__ __forceinline__ unsigned xcd_chunk(unsigned bid, unsigned nwg)
{
	constexpr unsigned S = XCD_SUPER;
	const unsigned full = nwg - nwg % (8u * S);
	if (bid >= full)
		return bid;
	const unsigned xcd = bid & 7u, local = bid >> 3;
	return (local / S) * (8u * S) + xcd * S + local % S;
}

// the same mapping with a run-time super-chunk size 2^shift (tuning "xcdsuper=N"; shift 4 = the default above)
__device__ __forceinline__ unsigned xcd_chunk(unsigned bid, unsigned nwg, unsigned shift)
{
	const unsigned S = 1u << shift;
	const unsigned full = nwg - nwg % (8u * S);
	if (bid >= full)
		return bid;
	const unsigned xcd = bid & 7u, local = bid >> 3;
	return ((local >> shift) << (shift + 3)) + (xcd << shift) + (local & (S - 1u));
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(const double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

// v[lane] + v[lane ^ 16] in every lane (v_permlane16_swap)
__device__ __forceinline__ double xor16_sum(const double v)
{
	typedef unsigned v2u __attribute__((ext_vector_type(2)));
	const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
	const v2u a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
	const v2u b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
	return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}

// v[lane] + v[lane ^ 32] in every lane (v_permlane32_swap)
__device__ __forceinline__ double xor32_sum(const double v)
{
	typedef unsigned v2u __attribute__((ext_vector_type(2)));
	const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
	const v2u a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
	const v2u b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
	return __hiloint2double((int)b.x, (int)a.x) + __hiloint2double((int)b.y, (int)a.y);
}

// v[lane ^ 4] inside every aligned group of 8 lanes: two row shifts and a select
__device__ __forceinline__ double xor4_value(const double v, const int lane)
{
	const double up = dpp_mov<0x104>(v);  // row_shl:4  lane i <- lane i+4
	const double dn = dpp_mov<0x114>(v);  // row_shr:4  lane i <- lane i-4
	return (lane & 4) ? dn : up;
}

// All-reduce (sum) over lane bits [LOBIT, HIBIT), every other bit kept; all on the VALU.
//  * bits 0 and 1: exact xor exchanges inside a quad (quad_perm);
//  * bits 2 and 3 (and bit 1 when the range reaches bit 3): row rotations.  Inside a 16-lane DPP row,
//    row_ror 8, 4, 2 applied in this order sum bits 3, 2, 1: after each step the value is periodic in the
//    bit just summed, so the wrap-around of the next rotation lands on an equal value.  This needs the
//    range to include bit 3 whenever it includes bit 2 (static_assert below);
//  * bits 4 and 5: v_permlane16_swap / v_permlane32_swap;
//  * the range [1, 3) (8-lane groups) is done with exact xor exchanges: quad_perm and xor4_value.
template <int LOBIT, int HIBIT>
__device__ __forceinline__ double allreduce_bits(double v)
{
	static_assert(HIBIT <= 6 && LOBIT >= 0 && LOBIT <= HIBIT, "lane bits");
	if (LOBIT == 1 && HIBIT == 3) {
		// bits 1 and 2 only (8-lane groups): exact xor exchanges, no rotation (bit 3 is not summed)
		v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
		v += xor4_value(v, (int)(threadIdx.x & 63));
		return v;
	}
	static_assert(!(LOBIT <= 2 && HIBIT > 2) || HIBIT > 3 || (LOBIT == 1 && HIBIT == 3),
	              "a range containing bit 2 must contain bit 3 (except the 8-lane case handled above)");
	constexpr bool rot = HIBIT > 3;  // rotations usable for bits 1..3
	if (LOBIT <= 3 && HIBIT > 3)
		v += dpp_mov<0x128>(v);  // row_ror:8
	if (LOBIT <= 2 && HIBIT > 2)
		v += dpp_mov<0x124>(v);  // row_ror:4
	if (LOBIT <= 1 && HIBIT > 1)
		v += rot ? dpp_mov<0x122>(v) : dpp_mov<0x4E>(v);  // row_ror:2 | quad_perm [2,3,0,1]
	if (LOBIT <= 0 && HIBIT > 0)
		v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
	if (LOBIT <= 4 && HIBIT > 4)
		v = xor16_sum(v);
	if (LOBIT <= 5 && HIBIT > 5)
		v = xor32_sum(v);
	return v;
}

}  // namespace bhip
