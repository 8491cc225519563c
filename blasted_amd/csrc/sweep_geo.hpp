// sweep_geo.hpp -- lane geometry of the general row-sweep kernels (kernels_sweep.hip, kernels_level.hip).
#pragma once

namespace bhip {

// BIG (scalar only): 256 instead of 128 rows per workgroup -- a scalar workgroup of 128 rows moves under 8 KB, and
// at 16.8 M rows the 131 072 prologues (staging, barrier) weigh: lower / upper sweep 3.7 / 3.8 -> 4.3 / 4.35 TB/s.
// Small problems keep 128 (64^3: 7 % faster with twice the workgroups).
template <int BS, bool BIG = false>
struct Geo {
	static constexpr int BSP = BS <= 1 ? 1 : (BS <= 2 ? 2 : (BS <= 4 ? 4 : 8));
	static constexpr int SUB = BSP * BSP;                         // lanes per block
	static constexpr int G = BS == 1 ? 4 : (BS <= 4 ? 16 : 64);  // lanes per block-row
	static constexpr int NB = G / SUB;                            // block slots per row
	static constexpr int RPW = 64 / G;                            // rows per wave and step
	static constexpr int RSTEP = 4 * RPW;                         // rows per workgroup and step
	static constexpr int RCHUNK = (BS == 1 && BIG) ? 256 : (BS <= 4 ? 128 : 64);  // rows per workgroup
	static constexpr int CAP = BS == 1 ? 4096 : 16 * RCHUNK;      // staged column indices
	static constexpr int LOBIT = BSP == 1 ? 0 : (BSP == 2 ? 1 : (BSP == 4 ? 2 : 3));
	static constexpr int HIBIT = G == 4 ? 2 : (G == 16 ? 4 : 6);
};

}  // namespace bhip
