// sweep_geo.hpp -- lane geometry of the general row-sweep kernels (kernels_sweep.hip, kernels_level.hip).
#pragma once

namespace bhip {

template <int BS>
struct Geo {
	static constexpr int BSP = BS <= 1 ? 1 : (BS <= 2 ? 2 : (BS <= 4 ? 4 : 8));
	static constexpr int SUB = BSP * BSP;                         // lanes per block
	static constexpr int G = BS == 1 ? 4 : (BS <= 4 ? 16 : 64);  // lanes per block-row
	static constexpr int NB = G / SUB;                            // block slots per row
	static constexpr int RPW = 64 / G;                            // rows per wave and step
	static constexpr int RSTEP = 4 * RPW;                         // rows per workgroup and step
	static constexpr int RCHUNK = BS <= 4 ? 128 : 64;             // rows per workgroup
	static constexpr int CAP = (BS == 1 ? 32 : 16) * RCHUNK;      // staged column indices
	static constexpr int LOBIT = BSP == 1 ? 0 : (BSP == 2 ? 1 : (BSP == 4 ? 2 : 3));
	static constexpr int HIBIT = G == 4 ? 2 : (G == 16 ? 4 : 6);
};

}  // namespace bhip
