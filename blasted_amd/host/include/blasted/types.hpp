// types.hpp -- small value types of the BLASTed operator API, restated without Eigen.
//   StorageOptions          include/blasted_config.hpp:17-19 (Eigen::ColMajor = 0, Eigen::RowMajor = 1)
//   StorageType             include/linearoperator.hpp:13
//   SolveParams             include/solverops_base.hpp:21-27
//   PrecInfo, PrecInfoList  include/preconditioner_diagnostics.hpp:14-58
//   FactInit, ApplyInit     include/async_initialization_decl.hpp:15-62
#pragma once

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

namespace blasted {

enum StorageOptions { ColMajor = 0, RowMajor = 1 };

enum StorageType { SPARSEROW, CSR, BSR, VIEWCSR, VIEWBSR, COO, MATRIXFREE, OTHERSTORAGE };

template <typename scalar>
struct SolveParams {
	scalar rtol, atol, dtol;  // relative / absolute / divergence tolerance
	bool ctol;                // check tolerances at all?
	int maxits;
};

/// Six diagnostics a preconditioner may report; slot order is part of the API (f_info is public).
struct PrecInfo {
	std::array<double, 6> f_info{{0, 0, 0, 0, 0, 0}};

	enum Slot { REMAINDER = 0, INITIAL_REMAINDER, UPPER_MIN_DD, UPPER_AVG_DD, LOWER_MIN_DD, LOWER_AVG_DD };

	double &prec_remainder_norm() { return f_info[REMAINDER]; }
	double &prec_rem_initial_norm() { return f_info[INITIAL_REMAINDER]; }
	double &upper_min_diag_dom() { return f_info[UPPER_MIN_DD]; }
	double &upper_avg_diag_dom() { return f_info[UPPER_AVG_DD]; }
	double &lower_min_diag_dom() { return f_info[LOWER_MIN_DD]; }
	double &lower_avg_diag_dom() { return f_info[LOWER_AVG_DD]; }
	const double &prec_remainder_norm() const { return f_info[REMAINDER]; }
	const double &prec_rem_initial_norm() const { return f_info[INITIAL_REMAINDER]; }
	const double &upper_min_diag_dom() const { return f_info[UPPER_MIN_DD]; }
	const double &upper_avg_diag_dom() const { return f_info[UPPER_AVG_DD]; }
	const double &lower_min_diag_dom() const { return f_info[LOWER_MIN_DD]; }
	const double &lower_avg_diag_dom() const { return f_info[LOWER_AVG_DD]; }
};

struct PrecInfoList {
	std::vector<PrecInfo> infolist;
	static const std::array<std::string, 6> descr;
	static const int field_width;
};

enum FactInit { INIT_F_ZERO, INIT_F_ORIGINAL, INIT_F_SGS, INIT_F_NONE };
enum ApplyInit { INIT_A_ZERO, INIT_A_JACOBI, INIT_A_NONE };

inline FactInit getFactInitFromString(const std::string itype)
{
	static const struct { const char *name; FactInit v; } tab[] = {
	    {"init_zero", INIT_F_ZERO}, {"init_original", INIT_F_ORIGINAL},
	    {"init_sgs", INIT_F_SGS},   {"init_none", INIT_F_NONE}};
	for (const auto &t : tab)
		if (itype == t.name)
			return t.v;
	throw std::invalid_argument("Factor initialization not recongnized!");
}

inline ApplyInit getApplyInitFromString(const std::string itype)
{
	static const struct { const char *name; ApplyInit v; } tab[] = {
	    {"init_zero", INIT_A_ZERO}, {"init_jacobi", INIT_A_JACOBI}, {"init_none", INIT_A_NONE}};
	for (const auto &t : tab)
		if (itype == t.name)
			return t.v;
	throw std::invalid_argument("Apply initialization not recongnized!");
}

}  // namespace blasted
