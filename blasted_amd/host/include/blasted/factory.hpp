// factory.hpp -- settings structs and the operator factory of the BLASTed API
// (include/solverfactory.hpp:21-112, src/solverfactory.cpp:35-230).
//
// Type strings and dispatch rules are the reference's: `ilu0` -> async factor + async apply,
// `seqilu0` -> sequential both, `sfilu0` -> sequential factor, `sapilu0` -> sequential apply; bs == 1
// always selects the scalar operators; `level_sgs` and `async_level_ilu0` are the exact (level-scheduled)
// operators, `gs` the chaotic forward relaxation, `jacobi`, `none`.  The one type outside the
// preconditioner-apply hot path, `cscbgs` (CSC storage), is recognised by solverTypeFromString but
// create_preconditioner rejects it with std::invalid_argument on this backend.
#pragma once

#include <string>

#include "operators.hpp"
#include "solvertypes.h"
#include "types.hpp"

namespace blasted {

const std::string noprecstr = "none";
const std::string jacobistr = "jacobi";
const std::string gsstr = "gs";
const std::string sgsstr = "sgs";
const std::string ilu0str = "ilu0";
const std::string seqilu0str = "seqilu0";
const std::string sfilu0str = "sfilu0";
const std::string sapilu0str = "sapilu0";
const std::string cscbgsstr = "cscbgs";
const std::string levelsgsstr = "level_sgs";
const std::string asynclevelilustr = "async_level_ilu0";

/// What every iteration needs to know
struct SolverSettings {
	BlastedSolverType prectype;
	int bs;                        ///< block size of the matrix
	StorageOptions blockstorage;   ///< RowMajor or ColMajor inside a block
	bool relax;                    ///< relaxation wanted instead of preconditioning
	int thread_chunk_size;         ///< OpenMP chunk of the reference; accepted, unused on the GPU
	virtual ~SolverSettings() = default;
};

/// Extra settings of the asynchronous iterations
struct AsyncSolverSettings : public SolverSettings {
	bool scale;                    ///< symmetric scaling before factorisation
	int nbuildsweeps;
	int napplysweeps;
	FactInit fact_inittype;
	ApplyInit apply_inittype;
	bool compute_precinfo;
};

template <typename scalar, typename index>
class FactoryBase {
public:
	FactoryBase() {}
	virtual ~FactoryBase() {}
	/// Returns an owning raw pointer (the caller deletes it)
	virtual SRPreconditioner<scalar, index> *
	create_preconditioner(SRMatrixStorage<const scalar, const index> &&prec_matrix,
	                      const SolverSettings &settings) const = 0;
	/// Throws std::invalid_argument for an unknown string
	virtual BlastedSolverType solverTypeFromString(const std::string precstr) const = 0;
};

template <typename scalar, typename index>
class SRFactory : public FactoryBase<scalar, index> {
public:
	SRPreconditioner<scalar, index> *create_preconditioner(SRMatrixStorage<const scalar, const index> &&prec_matrix,
	                                                       const SolverSettings &settings) const;
	BlastedSolverType solverTypeFromString(const std::string precstr) const;

private:
	template <int bs, StorageOptions stor>
	SRPreconditioner<scalar, index> *
	create_srpreconditioner_of_type(SRMatrixStorage<const scalar, const index> &&prec_matrix,
	                                const AsyncSolverSettings &opts) const;
};

}  // namespace blasted
