// factory.cpp -- SRFactory: strings -> BlastedSolverType, (bs, layout, type) -> operator object.
// Behaviour restated from src/solverfactory.cpp:35-230 (same exceptions and messages).
#include "blasted/factory.hpp"

#include <stdexcept>
#include <typeinfo>

namespace blasted {

template <typename scalar, typename index>
BlastedSolverType SRFactory<scalar, index>::solverTypeFromString(const std::string precstr) const
{
	static const struct { const std::string *name; BlastedSolverType type; } table[] = {
	    {&jacobistr, BLASTED_JACOBI},       {&gsstr, BLASTED_GS},
	    {&sgsstr, BLASTED_SGS},             {&ilu0str, BLASTED_ILU0},
	    {&seqilu0str, BLASTED_SEQILU0},     {&sfilu0str, BLASTED_SFILU0},
	    {&sapilu0str, BLASTED_SAPILU0},     {&cscbgsstr, BLASTED_CSC_BGS},
	    {&levelsgsstr, BLASTED_LEVEL_SGS},  {&asynclevelilustr, BLASTED_ASYNC_LEVEL_ILU0},
	    {&noprecstr, BLASTED_NO_PREC}};
	for (const auto &t : table)
		if (precstr == *t.name)
			return t.type;
	throw std::invalid_argument("BLASTed: Preconditioner type not available!");
}

namespace {

// which halves of an ILU(0) run asynchronously for a given type: {factor, apply}
bool ilu_threading(const BlastedSolverType t, bool &tf, bool &ta)
{
	switch (t) {
	case BLASTED_ILU0: tf = true; ta = true; return true;
	case BLASTED_SEQILU0: tf = false; ta = false; return true;
	case BLASTED_SFILU0: tf = false; ta = true; return true;
	case BLASTED_SAPILU0: tf = true; ta = false; return true;
	default: return false;
	}
}

[[noreturn]] void not_on_this_backend(const char *what)
{
	throw std::invalid_argument(std::string("BLASTed(HIP): preconditioner type '") + what +
	                            "' is outside the MI355X backend's scope (jacobi, gs, sgs, ilu0, "
	                            "seqilu0, sfilu0, sapilu0, level_sgs, async_level_ilu0, none are available)");
}

}  // namespace

template <typename scalar, typename index>
template <int bs, StorageOptions stor>
SRPreconditioner<scalar, index> *SRFactory<scalar, index>::create_srpreconditioner_of_type(
    SRMatrixStorage<const scalar, const index> &&mat, const AsyncSolverSettings &opts) const
{
	bool tf = true, ta = true;
	if (ilu_threading(opts.prectype, tf, ta))
		return new AsyncBlockILU0_SRPreconditioner<scalar, index, bs, stor>(
		    std::move(mat), opts.nbuildsweeps, opts.napplysweeps, opts.scale, opts.thread_chunk_size,
		    opts.fact_inittype, opts.apply_inittype, tf, ta, opts.compute_precinfo);
	switch (opts.prectype) {
	case BLASTED_JACOBI: return new BJacobiSRPreconditioner<scalar, index, bs, stor>(std::move(mat));
	case BLASTED_SGS:
		return new AsyncBlockSGS_SRPreconditioner<scalar, index, bs, stor>(
		    std::move(mat), opts.napplysweeps, opts.apply_inittype, opts.thread_chunk_size);
	case BLASTED_NO_PREC: return new NoPreconditioner<scalar, index>(std::move(mat), bs);
	case BLASTED_GS:
		return new ChaoticBlockRelaxation<scalar, index, bs, stor>(std::move(mat), opts.napplysweeps,
		                                                           opts.thread_chunk_size);
	case BLASTED_LEVEL_SGS: return new Level_BSGS<scalar, index, bs, stor>(std::move(mat));
	case BLASTED_ASYNC_LEVEL_ILU0:
		return new Async_Level_BlockILU0<scalar, index, bs, stor>(std::move(mat), opts.nbuildsweeps, opts.scale,
		                                                          opts.thread_chunk_size, opts.fact_inittype,
		                                                          true, opts.compute_precinfo);
	default: throw std::invalid_argument("Invalid preconditioner!");
	}
}

template <typename scalar, typename index>
SRPreconditioner<scalar, index> *
SRFactory<scalar, index>::create_preconditioner(SRMatrixStorage<const scalar, const index> &&mat,
                                                const SolverSettings &set) const
{
	// plain SolverSettings -> std::bad_cast, as in the reference (src/solverfactory.cpp:136)
	const AsyncSolverSettings &opts = dynamic_cast<const AsyncSolverSettings &>(set);

	if (opts.bs == 1) {
		bool tf = true, ta = true;
		if (ilu_threading(opts.prectype, tf, ta))
			return new AsyncILU0_SRPreconditioner<scalar, index>(
			    std::move(mat), opts.nbuildsweeps, opts.napplysweeps, opts.scale, opts.thread_chunk_size,
			    opts.fact_inittype, opts.apply_inittype, opts.compute_precinfo, tf, ta);
		switch (opts.prectype) {
		case BLASTED_JACOBI: return new JacobiSRPreconditioner<scalar, index>(std::move(mat));
		case BLASTED_SGS:
			return new AsyncSGS_SRPreconditioner<scalar, index>(std::move(mat), opts.napplysweeps,
			                                                    opts.apply_inittype, opts.thread_chunk_size);
		case BLASTED_NO_PREC: return new NoPreconditioner<scalar, index>(std::move(mat), 1);
		case BLASTED_GS:
			return new ChaoticRelaxation<scalar, index>(std::move(mat), opts.napplysweeps,
			                                            opts.thread_chunk_size);
		case BLASTED_CSC_BGS: not_on_this_backend("cscbgs");
		case BLASTED_LEVEL_SGS: return new Level_SGS<scalar, index>(std::move(mat));
		case BLASTED_ASYNC_LEVEL_ILU0:
			return new Async_Level_ILU0<scalar, index>(std::move(mat), opts.nbuildsweeps, opts.scale,
			                                           opts.thread_chunk_size, opts.fact_inittype, true,
			                                           opts.compute_precinfo);
		default: throw std::invalid_argument("Invalid preconditioner!");
		}
	}

	if (opts.blockstorage != RowMajor && opts.blockstorage != ColMajor)
		throw std::invalid_argument("Block ordering must be either rowmajor or colmajor!");
	const bool rm = opts.blockstorage == RowMajor;
#define BLASTED_BS_CASE(N)                                                                    \
	case N:                                                                                   \
		return rm ? create_srpreconditioner_of_type<N, RowMajor>(std::move(mat), opts)        \
		          : create_srpreconditioner_of_type<N, ColMajor>(std::move(mat), opts);
	switch (opts.bs) {
		BLASTED_BS_CASE(2)
		BLASTED_BS_CASE(3)
		BLASTED_BS_CASE(4)
		BLASTED_BS_CASE(5)
		BLASTED_BS_CASE(7)
		BLASTED_BS_CASE(8)
	default:
		throw std::invalid_argument("Block size " + std::to_string(opts.bs) + " not supported for " +
		                            (rm ? "row major!" : "column major!"));
	}
#undef BLASTED_BS_CASE
}

template class SRFactory<double, int>;

}  // namespace blasted
