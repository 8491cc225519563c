// operators.hpp -- the operator classes of the BLASTed API on the MI355X backend.
//
// Same class names, constructor argument lists, virtual interface and error behaviour as the reference:
//   AbstractLinearOperator, MatrixView          include/linearoperator.hpp:17-131
//   Preconditioner, SRPreconditioner, NoPreconditioner   include/solverops_base.hpp:31-106
//   JacobiSRPreconditioner, BJacobiSRPreconditioner      include/solverops_jacobi.hpp
//   AsyncSGS_SRPreconditioner, AsyncBlockSGS_SRPreconditioner   include/solverops_sgs.hpp:23-121
//   ChaoticRelaxation, ChaoticBlockRelaxation                   include/relaxation_chaotic.hpp:20-95
//   Level_BSGS, Level_SGS                                       include/solverops_levels_sgs.hpp:20-80
//   Async_Level_BlockILU0, Async_Level_ILU0                     include/solverops_levels_ilu0.hpp:20-85
//   AsyncILU0_SRPreconditioner, AsyncBlockILU0_SRPreconditioner include/solverops_ilu0.hpp:23-169
//   SRMatrixView, BSRMatrixView, CSRMatrixView (apply / gemv3 only)  include/blockmatrices.hpp:27-160
// What is different is where the work happens: every compute()/apply()/apply_relax()/gemv3() forwards to
// the C ABI of include/blasted_hip.h; there is no host arithmetic in these classes.  Vectors passed to
// the reference signatures are host pointers (as PETSc's VecGetArray gives them); the *_device variants
// take HBM pointers and never cross PCIe.
//
// Knobs without a GPU meaning keep their place in the signatures: thread_chunk_size is accepted and
// ignored; threadedfactor / threadedapply = false (the seq* / sf* / sap* factory types) select the exact
// sequential result, obtained on the device by level-scheduled exact passes.
#pragma once

#include <memory>
#include <string>

#include "storage.hpp"
#include "types.hpp"

struct blasted_hip_prec_s;

namespace blasted {

namespace detail {

/// RAII owner of one blasted_hip_prec; translates C-ABI error codes into the reference's exceptions
class HipOperator {
public:
	HipOperator();
	~HipOperator();
	HipOperator(const HipOperator &) = delete;
	HipOperator &operator=(const HipOperator &) = delete;

	/// Uploads the pattern on first use, (re-)sets the values every time
	void bind(const CRawBSRMatrix<double, int> &mat, int bs, StorageOptions stor);
	blasted_hip_prec_s *get() const { return h; }
	static void check(int rc);
	/// how ilu0 / sgs apply their sweeps: BLASTED_HIP_ASYNC (the reference's chaotic sweeps) unless set_sweep_mode()
	/// or BLASTED_HIP_SWEEP_MODE says otherwise
	static int sweep_mode();
	static int default_device();

private:
	blasted_hip_prec_s *h;
	bool pattern_set;
};

/// "async" | "deterministic" | "exact" -> the C ABI's mode; throws std::invalid_argument otherwise
int sweep_mode_from_string(const char *m);
/// process-wide choice of how the asynchronous types apply their sweeps (overrides BLASTED_HIP_SWEEP_MODE; NULL or
/// "" removes the override).  "async": the reference's chaotic sweeps (default; flexible outer solvers);
/// "deterministic": synchronous sweeps, a fixed operator (any Krylov method); "exact": level-scheduled solves.
void set_sweep_mode(const char *m);
const char *sweep_mode_name();
/// true when the process-wide mode was chosen by somebody (set_sweep_mode or one of the environment variables)
/// rather than being the built-in default
bool sweep_mode_is_explicit();

}  // namespace detail

template <typename scalar, typename index>
class AbstractLinearOperator {
public:
	explicit AbstractLinearOperator(const StorageType storagetype) : _type{storagetype} {}
	virtual ~AbstractLinearOperator() {}
	StorageType type() { return _type; }
	virtual index dim() const = 0;
	virtual void apply(const scalar *const x, scalar *const __restrict y) const = 0;

protected:
	StorageType _type;
};

template <typename scalar, typename index>
class MatrixView : public AbstractLinearOperator<scalar, index> {
public:
	explicit MatrixView(const StorageType storagetype) : AbstractLinearOperator<scalar, index>(storagetype) {}
	virtual ~MatrixView() {}
	virtual void apply(const scalar *const x, scalar *const __restrict y) const = 0;
	/// z := a A x + b y
	virtual void gemv3(const scalar a, const scalar *const __restrict x, const scalar b,
	                   const scalar *const y, scalar *const z) const = 0;
};

/// View of a sparse-row matrix (not owning unless the storage does)
template <typename scalar, typename index>
class SRMatrixView : public MatrixView<scalar, index> {
public:
	SRMatrixView(SRMatrixStorage<const scalar, const index> &&matrix, const StorageType storagetype,
	             const int block_size, const StorageOptions layout);
	virtual ~SRMatrixView();
	const SRMatrixStorage<const scalar, const index> &getSRStorage() const { return mat; }
	index dim() const { return mat.nbrows * bs_; }
	void apply(const scalar *const x, scalar *const __restrict y) const;
	void gemv3(const scalar a, const scalar *const __restrict x, const scalar b, const scalar *const y,
	           scalar *const z) const;
	/// HBM-resident vectors
	void apply_device(const scalar *const dx, scalar *const dy) const;

protected:
	SRMatrixStorage<const scalar, const index> mat;
	int bs_;
	std::unique_ptr<detail::HipOperator> op;
};

template <typename scalar, typename index, int bs, StorageOptions stor>
class BSRMatrixView : public SRMatrixView<scalar, index> {
	static_assert(bs > 0, "Block size must be positive!");

public:
	explicit BSRMatrixView(SRMatrixStorage<const scalar, const index> &&matrix)
	    : SRMatrixView<scalar, index>(std::move(matrix), VIEWBSR, bs, stor)
	{
	}
};

template <typename scalar, typename index>
class CSRMatrixView : public SRMatrixView<scalar, index> {
public:
	explicit CSRMatrixView(SRMatrixStorage<const scalar, const index> &&matrix)
	    : SRMatrixView<scalar, index>(std::move(matrix), VIEWCSR, 1, ColMajor)
	{
	}
};

template <typename scalar, typename index>
class Preconditioner : public AbstractLinearOperator<scalar, index> {
public:
	explicit Preconditioner(const StorageType storagetype) : AbstractLinearOperator<scalar, index>(storagetype) {}
	virtual ~Preconditioner() {}
	virtual index dim() const = 0;
	virtual PrecInfo compute() = 0;
	virtual void apply(const scalar *const x, scalar *const __restrict y) const = 0;
	virtual void apply_relax(const scalar *const x, scalar *const __restrict y) const = 0;
	virtual bool relaxationAvailable() const = 0;
	void setApplyParams(const SolveParams<scalar> sparams) { solveparams = sparams; }

protected:
	SolveParams<scalar> solveparams{};
};

/// Preconditioner built from a sparse-row matrix; holds the storage (pmat) and its raw view (mat)
template <typename scalar, typename index>
class SRPreconditioner : public Preconditioner<scalar, index> {
public:
	explicit SRPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix);
	virtual ~SRPreconditioner();
	/// apply / apply_relax on HBM-resident vectors: the same operation as the host-vector member of the same
	/// class, enqueued on the operator's stream (no host copy, no synchronisation).  Not in the reference.
	void apply_device(const scalar *const dx, scalar *const dy) const { apply_at(dx, dy, 1); }
	void apply_relax_device(const scalar *const db, scalar *const dx) const { relax_at(db, dx, 1); }
	/// false for operators that have nothing to run on the device (NoPreconditioner)
	virtual bool deviceVectorsAvailable() const { return true; }
	/// How THIS operator applies its asynchronous sweeps (ilu0, sgs): BLASTED_HIP_ASYNC, BLASTED_HIP_DETERMINISTIC or
	/// BLASTED_HIP_LEVEL; a negative value (the default) follows the process-wide choice (detail::set_sweep_mode,
	/// BLASTED_HIP_SWEEP_MODE).  Not in the reference; the PCSHELL glue sets it per KSP tree.
	/// HBM this operator holds now (pattern / value mirrors, factor, derived copies, vectors; 0 before compute()), and
	/// the device it was created on (ranks of a node share its GPUs round-robin: HipOperator::default_device).  Not in
	/// the reference.
	long deviceBytes() const;
	int deviceIndex() const { return detail::HipOperator::default_device(); }
	void setSweepMode(const int mode) { sweepmode_ = mode; }
	int sweepMode() const { return sweepmode_ >= 0 ? sweepmode_ : detail::HipOperator::sweep_mode(); }

protected:
	/// the one implementation behind apply / apply_device and apply_relax / apply_relax_device:
	/// loc = 0 host vectors (BLASTED_HIP_HOST), 1 device vectors (BLASTED_HIP_DEVICE)
	virtual void apply_at(const scalar *const x, scalar *const y, const int loc) const;
	virtual void relax_at(const scalar *const b, scalar *const x, const int loc) const;

	SRMatrixStorage<const scalar, const index> pmat;
	CRawBSRMatrix<scalar, index> mat;
	std::unique_ptr<detail::HipOperator> op;
	int sweepmode_ = -1;
};

template <typename scalar, typename index>
class NoPreconditioner : public SRPreconditioner<scalar, index> {
public:
	NoPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix, const index bs);
	index dim() const { return ndim; }
	bool relaxationAvailable() const { return false; }
	bool deviceVectorsAvailable() const { return false; }
	PrecInfo compute() { return PrecInfo(); }
	void apply(const scalar *const x, scalar *const __restrict y) const;
	void apply_relax(const scalar *const x, scalar *const __restrict y) const;

protected:
	index ndim;
};

/// Block-Jacobi: dblocks_i = A_ii^-1 (in HBM)
template <typename scalar, typename index, int bs, StorageOptions stor>
class BJacobiSRPreconditioner : public SRPreconditioner<scalar, index> {
	static_assert(bs > 0, "Block size must be positive!");

public:
	explicit BJacobiSRPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix);
	virtual ~BJacobiSRPreconditioner();
	index dim() const { return mat.nbrows * bs; }
	bool relaxationAvailable() const { return true; }
	PrecInfo compute();
	void apply(const scalar *const r, scalar *const __restrict z) const;
	/// synchronous Jacobi relaxation with the reference's optional convergence test on the step
	/// difference (src/solverops_jacobi.cpp:66-119): solveparams.maxits, ctol, rtol, atol, dtol
	void apply_relax(const scalar *const b, scalar *const __restrict x) const;

protected:
	using SRPreconditioner<scalar, index>::mat;
	using SRPreconditioner<scalar, index>::op;
	using SRPreconditioner<scalar, index>::solveparams;
	void bind_and_invert();
	void apply_at(const scalar *const r, scalar *const z, const int loc) const;
	void relax_at(const scalar *const b, scalar *const x, const int loc) const;
};

template <typename scalar, typename index>
class JacobiSRPreconditioner : public BJacobiSRPreconditioner<scalar, index, 1, ColMajor> {
public:
	explicit JacobiSRPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix)
	    : BJacobiSRPreconditioner<scalar, index, 1, ColMajor>(std::move(matrix))
	{
	}
};

/// Asynchronous block symmetric Gauss-Seidel
template <typename scalar, typename index, int bs, StorageOptions stor>
class AsyncBlockSGS_SRPreconditioner : public BJacobiSRPreconditioner<scalar, index, bs, stor> {
public:
	AsyncBlockSGS_SRPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix,
	                               const int napplysweeps, const ApplyInit apply_inittype,
	                               const int threadchunksize);
	~AsyncBlockSGS_SRPreconditioner();
	bool relaxationAvailable() const { return true; }
	PrecInfo compute();
	void apply(const scalar *const r, scalar *const __restrict z) const;
	void apply_relax(const scalar *const b, scalar *const __restrict x) const;

protected:
	using SRPreconditioner<scalar, index>::mat;
	using SRPreconditioner<scalar, index>::op;
	using SRPreconditioner<scalar, index>::solveparams;
	void apply_at(const scalar *const r, scalar *const z, const int loc) const;
	void relax_at(const scalar *const b, scalar *const x, const int loc) const;
	const int napplysweeps;
	const ApplyInit ainit;
	const int thread_chunk_size;
};

template <typename scalar, typename index>
class AsyncSGS_SRPreconditioner : public AsyncBlockSGS_SRPreconditioner<scalar, index, 1, ColMajor> {
public:
	AsyncSGS_SRPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix, const int napplysweeps,
	                          const ApplyInit apply_inittype, const int threadchunksize)
	    : AsyncBlockSGS_SRPreconditioner<scalar, index, 1, ColMajor>(std::move(matrix), napplysweeps,
	                                                                   apply_inittype, threadchunksize)
	{
	}
};

/// Chaotic (asynchronous forward Gauss-Seidel) block relaxation, the `gs` type
/// (include/relaxation_chaotic.hpp:20-55).  apply() and apply_relax() both relax in place: the output
/// vector is also the initial guess, exactly as in the reference.
template <typename scalar, typename index, int bs, StorageOptions stor>
class ChaoticBlockRelaxation : public BJacobiSRPreconditioner<scalar, index, bs, stor> {
public:
	ChaoticBlockRelaxation(SRMatrixStorage<const scalar, const index> &&matrix, const int napplysweeps,
	                       const int thread_chunk_size);
	bool relaxationAvailable() const { return true; }
	void apply(const scalar *const b, scalar *const __restrict x) const;
	void apply_relax(const scalar *const b, scalar *const __restrict x) const;

protected:
	using SRPreconditioner<scalar, index>::op;
	using SRPreconditioner<scalar, index>::solveparams;
	void apply_at(const scalar *const b, scalar *const x, const int loc) const;
	void relax_at(const scalar *const b, scalar *const x, const int loc) const;
	const int napplysweeps;
	const int thread_chunk_size;
};

template <typename scalar, typename index>
class ChaoticRelaxation : public ChaoticBlockRelaxation<scalar, index, 1, ColMajor> {
public:
	ChaoticRelaxation(SRMatrixStorage<const scalar, const index> &&matrix, const int napplysweeps,
	                  const int thread_chunk_size)
	    : ChaoticBlockRelaxation<scalar, index, 1, ColMajor>(std::move(matrix), napplysweeps, thread_chunk_size)
	{
	}
};

/// Asynchronous block ILU(0): async fixed-point factorisation + async triangular sweeps
template <typename scalar, typename index, int bs, StorageOptions stor>
class AsyncBlockILU0_SRPreconditioner : public SRPreconditioner<scalar, index> {
	static_assert(bs > 0, "Block size must be positive!");
	static_assert(stor == RowMajor || stor == ColMajor, "Invalid storage option!");

public:
	AsyncBlockILU0_SRPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix,
	                                const int nbuildsweeps, const int napplysweeps, const bool use_scaling,
	                                const int thread_chunk_size, const FactInit fact_inittype,
	                                const ApplyInit apply_inittype, const bool threadedfactor = true,
	                                const bool threadedapply = true, const bool compute_remainder = false);
	~AsyncBlockILU0_SRPreconditioner();
	index dim() const { return mat.nbrows * bs; }
	bool relaxationAvailable() const { return false; }
	/// successive calls must keep the sparsity pattern (values may change)
	PrecInfo compute();
	void apply(const scalar *const x, scalar *const __restrict y) const;
	/// throws std::runtime_error, as the reference does
	void apply_relax(const scalar *const x, scalar *const __restrict y) const;

protected:
	using SRPreconditioner<scalar, index>::mat;
	using SRPreconditioner<scalar, index>::op;
	void apply_at(const scalar *const r, scalar *const z, const int loc) const;
	void relax_at(const scalar *const b, scalar *const x, const int loc) const;
	const bool usescaling;
	const bool threadedfactor;
	const bool threadedapply;
	const int nbuildsweeps;
	const int napplysweeps;
	const int thread_chunk_size;
	const FactInit factinittype;
	const ApplyInit applyinittype;
	const bool compute_remainder;
};

/// Scalar ILU(0).  Note the reference's argument order: compute_preconditioner_info comes BEFORE the
/// threaded flags here (include/solverops_ilu0.hpp:113-118).
template <typename scalar, typename index>
class AsyncILU0_SRPreconditioner : public AsyncBlockILU0_SRPreconditioner<scalar, index, 1, ColMajor> {
public:
	AsyncILU0_SRPreconditioner(SRMatrixStorage<const scalar, const index> &&matrix, const int nbuildsweeps,
	                           const int napplysweeps, const bool use_scaling, const int thread_chunk_size,
	                           const FactInit fact_inittype, const ApplyInit apply_inittype,
	                           const bool compute_preconditioner_info, const bool threadedfactor = true,
	                           const bool threadedapply = true)
	    : AsyncBlockILU0_SRPreconditioner<scalar, index, 1, ColMajor>(
	          std::move(matrix), nbuildsweeps, napplysweeps, use_scaling, thread_chunk_size, fact_inittype,
	          apply_inittype, threadedfactor, threadedapply, compute_preconditioner_info)
	{
	}
};

/// Level-scheduled (exact) block SGS, the `level_sgs` type (include/solverops_levels_sgs.hpp:20-50).
/// The reference needs a matrix whose independent rows are consecutive (computeLevels); the device
/// schedule is built from the dependency graph, so any ordering works and gives the serial result.
template <typename scalar, typename index, int bs, StorageOptions stor>
class Level_BSGS : public BJacobiSRPreconditioner<scalar, index, bs, stor> {
public:
	Level_BSGS(SRMatrixStorage<const scalar, const index> &&matrix);
	bool relaxationAvailable() const { return true; }
	/// the first call also builds the level schedule
	PrecInfo compute();
	void apply(const scalar *const r, scalar *const __restrict z) const;
	void apply_relax(const scalar *const b, scalar *const __restrict x) const;
	/// number of dependency levels (one kernel launch each per pass)
	int numLevels() const;

protected:
	using SRPreconditioner<scalar, index>::op;
	using SRPreconditioner<scalar, index>::solveparams;
	void apply_at(const scalar *const r, scalar *const z, const int loc) const;
	void relax_at(const scalar *const b, scalar *const x, const int loc) const;
};

template <typename scalar, typename index>
class Level_SGS : public Level_BSGS<scalar, index, 1, ColMajor> {
public:
	Level_SGS(SRMatrixStorage<const scalar, const index> &&matrix)
	    : Level_BSGS<scalar, index, 1, ColMajor>(std::move(matrix))
	{
	}
};

/// Asynchronous factorisation with level-scheduled (exact) triangular solves, the `async_level_ilu0`
/// type (include/solverops_levels_ilu0.hpp:20-50).  Constructor arguments as in the reference:
/// one apply "sweep", INIT_A_NONE, threaded apply.
template <typename scalar, typename index, int bs, StorageOptions stor>
class Async_Level_BlockILU0 : public AsyncBlockILU0_SRPreconditioner<scalar, index, bs, stor> {
public:
	Async_Level_BlockILU0(SRMatrixStorage<const scalar, const index> &&matrix, const int nbuildsweeps,
	                      const bool use_scaling, const int thread_chunk_size, const FactInit fact_inittype,
	                      const bool threadedfactor = true, const bool compute_remainder = false);
	PrecInfo compute();
	void apply(const scalar *const x, scalar *const __restrict y) const;
	void apply_relax(const scalar *const x, scalar *const __restrict y) const;
	int numLevels() const;

protected:
	using SRPreconditioner<scalar, index>::op;
	void apply_at(const scalar *const r, scalar *const z, const int loc) const;
};

template <typename scalar, typename index>
class Async_Level_ILU0 : public Async_Level_BlockILU0<scalar, index, 1, ColMajor> {
public:
	Async_Level_ILU0(SRMatrixStorage<const scalar, const index> &&matrix, const int nbuildsweeps,
	                 const bool use_scaling, const int thread_chunk_size, const FactInit fact_inittype,
	                 const bool threadedfactor = true, const bool compute_remainder = false)
	    : Async_Level_BlockILU0<scalar, index, 1, ColMajor>(std::move(matrix), nbuildsweeps, use_scaling,
	                                                        thread_chunk_size, fact_inittype, threadedfactor,
	                                                        compute_remainder)
	{
	}
};

}  // namespace blasted
