// blasted_petsc.cpp -- PCSHELL glue between PETSc and the MI355X operators.
//
// Behaviour follows the reference's src/blasted_petsc.cpp (options :136-213, object creation :216-311,
// callbacks :403-575, tree walk :578-661, installation :663-721, timers :723-735), with two deliberate
// differences:
//  * the rank-local matrix is reached through PETSc's PUBLIC interface (MatGetRowIJ,
//    MatSeqAIJGetArrayRead, MatSeqBAIJGetArray) instead of the private Mat_SeqAIJ / Mat_SeqBAIJ structs
//    (src/blasted_petsc.cpp:14-15,285-297).  The (block-)row structure is copied once per operator and
//    handed back at once (MatRestoreRowIJ); the diagonal positions are found by one scan of the rows; the
//    value array is taken out and handed back around every compute(), which is when the operator reads it
//    (the values go to HBM there) -- a Mat whose value array moves gets a new operator;
//  * bctx->prectype is set before it is consulted when the Richardson callback is installed (the
//    reference reads it uninitialised);
//  * vectors that live in HBM (VECHIP / VECSEQHIP / VECMPIHIP, PETSc configured with HIP) are handed to the
//    operator as device pointers (VecHIPGetArrayRead / VecHIPGetArrayWrite): r and z never cross PCIe.
// Built where PETSc is available (make petsc PETSC_DIR=... PETSC_ARCH=...); in this repository it is also
// built and executed against the test-only mini-PETSc of tests/petsc_stub (tests/test_gpu_petsc.py).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "blasted_petsc_ext.hpp"

using namespace blasted;

typedef SRPreconditioner<PetscReal, PetscInt> BlastedPreconditioner;
typedef FactoryBase<PetscReal, PetscInt> BlastedFactory;

namespace {

/// what this glue owns next to an operator: its copy of the (block-)row structure, the diagonal positions,
/// and the Mat / value array the operator was created on
struct LocalMatrix {
	std::vector<PetscInt> ia, ja, diag;
	Mat A = NULL;
	const PetscScalar *vals = NULL;
	// -blasted_pin_host_arrays: host ranges this operator page-locked (the Mat's value array), released with
	// the operator
	bool pin = false;
	std::vector<const void *> pinned;
	void pin_range(const void *p, const size_t nbytes)
	{
		if (!pin || !p || nbytes < (1u << 16) || pinned.size() >= 8)
			return;
		for (const void *q : pinned)
			if (q == p)
				return;
		if (blasted::detail::host_register(p, nbytes))
			pinned.push_back(p);
	}
	~LocalMatrix()
	{
		for (const void *q : pinned)
			blasted::detail::host_unregister(q);
	}
};

// side table: operator -> arrays this glue allocated for it (they must outlive the operator)
std::map<void *, LocalMatrix *> g_local_matrices;
void register_local_matrix(void *op, LocalMatrix *lm)
{
	g_local_matrices[op] = lm;
}
void release_local_matrix(void *op)
{
	auto it = g_local_matrices.find(op);
	if (it != g_local_matrices.end()) {
		delete it->second;
		g_local_matrices.erase(it);
	}
}
LocalMatrix *local_matrix_of(void *op)
{
	auto it = g_local_matrices.find(op);
	return it == g_local_matrices.end() ? nullptr : it->second;
}

// side table: PCSHELL context -> how its operator applies asynchronous sweeps (the C struct is public ABI and has
// no field for it).  -1: follow the process-wide choice.  Filled by setup_blasted_stack, read at every compute().
struct NodeMode {
	int mode = -1;
	bool fixed_outer = false;  // some KSP between this PC and the top of the tree assumes a fixed preconditioner
	std::string outer_type;    // ... the outermost such KSP's type
};
std::map<const Blasted_data *, NodeMode> g_node_modes;

/// Krylov methods that tolerate a preconditioner which changes from one application to the next (and the
/// non-Krylov ones: a single application, Richardson)
bool ksp_tolerates_variable_pc(KSP ksp, std::string &type)
{
	KSPType kt = NULL;
	if (KSPGetType(ksp, &kt) || !kt)
		return true;
	type = kt;
	static const char *const flexible[] = {"fgmres", "gcr", "richardson", "preonly", "fcg", "pipefgmres", "pipefcg",
	                                       "pipegcr", "fbcgs", "fbcgsr"};
	for (const char *f : flexible)
		if (type == f)
			return true;
	return false;
}

/// The value array of the rank-local matrix, through the accessor of its type
PetscErrorCode get_values(Mat A, const int bs, const PetscScalar **vals)
{
	PetscErrorCode ierr = 0;
	if (bs == 1) {
		ierr = MatSeqAIJGetArrayRead(A, vals); CHKERRQ(ierr);
	} else {
		PetscScalar *v = NULL;
		ierr = MatSeqBAIJGetArray(A, &v); CHKERRQ(ierr);
		*vals = v;
	}
	return ierr;
}
PetscErrorCode restore_values(Mat A, const int bs, const PetscScalar **vals)
{
	PetscErrorCode ierr = 0;
	if (bs == 1) {
		ierr = MatSeqAIJRestoreArrayRead(A, vals); CHKERRQ(ierr);
	} else {
		PetscScalar *v = const_cast<PetscScalar *>(*vals);
		ierr = MatSeqBAIJRestoreArray(A, &v); CHKERRQ(ierr);
		*vals = NULL;
	}
	return ierr;
}

#if defined(PETSC_HAVE_HIP)
/// true when the vector's entries live in HBM (any of PETSc's HIP vector types)
bool vec_is_hip(Vec v)
{
	VecType t = NULL;
	if (VecGetType(v, &t) || !t)
		return false;
	return std::strcmp(t, VECSEQHIP) == 0 || std::strcmp(t, VECHIP) == 0 || std::strcmp(t, "mpihip") == 0;
}
#endif

struct StopWatch {
	std::chrono::steady_clock::time_point w0 = std::chrono::steady_clock::now();
	std::clock_t c0 = std::clock();
	void add_to(double &wall, double &cpu) const
	{
		wall += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
		cpu += (double)(std::clock() - c0) / CLOCKS_PER_SEC;
	}
};

void require(const PetscBool found, const char *const tag)
{
	if (!found) {  // missing mandatory option: the reference aborts (src/blasted_petsc.cpp:36-39,81-84)
		std::printf("BLASTed: %s not set!\n", tag);
		std::fflush(stdout);
		std::abort();
	}
}

int mandatory_int(const char *const tag)
{
	PetscBool set = PETSC_FALSE;
	PetscInt val = 0;
	PetscOptionsGetInt(NULL, NULL, tag, &val, &set);
	require(set, tag);
	return (int)val;
}

bool mandatory_bool(const char *const tag)
{
	PetscBool set = PETSC_FALSE, val = PETSC_FALSE;
	if (PetscOptionsGetBool(NULL, NULL, tag, &val, &set))
		throw std::runtime_error("Petsc could not get optional bool option!");
	if (!set)
		throw std::runtime_error("Bool option " + std::string(tag) + " not set!");
	return val == PETSC_TRUE;
}

bool optional_bool(const char *const tag, const bool dflt)
{
	PetscBool set = PETSC_FALSE, val = dflt ? PETSC_TRUE : PETSC_FALSE;
	if (PetscOptionsGetBool(NULL, NULL, tag, &val, &set))
		throw std::runtime_error("Petsc could not get optional bool option!");
	if (!set)
		std::printf(" BLASTed: %s not set; using default value of %d\n", tag, (int)dflt);
	return val == PETSC_TRUE;
}

void mandatory_string(const char *const tag, char out[BLASTED_OPT_STRLEN])
{
	PetscBool set = PETSC_FALSE;
	PetscOptionsGetString(NULL, NULL, tag, out, BLASTED_OPT_STRLEN, &set);
	require(set, tag);
}

bool needs_async_params(const BlastedSolverType t)
{
	return t != BLASTED_JACOBI && t != BLASTED_LEVEL_SGS && t != BLASTED_NO_PREC;
}

bool has_factorisation(const BlastedSolverType t)
{
	return t == BLASTED_ILU0 || t == BLASTED_SAPILU0 || t == BLASTED_ASYNC_LEVEL_ILU0;
}

/// sweep count -1 selects the sequential variants (src/blasted_petsc.cpp:88-133)
void resolve_sequential(const Blasted_data *const ctx, AsyncSolverSettings &s)
{
	static_assert(BLASTED_SEQUENTIAL_SYMBOL < 0, "Symbol of sequential build/apply must be -ve!");
	s.nbuildsweeps = ctx->nbuildsweeps;
	s.napplysweeps = ctx->napplysweeps;
	if (s.prectype == BLASTED_SEQILU0)
		return;
	const bool seqa = ctx->napplysweeps == BLASTED_SEQUENTIAL_SYMBOL;
	const bool seqb = ctx->nbuildsweeps == BLASTED_SEQUENTIAL_SYMBOL;
	if ((seqa && seqb) || (seqa && s.prectype == BLASTED_SFILU0) || (seqb && s.prectype == BLASTED_SAPILU0)) {
		s.prectype = BLASTED_SEQILU0;
		s.nbuildsweeps = s.napplysweeps = 1;
		return;
	}
	if (seqa) {
		if (s.prectype != BLASTED_ILU0 && s.prectype != BLASTED_SAPILU0)
			throw std::runtime_error(" Seq. appl. only supported with async ILU factorization!");
		s.napplysweeps = 1;
		s.prectype = BLASTED_SAPILU0;
		std::printf("  Sequential application requested.\n");
	}
	if (seqb) {
		if (s.prectype != BLASTED_ILU0 && s.prectype != BLASTED_SFILU0)
			throw std::runtime_error(" Seq. fact. only supported with async triangular application!");
		s.nbuildsweeps = 1;
		s.prectype = BLASTED_SFILU0;
		std::printf("  Sequential factorization requested.\n");
	}
}

PetscErrorCode read_options(PC pc)
{
	Blasted_data *ctx;
	PetscErrorCode ierr = PCShellGetContext(pc, (void **)&ctx); CHKERRQ(ierr);
	const BlastedFactory *const factory = (const BlastedFactory *)ctx->bfactory;

	mandatory_string("-blasted_pc_type", ctx->prectypestr);
	const BlastedSolverType ptype = factory->solverTypeFromString(ctx->prectypestr);

	PetscInt sweeps[2] = {1, 1};
	if (needs_async_params(ptype)) {
		PetscBool set = PETSC_FALSE;
		PetscInt nmax = 2;
		PetscOptionsGetIntArray(NULL, NULL, "-blasted_async_sweeps", sweeps, &nmax, &set);
		if (!set || nmax < 2) {
			std::printf("BLASTed: Number of async sweeps not set properly!\n");
			std::fflush(stdout);
			std::abort();
		}
		if (has_factorisation(ptype)) {
			ctx->scale = mandatory_bool("-blasted_use_symmetric_scaling");
			mandatory_string("-blasted_async_fact_init_type", ctx->factinittype);
		} else {
			ctx->scale = false;
			std::strcpy(ctx->factinittype, "NA");
		}
		mandatory_string("-blasted_async_apply_init_type", ctx->applyinittype);
		ctx->threadchunksize = mandatory_int("-blasted_thread_chunk_size");
	}
	ctx->compute_precinfo = optional_bool("-blasted_compute_preconditioner_info", false);

	ctx->bprec = nullptr;
	ctx->prectype = ptype;
	ctx->nbuildsweeps = (int)sweeps[0];
	ctx->napplysweeps = (int)sweeps[1];
	ctx->first_setup_done = true;
	ctx->cputime = ctx->walltime = ctx->factorcputime = ctx->factorwalltime = ctx->applycputime =
	    ctx->applywalltime = 0;

	const std::string pcname = std::string("Blasted-") + ctx->prectypestr;
	ierr = PCShellSetName(pc, pcname.c_str()); CHKERRQ(ierr);
	return ierr;
}

/// Wraps the rank-local (block-)row arrays of the PC's matrix, zero copy, and creates the operator
PetscErrorCode create_operator(PC pc)
{
	Blasted_data *ctx;
	PetscErrorCode ierr = PCShellGetContext(pc, (void **)&ctx); CHKERRQ(ierr);
	release_local_matrix(ctx->bprec);
	delete reinterpret_cast<BlastedPreconditioner *>(ctx->bprec);
	ctx->bprec = nullptr;

	Mat A;
	ierr = PCGetOperators(pc, NULL, &A); CHKERRQ(ierr);
	PetscInt localrows, localcols;
	ierr = MatGetLocalSize(A, &localrows, &localcols); CHKERRQ(ierr);
	if (localrows != localcols)
		SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_SIZ, "BLASTed: the local matrix must be square");

	PetscBool diagmissing = PETSC_FALSE;
	PetscInt badrow = -1;
	ierr = MatMissingDiagonal(A, &diagmissing, &badrow); CHKERRQ(ierr);
	if (diagmissing)
		SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "! Zero diagonal in a (block-)row!");

	const BlastedFactory *const factory = (const BlastedFactory *)ctx->bfactory;
	AsyncSolverSettings settings;
	settings.prectype = factory->solverTypeFromString(ctx->prectypestr);
	settings.bs = ctx->bs;
	settings.blockstorage = ColMajor;  // PETSc BAIJ blocks are column-major
	settings.scale = ctx->scale;
	resolve_sequential(ctx, settings);
	settings.thread_chunk_size = ctx->threadchunksize;
	settings.compute_precinfo = ctx->compute_precinfo;
	settings.fact_inittype = INIT_F_NONE;
	settings.apply_inittype = INIT_A_NONE;
	if (needs_async_params(settings.prectype)) {
		if (has_factorisation(settings.prectype))
			settings.fact_inittype = getFactInitFromString(ctx->factinittype);
		settings.apply_inittype = getApplyInitFromString(ctx->applyinittype);
	}
	settings.relax = false;

	if (ctx->bs <= 0 || ctx->bs == 6 || ctx->bs > 8)
		SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "BLASTed: this block size is not supported!");

	// (block-)row structure through the public interface: copied (integers, once per operator) and handed
	// back at once, so that nothing of the Mat is held between calls
	const PetscBool compressed = ctx->bs > 1 ? PETSC_TRUE : PETSC_FALSE;
	PetscInt nbr = 0;
	const PetscInt *ia = NULL, *ja = NULL;
	PetscBool done = PETSC_FALSE;
	ierr = MatGetRowIJ(A, 0, PETSC_FALSE, compressed, &nbr, &ia, &ja, &done); CHKERRQ(ierr);
	if (!done || nbr != localrows / ctx->bs) {
		if (done)
			MatRestoreRowIJ(A, 0, PETSC_FALSE, compressed, &nbr, &ia, &ja, &done);
		SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "BLASTed: could not access the (block-)row structure");
	}
	LocalMatrix *lm = new LocalMatrix;
	lm->A = A;
	lm->ia.assign(ia, ia + nbr + 1);
	lm->ja.assign(ja, ja + ia[nbr]);
	ierr = MatRestoreRowIJ(A, 0, PETSC_FALSE, compressed, &nbr, &ia, &ja, &done);
	if (ierr) {
		delete lm;
		return ierr;
	}
	nbr = (PetscInt)lm->ia.size() - 1;

	// diagonal positions (integer scan, once per pattern)
	lm->diag.resize(nbr);
	for (PetscInt i = 0; i < nbr; i++) {
		PetscInt d = -1;
		for (PetscInt j = lm->ia[i]; j < lm->ia[i + 1]; j++)
			if (lm->ja[j] == i) {
				d = j;
				break;
			}
		if (d < 0) {
			delete lm;
			SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "! Missing diagonal block!");
		}
		lm->diag[i] = d;
	}

	// the operator wraps the Mat's value array, zero copy on the host side, and reads it in compute() only
	// (compute_preconditioner_blasted takes it out again around that call and checks that it has not moved)
	ierr = get_values(A, ctx->bs, &lm->vals);
	if (ierr) {
		delete lm;
		return ierr;
	}
	const PetscScalar *vals = lm->vals;
	const PetscScalar *held = vals;
	ierr = restore_values(A, ctx->bs, &held);
	if (ierr) {
		delete lm;
		return ierr;
	}

	BlastedPreconditioner *precop = NULL;
	try {
		precop = factory->create_preconditioner(
		    SRMatrixStorage<const PetscReal, const PetscInt>(lm->ia.data(), lm->ja.data(), vals, lm->diag.data(),
		                                                     lm->ia.data() + 1, nbr, lm->ia[nbr], lm->ia[nbr], ctx->bs),
		    settings);
	} catch (...) {
		delete lm;
		throw;
	}
	ctx->bprec = reinterpret_cast<void *>(precop);
	register_local_matrix(ctx->bprec, lm);  // released in cleanup_blasted together with the operator
	// Not a reference option: page-lock the Mat's value array, so that every compute() uploads it by direct DMA.
	// The Mat is the caller's and outlives the PC (KSPDestroy -> PCDestroy -> cleanup_blasted comes before the
	// application's MatDestroy).  Vectors are NOT pinned: the vectors a PC is applied to are KSP work vectors, and
	// KSPDestroy resets the KSP -- freeing them -- BEFORE it destroys the PC, so a range registered for the
	// operator's lifetime would be freed while registered (the stale-range hazard described in blasted_hip.h).
	// Device-resident vectors (VecHIP) are the way around the vector transfers.  Off by default.
	{
		PetscBool set = PETSC_FALSE, val = PETSC_FALSE;
		PetscOptionsGetBool(NULL, NULL, "-blasted_pin_host_arrays", &val, &set);
		lm->pin = set && val;
		lm->pin_range(lm->vals, sizeof(PetscScalar) * (size_t)lm->ia[nbr] * ctx->bs * ctx->bs);
	}

	ctx->infolist = NULL;
	if (ctx->compute_precinfo) {
		PrecInfoList *bpinfo = new PrecInfoList;
		bpinfo->infolist.reserve(250);
		ctx->infolist = static_cast<void *>(bpinfo);
	}
	return ierr;
}

}  // namespace

extern "C" {

Blasted_data_list newBlastedDataList()
{
	Blasted_data_list b;
	b.ctxlist = NULL;
	b.size = 0;
	b.bfactory = NULL;
	b._defaultfactory = 0;
	b.factorcputime = b.factorwalltime = b.applycputime = b.applywalltime = 0.0;
	return b;
}

void destroyBlastedDataList(Blasted_data_list *const b)
{
	if (b->_defaultfactory == 1) {
		delete (BlastedFactory *)b->bfactory;
		b->_defaultfactory = 0;
	}
	while (b->ctxlist != NULL) {
		Blasted_data *node = b->ctxlist;
		b->ctxlist = node->next;
		g_node_modes.erase(node);
		delete static_cast<PrecInfoList *>(node->infolist);
		delete node;
		b->size--;
	}
	if (b->size != 0)
		throw std::logic_error("Could not delete Blasted_data_list properly!");
}

Blasted_data newBlastedDataContext()
{
	Blasted_data ctx;
	std::memset(&ctx, 0, sizeof(ctx));
	ctx.bprec = NULL;
	ctx.infolist = NULL;
	ctx.first_setup_done = false;
	ctx.next = NULL;
	return ctx;
}

void appendBlastedDataContext(Blasted_data_list *const bdl, const Blasted_data bd)
{
	Blasted_data *node = new Blasted_data;
	*node = bd;
	node->next = bdl->ctxlist;
	bdl->ctxlist = node;
	bdl->size++;
}

PetscErrorCode cleanup_blasted(PC pc)
{
	Blasted_data *ctx;
	PetscErrorCode ierr = PCShellGetContext(pc, (void **)&ctx); CHKERRQ(ierr);
	release_local_matrix(ctx->bprec);
	delete reinterpret_cast<BlastedPreconditioner *>(ctx->bprec);
	ctx->bprec = NULL;
	return ierr;
}

PetscErrorCode compute_preconditioner_blasted(PC pc)
{
	Blasted_data *ctx;
	PetscErrorCode ierr = PCShellGetContext(pc, (void **)&ctx); CHKERRQ(ierr);
	if (!ctx->first_setup_done) {
		ierr = read_options(pc); CHKERRQ(ierr);
		ierr = create_operator(pc); CHKERRQ(ierr);
	}
	// the operator reads the Mat's values now: take the array out for the duration of the call.  A new Mat,
	// or an array that has moved since the operator was created, gets a new operator.
	Mat A;
	ierr = PCGetOperators(pc, NULL, &A); CHKERRQ(ierr);
	LocalMatrix *lm = local_matrix_of(ctx->bprec);
	const PetscScalar *vals = NULL;
	ierr = get_values(A, ctx->bs, &vals); CHKERRQ(ierr);
	if (!lm || lm->A != A || lm->vals != vals) {
		ierr = restore_values(A, ctx->bs, &vals); CHKERRQ(ierr);
		ierr = create_operator(pc); CHKERRQ(ierr);
		ierr = get_values(A, ctx->bs, &vals); CHKERRQ(ierr);
	}
	{
		const StopWatch sw;
		BlastedPreconditioner *const precop = reinterpret_cast<BlastedPreconditioner *>(ctx->bprec);
		{
			const auto nm = g_node_modes.find(ctx);
			precop->setSweepMode(nm == g_node_modes.end() ? -1 : nm->second.mode);
		}
		const PrecInfo pinfo = precop->compute();  // values H2D + factorisation on the GPU
		if (ctx->compute_precinfo)
			static_cast<PrecInfoList *>(ctx->infolist)->infolist.push_back(pinfo);
		sw.add_to(ctx->factorwalltime, ctx->factorcputime);
	}
	ierr = restore_values(A, ctx->bs, &vals); CHKERRQ(ierr);
	return ierr;
}

PetscErrorCode apply_local_blasted(PC pc, Vec r, Vec z)
{
	Blasted_data *ctx;
	PetscErrorCode ierr = PCShellGetContext(pc, (void **)&ctx); CHKERRQ(ierr);
	const BlastedPreconditioner *const prec = reinterpret_cast<const BlastedPreconditioner *>(ctx->bprec);
#if defined(PETSC_HAVE_HIP)
	if (vec_is_hip(r) && vec_is_hip(z) && prec->deviceVectorsAvailable()) {
		// both vectors live in HBM: the operator works on them in place.  PETSc's own kernels may run on any
		// stream and the operator has a private one, so the device is drained on either side of the call.
		const PetscReal *rd;
		PetscReal *zd;
		ierr = VecHIPGetArrayRead(r, &rd); CHKERRQ(ierr);
		ierr = VecHIPGetArrayWrite(z, &zd); CHKERRQ(ierr);
		{
			const StopWatch sw;
			blasted::detail::device_synchronize();
			prec->apply_device(rd, zd);
			blasted::detail::device_synchronize();
			sw.add_to(ctx->applywalltime, ctx->applycputime);
		}
		ierr = VecHIPRestoreArrayWrite(z, &zd); CHKERRQ(ierr);
		ierr = VecHIPRestoreArrayRead(r, &rd); CHKERRQ(ierr);
		return ierr;
	}
#endif
	const PetscReal *ra;
	PetscReal *za;
	ierr = VecGetArray(z, &za); CHKERRQ(ierr);
	ierr = VecGetArrayRead(r, &ra); CHKERRQ(ierr);
	{
		const StopWatch sw;
		prec->apply(ra, za);  // host vectors: r H2D, sweeps, z D2H
		sw.add_to(ctx->applywalltime, ctx->applycputime);
	}
	ierr = VecRestoreArrayRead(r, &ra); CHKERRQ(ierr);
	ierr = VecRestoreArray(z, &za); CHKERRQ(ierr);
	return ierr;
}

PetscErrorCode relax_local_blasted(PC pc, Vec rhs, Vec x, Vec w, const PetscReal rtol, const PetscReal abstol,
                                   const PetscReal dtol, const PetscInt it, const PetscBool guesszero,
                                   PetscInt *const outits, PCRichardsonConvergedReason *const reason)
{
	Blasted_data *ctx;
	PetscErrorCode ierr = PCShellGetContext(pc, (void **)&ctx); CHKERRQ(ierr);
	BlastedPreconditioner *const relaxation = reinterpret_cast<BlastedPreconditioner *>(ctx->bprec);
	// only the iteration count is used; SGS never checks tolerances (src/solverops_sgs.cpp:96-115)
	relaxation->setApplyParams({rtol, abstol, dtol, false, (int)it});
	if (guesszero) {
		ierr = VecSet(x, 0.0); CHKERRQ(ierr);
	}
#if defined(PETSC_HAVE_HIP)
	if (vec_is_hip(rhs) && vec_is_hip(x) && relaxation->deviceVectorsAvailable()) {
		const PetscReal *bd;
		PetscReal *xd;
		ierr = VecHIPGetArrayRead(rhs, &bd); CHKERRQ(ierr);
		ierr = VecHIPGetArray(x, &xd); CHKERRQ(ierr);  // initial guess and result
		{
			const StopWatch sw;
			blasted::detail::device_synchronize();
			relaxation->apply_relax_device(bd, xd);
			blasted::detail::device_synchronize();
			sw.add_to(ctx->applywalltime, ctx->applycputime);
		}
		ierr = VecHIPRestoreArray(x, &xd); CHKERRQ(ierr);
		ierr = VecHIPRestoreArrayRead(rhs, &bd); CHKERRQ(ierr);
		*reason = PCRICHARDSON_CONVERGED_ITS;
		*outits = it;
		return ierr;
	}
#endif
	const PetscReal *ba;
	PetscReal *xa;
	ierr = VecGetArray(x, &xa); CHKERRQ(ierr);
	ierr = VecGetArrayRead(rhs, &ba); CHKERRQ(ierr);
	{
		const StopWatch sw;
		relaxation->apply_relax(ba, xa);
		sw.add_to(ctx->applywalltime, ctx->applycputime);
	}
	ierr = VecRestoreArrayRead(rhs, &ba); CHKERRQ(ierr);
	ierr = VecRestoreArray(x, &xa); CHKERRQ(ierr);
	*reason = PCRICHARDSON_CONVERGED_ITS;
	*outits = it;
	return ierr;
}

PetscErrorCode setup_localpreconditioner_blasted(KSP ksp, Blasted_data *const bctx)
{
	Mat A;
	PetscErrorCode ierr = KSPGetOperators(ksp, NULL, &A); CHKERRQ(ierr);
	PetscInt matbs;
	MatType mtype;
	ierr = MatGetBlockSize(A, &matbs); CHKERRQ(ierr);
	ierr = MatGetType(A, &mtype); CHKERRQ(ierr);
	auto is = [&](const char *t) { return std::strcmp(mtype, t) == 0; };
	const bool isblock = is(MATBAIJ) || is(MATMPIBAIJ) || is(MATSEQBAIJ);
	const bool islocal = is(MATSEQAIJ) || is(MATSEQBAIJ);

	PC pc;
	ierr = KSPGetPC(ksp, &pc); CHKERRQ(ierr);
	PetscBool isshell;
	ierr = PetscObjectTypeCompare((PetscObject)pc, PCSHELL, &isshell); CHKERRQ(ierr);
	if (!isshell)
		SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_ARG_WRONGSTATE, "Need SHELL preconditioner for BLASTed!\n");
	if (!islocal)
		SETERRQ(PETSC_COMM_WORLD, PETSC_ERR_SUP, "PC as PCSHELL is only supported for local solvers.");

	bctx->bs = isblock ? (int)matbs : 1;
	bctx->first_setup_done = false;
	// know the type before deciding on the Richardson callback
	{
		char tstr[BLASTED_OPT_STRLEN];
		PetscBool set = PETSC_FALSE;
		PetscOptionsGetString(NULL, NULL, "-blasted_pc_type", tstr, BLASTED_OPT_STRLEN, &set);
		bctx->prectype = BLASTED_NO_PREC;
		if (set && bctx->bfactory)
			bctx->prectype = ((const BlastedFactory *)bctx->bfactory)->solverTypeFromString(tstr);
	}
	ierr = PCShellSetContext(pc, (void *)bctx); CHKERRQ(ierr);
	ierr = PCShellSetSetUp(pc, &compute_preconditioner_blasted); CHKERRQ(ierr);
	ierr = PCShellSetApply(pc, &apply_local_blasted); CHKERRQ(ierr);
	ierr = PCShellSetDestroy(pc, &cleanup_blasted); CHKERRQ(ierr);
	// the reference registers it for every type but ilu0 / cscbgs / none (src/blasted_petsc.cpp:711-718) and
	// lets apply_relax throw where it is not implemented; here only the types whose relaxation exists
	if (bctx->prectype == BLASTED_SGS || bctx->prectype == BLASTED_GS || bctx->prectype == BLASTED_LEVEL_SGS ||
	    bctx->prectype == BLASTED_JACOBI) {
		ierr = PCShellSetApplyRichardson(pc, &relax_local_blasted); CHKERRQ(ierr);
	}
	return ierr;
}

/// Not a reference option: -blasted_sweep_mode async|deterministic|exact chooses how the asynchronous types (ilu0,
/// sgs) of THIS list apply their sweeps (operators.hpp, SRPreconditioner::setSweepMode).  Without the option, and
/// without a process-wide choice (BLASTED_HIP_SWEEP_MODE, detail::set_sweep_mode), the mode follows the KSP tree:
/// the reference's chaotic sweeps (async) where every Krylov method above the PC takes a preconditioner that
/// changes between applications (fgmres, gcr, richardson, preonly, ...), and the deterministic (synchronous) sweeps
/// where one of them assumes a fixed operator (PETSc's default gmres, bcgs, cg): on this GPU the chaotic sweeps
/// differ from one application to the next far more than under the reference's few threads, and few sweeps on a
/// large subdomain make such a method stall or diverge ("Which outer solver each mode supports").
static PetscErrorCode sweep_mode_option(const Blasted_data_list *const bctx)
{
	char mstr[BLASTED_OPT_STRLEN];
	PetscBool set = PETSC_FALSE;
	PetscOptionsGetString(NULL, NULL, "-blasted_sweep_mode", mstr, BLASTED_OPT_STRLEN, &set);
	int asked = -1;
	if (set) {
		try {
			asked = blasted::detail::sweep_mode_from_string(mstr);
		} catch (const std::invalid_argument &e) {  // never let an exception unwind into a C caller
			std::fprintf(stderr, "setup_blasted_stack(): -blasted_sweep_mode %s: %s\n", mstr, e.what());
			SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_WRONG, "-blasted_sweep_mode must be async, deterministic or exact");
		}
	}
	const bool process_choice = blasted::detail::sweep_mode_is_explicit();
	bool told = false;
	for (const Blasted_data *node = bctx->ctxlist; node != NULL; node = node->next) {
		NodeMode &nm = g_node_modes[node];
		const bool chaotic = node->prectype == BLASTED_ILU0 || node->prectype == BLASTED_SFILU0 ||
		                     node->prectype == BLASTED_SGS;  // types whose APPLICATION runs asynchronous sweeps
		if (asked >= 0)
			nm.mode = asked;
		else if (process_choice || !chaotic || !nm.fixed_outer)
			nm.mode = -1;
		else {
			nm.mode = blasted::detail::sweep_mode_from_string("deterministic");
			if (!told)
				std::printf("setup_blasted_stack(): -ksp_type %s assumes a fixed preconditioner: the %s sweeps are "
				            "applied in the deterministic (synchronous) mode.  -blasted_sweep_mode async gives the "
				            "reference's chaotic sweeps (flexible methods: fgmres, gcr), exact the level-scheduled "
				            "solves.\n", nm.outer_type.c_str(), node->prectype == BLASTED_SGS ? "sgs" : "ilu0");
			told = true;
		}
	}
	return 0;
}

PetscErrorCode setup_blasted_stack(KSP ksp, Blasted_data_list *const bctx)
{
	BlastedFactory *factory = new SRFactory<double, int>();
	bctx->bfactory = (void *)factory;
	bctx->_defaultfactory = 1;
	PetscErrorCode ierr = setup_blasted_stack_ext(ksp, factory, bctx); CHKERRQ(ierr);
	return sweep_mode_option(bctx);
}

void computeTotalTimes(Blasted_data_list *const bctv)
{
	bctv->factorcputime = bctv->factorwalltime = bctv->applycputime = bctv->applywalltime = 0.0;
	for (Blasted_data *node = bctv->ctxlist; node != NULL; node = node->next) {
		bctv->factorwalltime += node->factorwalltime;
		bctv->applywalltime += node->applywalltime;
		bctv->factorcputime += node->factorcputime;
		bctv->applycputime += node->applycputime;
	}
}

}  // extern "C"

static int walk_ksp_tree(KSP ksp, const BlastedFactory *const fctry, Blasted_data_list *const bctv, NodeMode outer);

int setup_blasted_stack_ext(KSP ksp, const BlastedFactory *const fctry, Blasted_data_list *const bctv)
{
	return walk_ksp_tree(ksp, fctry, bctv, NodeMode());
}

static int walk_ksp_tree(KSP ksp, const BlastedFactory *const fctry, Blasted_data_list *const bctv, NodeMode outer)
{
	PC pc;
	PetscErrorCode ierr = KSPGetPC(ksp, &pc); CHKERRQ(ierr);
	{
		std::string type;
		if (!ksp_tolerates_variable_pc(ksp, type) && !outer.fixed_outer) {
			outer.fixed_outer = true;
			outer.outer_type = type;
		}
	}
	auto is = [&](const char *type, PetscBool *flag) {
		return PetscObjectTypeCompare((PetscObject)pc, type, flag);
	};
	PetscBool isbjacobi, isasm, isshell, ismg, isgamg, isksp;
	ierr = is(PCBJACOBI, &isbjacobi); CHKERRQ(ierr);
	ierr = is(PCASM, &isasm); CHKERRQ(ierr);
	ierr = is(PCSHELL, &isshell); CHKERRQ(ierr);
	ierr = is(PCMG, &ismg); CHKERRQ(ierr);
	ierr = is(PCGAMG, &isgamg); CHKERRQ(ierr);
	ierr = is(PCKSP, &isksp); CHKERRQ(ierr);

	if (isbjacobi || isasm) {
		ierr = KSPSetUp(ksp); CHKERRQ(ierr);
		ierr = PCSetUp(pc); CHKERRQ(ierr);
		PetscInt nlocal, first;
		KSP *subksp;
		if (isbjacobi) {
			ierr = PCBJacobiGetSubKSP(pc, &nlocal, &first, &subksp); CHKERRQ(ierr);
		} else {
			ierr = PCASMGetSubKSP(pc, &nlocal, &first, &subksp); CHKERRQ(ierr);
		}
		if (nlocal != 1)
			SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_WRONGSTATE, "Only one subdomain per rank is supported.");
		ierr = walk_ksp_tree(subksp[0], fctry, bctv, outer); CHKERRQ(ierr);
	} else if (ismg || isgamg) {
		ierr = KSPSetUp(ksp); CHKERRQ(ierr);
		ierr = PCSetUp(pc); CHKERRQ(ierr);
		PetscInt nlevels;
		ierr = PCMGGetLevels(pc, &nlevels); CHKERRQ(ierr);
		for (PetscInt lvl = 1; lvl < nlevels; lvl++) {
			KSP smoother;
			ierr = PCMGGetSmoother(pc, lvl, &smoother); CHKERRQ(ierr);
			ierr = walk_ksp_tree(smoother, fctry, bctv, outer); CHKERRQ(ierr);
		}
		KSP coarse;
		ierr = PCMGGetCoarseSolve(pc, &coarse); CHKERRQ(ierr);
		ierr = walk_ksp_tree(coarse, fctry, bctv, outer); CHKERRQ(ierr);
	} else if (isksp) {
		ierr = KSPSetUp(ksp); CHKERRQ(ierr);
		ierr = PCSetUp(pc); CHKERRQ(ierr);
		KSP sub;
		ierr = PCKSPGetKSP(pc, &sub); CHKERRQ(ierr);
		ierr = walk_ksp_tree(sub, fctry, bctv, outer); CHKERRQ(ierr);
	} else if (isshell) {
		std::printf("setup_blasted_stack(): Found valid parent KSP for BLASTed.\n");
		appendBlastedDataContext(bctv, newBlastedDataContext());
		bctv->ctxlist->bfactory = bctv->bfactory ? bctv->bfactory : (void *)fctry;
		ierr = setup_localpreconditioner_blasted(ksp, bctv->ctxlist); CHKERRQ(ierr);
		g_node_modes[bctv->ctxlist] = outer;
	}
	return ierr;
}

namespace blasted {
int setup_blasted_stack_ext(KSP ksp, const FactoryBase<double, int> &factory, Blasted_data_list *const bctx)
{
	bctx->bfactory = (void *)&factory;
	PetscErrorCode ierr = ::setup_blasted_stack_ext(ksp, &factory, bctx); CHKERRQ(ierr);
	return sweep_mode_option(bctx);
}
}  // namespace blasted
