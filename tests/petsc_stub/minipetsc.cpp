// minipetsc.cpp -- TEST-ONLY implementation of tests/petsc_stub/petscksp.h (see there): just enough of PETSc's
// object model, single rank, for blasted_petsc.cpp's callbacks to be installed and EXECUTED by
// tests/cpp/petsc_driver.cpp.  Device memory of the HIP vectors comes from the backend's own C ABI
// (blasted_hip_buffer_*), so no HIP header is needed here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "blasted_hip.h"
#include "petscksp.h"

namespace {

enum ClassId { CLS_MAT = 1, CLS_VEC, CLS_PC, CLS_KSP };

struct Header {
	int classid;
	std::string type;
	int refs = 1;
};

std::map<std::string, std::string> g_options;
int g_outstanding = 0, g_copies = 0, g_hip_accesses = 0;

}  // namespace

struct _p_PetscObject {
	Header h;
};

struct _p_Mat {
	Header h;
	int bs = 1;
	int nbrows = 0;  // (block-)rows
	std::vector<PetscInt> i, j, diag;
	std::vector<PetscScalar> a;
};

struct _p_Vec {
	Header h;
	int n = 0;
	std::vector<PetscScalar> host;
	void *dev = nullptr;
	bool host_valid = true, dev_valid = false;
};

struct _p_PC {
	Header h;
	Mat A = nullptr, P = nullptr;
	bool setup = false;
	// shell
	void *ctx = nullptr;
	std::string name;
	PetscErrorCode (*fsetup)(PC) = nullptr;
	PetscErrorCode (*fapply)(PC, Vec, Vec) = nullptr;
	PetscErrorCode (*fdestroy)(PC) = nullptr;
	PetscErrorCode (*frich)(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *,
	                        PCRichardsonConvergedReason *) = nullptr;
	// bjacobi / asm / ksp: one inner solver on the same (rank-local) matrix
	KSP sub = nullptr;
	std::string subtype = PCSHELL;
};

struct _p_KSP {
	Header h;
	PC pc = nullptr;
	Mat A = nullptr, P = nullptr;
	bool setup = false;
};

namespace {

bool is_container(const PC pc)
{
	return pc->h.type == PCBJACOBI || pc->h.type == PCASM || pc->h.type == PCKSP;
}

const std::string *find_option(const char *name)
{
	auto it = g_options.find(name ? name : "");
	return it == g_options.end() ? nullptr : &it->second;
}

void to_host(Vec v)
{
	if (!v->host_valid) {
		blasted_hip_buffer_download(v->host.data(), v->dev, sizeof(PetscScalar) * (size_t)v->n);
		v->host_valid = true;
		g_copies++;
	}
}

PetscErrorCode to_device(Vec v, bool need_contents)
{
	if (!v->dev) {
		if (blasted_hip_buffer_alloc(&v->dev, sizeof(PetscScalar) * (size_t)(v->n ? v->n : 1), 0) != BLASTED_HIP_OK)
			return PETSC_ERR_LIB;
	}
	if (need_contents && !v->dev_valid) {
		blasted_hip_buffer_upload(v->dev, v->host.data(), sizeof(PetscScalar) * (size_t)v->n);
		g_copies++;
	}
	v->dev_valid = true;
	return 0;
}

void destroy_pc(PC pc);

void destroy_ksp(KSP ksp)
{
	if (!ksp)
		return;
	destroy_pc(ksp->pc);
	delete ksp;
}

void destroy_pc(PC pc)
{
	if (!pc)
		return;
	if (pc->h.type == PCSHELL && pc->fdestroy)
		pc->fdestroy(pc);
	destroy_ksp(pc->sub);
	delete pc;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------- options

PetscErrorCode PetscOptionsSetValue(PetscOptions, const char name[], const char value[])
{
	g_options[name] = value ? value : "";
	return 0;
}

PetscErrorCode PetscOptionsClear(PetscOptions)
{
	g_options.clear();
	return 0;
}

PetscErrorCode PetscOptionsGetInt(PetscOptions, const char[], const char name[], PetscInt *val, PetscBool *set)
{
	const std::string *s = find_option(name);
	if (set)
		*set = s ? PETSC_TRUE : PETSC_FALSE;
	if (s)
		*val = (PetscInt)std::atoi(s->c_str());
	return 0;
}

PetscErrorCode PetscOptionsGetBool(PetscOptions, const char[], const char name[], PetscBool *val, PetscBool *set)
{
	const std::string *s = find_option(name);
	if (set)
		*set = s ? PETSC_TRUE : PETSC_FALSE;
	if (s) {
		const std::string v = *s;
		if (v.empty() || v == "1" || v == "true" || v == "TRUE" || v == "yes" || v == "on")
			*val = PETSC_TRUE;
		else if (v == "0" || v == "false" || v == "FALSE" || v == "no" || v == "off")
			*val = PETSC_FALSE;
		else
			return PETSC_ERR_ARG_WRONG;
	}
	return 0;
}

PetscErrorCode PetscOptionsGetString(PetscOptions, const char[], const char name[], char out[], size_t len,
                                     PetscBool *set)
{
	const std::string *s = find_option(name);
	if (set)
		*set = s ? PETSC_TRUE : PETSC_FALSE;
	if (s && len > 0) {
		std::strncpy(out, s->c_str(), len - 1);
		out[len - 1] = '\0';
	}
	return 0;
}

PetscErrorCode PetscOptionsGetIntArray(PetscOptions, const char[], const char name[], PetscInt vals[], PetscInt *nmax,
                                       PetscBool *set)
{
	const std::string *s = find_option(name);
	if (set)
		*set = s ? PETSC_TRUE : PETSC_FALSE;
	if (!s) {
		*nmax = 0;
		return 0;
	}
	PetscInt n = 0;
	size_t pos = 0;
	while (pos <= s->size() && n < *nmax) {
		size_t comma = s->find(',', pos);
		if (comma == std::string::npos)
			comma = s->size();
		if (comma > pos)
			vals[n++] = (PetscInt)std::atoi(s->substr(pos, comma - pos).c_str());
		pos = comma + 1;
	}
	*nmax = n;
	return 0;
}

// ---------------------------------------------------------------------------------------------- PCSHELL

PetscErrorCode PCShellGetContext(PC pc, void **ctx)
{
	*ctx = pc->ctx;
	return 0;
}
PetscErrorCode PCShellSetContext(PC pc, void *ctx)
{
	pc->ctx = ctx;
	return 0;
}
PetscErrorCode PCShellSetName(PC pc, const char name[])
{
	pc->name = name;
	return 0;
}
PetscErrorCode PCShellGetName(PC pc, const char *name[])
{
	*name = pc->name.c_str();
	return 0;
}
PetscErrorCode PCShellSetSetUp(PC pc, PetscErrorCode (*f)(PC))
{
	pc->fsetup = f;
	return 0;
}
PetscErrorCode PCShellSetApply(PC pc, PetscErrorCode (*f)(PC, Vec, Vec))
{
	pc->fapply = f;
	return 0;
}
PetscErrorCode PCShellSetDestroy(PC pc, PetscErrorCode (*f)(PC))
{
	pc->fdestroy = f;
	return 0;
}
PetscErrorCode PCShellSetApplyRichardson(PC pc, PetscErrorCode (*f)(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal,
                                                                      PetscInt, PetscBool, PetscInt *,
                                                                      PCRichardsonConvergedReason *))
{
	pc->frich = f;
	return 0;
}

// ---------------------------------------------------------------------------------------------- PC / KSP

PetscErrorCode PCGetOperators(PC pc, Mat *A, Mat *P)
{
	if (A)
		*A = pc->A;
	if (P)
		*P = pc->P;
	return 0;
}

PetscErrorCode PCSetType(PC pc, PCType type)
{
	pc->h.type = type;
	pc->setup = false;
	return 0;
}

PetscErrorCode KSPCreate(MPI_Comm, KSP *out)
{
	KSP ksp = new _p_KSP;
	ksp->h.classid = CLS_KSP;
	ksp->h.type = "preonly";
	ksp->pc = new _p_PC;
	ksp->pc->h.classid = CLS_PC;
	ksp->pc->h.type = PCNONE;
	*out = ksp;
	return 0;
}

PetscErrorCode KSPSetOperators(KSP ksp, Mat A, Mat P)
{
	ksp->A = A;
	ksp->P = P;
	ksp->pc->A = A;
	ksp->pc->P = P;
	// new values: the next KSPSetUp sets the preconditioner up again, all the way down
	ksp->setup = false;
	for (PC pc = ksp->pc; pc; pc = (pc->sub ? pc->sub->pc : nullptr)) {
		pc->setup = false;
		if (pc->sub)
			pc->sub->setup = false;
	}
	return 0;
}

PetscErrorCode KSPGetType(KSP ksp, KSPType *type)
{
	*type = ksp->h.type.c_str();
	return 0;
}

PetscErrorCode KSPSetFromOptions(KSP ksp)
{
	ksp->h.type = "gmres";  // PETSc's default for a KSP that is asked for its options
	if (const std::string *t = find_option("-ksp_type"))
		ksp->h.type = *t;
	if (const std::string *t = find_option("-pc_type"))
		ksp->pc->h.type = *t;
	if (const std::string *t = find_option("-sub_pc_type"))
		ksp->pc->subtype = *t;
	if (const std::string *t = find_option("-ksp_pc_type"))
		ksp->pc->subtype = *t;
	return 0;
}

PetscErrorCode KSPGetPC(KSP ksp, PC *pc)
{
	*pc = ksp->pc;
	return 0;
}

PetscErrorCode KSPGetOperators(KSP ksp, Mat *A, Mat *P)
{
	if (A)
		*A = ksp->A;
	if (P)
		*P = ksp->P;
	return 0;
}

PetscErrorCode PCSetUp(PC pc)
{
	if (pc->setup)
		return 0;
	if (is_container(pc)) {
		if (!pc->sub) {  // one block per rank: the inner solver works on the rank-local matrix itself
			PetscErrorCode ierr = KSPCreate(PETSC_COMM_SELF, &pc->sub);
			CHKERRQ(ierr);
			pc->sub->pc->h.type = pc->subtype;
		}
		// as in PETSc, the inner solver is created here but set up later, "on blocks" (PCSetUpOnBlocks, which
		// KSPSolve / PCApply reach): the application installs its PCSHELL callbacks in between
		pc->sub->A = pc->sub->pc->A = pc->A;
		pc->sub->P = pc->sub->pc->P = pc->P;
	} else if (pc->h.type == PCSHELL) {
		if (pc->fsetup) {
			PetscErrorCode ierr = pc->fsetup(pc);
			CHKERRQ(ierr);
		}
	} else if (pc->h.type == PCMG || pc->h.type == PCGAMG) {
		return PETSC_ERR_SUP;
	}
	pc->setup = true;
	return 0;
}

PetscErrorCode PCSetUpOnBlocks(PC pc)
{
	if (is_container(pc) && pc->sub) {
		PetscErrorCode ierr = KSPSetUp(pc->sub);
		CHKERRQ(ierr);
		return PCSetUpOnBlocks(pc->sub->pc);
	}
	return 0;
}

PetscErrorCode KSPSetUp(KSP ksp)
{
	if (ksp->setup)
		return 0;
	PetscErrorCode ierr = PCSetUp(ksp->pc);
	CHKERRQ(ierr);
	ksp->setup = true;
	return 0;
}

PetscErrorCode PCApply(PC pc, Vec r, Vec z)
{
	PetscErrorCode ierr = PCSetUp(pc);
	CHKERRQ(ierr);
	ierr = PCSetUpOnBlocks(pc);
	CHKERRQ(ierr);
	if (is_container(pc))
		return PCApply(pc->sub->pc, r, z);  // inner KSP of type preonly
	if (pc->h.type == PCSHELL) {
		if (!pc->fapply)
			return PETSC_ERR_ARG_WRONGSTATE;
		return pc->fapply(pc, r, z);
	}
	return PETSC_ERR_SUP;
}

PetscErrorCode PCApplyRichardsonExists(PC pc, PetscBool *exists)
{
	if (is_container(pc) && pc->sub)
		return PCApplyRichardsonExists(pc->sub->pc, exists);
	*exists = (pc->h.type == PCSHELL && pc->frich) ? PETSC_TRUE : PETSC_FALSE;
	return 0;
}

PetscErrorCode PCApplyRichardson(PC pc, Vec b, Vec x, Vec w, PetscReal rtol, PetscReal abstol, PetscReal dtol,
                                 PetscInt its, PetscBool guesszero, PetscInt *outits,
                                 PCRichardsonConvergedReason *reason)
{
	PetscErrorCode ierr = PCSetUp(pc);
	CHKERRQ(ierr);
	ierr = PCSetUpOnBlocks(pc);
	CHKERRQ(ierr);
	if (is_container(pc))
		return PCApplyRichardson(pc->sub->pc, b, x, w, rtol, abstol, dtol, its, guesszero, outits, reason);
	if (pc->h.type != PCSHELL || !pc->frich)
		return PETSC_ERR_SUP;
	return pc->frich(pc, b, x, w, rtol, abstol, dtol, its, guesszero, outits, reason);
}

static PetscErrorCode get_sub(PC pc, const char *type, PetscInt *nlocal, PetscInt *first, KSP **sub)
{
	if (pc->h.type != type || !pc->sub)
		return PETSC_ERR_ARG_WRONGSTATE;
	if (nlocal)
		*nlocal = 1;
	if (first)
		*first = 0;
	*sub = &pc->sub;
	return 0;
}

PetscErrorCode PCBJacobiGetSubKSP(PC pc, PetscInt *nlocal, PetscInt *first, KSP **sub)
{
	return get_sub(pc, PCBJACOBI, nlocal, first, sub);
}
PetscErrorCode PCASMGetSubKSP(PC pc, PetscInt *nlocal, PetscInt *first, KSP **sub)
{
	return get_sub(pc, PCASM, nlocal, first, sub);
}
PetscErrorCode PCKSPGetKSP(PC pc, KSP *sub)
{
	if (pc->h.type != PCKSP || !pc->sub)
		return PETSC_ERR_ARG_WRONGSTATE;
	*sub = pc->sub;
	return 0;
}
PetscErrorCode PCMGGetLevels(PC, PetscInt *)
{
	return PETSC_ERR_SUP;
}
PetscErrorCode PCMGGetSmoother(PC, PetscInt, KSP *)
{
	return PETSC_ERR_SUP;
}
PetscErrorCode PCMGGetCoarseSolve(PC, KSP *)
{
	return PETSC_ERR_SUP;
}

PetscErrorCode KSPDestroy(KSP *ksp)
{
	destroy_ksp(*ksp);
	*ksp = nullptr;
	return 0;
}

// ---------------------------------------------------------------------------------------------- Mat

static PetscErrorCode make_mat(const char *type, PetscInt bs, PetscInt m, PetscInt n, const PetscInt i[],
                               const PetscInt j[], const PetscScalar a[], Mat *out)
{
	if (m != n || bs < 1 || m % bs)
		return PETSC_ERR_ARG_SIZ;
	Mat A = new _p_Mat;
	A->h.classid = CLS_MAT;
	A->h.type = type;
	A->bs = bs;
	A->nbrows = m / bs;
	A->i.assign(i, i + A->nbrows + 1);
	const PetscInt nz = i[A->nbrows];
	A->j.assign(j, j + nz);
	A->a.assign(a, a + (size_t)nz * bs * bs);
	A->diag.assign(A->nbrows, -1);
	for (PetscInt r = 0; r < A->nbrows; r++)
		for (PetscInt k = i[r]; k < i[r + 1]; k++)
			if (j[k] == r)
				A->diag[r] = k;
	*out = A;
	return 0;
}

PetscErrorCode MatCreateSeqAIJWithArrays(MPI_Comm, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[],
                                         Mat *out)
{
	return make_mat(MATSEQAIJ, 1, m, n, i, j, a, out);
}

PetscErrorCode MatCreateSeqBAIJWithArrays(MPI_Comm, PetscInt bs, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[],
                                          PetscScalar a[], Mat *out)
{
	return make_mat(MATSEQBAIJ, bs, m, n, i, j, a, out);
}

PetscErrorCode MatDestroy(Mat *A)
{
	if (*A && --(*A)->h.refs == 0)
		delete *A;
	*A = nullptr;
	return 0;
}

PetscErrorCode MatGetLocalSize(Mat A, PetscInt *m, PetscInt *n)
{
	if (m)
		*m = A->nbrows * A->bs;
	if (n)
		*n = A->nbrows * A->bs;
	return 0;
}

PetscErrorCode MatMissingDiagonal(Mat A, PetscBool *missing, PetscInt *row)
{
	*missing = PETSC_FALSE;
	for (PetscInt r = 0; r < A->nbrows; r++)
		if (A->diag[r] < 0) {
			*missing = PETSC_TRUE;
			if (row)
				*row = r;
			break;
		}
	return 0;
}

PetscErrorCode MatGetRowIJ(Mat A, PetscInt shift, PetscBool symmetric, PetscBool blockcompressed, PetscInt *n,
                           const PetscInt *ia[], const PetscInt *ja[], PetscBool *done)
{
	// what the glue asks for: 0-based, not symmetrised, block-compressed for BAIJ
	if (shift != 0 || symmetric || (A->bs > 1 && !blockcompressed)) {
		*done = PETSC_FALSE;
		return 0;
	}
	*n = A->nbrows;
	*ia = A->i.data();
	*ja = A->j.data();
	*done = PETSC_TRUE;
	g_outstanding++;
	return 0;
}

PetscErrorCode MatRestoreRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *ia[],
                               const PetscInt *ja[], PetscBool *done)
{
	if (ia)
		*ia = nullptr;
	if (ja)
		*ja = nullptr;
	if (done)
		*done = PETSC_TRUE;
	g_outstanding--;
	return 0;
}

PetscErrorCode MatSeqAIJGetArrayRead(Mat A, const PetscScalar **a)
{
	if (A->h.type != MATSEQAIJ)
		return PETSC_ERR_ARG_WRONG;
	*a = A->a.data();
	g_outstanding++;
	return 0;
}
PetscErrorCode MatSeqAIJRestoreArrayRead(Mat, const PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}
PetscErrorCode MatSeqAIJGetArray(Mat A, PetscScalar **a)
{
	if (A->h.type != MATSEQAIJ)
		return PETSC_ERR_ARG_WRONG;
	*a = A->a.data();
	g_outstanding++;
	return 0;
}
PetscErrorCode MatSeqAIJRestoreArray(Mat, PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}
PetscErrorCode MatSeqBAIJGetArray(Mat A, PetscScalar **a)
{
	if (A->h.type != MATSEQBAIJ)
		return PETSC_ERR_ARG_WRONG;
	*a = A->a.data();
	g_outstanding++;
	return 0;
}
PetscErrorCode MatSeqBAIJRestoreArray(Mat, PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}

PetscErrorCode MatGetBlockSize(Mat A, PetscInt *bs)
{
	*bs = A->bs;
	return 0;
}

PetscErrorCode MatGetType(Mat A, MatType *type)
{
	*type = A->h.type.c_str();
	return 0;
}

PetscErrorCode PetscObjectTypeCompare(PetscObject obj, const char type[], PetscBool *same)
{
	*same = (obj && obj->h.type == type) ? PETSC_TRUE : PETSC_FALSE;
	return 0;
}

PetscErrorCode PetscObjectReference(PetscObject obj)
{
	obj->h.refs++;
	return 0;
}

// ---------------------------------------------------------------------------------------------- Vec

static PetscErrorCode make_vec(const char *type, PetscInt n, Vec *out)
{
	Vec v = new _p_Vec;
	v->h.classid = CLS_VEC;
	v->h.type = type;
	v->n = n;
	v->host.assign((size_t)n, 0.0);
	*out = v;
	return 0;
}

PetscErrorCode VecCreateSeq(MPI_Comm, PetscInt n, Vec *v)
{
	return make_vec(VECSEQ, n, v);
}
PetscErrorCode VecCreateSeqHIP(MPI_Comm, PetscInt n, Vec *v)
{
	return make_vec(VECSEQHIP, n, v);
}

PetscErrorCode VecDestroy(Vec *v)
{
	if (*v && --(*v)->h.refs == 0) {
		if ((*v)->dev)
			blasted_hip_buffer_free((*v)->dev);
		delete *v;
	}
	*v = nullptr;
	return 0;
}

PetscErrorCode VecGetType(Vec v, VecType *type)
{
	*type = v->h.type.c_str();
	return 0;
}

PetscErrorCode VecGetLocalSize(Vec v, PetscInt *n)
{
	*n = v->n;
	return 0;
}

PetscErrorCode VecGetArray(Vec v, PetscScalar **a)
{
	to_host(v);
	v->dev_valid = false;  // read-write host access
	*a = v->host.data();
	g_outstanding++;
	return 0;
}
PetscErrorCode VecGetArrayRead(Vec v, const PetscScalar **a)
{
	to_host(v);
	*a = v->host.data();
	g_outstanding++;
	return 0;
}
PetscErrorCode VecRestoreArray(Vec, PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}

PetscErrorCode VecSet(Vec v, PetscScalar s)
{
	for (auto &x : v->host)
		x = s;
	v->host_valid = true;
	v->dev_valid = false;
	return 0;
}

PetscErrorCode VecHIPGetArrayRead(Vec v, const PetscScalar **a)
{
	if (v->h.type != VECSEQHIP)
		return PETSC_ERR_ARG_WRONG;
	PetscErrorCode ierr = to_device(v, true);
	CHKERRQ(ierr);
	*a = static_cast<const PetscScalar *>(v->dev);
	g_hip_accesses++;
	g_outstanding++;
	return 0;
}
PetscErrorCode VecHIPRestoreArrayRead(Vec, const PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}
PetscErrorCode VecHIPGetArrayWrite(Vec v, PetscScalar **a)
{
	if (v->h.type != VECSEQHIP)
		return PETSC_ERR_ARG_WRONG;
	PetscErrorCode ierr = to_device(v, false);
	CHKERRQ(ierr);
	v->host_valid = false;
	*a = static_cast<PetscScalar *>(v->dev);
	g_hip_accesses++;
	g_outstanding++;
	return 0;
}
PetscErrorCode VecHIPRestoreArrayWrite(Vec, PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}
PetscErrorCode VecHIPGetArray(Vec v, PetscScalar **a)
{
	if (v->h.type != VECSEQHIP)
		return PETSC_ERR_ARG_WRONG;
	PetscErrorCode ierr = to_device(v, true);
	CHKERRQ(ierr);
	v->host_valid = false;
	*a = static_cast<PetscScalar *>(v->dev);
	g_hip_accesses++;
	g_outstanding++;
	return 0;
}
PetscErrorCode VecHIPRestoreArray(Vec, PetscScalar **a)
{
	*a = nullptr;
	g_outstanding--;
	return 0;
}

int MiniPetscOutstandingAccesses(void)
{
	return g_outstanding;
}
int MiniPetscHostDeviceCopies(void)
{
	return g_copies;
}
int MiniPetscHipAccesses(void)
{
	return g_hip_accesses;
}

}  // extern "C"
