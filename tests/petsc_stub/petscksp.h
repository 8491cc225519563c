/* petscksp.h -- TEST-ONLY mini-PETSc: the PETSc names blasted_petsc.cpp uses (plus what a driver needs to
 * build the KSP / PC / Mat / Vec objects around it), with the signatures of PETSc's public headers, so that
 * the PCSHELL glue can be compiled AND EXECUTED in an image without PETSc.  The implementation is
 * tests/petsc_stub/minipetsc.cpp: a single-rank options database, SeqAIJ / SeqBAIJ matrices, host and
 * HIP-resident vectors, and a KSP -> PC(bjacobi | asm | ksp) -> sub-KSP -> PC(shell) tree whose PCSetUp /
 * PCApply / PCApplyRichardson / PCDestroy call the shell callbacks the way PETSc does.
 * It is test infrastructure (tests/test_gpu_petsc.py, tests/cpp/petsc_driver.cpp), not a PETSc replacement
 * and not part of the product. */
#ifndef BLASTED_TEST_PETSC_STUB_H
#define BLASTED_TEST_PETSC_STUB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef int PetscErrorCode;
typedef int PetscInt;
typedef double PetscReal;
typedef double PetscScalar;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef struct _p_KSP *KSP;
typedef struct _p_PC *PC;
typedef struct _p_Mat *Mat;
typedef struct _p_Vec *Vec;
typedef struct _p_PetscObject *PetscObject;
typedef struct _p_PetscOptions *PetscOptions;
typedef const char *MatType;
typedef const char *VecType;
typedef const char *PCType;
typedef int MPI_Comm;
typedef enum { PCRICHARDSON_CONVERGED_RTOL = 2, PCRICHARDSON_CONVERGED_ATOL = 3, PCRICHARDSON_CONVERGED_ITS = 4 } PCRichardsonConvergedReason;
#define PETSC_COMM_SELF 1
#define PETSC_COMM_WORLD 2
#define PETSC_ERR_SUP 56
#define PETSC_ERR_LIB 76
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_ARG_WRONG 62
#define PCBJACOBI "bjacobi"
#define PCASM "asm"
#define PCSHELL "shell"
#define PCMG "mg"
#define PCGAMG "gamg"
#define PCKSP "ksp"
#define PCNONE "none"
#define MATBAIJ "baij"
#define MATMPIBAIJ "mpibaij"
#define MATSEQBAIJ "seqbaij"
#define MATSEQAIJ "seqaij"
#define VECSEQ "seq"
#define VECSEQHIP "seqhip"
#define VECHIP "hip"
#ifndef BLASTED_TEST_NO_PETSC_HIP
#define PETSC_HAVE_HIP 1 /* this "installation" has HIP vectors: the glue's device branch is compiled */
#endif
#define CHKERRQ(ierr) do { if (ierr) return ierr; } while (0)
#define SETERRQ(comm, code, msg) return (code)

/* options database (single, global: every call of the glue passes NULL) */
PetscErrorCode PetscOptionsSetValue(PetscOptions, const char name[], const char value[]);
PetscErrorCode PetscOptionsClear(PetscOptions);
PetscErrorCode PetscOptionsGetInt(PetscOptions, const char[], const char[], PetscInt *, PetscBool *);
PetscErrorCode PetscOptionsGetBool(PetscOptions, const char[], const char[], PetscBool *, PetscBool *);
PetscErrorCode PetscOptionsGetString(PetscOptions, const char[], const char[], char[], size_t, PetscBool *);
PetscErrorCode PetscOptionsGetIntArray(PetscOptions, const char[], const char[], PetscInt[], PetscInt *, PetscBool *);

/* PCSHELL */
PetscErrorCode PCShellGetContext(PC, void **);
PetscErrorCode PCShellSetContext(PC, void *);
PetscErrorCode PCShellSetName(PC, const char[]);
PetscErrorCode PCShellGetName(PC, const char *[]);
PetscErrorCode PCShellSetSetUp(PC, PetscErrorCode (*)(PC));
PetscErrorCode PCShellSetApply(PC, PetscErrorCode (*)(PC, Vec, Vec));
PetscErrorCode PCShellSetDestroy(PC, PetscErrorCode (*)(PC));
PetscErrorCode PCShellSetApplyRichardson(PC, PetscErrorCode (*)(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *, PCRichardsonConvergedReason *));

/* PC / KSP */
PetscErrorCode PCGetOperators(PC, Mat *, Mat *);
PetscErrorCode PCSetUp(PC);
PetscErrorCode PCSetUpOnBlocks(PC);
PetscErrorCode PCSetType(PC, PCType);
PetscErrorCode PCApply(PC, Vec, Vec);
PetscErrorCode PCApplyRichardson(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *, PCRichardsonConvergedReason *);
PetscErrorCode PCApplyRichardsonExists(PC, PetscBool *);
PetscErrorCode PCBJacobiGetSubKSP(PC, PetscInt *, PetscInt *, KSP **);
PetscErrorCode PCASMGetSubKSP(PC, PetscInt *, PetscInt *, KSP **);
PetscErrorCode PCMGGetLevels(PC, PetscInt *);
PetscErrorCode PCMGGetSmoother(PC, PetscInt, KSP *);
PetscErrorCode PCMGGetCoarseSolve(PC, KSP *);
PetscErrorCode PCKSPGetKSP(PC, KSP *);
PetscErrorCode KSPCreate(MPI_Comm, KSP *);
PetscErrorCode KSPSetOperators(KSP, Mat, Mat);
PetscErrorCode KSPSetFromOptions(KSP); /* -pc_type, -sub_pc_type / -ksp_pc_type (the inner PC's type) */
PetscErrorCode KSPGetPC(KSP, PC *);
typedef const char *KSPType;
PetscErrorCode KSPGetType(KSP, KSPType *); /* -ksp_type of the outermost KSP (PETSc's default: gmres); inner ones: preonly */
PetscErrorCode KSPSetUp(KSP);
PetscErrorCode KSPGetOperators(KSP, Mat *, Mat *);
PetscErrorCode KSPDestroy(KSP *);

/* Mat: sequential AIJ and BAIJ (column-major blocks, as PETSc stores them); the arrays are copied */
PetscErrorCode MatCreateSeqAIJWithArrays(MPI_Comm, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *);
PetscErrorCode MatCreateSeqBAIJWithArrays(MPI_Comm, PetscInt bs, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *);
PetscErrorCode MatDestroy(Mat *);
PetscErrorCode MatGetLocalSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatMissingDiagonal(Mat, PetscBool *, PetscInt *);
PetscErrorCode MatGetRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *[], const PetscInt *[], PetscBool *);
PetscErrorCode MatRestoreRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *[], const PetscInt *[], PetscBool *);
PetscErrorCode MatSeqAIJGetArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatSeqAIJRestoreArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatSeqAIJGetArray(Mat, PetscScalar **);
PetscErrorCode MatSeqAIJRestoreArray(Mat, PetscScalar **);
PetscErrorCode MatSeqBAIJGetArray(Mat, PetscScalar **);
PetscErrorCode MatSeqBAIJRestoreArray(Mat, PetscScalar **);
PetscErrorCode MatGetBlockSize(Mat, PetscInt *);
PetscErrorCode MatGetType(Mat, MatType *);
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char[], PetscBool *);
PetscErrorCode PetscObjectReference(PetscObject);

/* Vec: host (VECSEQ) and HIP-resident (VECSEQHIP) */
PetscErrorCode VecCreateSeq(MPI_Comm, PetscInt n, Vec *);
PetscErrorCode VecCreateSeqHIP(MPI_Comm, PetscInt n, Vec *);
PetscErrorCode VecDestroy(Vec *);
PetscErrorCode VecGetType(Vec, VecType *);
PetscErrorCode VecGetLocalSize(Vec, PetscInt *);
PetscErrorCode VecGetArray(Vec, PetscScalar **);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArray(Vec, PetscScalar **);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecSet(Vec, PetscScalar);
PetscErrorCode VecHIPGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPGetArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPRestoreArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPGetArray(Vec, PetscScalar **);
PetscErrorCode VecHIPRestoreArray(Vec, PetscScalar **);

/* test instrumentation (not PETSc): Get... / Restore... calls still outstanding over all objects ever created,
 * and how often a HIP vector had to be copied between host and device */
int MiniPetscOutstandingAccesses(void);
int MiniPetscHostDeviceCopies(void);
int MiniPetscHipAccesses(void); /* VecHIPGetArray... calls made (by the glue: the harness only uses host accessors) */
#ifdef __cplusplus
}
#endif
#endif
