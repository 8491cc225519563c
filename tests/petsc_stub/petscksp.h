/* petscksp.h -- TEST-ONLY declaration stub: the handful of PETSc names blasted_petsc.cpp uses, so that
 * the PCSHELL glue can be syntax- and signature-checked in an image without PETSc
 * (tests/test_host_build.py).  Nothing here is linked or executed; it is not a PETSc replacement. */
#ifndef BLASTED_TEST_PETSC_STUB_H
#define BLASTED_TEST_PETSC_STUB_H
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
typedef int MPI_Comm;
typedef enum { PCRICHARDSON_CONVERGED_RTOL = 2, PCRICHARDSON_CONVERGED_ATOL = 3, PCRICHARDSON_CONVERGED_ITS = 4 } PCRichardsonConvergedReason;
#define PETSC_COMM_SELF 1
#define PETSC_COMM_WORLD 2
#define PETSC_ERR_SUP 56
#define PETSC_ERR_LIB 76
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PCBJACOBI "bjacobi"
#define PCASM "asm"
#define PCSHELL "shell"
#define PCMG "mg"
#define PCGAMG "gamg"
#define PCKSP "ksp"
#define MATBAIJ "baij"
#define MATMPIBAIJ "mpibaij"
#define MATSEQBAIJ "seqbaij"
#define MATSEQAIJ "seqaij"
#define CHKERRQ(ierr) do { if (ierr) return ierr; } while (0)
#define SETERRQ(comm, code, msg) return (code)
PetscErrorCode PetscOptionsGetInt(PetscOptions, const char[], const char[], PetscInt *, PetscBool *);
PetscErrorCode PetscOptionsGetBool(PetscOptions, const char[], const char[], PetscBool *, PetscBool *);
PetscErrorCode PetscOptionsGetString(PetscOptions, const char[], const char[], char[], size_t, PetscBool *);
PetscErrorCode PetscOptionsGetIntArray(PetscOptions, const char[], const char[], PetscInt[], PetscInt *, PetscBool *);
PetscErrorCode PCShellGetContext(PC, void **);
PetscErrorCode PCShellSetContext(PC, void *);
PetscErrorCode PCShellSetName(PC, const char[]);
PetscErrorCode PCShellSetSetUp(PC, PetscErrorCode (*)(PC));
PetscErrorCode PCShellSetApply(PC, PetscErrorCode (*)(PC, Vec, Vec));
PetscErrorCode PCShellSetDestroy(PC, PetscErrorCode (*)(PC));
PetscErrorCode PCShellSetApplyRichardson(PC, PetscErrorCode (*)(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *, PCRichardsonConvergedReason *));
PetscErrorCode PCGetOperators(PC, Mat *, Mat *);
PetscErrorCode PCSetUp(PC);
PetscErrorCode PCBJacobiGetSubKSP(PC, PetscInt *, PetscInt *, KSP **);
PetscErrorCode PCASMGetSubKSP(PC, PetscInt *, PetscInt *, KSP **);
PetscErrorCode PCMGGetLevels(PC, PetscInt *);
PetscErrorCode PCMGGetSmoother(PC, PetscInt, KSP *);
PetscErrorCode PCMGGetCoarseSolve(PC, KSP *);
PetscErrorCode PCKSPGetKSP(PC, KSP *);
PetscErrorCode KSPGetPC(KSP, PC *);
PetscErrorCode KSPSetUp(KSP);
PetscErrorCode KSPGetOperators(KSP, Mat *, Mat *);
PetscErrorCode MatGetLocalSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatMissingDiagonal(Mat, PetscBool *, PetscInt *);
PetscErrorCode MatGetRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *[], const PetscInt *[], PetscBool *);
PetscErrorCode MatSeqAIJGetArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatSeqBAIJGetArray(Mat, PetscScalar **);
PetscErrorCode MatGetBlockSize(Mat, PetscInt *);
PetscErrorCode MatGetType(Mat, MatType *);
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char[], PetscBool *);
PetscErrorCode VecGetArray(Vec, PetscScalar **);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArray(Vec, PetscScalar **);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecSet(Vec, PetscScalar);
#ifdef __cplusplus
}
#endif
#endif
