/* blasted_petsc.h -- PCSHELL surface of BLASTed for PETSc applications (C ABI).
 *
 * Same entry points, option names and public struct layouts as the reference's
 * include/blasted_petsc.h:31-167 (applications allocate Blasted_data / Blasted_data_list themselves, so
 * field order and types are part of the ABI).  Usage is unchanged (doc/user-doc.md:4-29):
 *   -pc_type bjacobi|asm  -sub_pc_type shell  -blasted_pc_type ilu0  -blasted_async_sweeps 3,3 ...
 * and, after KSPSetFromOptions,  setup_blasted_stack(ksp, &list).
 * On this backend the local preconditioner of every rank lives on a GPU of the node
 * (rank -> device: BLASTED_HIP_DEVICE, or the launcher's local-rank variable, modulo the device count).
 */
#ifndef BLASTED_PETSC_H
#define BLASTED_PETSC_H

#include <stdbool.h>
#include <petscksp.h>

#include "solvertypes.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BLASTED_OPT_STRLEN 20

/* context of one PCSHELL instance; next links the instances of one solver stack */
struct Blasted_node {
	void *bprec;                              /* the SRPreconditioner object */
	void *bfactory;                           /* the factory that made it */

	int bs;                                   /* block size of the local matrix */
	char prectypestr[BLASTED_OPT_STRLEN];     /* -blasted_pc_type */
	BlastedSolverType prectype;

	bool scale;                               /* -blasted_use_symmetric_scaling */
	int threadchunksize;                      /* -blasted_thread_chunk_size (no GPU meaning, kept) */

	int nbuildsweeps;                         /* -blasted_async_sweeps b,a */
	int napplysweeps;
	char factinittype[BLASTED_OPT_STRLEN];    /* -blasted_async_fact_init_type */
	char applyinittype[BLASTED_OPT_STRLEN];   /* -blasted_async_apply_init_type */

	bool compute_precinfo;                    /* -blasted_compute_preconditioner_info */
	void *infolist;                           /* PrecInfoList* when compute_precinfo */

	bool first_setup_done;                    /* must start false */

	double cputime;
	double walltime;
	double factorcputime;
	double factorwalltime;
	double applycputime;
	double applywalltime;

	struct Blasted_node *next;
};
typedef struct Blasted_node Blasted_data;

typedef struct {
	Blasted_data *ctxlist;
	int size;

	void *bfactory;
	int _defaultfactory;                      /* internal: 1 when the list owns bfactory */

	double factorcputime;
	double factorwalltime;
	double applycputime;
	double applywalltime;
} Blasted_data_list;

Blasted_data_list newBlastedDataList();
void computeTotalTimes(Blasted_data_list *const bctv);
/* call after KSPDestroy; throws std::logic_error if the list cannot be emptied */
void destroyBlastedDataList(Blasted_data_list *const bdv);

/* walks the KSP/PC tree (bjacobi, asm, mg, gamg, ksp) and installs BLASTed in every PCSHELL found */
PetscErrorCode setup_blasted_stack(KSP ksp, Blasted_data_list *const bctx);

Blasted_data newBlastedDataContext();
/* the new node becomes the head of the list */
void appendBlastedDataContext(Blasted_data_list *const bdl, const Blasted_data bd);

PetscErrorCode setup_localpreconditioner_blasted(KSP ksp, Blasted_data *const bctx);

/* the PCSHELL callbacks */
PetscErrorCode cleanup_blasted(PC pc);
PetscErrorCode compute_preconditioner_blasted(PC pc);
PetscErrorCode apply_local_blasted(PC pc, Vec r, Vec z);
PetscErrorCode relax_local_blasted(PC pc, Vec rhs, Vec x, Vec w, PetscReal rtol, PetscReal abstol,
                                   PetscReal dtol, PetscInt it, PetscBool guesszero, PetscInt *outits,
                                   PCRichardsonConvergedReason *reason);

#ifdef __cplusplus
}
#endif
#endif
