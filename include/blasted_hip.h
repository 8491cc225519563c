/* blasted_hip.h -- C ABI of the MI355X (gfx950) backend of BLASTed's preconditioner-apply hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  The C++ operator
 * classes of the host layer (blasted_amd/host: SRPreconditioner hierarchy, SRFactory, PCSHELL glue)
 * call ONLY these entry points; each one names the reference routine whose OpenMP loop nest it
 * replaces (paths relative to the BLASTed source tree).
 *
 * Object model: one `blasted_hip_prec` per reference operator object (SRPreconditioner instance).
 * It owns the device mirror of the sparsity pattern (uploaded once), borrows or mirrors the matrix
 * values (re-set before every compute(), include/solverops_ilu0.hpp:54-56), and owns what the
 * reference operator owns: iluvals / scale / ytemp (src/solverops_ilu0.cpp:150-183), dblocks
 * (src/solverops_jacobi.cpp:31-38) and the ILU position lists (include/ilu_pattern.hpp:39-52).
 * All work is enqueued on the object's HIP stream.  Calls taking host vectors return after the result
 * is in host memory; calls taking device vectors return once the work is enqueued (stream ordered).
 *
 * Storage conventions are the reference's (include/srmatrixdefs.hpp:98-125): browptr[nbrows+1],
 * bcolind[nnzb] ascending inside every block-row, diagind[nbrows] = storage position of the
 * diagonal block (present in every block-row), vals[nnzb*bs*bs]; vectors block-interleaved.
 * FP64 values, int32 indices (the reference instantiates <double,int> only).
 *
 * Every function returns BLASTED_HIP_OK or an error code; blasted_hip_last_error() gives the text.
 * There is NO CPU fallback: without a usable gfx950 device every compute entry point fails.
 */
#ifndef BLASTED_HIP_H
#define BLASTED_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct blasted_hip_prec_s *blasted_hip_prec;

enum {
	BLASTED_HIP_OK = 0,
	BLASTED_HIP_EINVAL = 1,   /* invalid argument (std::invalid_argument in the host layer)      */
	BLASTED_HIP_ENODEV = 2,   /* no usable HIP device                                            */
	BLASTED_HIP_ERUNTIME = 3, /* a HIP call failed (std::runtime_error with hipGetErrorString)   */
	BLASTED_HIP_ESTATE = 4,   /* call order violated (e.g. apply before factorize)               */
	BLASTED_HIP_ENOTIMPL = 5  /* block size / layout not instantiated                            */
};

/* in-block layout: Eigen::ColMajor / Eigen::RowMajor of include/blasted_config.hpp:27-28 */
enum { BLASTED_HIP_COLMAJOR = 0, BLASTED_HIP_ROWMAJOR = 1 };

/* where a caller's array lives */
enum { BLASTED_HIP_HOST = 0, BLASTED_HIP_DEVICE = 1 };

/* sweep semantics.  ASYNC: in-place chaotic sweeps, one launch per sweep (the GPU counterpart of the
 * reference's `omp for schedule(dynamic) nowait`, src/solverops_ilu0.cpp:99-108).  JACOBI_SYNC:
 * double-buffered synchronous Jacobi sweeps, deterministic; used by the parity tests.  LEVEL: every
 * sweep is one exact in-order pass, run as one launch per dependency level (the reference's
 * level-scheduled operators, src/solverops_levels_ilu0.cpp:58-105, src/solverops_levels_sgs.cpp:52-123);
 * accepted by the apply / relaxation entry points, not by ilu0_factorize.  DETERMINISTIC: the product mode for
 * callers that need a FIXED linear operator (any non-flexible Krylov method): synchronous sweeps like
 * JACOBI_SYNC -- the same sweep counts, run-to-run identical results -- except that the forward half of
 * sgs_apply is the exact in-order solve, as it is in the reference at every thread count
 * (src/solverops_sgs.cpp:62-66).  The host C++ layer passes ASYNC by default (the reference's semantics; round 2
 * passed DETERMINISTIC) and DETERMINISTIC / LEVEL on request (-blasted_sweep_mode, BLASTED_HIP_SWEEP_MODE). */
enum { BLASTED_HIP_ASYNC = 0, BLASTED_HIP_JACOBI_SYNC = 1, BLASTED_HIP_LEVEL = 2, BLASTED_HIP_DETERMINISTIC = 3 };

/* include/async_initialization_decl.hpp:15-34, same numeric values as FactInit / ApplyInit */
enum { BLASTED_HIP_INIT_F_ZERO = 0, BLASTED_HIP_INIT_F_ORIGINAL = 1, BLASTED_HIP_INIT_F_SGS = 2,
       BLASTED_HIP_INIT_F_NONE = 3 };
enum { BLASTED_HIP_INIT_A_ZERO = 0, BLASTED_HIP_INIT_A_JACOBI = 1, BLASTED_HIP_INIT_A_NONE = 2 };

const char *blasted_hip_last_error(void);
int blasted_hip_device_count(void);

/* Lifetime.  own_stream != 0: the object creates (and later destroys) a private non-blocking stream and
 * `stream` is ignored.  own_stream == 0: all work goes to `stream`, a caller-owned hipStream_t; NULL is
 * the device's default (null) stream -- what a framework that enqueues its own kernels there
 * (e.g. torch's default stream) must pass so that both are ordered.
 * Replaces: construction/destruction of an SRPreconditioner (src/solverops_base.cpp:20-24). */
int blasted_hip_create(blasted_hip_prec *out, int device, void *stream, int own_stream);
int blasted_hip_destroy(blasted_hip_prec p);
int blasted_hip_synchronize(blasted_hip_prec p);
/* hipDeviceSynchronize on `device`: waits for every stream, the operators' private ones and any other
 * library's.  For callers that hand over device vectors produced on streams this library does not know
 * (the PCSHELL glue on PETSc's HIP vectors). */
int blasted_hip_device_synchronize(int device);

/* Sparsity pattern, once per object.  Validates sorted columns and the diagonal positions.
 * bs: 1 (scalar CSR semantics of AsyncILU0/AsyncSGS/Jacobi) or one of the instantiated block sizes.
 * Replaces: the SRMatrixStorage/CRawBSRMatrix view held by SRPreconditioner
 * (include/solverops_base.hpp:67-78). */
int blasted_hip_set_pattern(blasted_hip_prec p, int nbrows, int nnzb, int bs, int layout,
                            const int *browptr, const int *bcolind, const int *diagind, int loc);

/* Matrix values, before every compute.  loc == DEVICE borrows the pointer (zero copy; it must stay
 * valid and unchanged until the next set_values); loc == HOST uploads into an owned mirror. */
int blasted_hip_set_values(blasted_hip_prec p, const double *vals, int loc);

/* ---- ILU(0) ------------------------------------------------------------------------------ */

/* compute_ILU_positions_CSR_CSR, src/ilu_pattern.cpp:32-163 (integer work; bit-exact). */
int blasted_hip_ilu0_positions(blasted_hip_prec p);
/* total number of (lower,upper) pairs, and a host copy of the three lists (tests, diagnostics) */
int blasted_hip_ilu0_positions_size(blasted_hip_prec p, long *npairs);
int blasted_hip_ilu0_get_positions(blasted_hip_prec p, int *posptr, int *lowerp, int *upperp);

/* block_ilu0_factorize, src/async_blockilu_factor.cpp:47-149 (bs>1: diagonal blocks left inverted)
 * scalar_ilu0_factorize, src/async_ilu_factor.cpp:36-98 (bs==1: diagonal not inverted; INIT_F_ZERO
 * falls through to INIT_F_ORIGINAL as in the reference).
 * use_scaling: symmetric scaling by getScalingVector, src/rawsrmatrixutils.cpp:343-350.
 * precinfo: NULL, or 6 doubles in PrecInfo order (include/preconditioner_diagnostics.hpp:14-43). */
int blasted_hip_ilu0_factorize(blasted_hip_prec p, int nbuildsweeps, int fact_init, int use_scaling,
                               int mode, double *precinfo);

/* block_ilu0_apply, src/solverops_ilu0.cpp:55-148 ; scalar_ilu0_apply, :239-321.
 * z = S U^-1 L^-1 S r by napplysweeps lower sweeps then napplysweeps upper sweeps.
 * apply_init other than ZERO / JACOBI -> BLASTED_HIP_EINVAL (the reference throws, :125-126).
 * A negative sweep count (BLASTED_SEQUENTIAL_SYMBOL) selects the reference's sequential variants
 * (threadedfactor / threadedapply = false), i.e. the result of one in-order serial pass: apply runs one
 * level-scheduled pass (mode LEVEL); factorize runs the exact ILU(0) as one launch per dependency level,
 * each row's entries computed in storage order from final values of the earlier levels. */
int blasted_hip_ilu0_apply(blasted_hip_prec p, const double *r, double *z, int napplysweeps,
                           int apply_init, int mode, int loc);

/* ---- (block-)Jacobi, SGS, relaxation ----------------------------------------------------- */

/* BJacobiSRPreconditioner::compute / JacobiSRPreconditioner::compute,
 * src/solverops_jacobi.cpp:31-48,150-162: dblocks_i = A_ii^-1. */
int blasted_hip_jacobi_compute(blasted_hip_prec p);
/* BJacobiSRPreconditioner::apply, src/solverops_jacobi.cpp:51-63 */
int blasted_hip_jacobi_apply(blasted_hip_prec p, const double *r, double *z, int loc);

/* BJacobiSRPreconditioner::apply_relax, src/solverops_jacobi.cpp:66-119: at most maxits synchronous
 * (block-)Jacobi steps on x (initial guess and result).  check_tol != 0: stop as the reference does on
 * the step difference d_k = ||x_k+1 - x_k||_2: d_k < atol, d_k/d_0 < rtol or d_k/d_0 > dtol.
 * steps_done (may be NULL) receives the number of steps taken. */
int blasted_hip_jacobi_relax(blasted_hip_prec p, const double *b, double *x, int maxits, int check_tol,
                             double rtol, double atol, double dtol, int *steps_done, int loc);

/* Async[Block]SGS_SRPreconditioner::apply, src/solverops_sgs.cpp:47-83,149-176.
 * With INIT_A_NONE z is read as the initial guess of the backward sweeps (as in the reference). */
int blasted_hip_sgs_apply(blasted_hip_prec p, const double *r, double *z, int napplysweeps,
                          int apply_init, int mode, int loc);
/* Async[Block]SGS_SRPreconditioner::apply_relax, src/solverops_sgs.cpp:85-116,178-203:
 * maxits x (ascending pass, descending pass) of x_i = D_i^-1 (b_i - sum_{j!=i} A_ij x_j); x in/out. */
int blasted_hip_sgs_relax(blasted_hip_prec p, const double *b, double *x, int maxits, int mode,
                          int loc);

/* Chaotic[Block]Relaxation::apply / apply_relax (the `gs` type), src/relaxation_chaotic.cpp:21-70,
 * 92-125: nsweeps ascending passes of the same row update; x is the initial guess and the result. */
int blasted_hip_gs_relax(blasted_hip_prec p, const double *b, double *x, int nsweeps, int mode,
                         int loc);

/* ---- level schedule ------------------------------------------------------------------------ */

/* Dependency levels of the pattern (the role of computeLevels, src/levelschedule.cpp:13-72, without
 * its need for a level-ordered matrix): level(i) = 1 + max level(j) over j < i coupled to i.  Built
 * lazily by the first LEVEL-mode call; this entry point builds it eagerly (once per pattern). */
int blasted_hip_level_schedule(blasted_hip_prec p);
int blasted_hip_level_count(blasted_hip_prec p, int *nlevels);
/* out4 = { levels, fixed-point passes of the build, single-launch exact passes run so far, how many of
 * them gave up waiting and were redone with per-level launches (expected: 0) } */
int blasted_hip_level_stats(blasted_hip_prec p, long *out4);
/* HBM footprint: out4 = { bytes of device memory this operator holds now (pattern / value mirrors, factor,
 * derived copies, vectors, schedules), the peak of that figure, how many derived two-triangle copies of the
 * factor / matrix are resident (natural-order for the asynchronous sweeps, level-ordered for the exact
 * solves; at most one per array unless tuning "copies=both"), bytes of caller-owned host memory page-locked
 * through blasted_hip_host_register (process-wide) }.  Borrowed device arrays are not counted. */
int blasted_hip_memory_stats(blasted_hip_prec p, long *out4);
/* Class-aware placement of the sweeps' large buffers (process-wide counters; DESIGN.md "Address classes"): the
 * triangle copies the asynchronous sweeps stream are built from 2 GiB pieces that are checked, one by one, not to
 * share their HBM address class with the vector the sweep writes.  out5 = { buffers built that way, pieces kept,
 * pieces turned down (wrong class), pieces kept unchecked because the search had already held back a whole class
 * (expected: 0), probe timings taken }.  Tuning "placement=0" / BLASTED_HIP_PLACEMENT=0 switches it off. */
int blasted_hip_placement_stats(long *out5);
/* Where the buffers of an asynchronous ILU application lie relative to a caller's device vectors r and z (measured
 * with the address-class probe, about 0.2 s at 256^3 bs=4): out8 = { 1 GiB pieces of the lower triangle copy, how many
 * of them share ytemp's class (0 wanted: the lower sweeps write ytemp), how many share r's (all wanted: they read r),
 * pieces of the upper copy, how many share z's class (0 wanted: the upper sweeps write z), how many share ytemp's (all
 * wanted), ytemp in r's class (0 / 1), ytemp in z's class }; -1 = not applicable (no copies yet, vectors under 64 MiB). */
int blasted_hip_placement_check(blasted_hip_prec p, const double *r_dev, const double *z_dev, long *out8);
/* host copies (any may be NULL): level_of_row[nbrows], rows_by_level[nbrows] (stable: ascending row
 * inside a level), level_ptr[nlevels+1] */
int blasted_hip_get_levels(blasted_hip_prec p, int *level_of_row, int *rows_by_level, int *level_ptr);

/* ---- SpMV --------------------------------------------------------------------------------- */

/* BLAS_BSR::matrix_apply / BLAS_CSR::matrix_apply, src/blas/matvecs.cpp:26-48,78-92: y = A x */
int blasted_hip_spmv(blasted_hip_prec p, const double *x, double *y, int loc);
/* BLAS_BSR::gemv3 / BLAS_CSR::gemv3, src/blas/matvecs.cpp:52-75,94-108: z = a A x + b y
 * (x must not alias z) */
int blasted_hip_gemv3(blasted_hip_prec p, double a, const double *x, double b, const double *y,
                      double *z, int loc);

/* ---- read-back of operator state (tests, -blasted_compute_preconditioner_info) ------------ */
int blasted_hip_get_iluvals(blasted_hip_prec p, double *out_host);  /* nnzb*bs*bs */
int blasted_hip_get_dblocks(blasted_hip_prec p, double *out_host);  /* nbrows*bs*bs */
int blasted_hip_get_scale(blasted_hip_prec p, double *out_host);    /* nbrows*bs */
int blasted_hip_get_ytemp(blasted_hip_prec p, double *out_host);    /* nbrows*bs */
/* device pointer of the factor storage (benchmarks, tests: initialise in HBM without a host copy).  The caller may
 * write through it before the next call into the operator; copies the operator derived from the factor (compact
 * triangles, its diagonal) are made again from the storage when next needed. */
int blasted_hip_iluvals_device(blasted_hip_prec p, double **dev_ptr);

/* ---- raw HBM buffers: the storage behind device_vector<T> (include/device_container.hpp:19-20,
 * which on this backend is the HIP buffer holder).  Synchronous copies on the given device. */
int blasted_hip_buffer_alloc(void **dev_ptr, unsigned long nbytes, int device);
int blasted_hip_buffer_free(void *dev_ptr);
int blasted_hip_buffer_upload(void *dev_ptr, const void *host_ptr, unsigned long nbytes);
int blasted_hip_buffer_download(void *host_ptr, const void *dev_ptr, unsigned long nbytes);
/* Page-locks a caller-owned host range in place (hipHostRegister), so that the HOST-vector entry points and
 * set_values copy from / to it by direct DMA instead of through the runtime's staged pageable copy.  For
 * arrays whose lifetime the caller controls (the PCSHELL glue: the Mat's value array, the KSP's work vectors):
 * unregister BEFORE the memory is freed.  Process-wide; the address is the key. */
int blasted_hip_host_register(void *host_ptr, unsigned long nbytes);
int blasted_hip_host_unregister(void *host_ptr);
/* Measurement aid (bench.py's roofline object): the rate in GB/s at which this device streams nbytes of an
 * existing device buffer through a read-only kernel shaped like the sweeps' value stream (64 KiB per
 * workgroup, 16-byte non-temporal loads) -- the practical ceiling a sweep's algorithmic rate is read
 * against.  Average of `reps` launches after two untimed ones, on the null stream. */
int blasted_hip_measure_read_stream(const void *dev_ptr, unsigned long nbytes, int reps, double *gbps);

/* ---- tuning hook (process-wide; measurements only).  spec: NULL = default, "generic" = always the
 * generic kernel family, or "r<128|256>,nt<0|1>,u<1|2>[,s<1|2|3>]" for the tuned bs=4/8 kernel (s: block
 * slots per row at bs=4 -- 1 (default), 2, or 3 = one slot for the triangular sweeps only; u: row steps in
 * flight for the bs=4 triangular sweeps -- 1 (default; best accuracy per sweep) or 2 (2 % faster sweeps)); the same strings
 * are read once from the environment variable BLASTED_HIP_SWEEPW.  "factor4=0" / "factor4=1"
 * switches the tuned bs=4 factorisation kernel off / on (environment: BLASTED_HIP_FACTOR4); "factor1=0" /
 * "factor1=1" the chunk-staged scalar (CSR) factorisation kernel (BLASTED_HIP_FACTOR1).
 * "factorodd=0" / "factorodd=1": tuned bs=5/7 factorisation kernel off / on (BLASTED_HIP_FACTORODD).
 * "sweepwr=0" / "sweepwr=1": tuned row-major bs=4/8 sweep kernel off / on (BLASTED_HIP_SWEEPWR).
 * "sweepodd=0" / "sweepodd=1": tuned bs=3/5/7 sweep kernel off / on (environment: BLASTED_HIP_SWEEPODD).
 * "level=syncfree" (default) / "level=launch": exact passes as one persistent launch or as one launch
 * per dependency level (environment: BLASTED_HIP_LEVEL).  "levelstore=1" (default) / "levelstore=0":
 * exact triangular solves read level-ordered copies of the factor's triangles (one extra copy of the
 * factor, permuted once per factorisation) or the factor in place (environment: BLASTED_HIP_LEVELSTORE);
 * "interleave=1": interleaved row order inside a chunk for the in-place triangular sweeps (closer to the exact
 * solves per sweep, 10 % slower per sweep, no better inside a flexible Krylov solver; default 0; environment:
 * BLASTED_HIP_INTERLEAVE).
 * "relaxsplit=1" (default) / "relaxsplit=0": an exact relaxation pass runs as a product with the other
 * triangle from the previous iterate plus an exact triangular solve, or as one whole-row exact kernel.
 * "compact=1" (default) / "compact=0": asynchronous ILU sweeps read natural-order compact copies of the
 * factor's triangles (one more copy of the factor, one copy pass per factorisation) or the factor in
 * place (environment: BLASTED_HIP_COMPACT).
 * "copies=one" (default) / "copies=both": keep only the derived copy last asked for, or both orderings
 * (environment: BLASTED_HIP_COPIES=both).  "sgsfwd=exact" (default) / "sgsfwd=async": forward half of an
 * ASYNC-mode SGS application as one exact pass (the reference's semantics) or as asynchronous sweeps.
 * "levelperm=0": the exact ILU solves keep natural-order vectors instead of a level-ordered iterate.
 * "levelwide=0" keeps the general single-launch kernel also for column-major bs 4 / 8; "sfonestep=0"
 * lets a wave of that kernel prefetch several row steps instead of one.
 * "factorsf=0|1|2|3": the exact factorisation as one launch per dependency level (0), as one dependency-polling
 * launch where that is faster (1, default: stencil-like rows of any block size but 2 through the plan kernels --
 * matrix-core at bs = 4, lane-per-row at bs = 1 -- and bs >= 5 in general), always as one launch (2), or always as
 * one launch of the general kernel (3); the factor is the same bits in every form.  "factorsf=p0" keeps the plan
 * kernel of block sizes 3, 5, 7, 8 off, "factor4=s0" the small-array instantiation of the bs = 4 one.  "levelfast=1" (default) / "levelfast=0": the level-schedule build starts
 * with one dependency-polling launch (plus a confirming pass) or runs relaxation passes only.  Test hooks:
 * "factorsf=a1" / "levelfast=2" make the single-launch factorisation / the polling launch of the schedule build
 * behave as if a wave had given up waiting, so that the fall-backs run.  "factorskip=1" (default) / "factorskip=0": in-place factorisation sweeps leave
 * upper blocks without position pairs alone once they hold their value (the scaled matrix block), or visit
 * every entry in every sweep.  "xcdsuper=N|auto", "levelserial=N", "sweepodd=nt0|nt1|occ0|occ1":
 * measurement switches described where they are read (capi.hip).
 * Round 3: "interleave=0" (default) / "1" (interleaved row order of the in-place bs=4/8 triangular sweeps, the
 * finished row handed on in registers at bs=4) / "2" (the round-1 form through memory) / "3" (that form for
 * relaxation passes too); "latestore=2" (default) / "0|1|4": the in-place bs=4 triangular sweeps store a workgroup's
 * rows once, with 2 (1, 4) row steps of a wave in flight, or step by step (0) -- bs=8: any non-zero value = stored
 * once (environment: BLASTED_HIP_LATESTORE); "invertrow=1" (default) / "0": diagonal blocks of size 5..8 inverted
 * by eight lanes per block or by one thread per block; "scalarlane=auto" (default) / "0|1|2|3|4": scalar (CSR)
 * triangular sweeps with one lane per row when they write a second buffer and the general kernel in place (auto),
 * the general kernel everywhere (0), one lane per row everywhere with one / two rows per lane (1 / 2), for the
 * whole-row operators too (3), or its one-wave sequential form (4) (environment: BLASTED_HIP_SCALARLANE);
 * "compactafter=N" (default -1 = by block size: 16 / 8 / 4 for bs 4 / other / 1): the compact triangle copies the
 * asynchronous sweeps read are made with the (N+1)-th application since the factor (matrix) last changed;
 * "factorfuse=1" (default) / "0": asynchronous builds from INIT_F_ORIGINAL (no scaling) read the matrix itself
 * in their first sweep instead of a copy made by an initialisation pass;
 * "scalarstage=1" (default) / "0": scalar product and relaxation passes into a second buffer with the products staged
 * through LDS (rows of at most 8 entries), or the general kernel; "factor1plan=1" (default) / "0": scalar in-place factorisation sweeps on the per-pattern plan or with the round-2
 * kernel; "gunroll=2": the general scalar kernel with 2 / 4 row steps in flight.  Measurement hooks that give WRONG results (timing
 * experiments only): "levelnowait=1" (exact passes with nobody waiting), "gatherprobe=1" (odd block sizes gather
 * their own row), "gatherprobe=2|3" (store probes of the interleaved sweeps). */
int blasted_hip_set_tuning(const char *spec);

/* ---- per-phase HIP-event timing (bench.py roofline) -------------------------------------- */
/* When enabled every apply/relax/spmv/factor call brackets its sweep kernels with hipEvents on the
 * object's stream; nothing is synchronised until blasted_hip_get_timing.
 * out[0..5] = {lower-sweep ms, lower launches, upper-sweep ms, upper launches, other ms, other launches}
 * ("lower" = ascending sweep kernel, "upper" = descending sweep kernel; for SpMV and the
 * factorisation sweeps the time is reported under "lower"). */
int blasted_hip_set_timing(blasted_hip_prec p, int enable);
int blasted_hip_get_timing(blasted_hip_prec p, double *out6, int reset);

#ifdef __cplusplus
}
#endif
#endif
