// ctx.hpp -- internal state of one blasted_hip_prec object and the launch interface between the
// C-ABI translation unit (capi.hip) and the kernel translation units.  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/blasted_hip.h"

namespace bhip {

void set_error(const std::string &msg);

struct HipFailure {
	int code;
};

#define BHIP_CHECK(expr)                                                                        \
	do {                                                                                        \
		hipError_t e_ = (expr);                                                                 \
		if (e_ != hipSuccess) {                                                                 \
			::bhip::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));               \
			throw ::bhip::HipFailure{BLASTED_HIP_ERUNTIME};                                      \
		}                                                                                       \
	} while (0)

#define BHIP_FAIL(code_, msg_)                  \
	do {                                        \
		::bhip::set_error(msg_);                \
		throw ::bhip::HipFailure{code_};        \
	} while (0)

// Which stored blocks of a block-row a sweep visits
enum Part { PART_LOWER = 0, PART_UPPER = 1, PART_OFFDIAG = 2, PART_ALL = 3, PART_NONE = 4 };
// What is done with acc = sum_j A_ij x_j
enum Post {
	POST_SUB = 0,      // out = rhs - acc                         (unit lower solve)
	POST_D_SUB = 1,    // out = D (rhs - acc)                     (upper solve, forward GS, relaxation)
	POST_SUB_D = 2,    // out = rhs - D acc                       (backward GS)
	POST_AXPBY = 3     // out = a acc + b y                       (SpMV, gemv3)
};
// Where D comes from
enum DSrc {
	D_NONE = 0,
	D_VALS_DIAG = 1,   // the (already inverted) diagonal block of the factor, vals[diagind[i]]
	D_DBLOCKS = 2,     // separate array of inverted diagonal blocks, dblocks[i]
	D_RECIP_DIAG = 3   // scalar ILU: 1 / vals[diagind[i]]
};

struct Pattern {
	int nbrows = 0, nnzb = 0, bs = 0, rowmajor = 0;
	int max_row_len = 0;  // longest block-row (set by set_pattern's validation pass)
	const int *browptr = nullptr, *bcolind = nullptr, *diagind = nullptr;
};

// Measurement variants that give WRONG results on purpose (timing experiments that take a kernel apart) exist only in
// the probes build of these sources (make probes -> libblasted_hip_probes.so, -DBHIP_PROBES; tools/probes/): in the
// product library BHIP_PROBE(x) is the constant 0, the tuning strings that select them are turned down, and the extra
// kernel instantiations are not built.
#ifdef BHIP_PROBES
#define BHIP_PROBE(x) (x)
#else
#define BHIP_PROBE(x) 0
#endif

struct SweepArgs {
	Pattern pat;
	const double *vals;     // block values the sweep multiplies with
	const double *dvals;    // D source array (vals or dblocks), may be null
	const double *rhs;      // right-hand side (or y of gemv3), may be null for POST_AXPBY with b==0
	const double *rscale;   // optional elementwise factor on rhs (z = S r fused into the sweep)
	const double *xin;      // gathered iterate
	double *xout;           // written iterate (== xin for in-place async sweeps)
	double *xnat;           // level-ordered exact solves: optional second output in natural row order
	double *z1out;          // scalar lower sweep (kernels_sweep.hip): optional second output z1_i = (1 / dvals_i) x_i, dvals =
	                        // the factor's diagonal by row -- the first upper sweep from a zero iterate, fused (small applications)
	double a, b;            // POST_AXPBY coefficients
	int descending;         // row order of the sweep
	int xcd_shift;          // log2 of the super-chunk size of the XCD-aware chunk numbering (lanes.hpp); -1 = the
	                        // launcher's default: 4 (16 chunks per turn), 6 for the odd block sizes (kernels_sweepodd.hip)
	int interleave;         // in-place sweeps: rows of one step are taken a step count apart (see kernels_sweepw.hip)
	int latestore;          // in-place bs=4 triangular sweeps in natural order store a chunk's results once (kernels_sweepw.hip, LS)
	int probe;              // measurements only (tuning "gatherprobe=1", WRONG results): kernels_sweepodd.hip gathers
	                        // every block's x segment from the row's OWN position -- the sweep at its algorithmic traffic
};

// Level schedule of a pattern (kernels_level.hip): rows sorted by dependency depth
struct LevelSchedule {
	bool built = false;
	int nlevels = 0;
	long build_passes = 0;
	int *rows = nullptr;   // device, nbrows: rows ordered by (level, row)
	int *level = nullptr;  // device, nbrows: level of each row
	std::vector<int> ptr;  // host, nlevels + 1: level l = rows[ptr[l] .. ptr[l+1])
	// single-launch (sync-free) passes
	int count = 0;                     // nbrows
	int4 *meta = nullptr;              // device, nbrows: {row, browptr[row], diagind[row], browptr[row+1]} in level order
	int *ctl = nullptr;                // device, 4 ints: claim counter, abort flag, (build: longest lower / upper part)
	int max_lower = 0, max_upper = 0;  // longest strictly-lower / strictly-upper row part
	int sf_grid = 1024;                // workgroups of a persistent launch
	long sf_launches = 0, sf_aborts = 0;
	// level-ordered compact copies of the two triangles (pattern part; values live in the operator):
	// position p of the level order owns blocks [lptr[p], lptr[p+1]) of the strictly-lower copy and
	// [uptr[p], uptr[p+1]) of the diagonal+upper copy, so an exact pass streams contiguous memory
	bool storage_built = false;
	int *lptr = nullptr, *uptr = nullptr;    // device, nbrows + 1
	int *lcol = nullptr, *ucol = nullptr;    // device, nnzL / nnzU + nbrows
	int4 *lmeta = nullptr, *umeta = nullptr; // device, nbrows: {row, lptr[p], lptr[p+1], -} / {row, -, uptr[p], uptr[p+1]}
	int4 *lhead = nullptr, *uhead = nullptr; // device, 2 * nbrows: column indices of a position's first 8 blocks (-1: none)
	// the same indices as POSITIONS in the level order (exact solves that keep their iterate level-ordered)
	int *posof = nullptr;                      // device, nbrows: position of each row
	int *lcolp = nullptr, *ucolp = nullptr;    // device: lcol / ucol mapped through posof
	int4 *lheadp = nullptr, *uheadp = nullptr; // device, 2 * nbrows: lhead / uhead mapped through posof
	long nnz_lower = 0, nnz_dupper = 0;
	// single-launch exact factorisation at bs = 4 (kernels_factor4.hip): longest pair list of a row (-1: not looked
	// at yet), per-level workgroup layout
	int f4_maxpairs = -1, f4_grid = 0, f4_maxtodo = 0;
	int *f4_desc = nullptr;  // device: the rows' plans, 16 ints per row in padded level order
};

// Arrays a single-launch pass reads instead of the natural-order ones (level-ordered copies)
struct LevelView {
	const int4 *meta = nullptr;
	const int4 *head = nullptr;
	const int4 *headp = nullptr;  // head as positions
	const int *colp = nullptr;    // column indices as positions
	const int *ptr = nullptr;
	const int *bcolind = nullptr;
	const double *vals = nullptr;
};

struct FactorArgs {
	Pattern pat;
	const double *avals;    // original matrix values
	const double *scale;    // symmetric scaling vector or null
	const int *posptr, *lowerp, *upperp;
	const double *in;       // factor values read
	double *out;            // factor values written (== in for async)
	double *dinv_scratch;   // optional nbrows*bs*bs array (bs >= 5): receives the inverted diagonal blocks of `in`
	const int *rows;        // optional row list (level-scheduled exact factorisation), else all rows
	int nrows;              // length of `rows`
	int diag_inverted;      // exact factorisation, bs > 1: diagonal blocks are stored inverted as soon as they are
	                        // final, and lower blocks multiply with the stored inverse (general kernel only)
	int skip_fixed;         // in-place sweeps after the first: an upper block without position pairs is the (scaled)
	                        // matrix block, which the sweep before has stored -- neither read nor written again
	int lrow_fresh = 0;     // sweeps with in != out: a pair's l_ik, a lower block of the row being computed, is read from
	                        // `out` (what this sweep has just stored) instead of `in` -- the first sweep of a build whose
	                        // initialisation pass is fused into it (capi.hip: fuse_init)
	const int *f1_dcol = nullptr;    // scalar in-place sweeps (kernels_factor1.hip, factor1p_kernel): per entry, the
	const int4 *f1_chunks = nullptr; // position of its column's diagonal entry (-1: not lower); per chunk, its ranges
};

// kernels_sweep.hip
void launch_sweep(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s);
bool sweep_supported(int bs);
bool sweep_offsets_fit(const Pattern &pat);
void set_sweep_unroll(int u);
// kernels_sweepw.hip (tuned bs=4/8 column-major path; false = not covered, use the generic family)
bool launch_sweepw(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s);
void set_sweepw_variant(const char *spec);
// kernels_sweepwr.hip (tuned bs=4/8 ROW-major path; false = not covered)
bool launch_sweepwr(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s);
void set_sweepwr_enabled(int on);
// kernels_sweep1.hip (scalar CSR with short rows, one lane per row; false = not covered)
bool launch_sweep1(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s);
void set_scalar_lane(int v);
void set_scalar_stage(int v);
// kernels_sweepodd.hip (tuned bs=3/5/7 column-major path; false = not covered)
bool launch_sweepodd(const SweepArgs &a, Part part, Post post, DSrc dsrc, hipStream_t s);
void set_sweepodd_enabled(int on);
// kernels_level.hip (exact in-order passes, one launch per dependency level)
void build_level_schedule(const Pattern &pat, LevelSchedule &ls, hipStream_t s);
void free_level_schedule(LevelSchedule &ls);
int launch_level_sweep(const SweepArgs &a, Part part, Post post, DSrc dsrc, const LevelSchedule &ls,
                       hipStream_t s);
void launch_syncfree_fill(double *x, long n, hipStream_t s);
bool launch_syncfree_sweep(const SweepArgs &a, Part part, Post post, DSrc dsrc, const LevelSchedule &ls,
                           hipStream_t s, const LevelView *view = nullptr);
// kernels_levelw.hip (streaming exact triangular pass, column-major bs 4 / 8, level-ordered copies)
bool launch_syncfree_wide(const SweepArgs &a, bool upper, const LevelSchedule &ls, const int *ptr, const int *cols,
                          const int4 *head, hipStream_t s, bool permuted = false, bool sgs = false);
bool syncfree_wide_supported(const Pattern &pat);
void launch_level_unpermute(const LevelSchedule &ls, int bs, const double *xperm, double *xnat, hipStream_t s);
void set_levelw_enabled(int on);
void set_syncfree_one_step(int on);
void set_level_fast(int on);
void set_invert_rowlane(int on);
void set_syncfree_nowait(int on);
void set_level_serial_after(long n);
void build_natural_storage(const Pattern &pat, LevelSchedule &ns, hipStream_t s);
void build_level_storage(const Pattern &pat, LevelSchedule &ls, hipStream_t s);
void launch_level_permute_values(const Pattern &pat, const LevelSchedule &ls, const double *vals,
                                 double *lvals, double *uvals, hipStream_t s);
// kernels_factor1.hip (scalar CSR factorisation sweep with chunk-staged operands)
bool launch_factor1(const FactorArgs &a, hipStream_t s);
long factor1_plan_chunks(int nbrows);
void build_factor1_plan(const Pattern &pat, const int *posptr, int *dcol, int4 *chunks, hipStream_t s);
void set_factor1_enabled(int on);
// kernels_factor4.hip (tuned bs=4 column-major factorisation sweep on the matrix core)
bool launch_factor4(const FactorArgs &a, hipStream_t s);
void set_factor4_enabled(int on);
// kernels_factor8.hip (bs=8 column-major factorisation sweep on the matrix core)
bool launch_factor8(const FactorArgs &a, double *dinv_scratch, hipStream_t s);
void set_factor8_enabled(int on);
// kernels_factorodd.hip (bs=5/7 column-major factorisation sweep in the 16-byte pair layout)
bool launch_factorodd(const FactorArgs &a, double *dinv_scratch, hipStream_t s);
void set_factorodd_enabled(int on);
void set_factor_probe(int v);
// kernels_factor.hip
void launch_factor_sweep(const FactorArgs &a, hipStream_t s);
int launch_factor_levels(FactorArgs a, const LevelSchedule &ls, hipStream_t s);
int launch_factor_syncfree(const FactorArgs &a, LevelSchedule &ls, hipStream_t s);
int launch_factor4_syncfree(const FactorArgs &a, LevelSchedule &ls, hipStream_t s);
int launch_factor1_syncfree(const FactorArgs &a, LevelSchedule &ls, hipStream_t s);
bool build_row_plans(const FactorArgs &a, LevelSchedule &ls, int rpwg, hipStream_t s);
void launch_factor_pending_fill(const Pattern &pat, double *f, hipStream_t s);
void set_factor_syncfree(int on);
void launch_invert_diag_blocks(const Pattern &pat, const double *src, long src_is_indexed_by_diag,
                               double *dst, long dst_is_indexed_by_diag, hipStream_t s);
void launch_fact_init(const Pattern &pat, const double *avals, const double *scale, int init_type,
                      double *iluvals, double *dblk_scratch, hipStream_t s);
void launch_scaling_vector(const Pattern &pat, const double *vals, double *scale, hipStream_t s);
double run_nonlinear_res(const FactorArgs &a, double *dev_scratch, hipStream_t s);
void run_diag_dominance(const Pattern &pat, const double *fvals, double *dev_scratch, double *out4,
                        hipStream_t s);
// kernels_aux.hip
long run_ilu_positions(const Pattern &pat, int **posptr, int **lowerp, int **upperp, hipStream_t s);
void launch_scale_vec(double *z, const double *scale, long n, hipStream_t s);
void launch_gather_diag_blocks(const Pattern &pat, const double *vals, double *dst, hipStream_t s);
void launch_read_stream(const void *buf, unsigned long nbytes, double *sink, hipStream_t s);
double run_diff_norm(const double *x, const double *y, long n, double *dev_scratch, hipStream_t s);
int validate_pattern_device(const Pattern &pat, hipStream_t s, int *max_row_len);

struct Timing {
	struct Rec {
		hipEvent_t e0, e1;
		int kind;
		int launches;
	};
	bool enabled = false;
	std::vector<Rec> recs;
	std::vector<hipEvent_t> pool;
	double ms[3] = {0, 0, 0};
	double launches[3] = {0, 0, 0};
};

void launch_read2_probe(const void *r0, const void *r1, long bytes_each, long piece_bytes, double *sink, hipStream_t s);  // probes build
void launch_rw_probe(const void *rd, long rd_bytes, void *wr, long wr_bytes, double *sink, hipStream_t s, long passes = 1);  // kernels_aux.hip: address-class probe
void launch_page_walk(const void *buf, long nloads, long stride_bytes, double *sink, hipStream_t s);  // probes build
void trace_alloc(const void *p, size_t bytes);  // devmem.hip

// Device allocations of the library go through these two (devmem.hip): same contract as hipMalloc / hipFree,
// plus per-operator accounting -- the bytes are booked on the operator whose entry point is running on this
// thread (blasted_hip_memory_stats), or on nobody (raw buffers handed to the caller).
hipError_t tracked_malloc(void **p, size_t bytes);
hipError_t tracked_free(void *p);
template <typename T>
inline hipError_t tracked_malloc(T **p, size_t bytes)
{
	return tracked_malloc(reinterpret_cast<void **>(p), bytes);
}

}  // namespace bhip

struct blasted_hip_prec_s {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;

	bhip::Pattern pat;
	int *browptr_own = nullptr, *bcolind_own = nullptr, *diagind_own = nullptr;
	bool have_pattern = false;

	const double *vals = nullptr;
	double *vals_own = nullptr;

	int *posptr = nullptr, *lowerp = nullptr, *upperp = nullptr;
	long npairs = -1;
	int *f1_dcol = nullptr;     // scalar in-place factorisation sweeps: the plan of factor1p_kernel
	int4 *f1_chunks = nullptr;

	double *iluvals = nullptr, *iluvals2 = nullptr;
	double *finv = nullptr;  // inverted diagonal blocks of the current iterate (bs >= 5 factorisation sweeps)
	double *fdiag = nullptr;  // the factor's (inverted) diagonal blocks, contiguous: first synchronous upper sweep
	bool fdiag_valid = false;
	double *scale = nullptr;
	bool factored = false, scaled = false;
	double *ytemp = nullptr;
	double *dblocks = nullptr;
	bool jacobi_done = false;

	double *tmp[3] = {nullptr, nullptr, nullptr};    // n-vectors: Jacobi-sync ping-pong
	double *stage[3] = {nullptr, nullptr, nullptr};  // n-vectors: device copies of host vectors
	double *red = nullptr;                           // small reduction scratch

	bhip::LevelSchedule levels;
	// second copies of the factor and of the matrix, split into the strictly-lower and the diagonal+upper
	// triangle: natural row order (asynchronous sweeps) and level order (exact passes)
	struct TriCopy {
		double *l = nullptr, *u = nullptr;  // each triangle is allocated and refreshed on its own
		bool valid_l = false, valid_u = false;
		void invalidate() { valid_l = valid_u = false; }
	};
	long fac_applies = 0, mat_applies = 0;      // sweep applications since the factor / the matrix values last changed
	                                            // (the compact triangle copies are made once they pay: capi.hip, g_compact_after)
	double *yperm = nullptr, *zperm = nullptr;  // level-ordered iterates of the exact ILU solves
	bool y_in_level_order = false;              // yperm holds L^-1 r of the last exact apply, ytemp is stale
	bool y_natural_too = false;                 // the next exact lower solve also writes y by row (into its x)
	bhip::LevelSchedule natstore;  // natural-order compact triangle storage (pattern part)
	TriCopy fac_nat, fac_lvl, mat_nat, mat_lvl;

	bhip::Timing timing;

	// thorough placement only ("placement=2"): a copy of the matrix values out of the class of the vector the asynchronous
	// relaxation passes write (capi.hip, relax_impl) -- the borrowed values lie where the caller put them
	double *relax_vals = nullptr;
	bool relax_vals_valid = false;
	long relax_passes = 0;  // passes since the matrix values last changed

	double *zeros = nullptr;  // n zeros, never written: the iterate the first sweep of a small application reads (capi.hip)

	long bytes_owned = 0, bytes_peak = 0;  // device memory this operator holds (tracked_malloc)

	// class-aware placement (devmem.hip, placed_alloc): what the next derived copy should avoid / share its address
	// class with -- set by the entry point around the call that may allocate the copy
	const void *place_avoid = nullptr, *place_same = nullptr;
	size_t place_ref_bytes = 0;
	bool fac_placed = false;    // the factor's compact copies lie in storage that a placement search chose
	long ilu_apps_life = 0;     // asynchronous ILU applications since the pattern was set (refactorisations do not reset it)
	bool ytemp_placed = false;  // ytemp has been checked against (and moved out of) the classes of a caller's r and z

	long n() const { return (long)pat.nbrows * pat.bs; }
	long nvals() const { return (long)pat.nnzb * pat.bs * pat.bs; }
};
