// capi.hip -- implementation of the C ABI declared in include/blasted_hip.h.
// Host-side orchestration only: buffer ownership, sweep sequencing, init/prologue handling exactly
// as the reference's operator methods do it (citations at each entry point).  All arithmetic is in
// the kernel translation units; there is no CPU compute path in this library.
#include "ctx.hpp"
#include "devmem.hpp"

#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace bhip {

static thread_local std::string g_last_error;

void set_error(const std::string &msg)
{
	g_last_error = msg;
}

const char *last_error_text()
{
	return g_last_error.c_str();
}

template <typename F>
static int guarded(F &&f)
{
	struct OwnerReset {
		~OwnerReset()
		{
			tl_owner = nullptr;
			(void)release_deferred();  // what a placement search of this call held back goes back to the driver now
		}
	} reset;
	try {
		f();
		return BLASTED_HIP_OK;
	} catch (const HipFailure &e) {
		return e.code;
	} catch (const std::exception &e) {
		set_error(std::string("unexpected exception: ") + e.what());
		return BLASTED_HIP_ERUNTIME;
	}
}

static void use_device(blasted_hip_prec p)
{
	if (!p)
		BHIP_FAIL(BLASTED_HIP_EINVAL, "null blasted_hip_prec");
	BHIP_CHECK(hipSetDevice(p->device));
	tl_owner = p;
}

static void need_pattern(blasted_hip_prec p)
{
	if (!p->have_pattern)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "set_pattern has not been called");
}

static void need_values(blasted_hip_prec p)
{
	need_pattern(p);
	if (!p->vals)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "set_values has not been called");
}

static double *ensure(double *&buf, long count)
{
	if (!buf)
		buf = dev_alloc<double>((size_t)count);
	return buf;
}

// ---- timing ------------------------------------------------------------------------------

static hipEvent_t take_event(Timing &t)
{
	if (!t.pool.empty()) {
		hipEvent_t e = t.pool.back();
		t.pool.pop_back();
		return e;
	}
	hipEvent_t e;
	BHIP_CHECK(hipEventCreate(&e));
	return e;
}

static void fold_timing(blasted_hip_prec p)
{
	Timing &t = p->timing;
	if (t.recs.empty())
		return;
	BHIP_CHECK(hipStreamSynchronize(p->stream));
	for (auto &r : t.recs) {
		float ms = 0;
		BHIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
		t.ms[r.kind] += ms;
		t.launches[r.kind] += r.launches;
		t.pool.push_back(r.e0);
		t.pool.push_back(r.e1);
	}
	t.recs.clear();
}

struct Phase {
	blasted_hip_prec p;
	int kind, launches = 0;
	hipEvent_t e0 = nullptr;
	Phase(blasted_hip_prec p_, int kind_) : p(p_), kind(kind_)
	{
		if (p->timing.enabled) {
			if (p->timing.recs.size() >= 2048)
				fold_timing(p);
			e0 = take_event(p->timing);
			BHIP_CHECK(hipEventRecord(e0, p->stream));
		}
	}
	void done()
	{
		if (e0) {
			hipEvent_t e1 = take_event(p->timing);
			BHIP_CHECK(hipEventRecord(e1, p->stream));
			p->timing.recs.push_back({e0, e1, kind, launches});
			e0 = nullptr;
		}
	}
};

// ---- pinned staging of caller-owned host arrays ----------------------------------------------
// PETSc hands the PCSHELL callbacks plain pageable arrays (VecGetArray, the Mat's value array).  A caller that
// controls the lifetime of such an array can page-lock it in place (blasted_hip_host_register /
// _unregister), so that copies from and to it are direct DMA instead of the runtime's staged pageable copy.
// It is the CALLER's decision: a range that is freed while registered leaves a stale entry in the HIP
// runtime's table, and a later copy that touches a new allocation overlapping it fails with "invalid
// argument" -- which is what an automatic "register whatever is seen twice" table did to a numpy test that
// frees and reallocates its arrays (round 2, tests/test_gpu_fuzz.py), so there is no such automatism.
namespace {

struct PinTable {
	std::mutex mu;
	std::map<uintptr_t, size_t> ranges;
	long registered_bytes = 0;
};

// Never destroyed: a static destructor would run at process exit in an order that is undefined with respect to
// the HIP runtime's own teardown, and must not call into it (hipHostUnregister after the runtime is gone).  Ranges
// still registered at exit are the caller's to release (blasted_hip_host_unregister); the OS unlocks the pages.
PinTable &g_pins = *new PinTable;

}  // namespace

// ---- host <-> device vector staging --------------------------------------------------------

// Returns a device pointer holding the caller's input vector.
static const double *in_vec(blasted_hip_prec p, const double *v, int loc, int slot)
{
	if (loc == BLASTED_HIP_DEVICE)
		return v;
	double *d = ensure(p->stage[slot], p->n());
	BHIP_CHECK(hipMemcpyAsync(d, v, sizeof(double) * p->n(), hipMemcpyHostToDevice, p->stream));
	return d;
}

static double *out_vec(blasted_hip_prec p, double *v, int loc, int slot)
{
	if (loc == BLASTED_HIP_DEVICE)
		return v;
	return ensure(p->stage[slot], p->n());
}

static void finish_out(blasted_hip_prec p, double *host, const double *dev, int loc)
{
	if (loc == BLASTED_HIP_DEVICE)
		return;
	BHIP_CHECK(hipMemcpyAsync(host, dev, sizeof(double) * p->n(), hipMemcpyDeviceToHost, p->stream));
	BHIP_CHECK(hipStreamSynchronize(p->stream));
}

static void check_loc(int loc)
{
	if (loc != BLASTED_HIP_HOST && loc != BLASTED_HIP_DEVICE)
		BHIP_FAIL(BLASTED_HIP_EINVAL, "loc must be BLASTED_HIP_HOST or BLASTED_HIP_DEVICE");
}

static void check_mode(int mode)
{
	if (mode != BLASTED_HIP_ASYNC && mode != BLASTED_HIP_JACOBI_SYNC && mode != BLASTED_HIP_LEVEL &&
	    mode != BLASTED_HIP_DETERMINISTIC)
		BHIP_FAIL(BLASTED_HIP_EINVAL, "mode must be BLASTED_HIP_ASYNC, BLASTED_HIP_JACOBI_SYNC, BLASTED_HIP_LEVEL or "
		                              "BLASTED_HIP_DETERMINISTIC");
}

static LevelSchedule &need_levels(blasted_hip_prec p)
{
	if (!p->levels.built)
		build_level_schedule(p->pat, p->levels, p->stream);
	return p->levels;
}

// 0: single persistent launch per exact pass, per-level launches as the fallback; 1: per-level launches
static int g_level_impl = [] {
	const char *e = std::getenv("BLASTED_HIP_LEVEL");
	return (e && std::strcmp(e, "launch") == 0) ? 1 : 0;
}();

// 1: exact triangular solves with the factor stream from level-ordered copies of its two triangles
// (contiguous reads; costs a second copy of the factor and one permutation pass per factorisation)
static int g_level_store = [] {
	const char *e = std::getenv("BLASTED_HIP_LEVELSTORE");
	return (e && std::strcmp(e, "0") == 0) ? 0 : 1;
}();

// 1 (default): asynchronous ILU sweeps read natural-order compact copies of the factor's two triangles,
// so a sweep streams exactly its triangle's blocks and column indices (in place, the staged index range
// of a chunk also covers the other triangle: +2.4 % HBM traffic at 256^3 bs=4, +7 % time on the
// unstructured bs=5 case).  Cost: a second copy of the factor and one copy pass per factorisation
// (6.1 ms at 256^3, repaid after about 20 three-sweep applies).  Results are bit-identical.
// The copy pass costs what about 20 three-sweep applications gain from it, and a caller that refactorises every few
// applications (a Newton or time-stepping loop) would pay it every time for nothing: the copies are made when the
// g_compact_after-th sweep application since the factor (the matrix, for SGS) last changed comes along -- at the
// break-even point, so whatever the caller does costs at most twice the better of "never" and "at once".  Until then
// the sweeps read the factor in place (bit-identical results).  tuning "compactafter=N" / BLASTED_HIP_COMPACT_AFTER
// (0 = with the first application, as rounds 1-2 did and as bench.py asks for: it measures the steady state).
// Default (-1): by block size, from the measured break-even points (tools/compact_lazy_ab.py,
// profiles/r03_compact_lazy_ab.txt: a three-sweep build followed by K applications of 3+3 sweeps) -- 16 at bs = 4
// (256^3: copies at once 40.1 / 187.3 ms at K = 1 / 16 against 34.0 / 189.8 without), 8 at the other block sizes
// (unstructured bs = 5: even at K ~ 11), 4 for scalar rows (even at K ~ 5: in place a triangular sweep fetches whole
// rows' lines for half their entries).
// The quick placement search (the default) costs the application that runs it 105-145 ms at 256^3 bs=4 and returns
// about 0.35 ms per application: it pays after a few hundred applications, the copy pass alone after twenty.  So the
// default makes PLAIN copies when they pay (compactafter) and PLACED ones -- allocated beside the plain ones, which are
// freed afterwards -- only once the operator has been applied `placeafter` times in its life (refactorisations keep the
// count and the storage).  tuning "placeafter=N" / BLASTED_HIP_PLACE_AFTER; the thorough search ("placement=2") places
// at once.
static long g_place_after = [] {
	const char *e = std::getenv("BLASTED_HIP_PLACE_AFTER");
	return e ? std::atol(e) : 256L;
}();

static long g_compact_after = [] {
	const char *e = std::getenv("BLASTED_HIP_COMPACT_AFTER");
	return e ? std::atol(e) : -1L;
}();

// true: this application uses the compact copies (they exist already, or it is time to make them)
static bool compact_now(const blasted_hip_prec p, long &applies, const blasted_hip_prec_s::TriCopy &c)
{
	const long after = g_compact_after >= 0 ? g_compact_after : (p->pat.bs == 4 ? 16 : (p->pat.bs == 1 ? 4 : 8));
	const bool yes = applies >= after || (c.valid_l && c.valid_u);
	applies++;
	return yes;
}

static int g_compact = [] {
	const char *e = std::getenv("BLASTED_HIP_COMPACT");
	return (e && std::strcmp(e, "0") == 0) ? 0 : 1;
}();

// tuning ("copies=one" default / "copies=both"): the natural-order copy (asynchronous sweeps) and the
// level-ordered copy (exact solves) of the same array are alternatives -- an operator type uses one of them
// -- so by default only the one last asked for stays resident: block array + one derived copy = two copies
// (256^3 bs=4: 30 GB instead of 45 GB for an operator that has been applied both ways).  The rule is applied
// per TRIANGLE: the product-mode SGS application, whose forward half is an exact solve and whose backward half
// are asynchronous sweeps, holds the level-ordered lower and the natural-order upper triangle -- one copy's
// worth.  A program that alternates the two kinds of application on ONE operator pays a copy pass per
// switch (6 ms for both triangles at 256^3); "copies=both" keeps both orderings.
static int g_keep_both_copies = [] {
	const char *e = std::getenv("BLASTED_HIP_COPIES");
	return (e && std::strcmp(e, "both") == 0) ? 1 : 0;
}();

// Brings one triangle of the two-triangle copy `c` of the block array `src` up to date in the storage `st`;
// `other` is the copy of the same array in the other ordering, whose same triangle is given up (see above).
static void refresh_copy(blasted_hip_prec p, const LevelSchedule &st, const double *src,
                         blasted_hip_prec_s::TriCopy &c, blasted_hip_prec_s::TriCopy &other, const bool upper)
{
	const long bs2 = (long)p->pat.bs * p->pat.bs;
	double *&mine = upper ? c.u : c.l;
	bool &valid = upper ? c.valid_u : c.valid_l;
	if (!mine && !g_keep_both_copies) {
		// the same triangle in the other ordering has the same size: take its storage over instead of freeing it and
		// allocating anew (round 4: no allocation when an operator alternates between the two kinds of application, and a
		// buffer that was placed with care -- placed_alloc -- stays where it is)
		double *&theirs = upper ? other.u : other.l;
		mine = theirs;
		theirs = nullptr;
		(upper ? other.valid_u : other.valid_l) = false;
		valid = false;
	}
	if (!mine) {
		PlaceHint hint;
		hint.avoid = p->place_avoid;
		hint.prefer = p->place_same;  // (a preference: a piece that only stays out of `avoid`'s class is second best)
		hint.ref_bytes = p->place_ref_bytes;
		if (trace_placement())
			std::fprintf(stderr, "[blasted_hip] %s triangle copy, %s order:\n", upper ? "upper" : "lower", &st == &p->natstore ? "natural" : "level");
		mine = dev_alloc_placed<double>((size_t)((upper ? st.nnz_dupper : st.nnz_lower) * bs2), hint, p->stream);
		valid = false;
	}
	if (!valid) {
		launch_level_permute_values(p->pat, st, src, upper ? nullptr : c.l, upper ? c.u : nullptr, p->stream);
		valid = true;
	}
}

// Points the sweep arguments at the natural-order compact copy of one triangle of `src` (the factor or
// the matrix).  In the copy a row's part is its whole row: browptr/diagind are set so that the LOWER /
// UPPER kernels see exactly that.
static void compact_args(blasted_hip_prec p, bool upper, SweepArgs &a, const double *src,
                         blasted_hip_prec_s::TriCopy &c)
{
	LevelSchedule &ns = p->natstore;
	build_natural_storage(p->pat, ns, p->stream);
	refresh_copy(p, ns, src, c, &c == &p->fac_nat ? p->fac_lvl : p->mat_lvl, upper);
	if (upper) {
		a.pat.browptr = ns.uptr;
		a.pat.diagind = ns.uptr;  // the diagonal block is the first of a row of this copy
		a.pat.bcolind = ns.ucol;
		a.vals = c.u;
		if (a.dvals == src)
			a.dvals = c.u;
	} else {
		a.pat.browptr = ns.lptr;
		a.pat.diagind = ns.lptr + 1;  // the lower part ends where the next row starts
		a.pat.bcolind = ns.lcol;
		a.vals = c.l;
	}
}

// The level-ordered view of the lower or diagonal+upper triangle of `src`; false when the copies are
// switched off.
static bool level_view(blasted_hip_prec p, bool upper, LevelView &v, const double *src,
                       blasted_hip_prec_s::TriCopy &c)
{
	if (!g_level_store)
		return false;
	LevelSchedule &ls = need_levels(p);
	build_level_storage(p->pat, ls, p->stream);
	refresh_copy(p, ls, src, c, &c == &p->fac_lvl ? p->fac_nat : p->mat_nat, upper);
	v.meta = upper ? ls.umeta : ls.lmeta;
	v.ptr = upper ? ls.uptr : ls.lptr;
	v.head = upper ? ls.uhead : ls.lhead;
	v.headp = upper ? ls.uheadp : ls.lheadp;
	v.colp = upper ? ls.ucolp : ls.lcolp;
	v.bcolind = upper ? ls.ucol : ls.lcol;
	v.vals = upper ? c.u : c.l;
	return true;
}

// Address classes of a triangular-sweep application (profiles/r04_placement_combo.txt, 256^3 bs=4): a sweep is fastest
// when everything it READS lies in one class and the vector it WRITES in another --
//   lower sweep (streams the lower copy, reads r, writes ytemp): ytemp in the copy's class 1.46 ms; elsewhere 1.31 (r
//     in the copy's class) / 1.34 (r in the third class) / 1.36 (r in ytemp's class);
//   upper sweep (streams the upper copy, reads ytemp, writes z): z in the copy's class 1.83 ms; elsewhere 1.62 (ytemp in
//     the copy's class) / 1.65-1.67 (ytemp elsewhere).
// So, when the compact copies are made (the caller's r and z are at hand): the lower copy goes into r's class, ytemp is
// moved out of r's and z's classes if it is in one of them, and the upper copy goes into ytemp's class.  With r and z
// in one class -- the usual case for a caller's vectors -- that is lower 1.31 / upper 1.62 ms every time.
static void set_place_hint(blasted_hip_prec p, const void *avoid, const void *same)
{
	p->place_avoid = avoid;
	p->place_same = same;
	p->place_ref_bytes = (avoid || same) ? sizeof(double) * (size_t)p->n() : 0;
}

// BLASTED_HIP_TRACE_PLACEMENT: where everything ended up -- every stream of the application against r, z and ytemp
static void report_classes(blasted_hip_prec p, const double *dr, const double *dz)
{
	if (!trace_placement())
		return;
	const size_t nbytes = sizeof(double) * (size_t)p->n();
	double *sink = dev_alloc<double>(1);
	struct Item {
		const char *name;
		void *ptr;
		size_t bytes;
	};
	const LevelSchedule &ns = p->natstore;
	const size_t lb = (size_t)ns.nnz_lower * p->pat.bs * p->pat.bs * 8, ub = (size_t)ns.nnz_dupper * p->pat.bs * p->pat.bs * 8;
	std::vector<Item> items = {{"ytemp", p->ytemp, nbytes}, {"r", const_cast<double *>(dr), nbytes}, {"z", const_cast<double *>(dz), nbytes}};
	for (size_t at = 0; p->fac_nat.l && at + ((size_t)512 << 20) <= lb; at += (size_t)1 << 30)
		items.push_back({"lower copy piece", reinterpret_cast<char *>(p->fac_nat.l) + at, lb - at < ((size_t)1 << 30) ? lb - at : ((size_t)1 << 30)});
	for (size_t at = 0; p->fac_nat.u && at + ((size_t)512 << 20) <= ub; at += (size_t)1 << 30)
		items.push_back({"upper copy piece", reinterpret_cast<char *>(p->fac_nat.u) + at, ub - at < ((size_t)1 << 30) ? ub - at : ((size_t)1 << 30)});
	if (ns.lcol)
		items.push_back({"lower column indices", ns.lcol, (size_t)ns.nnz_lower * 4});
	if (ns.ucol)
		items.push_back({"upper column indices", ns.ucol, (size_t)ns.nnz_dupper * 4});
	const void *refs[3] = {dr, dz, p->ytemp};
	const char *rn[3] = {"r", "z", "ytemp"};
	for (const Item &it : items) {
		std::string line;
		for (int k = 0; k < 3; k++) {
			if (it.ptr == refs[k]) {
				line += std::string(" ") + rn[k] + ":itself";
				continue;
			}
			size_t rd = it.bytes & ~(size_t)0xffff, wr = (rd >> 4) & ~(size_t)0xfff;
			if (rd < ((size_t)32 << 20))
				continue;
			const double t_self = probe_ms(it.ptr, rd, static_cast<char *>(it.ptr) + rd - wr, wr, 4, sink, p->stream);
			const double t_ref = probe_ms(it.ptr, rd, const_cast<void *>(refs[k]), wr, 4, sink, p->stream);
			char buf[64];
			std::snprintf(buf, sizeof buf, " %s:%.3f", rn[k], t_ref / t_self);
			line += buf;
		}
		std::fprintf(stderr, "[blasted_hip] classes: %-22s %p (%5zu MiB) time ratio against%s   (about 1: same class, below 0.96: another)\n",
		             it.name, it.ptr, it.bytes >> 20, line.c_str());
	}
	dev_free(sink);
}

static void place_ytemp(blasted_hip_prec p, const double *dr, const double *dz)
{
	const size_t nbytes = sizeof(double) * (size_t)p->n();
	if (!g_placement || p->ytemp_placed || !p->ytemp || nbytes < ((size_t)64 << 20))
		return;
	p->ytemp_placed = true;
	PlaceHint h;
	h.avoid = dr;
	h.avoid2 = dz;
	h.ref_bytes = nbytes;
	double *sink = dev_alloc<double>(1);
	if (trace_placement())
		std::fprintf(stderr, "[blasted_hip] ytemp against r and z:\n");
	const int fits = class_fits(p->ytemp, nbytes, h, sink, p->stream);
	dev_free(sink);
	if (fits >= 0)
		return;
	double *moved = static_cast<double *>(placed_alloc(nbytes, h, p->stream));
	if (!moved)
		return;
	BHIP_CHECK(hipMemcpyAsync(moved, p->ytemp, nbytes, hipMemcpyDeviceToDevice, p->stream));
	BHIP_CHECK(hipStreamSynchronize(p->stream));
	dev_free(p->ytemp);
	p->ytemp = moved;
}

static int g_small_apply = 2;  // tuning "smallapply=0|1|2": see blasted_hip_ilu0_apply (1: without the fused first upper sweep)
static int g_level_perm = 1;  // tuning: exact ILU solves keep their iterate level-ordered (bs 4/8 column-major)

// tuning "applynone=1" (tests: the reference's "-initialization exact" fixed-point cases of the triangular sweeps,
// tests/solverops/async_triangular_factors_convergence.cpp:121-141): blasted_hip_ilu0_apply accepts INIT_A_NONE --
// the lower sweeps start from the operator's current ytemp, the upper sweeps from the caller's z.  Off by default:
// the reference's apply throws for that init type (src/solverops_ilu0.cpp:125-126), and so does this one.
static int g_apply_allow_none = 0;

// ytemp in natural order again (after an exact apply that kept y level-ordered)
static void restore_ytemp(blasted_hip_prec p)
{
	if (p->y_in_level_order) {
		launch_level_unpermute(p->levels, p->pat.bs, p->yperm, p->ytemp, p->stream);
		p->y_in_level_order = false;
	}
}

// The exact ILU solves with a level-ordered iterate (kernels_levelw.hip, PERM): the lower solve leaves y in
// p->yperm, the upper solve reads it there, gathers from p->zperm and also writes z in natural order to x.
// Returns 0 when this form does not apply (the caller continues with the natural-order forms).
constexpr long SF_ABORT_LIMIT = 3;  // see exact_pass

static int exact_pass_permuted(blasted_hip_prec p, SweepArgs a, bool upper, double *x, LevelSchedule &ls,
                               const LevelView &view, bool sgs)
{
	if (!g_level_perm || !syncfree_wide_supported(p->pat))
		return 0;
	if (upper && !p->y_in_level_order)
		return 0;
	const long n = p->n();
	// (round 4: the level-ordered iterates placed out of the address class of the triangle copy the pass streams --
	// placed_alloc -- changed nothing: 5.55 against 5.64 ms per exact application at 256^3 bs=4; these passes wait on
	// dependencies, not on the memory system.  Plain allocations.)
	double *out = upper ? ensure(p->zperm, n) : ensure(p->yperm, n);
	launch_syncfree_fill(out, n, p->stream);
	a.vals = view.vals;
	a.xin = out;
	a.xout = out;
	a.xnat = (upper || p->y_natural_too) ? x : nullptr;
	if (upper)
		a.rhs = p->yperm;
	if (!launch_syncfree_wide(a, upper, ls, view.ptr, view.colp, view.headp, p->stream, true, sgs))
		return 0;
	int ctl[2] = {0, 0};
	BHIP_CHECK(hipMemcpyAsync(ctl, ls.ctl, sizeof(ctl), hipMemcpyDeviceToHost, p->stream));
	BHIP_CHECK(hipStreamSynchronize(p->stream));
	ls.sf_launches++;
	if (ctl[1]) {
		ls.sf_aborts++;
		if (upper)
			restore_ytemp(p);  // the natural-order forms below read y from ytemp
		return 0;
	}
	if (!upper)
		p->y_in_level_order = !p->y_natural_too;  // (with the second output ytemp is current as well)
	return 2;
}

// One exact in-order pass of an operator producing `x` (for relaxation: from the previous iterate
// `xold`, a different vector; otherwise xold is ignored).  Returns the number of launches.
static int exact_pass(blasted_hip_prec p, SweepArgs a, Part part, Post post, DSrc dsrc, double *x,
                      const double *xold)
{
	if (p->pat.nbrows == 0)
		return 0;  // an empty subdomain: nothing to solve
	LevelSchedule &ls = need_levels(p);
	// A dependency-polling launch that has to give up costs its whole spin budget (seconds) before the per-level
	// launches redo the pass: an operator whose single-launch passes have given up SF_ABORT_LIMIT times (another
	// tenant holding compute units, a device where the dispatch order assumed here does not hold) stops trying.
	if (g_level_impl == 0 && ls.sf_aborts < SF_ABORT_LIMIT) {
		LevelView view;
		const bool triangular = part == PART_LOWER || part == PART_UPPER;
		bool use_view = false;
		if (triangular && a.vals == p->iluvals)
			use_view = level_view(p, part == PART_UPPER, view, p->iluvals, p->fac_lvl);
		else if (triangular && a.vals == p->vals)
			use_view = level_view(p, part == PART_UPPER, view, p->vals, p->mat_lvl);
		const bool ilu_lower = part == PART_LOWER && post == POST_SUB && dsrc == D_NONE;
		const bool ilu_upper = part == PART_UPPER && post == POST_D_SUB && dsrc == D_VALS_DIAG;
		const bool sgs_fwd = part == PART_LOWER && post == POST_D_SUB && dsrc == D_DBLOCKS;
		const bool sgs_bwd = part == PART_UPPER && post == POST_SUB_D && dsrc == D_DBLOCKS;
		const bool ilu_ops = a.vals == p->iluvals && (ilu_lower || ilu_upper);
		const bool sgs_ops = a.vals == p->vals && (sgs_fwd || sgs_bwd) && (p->pat.bs == 4 || p->pat.bs == 8);
		if (use_view && (ilu_ops || sgs_ops)) {
			const bool up = ilu_upper || sgs_bwd;
			if (!up)
				p->y_in_level_order = false;
			const int done = exact_pass_permuted(p, a, up, x, ls, view, sgs_ops);
			if (done)
				return done;
			if (up)
				restore_ytemp(p);
		}
		launch_syncfree_fill(x, p->n(), p->stream);
		a.xin = xold ? xold : x;
		a.xout = x;
		if (launch_syncfree_sweep(a, part, post, dsrc, ls, p->stream, use_view ? &view : nullptr)) {
			// the pass is only valid if no wave gave up waiting: look at the abort flag before going on
			int ctl[2] = {0, 0};
			BHIP_CHECK(hipMemcpyAsync(ctl, ls.ctl, sizeof(ctl), hipMemcpyDeviceToHost, p->stream));
			BHIP_CHECK(hipStreamSynchronize(p->stream));
			ls.sf_launches++;
			if (!ctl[1])
				return 2;
			ls.sf_aborts++;
		}
	}
	// per-level launches, in place: a relaxation pass starts from the previous iterate
	if (xold && xold != x)
		BHIP_CHECK(hipMemcpyAsync(x, xold, sizeof(double) * (size_t)p->n(), hipMemcpyDeviceToDevice, p->stream));
	a.xin = x;
	a.xout = x;
	return launch_level_sweep(a, part, post, dsrc, ls, p->stream);
}

// "interleave=1" / "interleave=0" (default; environment BLASTED_HIP_INTERLEAVE=1|0): the rows a workgroup of the
// bs=4/8 column-major in-place triangular sweeps computes side by side are taken a step count apart, so a row's
// predecessor belongs to the previous step instead of to the same step (stale): Gauss-Seidel-like along a chunk, as the
// reference's threads are inside their chunks.  Round 3 rebuilt it for bs = 4 (kernels_sweepw.hip, IW: right-hand
// side window, the just-finished row forwarded in registers: +10 % per sweep pair instead of +13..15 %) and measured
// it two ways (profiles/r03_sweep_order_quality.txt, 256^3 / 160^3 bs=4):
//  * distance of z to the exact solves: after 3+3 / 10+10 sweeps 0.160 / 5.4e-4 -> 0.058 / 1.2e-5, contraction per
//    sweep 0.445 -> 0.30: 41-43 ms instead of 56-59 ms to bring the distance to 1e-6;
//  * time to solution inside the reference's flexible solver, GCR(30) at 160^3, three repetitions: 3 sweeps 943-960
//    iterations in natural order against 1173-1196 interleaved, 5 sweeps 516-517 against 510-511 -- the interleaved
//    sweeps leave a smaller but differently structured error (every fourth row still sees a stale predecessor), and
//    as a preconditioner they are no better per application and 10 % dearer.
// The default is decided by the second measurement: natural order.  "interleave=2" is the round-1 form that goes
// through memory for everything, "interleave=3" applies that form to relaxation passes too (no gain measured).
// Where the round-1 form's +13..15 % came from (tools/probes): the timing-only variant without gathers pays +16 %
// for the 32-byte rhs / result pieces a step takes 4 rows apart; storing the triangles in sweep order does not
// help (the value stream was not the problem); and sweeping the symmetrically permuted system, on which every
// access is contiguous again, still pays +10 % per sweep pair -- fresh neighbours are lines another wave has
// just written -- plus two vector permutation passes per application.  Both were built, measured and removed.
static int g_interleave = [] {
	const char *e = std::getenv("BLASTED_HIP_INTERLEAVE");
	return (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 0;
}();

// tuning ("xcdsuper=N", N a power of two): the XCDs take turns on super-chunks of N consecutive chunks (lanes.hpp)
static int g_xcd_shift = -1;  // -1: each kernel family's own default (16 chunks per turn; 64 for the odd block sizes)

// tuning ("relaxsplit=0|1"): exact relaxation passes as product + exact triangular solve (default) or as one
// whole-row exact kernel
static int g_relax_split = 1;

// tuning ("factorskip=0|1"): in-place factorisation sweeps after the first leave upper blocks without position
// pairs alone (their value, the scaled matrix block, does not change from sweep to sweep)
static int g_factor_skip_fixed = 1;
// tuning ("factorfuse=1" default / "0"): the initialisation pass of asynchronous INIT_F_ORIGINAL builds (bs >= 2) fused
// into the first sweep (see blasted_hip_ilu0_factorize)
static int g_factor_fuse_init = 1;
// tuning ("factor1plan=0|1"): scalar in-place factorisation sweeps on the precomputed plan (kernels_factor1.hip,
// factor1p_kernel; default) or with the round-2 kernel
static int g_factor1_plan = 1;

// tuning ("sgsfwd=exact|async"): the forward half of an ASYNC-mode SGS application as one exact in-order pass (the
// reference's semantics, default) or as napplysweeps asynchronous sweeps
static int g_sgs_exact_fwd = 1;

static int g_gather_probe = 0;  // tuning "gatherprobe=1": see SweepArgs::probe

// tuning "latestore=2" (default) / "latestore=0|1|4" (environment BLASTED_HIP_LATESTORE): the in-place bs=4 triangular
// sweeps store a workgroup's rows once, with 2 (1, 4) row steps of a wave in flight; 0 = step by step (kernels_sweepw.hip, LS)
static int g_late_store = [] {
	const char *e = std::getenv("BLASTED_HIP_LATESTORE");
	return (e && (e[0] == '0' || e[0] == '1' || e[0] == '4')) ? e[0] - '0' : 2;
}();

static SweepArgs base_args(blasted_hip_prec p)
{
	SweepArgs a;
	std::memset(&a, 0, sizeof(a));
	a.pat = p->pat;
	a.interleave = g_interleave;
	a.probe = g_gather_probe;
	a.latestore = g_late_store;
	a.xcd_shift = g_xcd_shift;
	a.a = 1.0;
	a.b = 0.0;
	return a;
}

// `nsweeps` sweeps of one operator on iterate `x`.
// ASYNC: in place.  JACOBI_SYNC: ping-pong between x and `other`; first_in (optional) is read by the
// first sweep instead of x.  Returns the buffer holding the final iterate (x or other).
static double *run_sweeps(blasted_hip_prec p, SweepArgs a, Part part, Post post, DSrc dsrc, double *x,
                          double *other, const double *first_in, int nsweeps, int mode, int kind,
                          double *z1_of_last = nullptr)
{
	Phase ph(p, kind);
	if (nsweeps < 0 || mode == BLASTED_HIP_LEVEL) {
		// exact in-order passes (sequential variants and the level-scheduled types), kernels_level.hip.
		// A triangular pass reads only rows it has already written, so the initial content of x is never
		// used and repeating the pass changes nothing: one pass is enough.
		const bool triangular = part == PART_LOWER || part == PART_UPPER;
		if (!triangular)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "run_sweeps: exact passes of this operator go through relax_impl");
		const int passes = nsweeps < 0 ? 1 : (nsweeps > 1 ? 1 : nsweeps);
		for (int s = 0; s < passes; s++)
			ph.launches += exact_pass(p, a, part, post, dsrc, x, nullptr);
		ph.done();
		return x;
	}
	double *cur = x;
	for (int s = 0; s < nsweeps; s++) {
		const double *in = (s == 0 && first_in) ? first_in : cur;
		double *out;
		if (mode == BLASTED_HIP_ASYNC)
			out = x;
		else
			out = (s == 0 && first_in) ? x : (cur == x ? other : x);
		a.xin = in;
		a.xout = out;
		a.z1out = (s == nsweeps - 1) ? z1_of_last : nullptr;  // (scalar lower sweeps of a small application, see there)
		launch_sweep(a, part, post, dsrc, p->stream);
		ph.launches++;
		cur = out;
	}
	ph.done();
	return cur;
}

}  // namespace bhip

using namespace bhip;

extern "C" {

const char *blasted_hip_last_error(void)
{
	return g_last_error.c_str();
}

int blasted_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int blasted_hip_create(blasted_hip_prec *out, int device, void *stream, int own_stream)
{
	return guarded([&] {
		if (!out)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "null output handle");
		*out = nullptr;
		int n = 0;
		if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
			BHIP_FAIL(BLASTED_HIP_ENODEV, "no HIP device available: the MI355X backend has no CPU fallback");
		if (device < 0 || device >= n)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "device index out of range");
		BHIP_CHECK(hipSetDevice(device));
		auto *p = new blasted_hip_prec_s();
		p->device = device;
		if (own_stream) {
			BHIP_CHECK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
			p->own_stream = true;
		} else {
			p->stream = (hipStream_t)stream;  // NULL = the null stream
			p->own_stream = false;
		}
		*out = p;
	});
}

int blasted_hip_destroy(blasted_hip_prec p)
{
	return guarded([&] {
		if (!p)
			return;
		(void)hipSetDevice(p->device);
		(void)hipStreamSynchronize(p->stream);
		tl_owner = p;
		for (auto &r : p->timing.recs) {
			(void)hipEventDestroy(r.e0);
			(void)hipEventDestroy(r.e1);
		}
		for (auto e : p->timing.pool)
			(void)hipEventDestroy(e);
		dev_free(p->browptr_own);
		dev_free(p->bcolind_own);
		dev_free(p->diagind_own);
		dev_free(p->vals_own);
		dev_free(p->posptr);
		dev_free(p->lowerp);
		dev_free(p->upperp);
		dev_free(p->f1_dcol);
		dev_free(p->f1_chunks);
		dev_free(p->iluvals);
		dev_free(p->iluvals2);
		dev_free(p->finv);
		dev_free(p->fdiag);
		dev_free(p->scale);
		dev_free(p->ytemp);
		dev_free(p->yperm);
		dev_free(p->zperm);
		dev_free(p->zeros);
		dev_free(p->relax_vals);
		dev_free(p->dblocks);
		for (int i = 0; i < 3; i++) {
			dev_free(p->tmp[i]);
			dev_free(p->stage[i]);
		}
		dev_free(p->red);
		free_level_schedule(p->levels);
		free_level_schedule(p->natstore);
		for (auto *c : {&p->fac_nat, &p->fac_lvl, &p->mat_nat, &p->mat_lvl}) {
			dev_free(c->l);
			dev_free(c->u);
		}
		if (p->own_stream)
			(void)hipStreamDestroy(p->stream);
		forget_owner(p);
		tl_owner = nullptr;
		delete p;
	});
}

int blasted_hip_synchronize(blasted_hip_prec p)
{
	return guarded([&] {
		use_device(p);
		BHIP_CHECK(hipStreamSynchronize(p->stream));
	});
}

int blasted_hip_device_synchronize(int device)
{
	return guarded([&] {
		int n = 0;
		if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
			BHIP_FAIL(BLASTED_HIP_ENODEV, "no HIP device available: the MI355X backend has no CPU fallback");
		if (device < 0 || device >= n)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "device index out of range");
		BHIP_CHECK(hipSetDevice(device));
		BHIP_CHECK(hipDeviceSynchronize());
	});
}

int blasted_hip_set_pattern(blasted_hip_prec p, int nbrows, int nnzb, int bs, int layout,
                            const int *browptr, const int *bcolind, const int *diagind, int loc)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		if (p->have_pattern)
			BHIP_FAIL(BLASTED_HIP_ESTATE, "the sparsity pattern of an operator is set once");
		// (an empty subdomain may come with null bcolind / diagind: there is nothing behind them)
		if (nbrows < 0 || nnzb < 0 || !browptr || (nnzb > 0 && !bcolind) || (nbrows > 0 && !diagind))
			BHIP_FAIL(BLASTED_HIP_EINVAL, "set_pattern: null array or negative size");
		if (layout != BLASTED_HIP_COLMAJOR && layout != BLASTED_HIP_ROWMAJOR)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "Block ordering must be either rowmajor or colmajor!");
		if (!sweep_supported(bs))
			BHIP_FAIL(BLASTED_HIP_ENOTIMPL, "Block size " + std::to_string(bs) + " not supported");
		if ((double)nnzb * bs * bs >= 2147483647.0 * 16)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "matrix too large");
		Pattern pat;
		pat.nbrows = nbrows;
		pat.nnzb = nnzb;
		pat.bs = bs;
		pat.rowmajor = (layout == BLASTED_HIP_ROWMAJOR) ? 1 : 0;
		if (loc == BLASTED_HIP_DEVICE) {
			pat.browptr = browptr;
			pat.bcolind = bcolind;
			pat.diagind = diagind;
		} else {
			p->browptr_own = dev_alloc<int>((size_t)nbrows + 1);
			p->bcolind_own = dev_alloc<int>((size_t)nnzb);
			p->diagind_own = dev_alloc<int>((size_t)nbrows);
			BHIP_CHECK(hipMemcpyAsync(p->browptr_own, browptr, sizeof(int) * ((size_t)nbrows + 1),
			                          hipMemcpyHostToDevice, p->stream));
			BHIP_CHECK(hipMemcpyAsync(p->bcolind_own, bcolind, sizeof(int) * (size_t)nnzb,
			                          hipMemcpyHostToDevice, p->stream));
			BHIP_CHECK(hipMemcpyAsync(p->diagind_own, diagind, sizeof(int) * (size_t)nbrows,
			                          hipMemcpyHostToDevice, p->stream));
			pat.browptr = p->browptr_own;
			pat.bcolind = p->bcolind_own;
			pat.diagind = p->diagind_own;
		}
		const int flags = validate_pattern_device(pat, p->stream, &pat.max_row_len);
		if (flags) {
			std::string msg = "set_pattern: invalid sparse-row pattern:";
			if (flags & 8) msg += " [browptr not monotone / wrong nnzb]";
			if (flags & 4) msg += " [column index out of range]";
			if (flags & 1) msg += " [block columns not strictly ascending inside a row]";
			if (flags & 2) msg += " [diagind does not address the diagonal block of its row]";
			dev_free(p->browptr_own);
			dev_free(p->bcolind_own);
			dev_free(p->diagind_own);
			p->browptr_own = p->bcolind_own = p->diagind_own = nullptr;
			BHIP_FAIL(BLASTED_HIP_EINVAL, msg);
		}
		if (!sweep_offsets_fit(pat)) {
			dev_free(p->browptr_own);
			dev_free(p->bcolind_own);
			dev_free(p->diagind_own);
			p->browptr_own = p->bcolind_own = p->diagind_own = nullptr;
			BHIP_FAIL(BLASTED_HIP_EINVAL, "set_pattern: matrix exceeds this build's 32-bit in-chunk offsets "
			                              "(vector must be < 4 GiB, 256 * longest row * block bytes < 4 GiB)");
		}
		p->pat = pat;
		p->have_pattern = true;
	});
}

int blasted_hip_set_values(blasted_hip_prec p, const double *vals, int loc)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		need_pattern(p);
		if (!vals && p->nvals() > 0)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "set_values: null array");
		if (loc == BLASTED_HIP_DEVICE && vals) {
			p->vals = vals;
		} else {
			if (!p->vals_own)
				p->vals_own = dev_alloc<double>((size_t)p->nvals());
			BHIP_CHECK(hipMemcpyAsync(p->vals_own, vals, sizeof(double) * (size_t)p->nvals(),
			                          hipMemcpyHostToDevice, p->stream));
			BHIP_CHECK(hipStreamSynchronize(p->stream));
			p->vals = p->vals_own;
		}
		p->mat_nat.invalidate();
		p->mat_lvl.invalidate();
		p->mat_applies = 0;
		p->relax_vals_valid = false;
		p->relax_passes = 0;
	});
}

/* ---- ILU(0) ------------------------------------------------------------------------------ */

int blasted_hip_ilu0_positions(blasted_hip_prec p)
{
	return guarded([&] {
		use_device(p);
		need_pattern(p);
		if (p->npairs >= 0)
			return;  // once per pattern, src/solverops_ilu0.cpp:193-196
		Phase ph(p, 2);
		p->npairs = run_ilu_positions(p->pat, &p->posptr, &p->lowerp, &p->upperp, p->stream);
		ph.launches = 3;
		ph.done();
	});
}

int blasted_hip_ilu0_positions_size(blasted_hip_prec p, long *npairs)
{
	return guarded([&] {
		use_device(p);
		if (p->npairs < 0 || !npairs)
			BHIP_FAIL(BLASTED_HIP_ESTATE, "ilu0_positions has not been computed");
		*npairs = p->npairs;
	});
}

int blasted_hip_ilu0_get_positions(blasted_hip_prec p, int *posptr, int *lowerp, int *upperp)
{
	return guarded([&] {
		use_device(p);
		if (p->npairs < 0)
			BHIP_FAIL(BLASTED_HIP_ESTATE, "ilu0_positions has not been computed");
		BHIP_CHECK(hipMemcpyAsync(posptr, p->posptr, sizeof(int) * ((size_t)p->pat.nnzb + 1),
		                          hipMemcpyDeviceToHost, p->stream));
		if (p->npairs > 0) {
			BHIP_CHECK(hipMemcpyAsync(lowerp, p->lowerp, sizeof(int) * (size_t)p->npairs,
			                          hipMemcpyDeviceToHost, p->stream));
			BHIP_CHECK(hipMemcpyAsync(upperp, p->upperp, sizeof(int) * (size_t)p->npairs,
			                          hipMemcpyDeviceToHost, p->stream));
		}
		BHIP_CHECK(hipStreamSynchronize(p->stream));
	});
}

int blasted_hip_ilu0_factorize(blasted_hip_prec p, int nbuildsweeps, int fact_init, int use_scaling,
                               int mode, double *precinfo)
{
	return guarded([&] {
		use_device(p);
		need_values(p);
		check_mode(mode);
		const bool deterministic = mode == BLASTED_HIP_DETERMINISTIC;
		if (deterministic)
			mode = BLASTED_HIP_JACOBI_SYNC;  // a fixed operator: synchronous sweeps (SGS apply: after an exact forward half)
		(void)deterministic;
		p->fac_lvl.invalidate();
		p->fac_nat.invalidate();
		p->fac_applies = 0;
		p->fdiag_valid = false;
		if (mode == BLASTED_HIP_LEVEL)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "ilu0_factorize: mode LEVEL applies to the apply / relaxation entry "
			                              "points; the exact factorisation is nbuildsweeps < 0");
		if (fact_init < BLASTED_HIP_INIT_F_ZERO || fact_init > BLASTED_HIP_INIT_F_NONE)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "Factor initialization not recongnized!");
		const Pattern &pat = p->pat;
		const long nv = p->nvals();

		// first-time setup, src/solverops_ilu0.cpp:150-183,190-196
		if (p->npairs < 0)
			p->npairs = run_ilu_positions(pat, &p->posptr, &p->lowerp, &p->upperp, p->stream);
		if (!p->iluvals) {
			p->iluvals = dev_alloc<double>((size_t)nv);
			BHIP_CHECK(hipMemcpyAsync(p->iluvals, p->vals, sizeof(double) * (size_t)nv,
			                          hipMemcpyDeviceToDevice, p->stream));
		}
		if (!p->ytemp) {
			p->ytemp = dev_alloc<double>((size_t)p->n());
			BHIP_CHECK(hipMemsetAsync(p->ytemp, 0, sizeof(double) * (size_t)p->n(), p->stream));
		}
		if (use_scaling) {
			ensure(p->scale, p->n());
			launch_scaling_vector(pat, p->vals, p->scale, p->stream);
		}
		p->scaled = use_scaling != 0;
		const double *scale = use_scaling ? p->scale : nullptr;

		// The exact in-order factorisation writes every entry before anything reads it (an entry needs earlier
		// entries of its own row and rows of earlier levels only), so its result does not depend on the initial
		// guess: the 15 GB initialisation pass (4.4 ms of 29.6 at 256^3 bs=4) is skipped unless the initial
		// remainder is asked for.
		// Asynchronous sweeps from INIT_F_ORIGINAL, no scaling (round 3): the initialisation pass -- a copy of the matrix
		// into the factor, 4.4 ms of a 28.5 ms three-sweep build at 256^3 bs=4 -- is fused into the first sweep, which reads
		// its iterate from the MATRIX (in = A, out = factor: every read sees the initial guess F0 = A, a valid schedule of
		// the chaotic iteration, and every entry is written, the pair-less upper blocks with their final value a_ij); the
		// sweeps after it run in place as before (same convergence per sweep as with the separate pass, measured on
		// Poisson and unstructured patterns at bs 3, 4, 5, 8: profiles/r03_factor_fuse_ab.txt).  The row being computed reads its OWN finished lower blocks back from
		// the factor, as an in-place sweep does (FactorArgs::lrow_fresh): without that the diagonal block of the first
		// sweep is built from the raw a_ik instead of l_ik and the build falls two sweeps behind (measured).  Scalar matrices take part through the plan kernel
		// (kernels_factor1.hip), which reads none of the chunk's old factor values anyway; not when the initial remainder is
		// asked for (it is evaluated on the initialised factor).
		const bool fuse_init = g_factor_fuse_init && fact_init == BLASTED_HIP_INIT_F_ORIGINAL && !use_scaling &&
		                       mode == BLASTED_HIP_ASYNC && nbuildsweeps >= 1 && !precinfo &&
		                       (pat.bs >= 2 || g_factor1_plan);  // (scalar: on the plan kernel, which reads no old factor value)
		const bool needs_init = !(nbuildsweeps < 0 && !precinfo) && !fuse_init;
		if (needs_init) {
			double *dscratch = nullptr;
			if (fact_init == BLASTED_HIP_INIT_F_SGS)
				dscratch = ensure(p->dblocks, (long)pat.nbrows * pat.bs * pat.bs);
			launch_fact_init(pat, p->vals, scale, fact_init, p->iluvals, dscratch, p->stream);
			if (fact_init == BLASTED_HIP_INIT_F_SGS)
				p->jacobi_done = false;  // dblocks was used as scratch
		}

		FactorArgs fa;
		fa.pat = pat;
		fa.avals = p->vals;
		fa.scale = scale;
		fa.posptr = p->posptr;
		fa.lowerp = p->lowerp;
		fa.upperp = p->upperp;
		fa.rows = nullptr;
		fa.nrows = 0;
		fa.diag_inverted = 0;
		fa.skip_fixed = 0;
		fa.dinv_scratch = nullptr;
		if (pat.bs == 1 && nbuildsweeps > 0 && mode != BLASTED_HIP_JACOBI_SYNC && !scale && g_factor1_plan) {  // (before the sweeps below)
			// once per pattern: what the scalar in-place sweep kernel reads instead of column indices
			if (!p->f1_dcol) {
				p->f1_dcol = dev_alloc<int>((size_t)pat.nnzb);
				p->f1_chunks = dev_alloc<int4>((size_t)factor1_plan_chunks(pat.nbrows));
				build_factor1_plan(pat, p->posptr, p->f1_dcol, p->f1_chunks, p->stream);
			}
			fa.f1_dcol = p->f1_dcol;
			fa.f1_chunks = p->f1_chunks;
		}
		if (pat.bs >= 5)
			fa.dinv_scratch = ensure(p->finv, (long)pat.nbrows * pat.bs * pat.bs);

		const long ngroups = (long)pat.nbrows + 8;
		if (precinfo) {
			for (int i = 0; i < 6; i++)
				precinfo[i] = 0;
			ensure(p->red, ngroups * 4);
			fa.in = p->iluvals;
			fa.out = nullptr;
			precinfo[1] = run_nonlinear_res(fa, p->red, p->stream);
		}

		// async_bilu0_sweeps, src/async_blockilu_factor.cpp:186-204
		double *cur = p->iluvals;
		if (nbuildsweeps < 0) {
			// sequential factorisation (seqilu0 / sfilu0): the exact ILU(0) of one in-order pass, run as
			// one launch per dependency level (kernels_factor.hip: launch_factor_levels)
			Phase ph(p, 0);
			fa.in = p->iluvals;
			fa.out = p->iluvals;
			// a diagonal block is final once its row is done: store it inverted right away (no inverse per
			// dependent lower block, no inversion pass at the end) -- unless the remainder, which is defined
			// on the un-inverted factor, is asked for
			fa.diag_inverted = (pat.bs > 1 && !precinfo) ? 1 : 0;
			LevelSchedule &ls = need_levels(p);
			// one launch whose rows wait for their predecessors, so that levels overlap ("level=launch" keeps one
			// launch per level, as for the exact solves)
			const int sf = (g_level_impl == 0 && ls.sf_aborts < SF_ABORT_LIMIT) ? launch_factor_syncfree(fa, ls, p->stream) : 0;
			if (sf != 0)
				ls.sf_launches++;
			if (sf == 1)
				ph.launches = 2;
			else {
				if (sf < 0)
					ls.sf_aborts++;  // a wave gave up waiting: redone with one launch per level
				ph.launches = launch_factor_levels(fa, ls, p->stream);
			}
			ph.done();
		} else {
			Phase ph(p, 0);
			for (int s = 0; s < nbuildsweeps; s++) {
				double *out = cur;
				if (mode == BLASTED_HIP_JACOBI_SYNC) {
					if (!p->iluvals2)
						p->iluvals2 = dev_alloc<double>((size_t)nv);
					out = (cur == p->iluvals) ? p->iluvals2 : p->iluvals;
				}
				fa.in = cur;
				fa.out = out;
				// (from the first sweep on after INIT_F_ORIGINAL: the initialisation pass has stored exactly those values)
				fa.skip_fixed = ((s > 0 || fact_init == BLASTED_HIP_INIT_F_ORIGINAL) && out == cur && g_factor_skip_fixed) ? 1 : 0;
				fa.lrow_fresh = 0;
				if (fuse_init && s == 0) {
					fa.in = p->vals;  // the initial guess itself; every entry of the factor is written
					fa.skip_fixed = 0;
					fa.lrow_fresh = 1;  // ... and a row's own lower blocks are read back fresh, as in place
					// (bs = 4, tried and removed: the pair-less upper blocks copied ahead of the block loop, a row's all at
					// once, the loop then skipping them -- 256^3 one-sweep build 10.1 -> 12.1 ms, slower)
				}
				launch_factor_sweep(fa, p->stream);
				ph.launches++;
				cur = out;
			}
			fa.lrow_fresh = 0;
			ph.done();
		}
		if (cur != p->iluvals)
			BHIP_CHECK(hipMemcpyAsync(p->iluvals, cur, sizeof(double) * (size_t)nv,
			                          hipMemcpyDeviceToDevice, p->stream));

		if (precinfo) {
			fa.in = p->iluvals;
			fa.out = nullptr;
			precinfo[0] = run_nonlinear_res(fa, p->red, p->stream);
			double dd[4];
			run_diag_dominance(pat, p->iluvals, p->red, dd, p->stream);
			precinfo[5] = dd[0];
			precinfo[4] = dd[1];
			precinfo[3] = dd[2];
			precinfo[2] = dd[3];
		}

		// block version only: invert diagonal blocks in place, src/async_blockilu_factor.cpp:143-146
		if (pat.bs > 1 && !fa.diag_inverted) {
			Phase ph(p, 2);
			launch_invert_diag_blocks(pat, p->iluvals, 1, p->iluvals, 1, p->stream);
			ph.launches = 1;
			ph.done();
		}
		p->factored = true;
	});
}

int blasted_hip_ilu0_apply(blasted_hip_prec p, const double *r, double *z, int napplysweeps,
                           int apply_init, int mode, int loc)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		check_mode(mode);
		const bool deterministic = mode == BLASTED_HIP_DETERMINISTIC;
		if (deterministic)
			mode = BLASTED_HIP_JACOBI_SYNC;  // a fixed operator: synchronous sweeps (SGS apply: after an exact forward half)
		(void)deterministic;
		need_pattern(p);
		if (!p->factored)
			BHIP_FAIL(BLASTED_HIP_ESTATE, "ilu0_apply before ilu0_factorize");
		const bool keep_iterates = g_apply_allow_none && apply_init == BLASTED_HIP_INIT_A_NONE && mode == BLASTED_HIP_ASYNC &&
		                           loc == BLASTED_HIP_DEVICE && napplysweeps > 0;
		if (apply_init != BLASTED_HIP_INIT_A_ZERO && apply_init != BLASTED_HIP_INIT_A_JACOBI && !keep_iterates)
			BHIP_FAIL(BLASTED_HIP_EINVAL, " scalar_ilu0_apply: Invalid init type!");
		if (!r || !z)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "ilu0_apply: null vector");
		if (napplysweeps < 0)
			mode = BLASTED_HIP_LEVEL;  // sequential variant: one exact in-order pass
		const long n = p->n();
		const size_t nbytes = sizeof(double) * (size_t)n;
		const double *dr = in_vec(p, r, loc, 0);
		double *dz = out_vec(p, z, loc, 1);
		const bool scalar = p->pat.bs == 1;
		const bool jac = mode == BLASTED_HIP_JACOBI_SYNC;
		if (keep_iterates)
			restore_ytemp(p);  // (an exact apply may have left y level-ordered)
		p->y_in_level_order = false;  // this call rewrites ytemp

		// Synchronous sweeps start from a known iterate, so their first sweep needs no matrix: from y0 = 0 the
		// lower sweep gives y1 = S r exactly, and from z0 = 0 the upper one gives z1 = D^-1 y, a product with
		// the diagonal blocks alone (bit-identical: the skipped products are all with zeros).  3+3 synchronous
		// sweeps then cost two lower sweeps, one diagonal pass and two upper sweeps.
		const bool skip_first = jac && napplysweeps >= 1;

		// y := 0 (both init types), src/solverops_ilu0.cpp:83-94.  The prologue z := S r is fused: the
		// lower sweeps read r (times scale) directly as their right-hand side.
		// A small (cache-resident) problem is launch-bound -- 64^3 scalar: eight launches 5.7 us apart for 46 us of
		// kernels -- so its two fills go: the first sweep of each triangle READS an operator-owned vector of zeros that
		// nobody ever writes (in = zeros, out = the iterate), the sweeps after it run in place.  With every row of such a
		// problem in flight at once the first in-place sweep from a zeroed iterate was a Jacobi sweep from zero anyway.
		// (Round 4; the whole application as ONE cooperative launch with grid barriers was built, verified and removed: its
		// iterate needs agent-scope accesses, which go past the L2s -- 270 us per application against 47:
		// tools/probes/removed/kernels_fused1.hip.txt.)  tuning "smallapply=0": the fills as before.
		const bool small_apply = g_small_apply && mode == BLASTED_HIP_ASYNC && napplysweeps > 0 && !keep_iterates && n > 0 &&
		                         (long)p->pat.nnzb * (8L * p->pat.bs * p->pat.bs + 4) + n * 32 < (96L << 20);
		if (small_apply && !p->zeros) {
			p->zeros = dev_alloc<double>((size_t)n);
			BHIP_CHECK(hipMemsetAsync(p->zeros, 0, nbytes, p->stream));
		}
		if (!skip_first && !keep_iterates && !small_apply) {
			Phase ph(p, 2);
			BHIP_CHECK(hipMemsetAsync(p->ytemp, 0, nbytes, p->stream));
			ph.launches = 1;
			ph.done();
		}
		SweepArgs a = base_args(p);
		a.vals = p->iluvals;
		a.rhs = dr;
		a.rscale = p->scaled ? p->scale : nullptr;
		a.descending = 0;
		const bool compact = g_compact && mode != BLASTED_HIP_LEVEL && napplysweeps > 0 && compact_now(p, p->fac_applies, p->fac_nat);
		// (see g_place_after: the quick search waits until the operator has shown that it lives long enough to pay for it)
		const bool want_placed = compact && g_placement && (g_placement >= 2 || p->ilu_apps_life >= g_place_after);
		double *old_copies[4] = {nullptr, nullptr, nullptr, nullptr};
		if (compact) {
			p->ilu_apps_life++;
			if (want_placed && !p->fac_placed && (p->fac_nat.l || p->fac_nat.u || p->fac_lvl.l || p->fac_lvl.u)) {
				// plain copies exist.  The quick search first LOOKS at them: a triangle whose pieces all meet its must stays
				// where it is (one process in three, the plain allocation happened to lie better than what a 32 GiB search
				// then found).  For the others new, placed copies are made beside the old ones (a release first would make
				// the allocation wait for the driver's wipe), the old storage goes when both triangles have been copied.
				const size_t bs2b = sizeof(double) * (size_t)p->pat.bs * p->pat.bs;
				const LevelSchedule &ns = p->natstore;
				bool keep[2] = {false, false};
				if (g_placement == 1 && ns.storage_built) {
					double *sink = dev_alloc<double>(1);
					for (int t = 0; t < 2; t++) {
						double *buf = t ? (p->fac_nat.u ? p->fac_nat.u : p->fac_lvl.u) : (p->fac_nat.l ? p->fac_nat.l : p->fac_lvl.l);
						const size_t bytes = (size_t)(t ? ns.nnz_dupper : ns.nnz_lower) * bs2b;
						if (!buf || bytes < ((size_t)64 << 20))
							continue;
						PlaceHint h;
						h.avoid = t ? static_cast<const void *>(dz) : static_cast<const void *>(p->ytemp);
						h.ref_bytes = nbytes;
						bool ok = true;
						for (size_t at = 0; ok && at + ((size_t)64 << 20) <= bytes; at += (size_t)1 << 30) {
							const size_t len = bytes - at < ((size_t)1 << 30) ? bytes - at : ((size_t)1 << 30);
							ok = class_fits(reinterpret_cast<char *>(buf) + at, len, h, sink, p->stream) >= 0;
						}
						keep[t] = ok;
						if (trace_placement())
							std::fprintf(stderr, "[blasted_hip] the plain %s copy %s its must\n", t ? "upper" : "lower", ok ? "meets" : "does not meet");
					}
					BHIP_CHECK(hipStreamSynchronize(p->stream));
					dev_free(sink);
				}
				if (!keep[0]) {
					old_copies[0] = p->fac_nat.l;
					old_copies[2] = p->fac_lvl.l;
					p->fac_nat.l = p->fac_lvl.l = nullptr;
					p->fac_nat.valid_l = p->fac_lvl.valid_l = false;
				}
				if (!keep[1]) {
					old_copies[1] = p->fac_nat.u;
					old_copies[3] = p->fac_lvl.u;
					p->fac_nat.u = p->fac_lvl.u = nullptr;
					p->fac_nat.valid_u = p->fac_lvl.valid_u = false;
				}
			}
			// Musts (10 % each when they go wrong): the lower copy not in ytemp's class, the upper copy not in z's -- two of
			// three classes will do, a search that holds back little finds them: the default.  Thorough ("placement=2") adds
			// the preferences: ytemp moved out of r's and z's classes first (place_ytemp), then the lower copy preferably in
			// r's class and the upper copy preferably in ytemp's; pieces that only meet the must are second best.
			if (g_placement >= 2 && !p->fac_nat.l)
				place_ytemp(p, dr, dz);
			if (want_placed)
				set_place_hint(p, p->ytemp, g_placement >= 2 ? dr : nullptr);
			compact_args(p, false, a, p->iluvals, p->fac_nat);
			set_place_hint(p, nullptr, nullptr);
		}
		// Small scalar applications, one launch less again: the first upper sweep of a small application reads the zeros
		// vector, i.e. it is z1 = D^-1 y and nothing else -- the LAST lower sweep stores it beside y_i (same arithmetic:
		// (1 / u_ii) * y_i, u_ii from the contiguous copy of the factor's diagonal), and the upper sweeps start in place
		// from z1.  64^3, 3+3 sweeps: 5 -> 4 launches.
		const bool fuse_z1 = small_apply && g_small_apply >= 2 && scalar && apply_init == BLASTED_HIP_INIT_A_ZERO &&
		                     p->pat.nbrows < (1 << 20);
		if (fuse_z1) {
			if (!p->fdiag_valid) {
				ensure(p->fdiag, (long)p->pat.nbrows);
				launch_gather_diag_blocks(p->pat, p->iluvals, p->fdiag, p->stream);
				p->fdiag_valid = true;
			}
			a.dvals = p->fdiag;
		}
		double *yother = jac ? ensure(p->tmp[0], n) : nullptr;
		double *y;
		if (skip_first) {
			const double *y1 = dr;  // y1 = S r
			if (p->scaled) {
				Phase ph(p, 2);
				BHIP_CHECK(hipMemcpyAsync(yother, dr, nbytes, hipMemcpyDeviceToDevice, p->stream));
				launch_scale_vec(yother, p->scale, n, p->stream);
				ph.launches = 2;
				ph.done();
				y1 = yother;
			}
			if (napplysweeps == 1) {
				BHIP_CHECK(hipMemcpyAsync(p->ytemp, y1, nbytes, hipMemcpyDeviceToDevice, p->stream));
				y = p->ytemp;
			} else
				y = run_sweeps(p, a, PART_LOWER, POST_SUB, D_NONE, p->ytemp, yother, y1, napplysweeps - 1, mode, 0);
		} else if (small_apply && !p->scaled && napplysweeps >= 2)
			// (from y0 = 0 the first lower sweep gives y1 = r exactly -- the skipped products are with zeros -- so the
			// second one can read r as its iterate: one launch less)
			y = run_sweeps(p, a, PART_LOWER, POST_SUB, D_NONE, p->ytemp, yother, dr, napplysweeps - 1, mode, 0, fuse_z1 ? dz : nullptr);
		else
			y = run_sweeps(p, a, PART_LOWER, POST_SUB, D_NONE, p->ytemp, yother, small_apply ? p->zeros : nullptr, napplysweeps, mode, 0,
			               fuse_z1 ? dz : nullptr);
		double *yfree = (y == p->ytemp) ? yother : p->ytemp;  // Jacobi mode: the non-final y buffer

		// z := y or z := 0, then upper sweeps, src/solverops_ilu0.cpp:110-141
		a = base_args(p);
		a.vals = p->iluvals;
		a.dvals = p->iluvals;
		a.rhs = y;
		a.descending = 1;
		if (compact) {
			if (want_placed)
				set_place_hint(p, dz, g_placement >= 2 ? p->ytemp : nullptr);
			const bool fresh = !p->fac_nat.u;
			compact_args(p, true, a, p->iluvals, p->fac_nat);
			set_place_hint(p, nullptr, nullptr);
			if (want_placed)
				p->fac_placed = true;
			for (double *o : old_copies)
				dev_free(o);
			if (fresh && want_placed)
				report_classes(p, dr, dz);
		}
		const DSrc dsrc = scalar ? D_RECIP_DIAG : D_VALS_DIAG;
		double *zfinal = dz;
		if (napplysweeps == 0) {
			Phase ph(p, 2);
			if (apply_init == BLASTED_HIP_INIT_A_JACOBI)
				BHIP_CHECK(hipMemcpyAsync(dz, y, nbytes, hipMemcpyDeviceToDevice, p->stream));
			else
				BHIP_CHECK(hipMemsetAsync(dz, 0, nbytes, p->stream));
			ph.launches = 1;
			ph.done();
		} else if (!jac) {
			const double *first_in = nullptr;
			if (apply_init == BLASTED_HIP_INIT_A_JACOBI)
				first_in = y;  // z0 = y: the first sweep gathers from y, no copy needed
			else if (small_apply)
				first_in = p->zeros;  // z0 = 0 without a fill (see above)
			else if (!keep_iterates) {
				Phase ph(p, 2);
				BHIP_CHECK(hipMemsetAsync(dz, 0, nbytes, p->stream));
				ph.launches = 1;
				ph.done();
			}
			if (fuse_z1)  // dz holds z1 already: the remaining sweeps run in place
				run_sweeps(p, a, PART_UPPER, POST_D_SUB, dsrc, dz, nullptr, nullptr, napplysweeps - 1, mode, 1);
			else
				run_sweeps(p, a, PART_UPPER, POST_D_SUB, dsrc, dz, nullptr, first_in, napplysweeps, mode, 1);
		} else if (apply_init == BLASTED_HIP_INIT_A_ZERO && !scalar) {
			// first sweep from z0 = 0: z1 = D^-1 y through the contiguous copy of the factor's diagonal blocks
			if (!p->fdiag_valid) {
				ensure(p->fdiag, (long)p->pat.nbrows * p->pat.bs * p->pat.bs);
				launch_gather_diag_blocks(p->pat, p->iluvals, p->fdiag, p->stream);
				p->fdiag_valid = true;
			}
			SweepArgs d = base_args(p);
			d.vals = p->iluvals;
			d.dvals = p->fdiag;
			d.rhs = y;
			run_sweeps(p, d, PART_NONE, POST_D_SUB, D_DBLOCKS, dz, nullptr, y, 1, BLASTED_HIP_ASYNC, 1);
			if (napplysweeps > 1) {
				double *zo = ensure(p->tmp[1], n);  // sweep 2: dz -> zo ; sweep 3: zo -> dz ; ...
				zfinal = run_sweeps(p, a, PART_UPPER, POST_D_SUB, dsrc, dz, zo, nullptr, napplysweeps - 1, mode, 1);
				if (zfinal != dz) {
					BHIP_CHECK(hipMemcpyAsync(dz, zfinal, nbytes, hipMemcpyDeviceToDevice, p->stream));
					zfinal = dz;
				}
			}
		} else {
			const double *first_in = y;
			if (apply_init == BLASTED_HIP_INIT_A_ZERO) {
				BHIP_CHECK(hipMemsetAsync(yfree, 0, nbytes, p->stream));
				first_in = yfree;
			}
			// sweep 1: first_in -> dz ; sweep 2: dz -> zo ; ...  (zo must differ from y and first_in)
			double *zo = ensure(p->tmp[1], n);
			zfinal = run_sweeps(p, a, PART_UPPER, POST_D_SUB, dsrc, dz, zo, first_in, napplysweeps, mode, 1);
			if (zfinal != dz) {
				BHIP_CHECK(hipMemcpyAsync(dz, zfinal, nbytes, hipMemcpyDeviceToDevice, p->stream));
				zfinal = dz;
			}
		}
		if (y != p->ytemp)  // keep the operator's ytemp = L^-1 r as the reference leaves it
			BHIP_CHECK(hipMemcpyAsync(p->ytemp, y, nbytes, hipMemcpyDeviceToDevice, p->stream));
		if (p->scaled) {  // z := S z, src/solverops_ilu0.cpp:143-147
			Phase ph(p, 2);
			launch_scale_vec(dz, p->scale, n, p->stream);
			ph.launches = 1;
			ph.done();
		}
		finish_out(p, z, dz, loc);
	});
}

/* ---- Jacobi / SGS ------------------------------------------------------------------------ */

int blasted_hip_jacobi_compute(blasted_hip_prec p)
{
	return guarded([&] {
		use_device(p);
		need_values(p);
		ensure(p->dblocks, (long)p->pat.nbrows * p->pat.bs * p->pat.bs);
		Phase ph(p, 2);
		launch_invert_diag_blocks(p->pat, p->vals, 1, p->dblocks, 0, p->stream);
		ph.launches = 1;
		ph.done();
		p->jacobi_done = true;
		p->mat_nat.invalidate();  // compute(): the borrowed values may have changed in place
		p->mat_lvl.invalidate();
		p->mat_applies = 0;
		p->relax_vals_valid = false;
		p->relax_passes = 0;
		if (!p->ytemp) {  // AsyncBlockSGS::compute, src/solverops_sgs.cpp:33-45
			p->ytemp = dev_alloc<double>((size_t)p->n());
			BHIP_CHECK(hipMemsetAsync(p->ytemp, 0, sizeof(double) * (size_t)p->n(), p->stream));
		}
	});
}

static void need_jacobi(blasted_hip_prec p)
{
	need_values(p);
	if (!p->jacobi_done)
		BHIP_FAIL(BLASTED_HIP_ESTATE, "jacobi_compute has not been called (or its dblocks were reused)");
}

int blasted_hip_jacobi_apply(blasted_hip_prec p, const double *r, double *z, int loc)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		need_jacobi(p);
		if (!r || !z)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "null vector");
		const double *dr = in_vec(p, r, loc, 0);
		double *dz = out_vec(p, z, loc, 1);
		SweepArgs a = base_args(p);
		a.vals = p->vals;
		a.dvals = p->dblocks;
		a.rhs = dr;
		run_sweeps(p, a, PART_NONE, POST_D_SUB, D_DBLOCKS, dz, nullptr, dr, 1, BLASTED_HIP_ASYNC, 0);
		finish_out(p, z, dz, loc);
	});
}

int blasted_hip_sgs_apply(blasted_hip_prec p, const double *r, double *z, int napplysweeps,
                          int apply_init, int mode, int loc)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		check_mode(mode);
		const bool deterministic = mode == BLASTED_HIP_DETERMINISTIC;
		if (deterministic)
			mode = BLASTED_HIP_JACOBI_SYNC;  // a fixed operator: synchronous sweeps (SGS apply: after an exact forward half)
		(void)deterministic;
		need_jacobi(p);
		if (!r || !z)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "sgs_apply: null vector");
		if (napplysweeps < 0)
			mode = BLASTED_HIP_LEVEL;
		if (apply_init < BLASTED_HIP_INIT_A_ZERO || apply_init > BLASTED_HIP_INIT_A_NONE)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "Apply initialization not recongnized!");
		const long n = p->n();
		const size_t nbytes = sizeof(double) * (size_t)n;
		const bool jac = mode == BLASTED_HIP_JACOBI_SYNC;
		const double *dr = in_vec(p, r, loc, 0);
		double *dz;
		if (loc == BLASTED_HIP_DEVICE)
			dz = z;
		else {
			dz = ensure(p->stage[1], n);
			if (apply_init == BLASTED_HIP_INIT_A_NONE)  // z is the initial guess of the backward sweeps
				BHIP_CHECK(hipMemcpyAsync(dz, z, nbytes, hipMemcpyHostToDevice, p->stream));
		}
		const bool reinit = apply_init == BLASTED_HIP_INIT_A_JACOBI || apply_init == BLASTED_HIP_INIT_A_ZERO;
		p->y_in_level_order = false;  // this call rewrites ytemp
		// The reference's forward loop stands OUTSIDE any parallel region (src/solverops_sgs.cpp:62-66 calls
		// perform_block_fgs, whose `omp for` is then orphaned, src/kernels/kernels_sgs.hpp:127): whatever the
		// thread and sweep counts, y = (D+L)^-1 r is the exact serial solve and only the backward sweeps are
		// asynchronous.  The product mode does the same: ONE exact in-order forward pass (repeating it changes
		// nothing, and it does not read the initial ytemp), then napplysweeps asynchronous backward sweeps.
		// "sgsfwd=async" restores napplysweeps asynchronous forward sweeps (round 1's form, a tuning variant).
		if (((mode == BLASTED_HIP_ASYNC && g_sgs_exact_fwd) || deterministic) && napplysweeps >= 1 && p->pat.nbrows > 0) {
			SweepArgs f = base_args(p);
			f.vals = p->vals;
			f.dvals = p->dblocks;
			f.rhs = dr;
			f.descending = 0;
			{
				Phase ph(p, 0);
				p->y_natural_too = true;  // the backward sweeps read y by row
				ph.launches = exact_pass(p, f, PART_LOWER, POST_D_SUB, D_DBLOCKS, p->ytemp, nullptr);
				p->y_natural_too = false;
				restore_ytemp(p);
				ph.done();
			}
			SweepArgs a = base_args(p);
			a.vals = p->vals;
			a.dvals = p->dblocks;
			a.rhs = p->ytemp;
			a.descending = 1;
			if (g_compact && compact_now(p, p->mat_applies, p->mat_nat))
				compact_args(p, true, a, p->vals, p->mat_nat);
			const double *first_in = nullptr;
			int nbwd = napplysweeps;
			if (apply_init == BLASTED_HIP_INIT_A_JACOBI)
				first_in = p->ytemp;  // z0 = y: the first sweep gathers from y
			else if (apply_init == BLASTED_HIP_INIT_A_ZERO) {
				if (jac) {
					// a synchronous backward sweep from z0 = 0 is z1 = y (the skipped products are with zeros):
					// continue as from z0 = y with one sweep less
					first_in = p->ytemp;
					nbwd = napplysweeps - 1;
				} else {
					Phase ph(p, 2);
					BHIP_CHECK(hipMemsetAsync(dz, 0, nbytes, p->stream));
					ph.launches = 1;
					ph.done();
				}
			} else if (jac) {
				// INIT_A_NONE: synchronous sweeps need the initial z in a buffer sweep 1 does not write
				double *z0 = ensure(p->tmp[0], n);
				BHIP_CHECK(hipMemcpyAsync(z0, dz, nbytes, hipMemcpyDeviceToDevice, p->stream));
				first_in = z0;
			}
			if (nbwd == 0)
				BHIP_CHECK(hipMemcpyAsync(dz, p->ytemp, nbytes, hipMemcpyDeviceToDevice, p->stream));
			else {
				double *zo = jac ? ensure(p->tmp[1], n) : nullptr;
				double *zfinal = run_sweeps(p, a, PART_UPPER, POST_SUB_D, D_DBLOCKS, dz, zo, first_in, nbwd, mode, 1);
				if (zfinal != dz)
					BHIP_CHECK(hipMemcpyAsync(dz, zfinal, nbytes, hipMemcpyDeviceToDevice, p->stream));
			}
			finish_out(p, z, dz, loc);
			return;
		}
		// Synchronous sweeps from a zero iterate need no matrix for their first sweep (as in ilu0_apply): the
		// forward one is y1 = D^-1 r, the backward one z1 = y -- bit-identical, the skipped products are with zeros.
		const bool skip_fwd = jac && reinit && napplysweeps >= 1;
		const bool skip_bwd = jac && apply_init == BLASTED_HIP_INIT_A_ZERO && napplysweeps >= 1;
		if (reinit && !skip_fwd) {  // src/solverops_sgs.cpp:57-60
			Phase ph(p, 2);
			BHIP_CHECK(hipMemsetAsync(p->ytemp, 0, nbytes, p->stream));
			ph.launches = 1;
			ph.done();
		}
		// forward sweeps ytemp := D^-1 (r - L ytemp), src/solverops_sgs.cpp:62-66
		SweepArgs a = base_args(p);
		a.vals = p->vals;
		a.dvals = p->dblocks;
		a.rhs = dr;
		a.descending = 0;
		const bool compact = g_compact && mode != BLASTED_HIP_LEVEL && napplysweeps > 0 && compact_now(p, p->mat_applies, p->mat_nat);
		if (skip_fwd)
			run_sweeps(p, a, PART_NONE, POST_D_SUB, D_DBLOCKS, p->ytemp, nullptr, dr, 1, BLASTED_HIP_ASYNC, 0);
		if (compact)
			compact_args(p, false, a, p->vals, p->mat_nat);
		double *yother = jac ? ensure(p->tmp[0], n) : nullptr;
		double *y = run_sweeps(p, a, PART_LOWER, POST_D_SUB, D_DBLOCKS, p->ytemp, yother, nullptr,
		                       skip_fwd ? napplysweeps - 1 : napplysweeps, mode, 0);
		if (y != p->ytemp) {
			BHIP_CHECK(hipMemcpyAsync(p->ytemp, y, nbytes, hipMemcpyDeviceToDevice, p->stream));
			y = p->ytemp;
		}
		// z init, src/solverops_sgs.cpp:68-75, then backward sweeps z := y - D^-1 U z, :77-82
		a = base_args(p);
		a.vals = p->vals;
		a.dvals = p->dblocks;
		a.rhs = y;
		a.descending = 1;
		if (compact)
			compact_args(p, true, a, p->vals, p->mat_nat);
		const double *first_in = nullptr;
		const int nbwd = skip_bwd ? napplysweeps - 1 : napplysweeps;  // skip_bwd: z1 = y, continue as from z0 = y
		if (apply_init == BLASTED_HIP_INIT_A_JACOBI || skip_bwd) {
			if (nbwd == 0)
				BHIP_CHECK(hipMemcpyAsync(dz, y, nbytes, hipMemcpyDeviceToDevice, p->stream));
			else
				first_in = y;
		} else if (apply_init == BLASTED_HIP_INIT_A_ZERO) {
			Phase ph(p, 2);
			BHIP_CHECK(hipMemsetAsync(dz, 0, nbytes, p->stream));
			ph.launches = 1;
			ph.done();
		}
		double *zo = jac ? ensure(p->tmp[1], n) : nullptr;
		if (jac && !first_in && nbwd > 0) {
			// synchronous sweeps need the initial z in a buffer that is not written by sweep 1
			BHIP_CHECK(hipMemcpyAsync(p->tmp[0] ? p->tmp[0] : ensure(p->tmp[0], n), dz, nbytes,
			                          hipMemcpyDeviceToDevice, p->stream));
			first_in = p->tmp[0];
		}
		double *zfinal = run_sweeps(p, a, PART_UPPER, POST_SUB_D, D_DBLOCKS, dz, zo, first_in, nbwd, mode, 1);
		if (zfinal != dz)
			BHIP_CHECK(hipMemcpyAsync(dz, zfinal, nbytes, hipMemcpyDeviceToDevice, p->stream));
		finish_out(p, z, dz, loc);
	});
}

// x_i <- D_i^-1 (b_i - sum_{j != i} A_ij x_j) in place, `maxits` steps of one ascending pass and, when
// `symmetric`, one descending pass
static int relax_impl(blasted_hip_prec p, const double *b, double *x, int maxits, int mode, int loc,
                      bool symmetric)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		check_mode(mode);
		const bool deterministic = mode == BLASTED_HIP_DETERMINISTIC;
		if (deterministic)
			mode = BLASTED_HIP_JACOBI_SYNC;  // a fixed operator: synchronous sweeps (SGS apply: after an exact forward half)
		(void)deterministic;
		need_jacobi(p);
		if (!b || !x || maxits < 0)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "relaxation: null vector or negative iteration count");
		const long n = p->n();
		const size_t nbytes = sizeof(double) * (size_t)n;
		const double *db = in_vec(p, b, loc, 0);
		double *dx;
		if (loc == BLASTED_HIP_DEVICE)
			dx = x;
		else {
			dx = ensure(p->stage[1], n);
			BHIP_CHECK(hipMemcpyAsync(dx, x, nbytes, hipMemcpyHostToDevice, p->stream));
		}
		SweepArgs a = base_args(p);
		a.vals = p->vals;
		a.dvals = p->dblocks;
		a.rhs = db;
		// Thorough placement ("placement=2", one rank per GPU, memory to spare): an asynchronous relaxation pass streams
		// the whole matrix and writes x, and where the MATRIX lies is the caller's choice -- in x's address class the pass
		// is 10 % slower (config 3: 3.03 against 2.74 ms, 0.71 / 0.81 of peak from one process to the next).  Once the
		// operator has done a few passes it keeps its own copy of the values out of x's class and streams that.
		const size_t mbytes = sizeof(double) * (size_t)p->nvals();
		if (g_placement >= 2 && mode == BLASTED_HIP_ASYNC && mbytes >= ((size_t)64 << 20)) {
			if (!p->relax_vals_valid && !p->relax_vals && p->relax_passes >= 8) {
				// only with room to spare: the copy is a convenience, never a reason for an application to fail
				size_t free_b = 0, total_b = 0;
				if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < mbytes + total_b / 8)
					p->relax_passes = -(1L << 60);  // (not asked again for this operator)
			}
			if (!p->relax_vals_valid && p->relax_passes >= 8) {
				if (!p->relax_vals) {
					PlaceHint h;
					h.avoid = dx;
					h.ref_bytes = nbytes;
					if (trace_placement())
						std::fprintf(stderr, "[blasted_hip] a copy of the matrix values for the relaxation passes, out of the class of x:\n");
					p->relax_vals = dev_alloc_placed<double>((size_t)p->nvals(), h, p->stream);
				}
				BHIP_CHECK(hipMemcpyAsync(p->relax_vals, p->vals, mbytes, hipMemcpyDeviceToDevice, p->stream));
				p->relax_vals_valid = true;
			}
			if (p->relax_vals_valid)
				a.vals = p->relax_vals;
			p->relax_passes += (long)maxits * (symmetric ? 2 : 1);
		}
		double *other = (mode != BLASTED_HIP_ASYNC) ? ensure(p->tmp[0], n) : nullptr;
		double *cur = dx;
		if (mode == BLASTED_HIP_LEVEL)
			restore_ytemp(p);  // the split exact passes use the level-ordered y buffer as scratch
		for (int step = 0; step < maxits; step++) {
			for (int dir = 0; dir < (symmetric ? 2 : 1); dir++) {
				a.descending = dir;
				double *o = (mode == BLASTED_HIP_JACOBI_SYNC) ? (cur == dx ? other : dx) : dx;
				Phase ph(p, dir);
				a.xin = cur;
				a.xout = o;
				if (mode == BLASTED_HIP_LEVEL && g_relax_split) {
					// exact ascending / descending Gauss-Seidel pass, split into a product with the other
					// triangle taken from the previous iterate (full-speed out-of-place sweep) and an exact
					// triangular solve (the streaming exact kernels of the SGS application):
					//   ascending : (D + L) x_new = b - U x_old        descending: (D + U) x_new = b - L x_old
					o = (cur == dx) ? other : dx;
					double *t = ensure(p->tmp[1], n);
					SweepArgs m = base_args(p);
					m.vals = p->vals;
					m.rhs = db;
					m.xin = cur;
					m.xout = t;
					m.descending = dir;
					launch_sweep(m, dir == 0 ? PART_UPPER : PART_LOWER, POST_SUB, D_NONE, p->stream);
					int nl = 1;
					SweepArgs e = base_args(p);
					e.vals = p->vals;
					e.dvals = p->dblocks;
					e.descending = dir;
					if (dir == 0) {
						e.rhs = t;  // x_new = D^-1 (t - L x_new)
						nl += exact_pass(p, e, PART_LOWER, POST_D_SUB, D_DBLOCKS, o, nullptr);
						if (p->y_in_level_order) {  // the level-ordered form left its result in yperm
							launch_level_unpermute(p->levels, p->pat.bs, p->yperm, o, p->stream);
							p->y_in_level_order = false;
							nl++;
						}
					} else {
						double *yd = ensure(p->tmp[2], n);  // x_new = D^-1 t - D^-1 U x_new
						SweepArgs j = base_args(p);
						j.vals = p->vals;
						j.dvals = p->dblocks;
						j.rhs = t;
						j.xin = t;
						j.xout = yd;
						launch_sweep(j, PART_NONE, POST_D_SUB, D_DBLOCKS, p->stream);
						e.rhs = yd;
						nl += 1 + exact_pass(p, e, PART_UPPER, POST_SUB_D, D_DBLOCKS, o, nullptr);
					}
					ph.launches = nl;
				} else if (mode == BLASTED_HIP_LEVEL) {  // the same pass as one whole-row exact kernel
					o = (cur == dx) ? other : dx;
					ph.launches = exact_pass(p, a, PART_OFFDIAG, POST_D_SUB, D_DBLOCKS, o, cur);
				} else {
					launch_sweep(a, PART_OFFDIAG, POST_D_SUB, D_DBLOCKS, p->stream);
					ph.launches = 1;
				}
				ph.done();
				cur = o;
			}
		}
		if (cur != dx)
			BHIP_CHECK(hipMemcpyAsync(dx, cur, nbytes, hipMemcpyDeviceToDevice, p->stream));
		finish_out(p, x, dx, loc);
	});
}

// src/solverops_sgs.cpp:96-115: per step an ascending and a descending pass
int blasted_hip_sgs_relax(blasted_hip_prec p, const double *b, double *x, int maxits, int mode, int loc)
{
	return relax_impl(p, b, x, maxits, mode, loc, true);
}

// src/relaxation_chaotic.cpp:21-70: ascending passes only
int blasted_hip_gs_relax(blasted_hip_prec p, const double *b, double *x, int nsweeps, int mode, int loc)
{
	return relax_impl(p, b, x, nsweeps, mode, loc, false);
}

// BJacobiSRPreconditioner::apply_relax, src/solverops_jacobi.cpp:66-119: synchronous (block-)Jacobi steps
// x <- D^-1 (b - (A - D) x); with check_tol the step difference ||x_new - x_old||_2 is tested against
// atol, and relative to the first step's against rtol (converged) and dtol (diverged).
int blasted_hip_jacobi_relax(blasted_hip_prec p, const double *b, double *x, int maxits, int check_tol,
                             double rtol, double atol, double dtol, int *steps_done, int loc)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		need_jacobi(p);
		if (!b || !x || maxits < 0)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "jacobi_relax: null vector or negative iteration count");
		const long n = p->n();
		const size_t nbytes = sizeof(double) * (size_t)n;
		const double *db = in_vec(p, b, loc, 0);
		double *dx;
		if (loc == BLASTED_HIP_DEVICE)
			dx = x;
		else {
			dx = ensure(p->stage[1], n);
			BHIP_CHECK(hipMemcpyAsync(dx, x, nbytes, hipMemcpyHostToDevice, p->stream));
		}
		double *other = ensure(p->tmp[0], n);
		if (check_tol)
			ensure(p->red, 1024);
		SweepArgs a = base_args(p);
		a.vals = p->vals;
		a.dvals = p->dblocks;
		a.rhs = db;
		a.descending = 0;
		double *cur = dx;
		double refdiffnorm = 1.0;
		int step = 0;
		for (; step < maxits; step++) {
			double *o = (cur == dx) ? other : dx;
			Phase ph(p, 0);
			a.xin = cur;
			a.xout = o;
			launch_sweep(a, PART_OFFDIAG, POST_D_SUB, D_DBLOCKS, p->stream);
			ph.launches = 1;
			ph.done();
			if (check_tol) {
				const double diffnorm = run_diff_norm(o, cur, n, p->red, p->stream);
				cur = o;
				if (step == 0)
					refdiffnorm = diffnorm;
				if (diffnorm < atol || diffnorm / refdiffnorm < rtol || diffnorm / refdiffnorm > dtol) {
					step++;
					break;
				}
			} else
				cur = o;
		}
		if (steps_done)
			*steps_done = step;
		if (cur != dx)
			BHIP_CHECK(hipMemcpyAsync(dx, cur, nbytes, hipMemcpyDeviceToDevice, p->stream));
		finish_out(p, x, dx, loc);
	});
}

/* ---- level schedule -------------------------------------------------------------------------- */

int blasted_hip_level_schedule(blasted_hip_prec p)
{
	return guarded([&] {
		use_device(p);
		need_pattern(p);
		need_levels(p);
	});
}

int blasted_hip_level_count(blasted_hip_prec p, int *nlevels)
{
	return guarded([&] {
		use_device(p);
		need_pattern(p);
		if (!nlevels)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "level_count: null output");
		*nlevels = need_levels(p).nlevels;
	});
}

int blasted_hip_level_stats(blasted_hip_prec p, long *out4)
{
	return guarded([&] {
		use_device(p);
		need_pattern(p);
		if (!out4)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "level_stats: null output");
		const LevelSchedule &ls = need_levels(p);
		out4[0] = ls.nlevels;
		out4[1] = ls.build_passes;
		out4[2] = ls.sf_launches;
		out4[3] = ls.sf_aborts;
	});
}

int blasted_hip_memory_stats(blasted_hip_prec p, long *out4)
{
	return guarded([&] {
		use_device(p);
		if (!out4)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "memory_stats: null output");
		out4[0] = p->bytes_owned;
		out4[1] = p->bytes_peak;
		int copies = 0;
		for (const auto *c : {&p->fac_nat, &p->fac_lvl, &p->mat_nat, &p->mat_lvl})
			copies += (c->l ? 1 : 0) + (c->u ? 1 : 0);
		copies += p->relax_vals ? 2 : 0;  // (thorough placement: the relaxation passes' own copy of the matrix values)
		out4[2] = copies;  // in triangles: two = one copy's worth
		std::lock_guard<std::mutex> lk(g_pins.mu);
		out4[3] = g_pins.registered_bytes;
	});
}

int blasted_hip_placement_check(blasted_hip_prec p, const double *r, const double *z, long *out8)
{
	return guarded([&] {
		use_device(p);
		if (!out8 || !r || !z)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "placement_check: null argument");
		for (int i = 0; i < 8; i++)
			out8[i] = -1;
		const size_t nbytes = sizeof(double) * (size_t)p->n();
		if (!p->ytemp || nbytes < ((size_t)64 << 20))
			return;  // too small for the probe to tell
		BHIP_CHECK(hipStreamSynchronize(p->stream));
		double *sink = dev_alloc<double>(1);
		auto same = [&](void *piece, size_t bytes, const void *ref) {
			PlaceHint h;
			h.same = ref;
			h.ref_bytes = nbytes;
			return class_fits(piece, bytes, h, sink, p->stream) > 0;
		};
		const LevelSchedule &ns = p->natstore;
		const size_t bsz = (size_t)p->pat.bs * p->pat.bs * 8, G1 = (size_t)1 << 30;
		const size_t lb = p->fac_nat.l ? (size_t)ns.nnz_lower * bsz : 0, ub = p->fac_nat.u ? (size_t)ns.nnz_dupper * bsz : 0;
		out8[0] = out8[1] = out8[2] = out8[3] = out8[4] = out8[5] = 0;
		for (size_t at = 0; at + ((size_t)256 << 20) <= lb; at += G1) {
			char *pc = reinterpret_cast<char *>(p->fac_nat.l) + at;
			const size_t sz = lb - at < G1 ? lb - at : G1;
			out8[0]++;
			out8[1] += same(pc, sz, p->ytemp);
			out8[2] += same(pc, sz, r);
		}
		for (size_t at = 0; at + ((size_t)256 << 20) <= ub; at += G1) {
			char *pc = reinterpret_cast<char *>(p->fac_nat.u) + at;
			const size_t sz = ub - at < G1 ? ub - at : G1;
			out8[3]++;
			out8[4] += same(pc, sz, z);
			out8[5] += same(pc, sz, p->ytemp);
		}
		out8[6] = same(p->ytemp, nbytes, r);
		out8[7] = same(p->ytemp, nbytes, z);
		dev_free(sink);
	});
}

int blasted_hip_placement_stats(long *out5)
{
	return guarded([&] {
		if (!out5)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "placement_stats: null output");
		out5[0] = g_place_stats.placed_buffers;
		out5[1] = g_place_stats.pieces;
		out5[2] = g_place_stats.rejected;
		out5[3] = g_place_stats.gave_up;
		out5[4] = g_place_stats.probes;
	});
}

int blasted_hip_get_levels(blasted_hip_prec p, int *level_of_row, int *rows_by_level, int *level_ptr)
{
	return guarded([&] {
		use_device(p);
		need_pattern(p);
		const LevelSchedule &ls = need_levels(p);
		const size_t nb = sizeof(int) * (size_t)p->pat.nbrows;
		if (level_of_row && nb)
			BHIP_CHECK(hipMemcpyAsync(level_of_row, ls.level, nb, hipMemcpyDeviceToHost, p->stream));
		if (rows_by_level && nb)
			BHIP_CHECK(hipMemcpyAsync(rows_by_level, ls.rows, nb, hipMemcpyDeviceToHost, p->stream));
		BHIP_CHECK(hipStreamSynchronize(p->stream));
		if (level_ptr)
			std::memcpy(level_ptr, ls.ptr.data(), sizeof(int) * ls.ptr.size());
	});
}

/* ---- SpMV --------------------------------------------------------------------------------- */

int blasted_hip_gemv3(blasted_hip_prec p, double a_, const double *x, double b_, const double *y,
                      double *z, int loc)
{
	return guarded([&] {
		use_device(p);
		check_loc(loc);
		need_values(p);
		if (!x || !z || (b_ != 0.0 && !y))
			BHIP_FAIL(BLASTED_HIP_EINVAL, "gemv3: null vector");
		if (x == z)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "gemv3: x must not alias z");
		const double *dx = in_vec(p, x, loc, 0);
		const double *dy = (b_ != 0.0) ? in_vec(p, y, loc, 2) : nullptr;
		double *dz = out_vec(p, z, loc, 1);
		SweepArgs a = base_args(p);
		a.vals = p->vals;
		a.rhs = dy;
		a.a = a_;
		a.b = b_;
		run_sweeps(p, a, PART_ALL, POST_AXPBY, D_NONE, dz, nullptr, dx, 1, BLASTED_HIP_ASYNC, 0);
		finish_out(p, z, dz, loc);
	});
}

int blasted_hip_spmv(blasted_hip_prec p, const double *x, double *y, int loc)
{
	return blasted_hip_gemv3(p, 1.0, x, 0.0, nullptr, y, loc);
}

/* ---- read-back ---------------------------------------------------------------------------- */

static int get_array(blasted_hip_prec p, const double *dev, long count, double *out, const char *what)
{
	return guarded([&] {
		use_device(p);
		if (!dev || !out)
			BHIP_FAIL(BLASTED_HIP_ESTATE, std::string(what) + " is not available");
		BHIP_CHECK(hipMemcpyAsync(out, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, p->stream));
		BHIP_CHECK(hipStreamSynchronize(p->stream));
	});
}

int blasted_hip_get_iluvals(blasted_hip_prec p, double *out)
{
	return get_array(p, p ? p->iluvals : nullptr, p ? p->nvals() : 0, out, "iluvals");
}

int blasted_hip_get_dblocks(blasted_hip_prec p, double *out)
{
	return get_array(p, p ? p->dblocks : nullptr, p ? (long)p->pat.nbrows * p->pat.bs * p->pat.bs : 0, out,
	                 "dblocks");
}

int blasted_hip_get_scale(blasted_hip_prec p, double *out)
{
	return get_array(p, p ? p->scale : nullptr, p ? p->n() : 0, out, "scale");
}

int blasted_hip_get_ytemp(blasted_hip_prec p, double *out)
{
	if (p && p->y_in_level_order) {  // the last exact apply kept y level-ordered: natural order on demand
		const int rc = guarded([&] {
			use_device(p);
			restore_ytemp(p);
		});
		if (rc != BLASTED_HIP_OK)
			return rc;
	}
	return get_array(p, p ? p->ytemp : nullptr, p ? p->n() : 0, out, "ytemp");
}

int blasted_hip_iluvals_device(blasted_hip_prec p, double **dev_ptr)
{
	return guarded([&] {
		use_device(p);
		if (!p->iluvals || !dev_ptr)
			BHIP_FAIL(BLASTED_HIP_ESTATE, "iluvals is not available");
		// the caller may write through the pointer: whatever was derived from the factor is made again when next needed
		p->fac_lvl.invalidate();
		p->fac_nat.invalidate();
		p->fac_applies = 0;
		p->fdiag_valid = false;
		*dev_ptr = p->iluvals;
	});
}

/* ---- raw buffers --------------------------------------------------------------------------- */

int blasted_hip_buffer_alloc(void **dev_ptr, unsigned long nbytes, int device)
{
	return guarded([&] {
		if (!dev_ptr)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "null output pointer");
		int n = 0;
		if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
			BHIP_FAIL(BLASTED_HIP_ENODEV, "no HIP device available: the MI355X backend has no CPU fallback");
		BHIP_CHECK(hipSetDevice(device));
		BHIP_CHECK(tracked_malloc(dev_ptr, nbytes ? nbytes : 1));  // no operator is running: booked on nobody
	});
}

int blasted_hip_buffer_free(void *dev_ptr)
{
	return guarded([&] {
		if (dev_ptr)
			BHIP_CHECK(tracked_free(dev_ptr));
	});
}

int blasted_hip_buffer_upload(void *dev_ptr, const void *host_ptr, unsigned long nbytes)
{
	return guarded([&] {
		if (nbytes)
			BHIP_CHECK(hipMemcpy(dev_ptr, host_ptr, nbytes, hipMemcpyHostToDevice));
	});
}

int blasted_hip_buffer_download(void *host_ptr, const void *dev_ptr, unsigned long nbytes)
{
	return guarded([&] {
		if (nbytes)
			BHIP_CHECK(hipMemcpy(host_ptr, dev_ptr, nbytes, hipMemcpyDeviceToHost));
	});
}

int blasted_hip_host_register(void *host_ptr, unsigned long nbytes)
{
	return guarded([&] {
		if (!host_ptr || nbytes == 0)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "host_register: null pointer or empty range");
		int n = 0;
		if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
			BHIP_FAIL(BLASTED_HIP_ENODEV, "no HIP device available: the MI355X backend has no CPU fallback");
		std::lock_guard<std::mutex> lk(g_pins.mu);
		const uintptr_t key = reinterpret_cast<uintptr_t>(host_ptr);
		if (g_pins.ranges.count(key))
			BHIP_FAIL(BLASTED_HIP_ESTATE, "host_register: this address is registered already");
		BHIP_CHECK(hipHostRegister(host_ptr, nbytes, hipHostRegisterDefault));
		g_pins.ranges[key] = nbytes;
		g_pins.registered_bytes += (long)nbytes;
	});
}

int blasted_hip_host_unregister(void *host_ptr)
{
	return guarded([&] {
		std::lock_guard<std::mutex> lk(g_pins.mu);
		auto it = g_pins.ranges.find(reinterpret_cast<uintptr_t>(host_ptr));
		if (it == g_pins.ranges.end())
			BHIP_FAIL(BLASTED_HIP_ESTATE, "host_unregister: this address is not registered");
		g_pins.registered_bytes -= (long)it->second;
		g_pins.ranges.erase(it);
		BHIP_CHECK(hipHostUnregister(host_ptr));
	});
}

int blasted_hip_measure_read_stream(const void *dev_ptr, unsigned long nbytes, int reps, double *gbps)
{
	return guarded([&] {
		if (!dev_ptr || !gbps || reps < 1 || (reinterpret_cast<uintptr_t>(dev_ptr) & 15u))
			BHIP_FAIL(BLASTED_HIP_EINVAL, "measure_read_stream: 16-byte aligned device buffer, reps >= 1");
		hipPointerAttribute_t attr;
		BHIP_CHECK(hipPointerGetAttributes(&attr, dev_ptr));
		BHIP_CHECK(hipSetDevice(attr.device));  // measure on the device that owns the buffer
		double *sink = dev_alloc<double>(1);
		hipEvent_t e0, e1;
		BHIP_CHECK(hipEventCreate(&e0));
		BHIP_CHECK(hipEventCreate(&e1));
		for (int r = 0; r < 2; r++)
			launch_read_stream(dev_ptr, nbytes, sink, nullptr);
		BHIP_CHECK(hipEventRecord(e0, nullptr));
		for (int r = 0; r < reps; r++)
			launch_read_stream(dev_ptr, nbytes, sink, nullptr);
		BHIP_CHECK(hipEventRecord(e1, nullptr));
		BHIP_CHECK(hipEventSynchronize(e1));
		float ms = 0.f;
		BHIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
		BHIP_CHECK(hipEventDestroy(e0));
		BHIP_CHECK(hipEventDestroy(e1));
		dev_free(sink);
		*gbps = ms > 0.f ? (double)(nbytes / 16 * 16) * reps / (ms * 1e-3) / 1e9 : 0.0;
	});
}

/* ---- tuning -------------------------------------------------------------------------------- */

int blasted_hip_set_tuning(const char *spec)
{
	return guarded([&] {
		if (spec && std::strncmp(spec, "levelperm=", 10) == 0)
			g_level_perm = spec[10] != '0';
		else if (spec && std::strncmp(spec, "interleave=", 11) == 0)
			g_interleave = spec[11] == '0' ? 0 : (spec[11] == '2' ? 2 : (spec[11] == '3' ? 3 : 1));  // 2: the round-1 form (through memory); 3: that form for relaxation passes too
		else if (spec && std::strncmp(spec, "compact=", 8) == 0)
			g_compact = spec[8] != '0';
		else if (spec && std::strncmp(spec, "compactafter=", 13) == 0)
			g_compact_after = std::atol(spec + 13);
		else if (spec && std::strncmp(spec, "placeafter=", 11) == 0)
			g_place_after = std::atol(spec + 11);
		else if (spec && std::strncmp(spec, "smallapply=", 11) == 0)
			g_small_apply = spec[11] - '0';
		else if (spec && std::strncmp(spec, "placement=", 10) == 0)
			g_placement = spec[10] == '0' ? 0 : (spec[10] == '2' ? 2 : 1);
		else if (spec && std::strncmp(spec, "applynone=", 10) == 0)
			g_apply_allow_none = spec[10] != '0';
		else if (spec && std::strncmp(spec, "allocoff=", 9) == 0)
			g_alloc_offset = (size_t)std::atol(spec + 9) & ~(size_t)255;
		else if (spec && std::strncmp(spec, "sfonestep=", 10) == 0)
			set_syncfree_one_step(spec[10] != '0');
		else if (spec && std::strncmp(spec, "levelwide=", 10) == 0)
			set_levelw_enabled(spec[10] - '0');
		else if (spec && std::strncmp(spec, "invertrow=", 10) == 0)
			set_invert_rowlane(spec[10] != '0');  // eight-lanes-per-block inversion of 5 <= bs <= 8 diagonal blocks
		else if (spec && std::strncmp(spec, "latestore=", 10) == 0)
			g_late_store = (spec[10] == '0' || spec[10] == '1' || spec[10] == '2' || spec[10] == '4') ? spec[10] - '0' : 2;  // row steps in flight; 0: stores step by step
#ifdef BHIP_PROBES
		else if (spec && std::strncmp(spec, "gatherprobe=", 12) == 0)
			g_gather_probe = spec[12] - '0';  // measurements only: wrong results (1: sweepodd gathers its own row; 2, 3: store probes of the interleaved bs=4 sweeps)
		else if (spec && std::strncmp(spec, "levelnowait=", 12) == 0)
			set_syncfree_nowait(spec[12] - '0');  // measurements only: wrong results
		else if (spec && std::strncmp(spec, "factorprobe=", 12) == 0)
			set_factor_probe(spec[12] - '0');
#else
		else if (spec && (std::strncmp(spec, "gatherprobe=", 12) == 0 || std::strncmp(spec, "levelnowait=", 12) == 0 ||
		                  std::strncmp(spec, "factorprobe=", 12) == 0))
			BHIP_FAIL(BLASTED_HIP_EINVAL, "this tuning string selects a timing experiment with WRONG results: it exists in the "
			                              "probes build only (make -C blasted_amd/csrc probes; BLASTED_HIP_PROBES=1)");
#endif
		else if (spec && std::strncmp(spec, "levelfast=", 10) == 0)
			set_level_fast(spec[10] - '0');  // (2: tests -- behave as if the polling launch had given up)
		else if (spec && std::strncmp(spec, "levelserial=", 12) == 0)
			set_level_serial_after(std::atol(spec + 12));
		else if (spec && std::strncmp(spec, "levelstore=", 11) == 0)
			g_level_store = spec[11] != '0';
		else if (spec && std::strncmp(spec, "level=", 6) == 0)
			g_level_impl = std::strcmp(spec + 6, "launch") == 0 ? 1 : 0;
		else if (spec && std::strncmp(spec, "sweepwr=", 8) == 0)
			set_sweepwr_enabled(spec[8] != '0');
		else if (spec && std::strncmp(spec, "sweepodd=", 9) == 0)
			set_sweepodd_enabled(std::strcmp(spec + 9, "nt1") == 0 ? 2 : (std::strcmp(spec + 9, "nt0") == 0 ? 3 : (std::strcmp(spec + 9, "occ1") == 0 ? 4 : (std::strcmp(spec + 9, "occ0") == 0 ? 5 : spec[9] != '0'))));
		else if (spec && std::strncmp(spec, "factorfuse=", 11) == 0)
			g_factor_fuse_init = spec[11] - '0';
		else if (spec && std::strncmp(spec, "factor1plan=", 12) == 0)
			g_factor1_plan = spec[12] != '0';
		else if (spec && std::strncmp(spec, "scalarstage=", 12) == 0)
			set_scalar_stage(spec[12] != '0');
		else if (spec && std::strncmp(spec, "scalarlane=", 11) == 0)
			set_scalar_lane(std::strcmp(spec + 11, "auto") == 0 ? -1 : spec[11] - '0');
		else if (spec && std::strncmp(spec, "gunroll=", 8) == 0)
			set_sweep_unroll(spec[8] == '1' ? 1 : (spec[8] == '2' ? 2 : 0));
		else if (spec && std::strncmp(spec, "factorodd=", 10) == 0)
			set_factorodd_enabled(spec[10] != '0');
		else if (spec && std::strncmp(spec, "relaxsplit=", 11) == 0)
			g_relax_split = spec[11] != '0';
		else if (spec && std::strncmp(spec, "xcdsuper=", 9) == 0) {
			const int n = std::atoi(spec + 9);
			int sh = 0;
			while ((1 << sh) < n && sh < 12)
				sh++;
			if (std::strcmp(spec + 9, "auto") == 0)
				g_xcd_shift = -1;
			else {
				if (n < 1 || (1 << sh) != n)
					BHIP_FAIL(BLASTED_HIP_EINVAL, "xcdsuper: a power of two between 1 and 4096, or auto");
				g_xcd_shift = sh;
			}
		} else if (spec && std::strncmp(spec, "copies=", 7) == 0)
			g_keep_both_copies = std::strcmp(spec + 7, "both") == 0;
		else if (spec && std::strncmp(spec, "sgsfwd=", 7) == 0)
			g_sgs_exact_fwd = std::strcmp(spec + 7, "async") != 0;
		else if (spec && std::strncmp(spec, "factorskip=", 11) == 0)
			g_factor_skip_fixed = spec[11] != '0';
		else if (spec && std::strncmp(spec, "factorsf=", 9) == 0)
			set_factor_syncfree(spec[9] == 'p' ? 10 + (spec[10] - '0') : (spec[9] == 'a' ? 20 + (spec[10] - '0') : spec[9] - '0'));
		else if (spec && std::strncmp(spec, "factor1=", 8) == 0)
			set_factor1_enabled(spec[8] != '0');
		else if (spec && std::strncmp(spec, "factor4=", 8) == 0)
			set_factor4_enabled(spec[8] == 's' ? 10 + (spec[9] - '0') : (spec[8] != '0'));
		else if (spec && std::strncmp(spec, "factor8=", 8) == 0)
			set_factor8_enabled(spec[8] == '2' ? 2 : (spec[8] != '0'));
		else
			set_sweepw_variant(spec);
	});
}

#ifdef BHIP_PROBES
/* ---- placement probes (libblasted_hip_probes.so only; tools/probes/placement_streams.py) ------------------- */
// Re-allocates one of the operator's buffers (under the current "allocoff"), contents preserved or rebuilt by the
// next application: "ytemp", "ucopy" / "lcopy" (natural-order triangle copies of the factor), "nat" (the compact
// pattern of those copies and both copies), "iluvals".
int blasted_hip_probe_move(blasted_hip_prec p, const char *what)
{
	return guarded([&] {
		use_device(p);
		BHIP_CHECK(hipStreamSynchronize(p->stream));
		const std::string w = what ? what : "";
		auto move = [&](double *&buf, size_t count) {
			if (!buf)
				return;
			double *nb = dev_alloc<double>(count);
			BHIP_CHECK(hipMemcpy(nb, buf, sizeof(double) * count, hipMemcpyDeviceToDevice));
			dev_free(buf);
			buf = nb;
		};
		if (w == "ytemp")
			move(p->ytemp, (size_t)p->n());
		else if (w == "iluvals") {
			move(p->iluvals, (size_t)p->nvals());
			p->fac_nat.invalidate();
		} else if (w == "ucopy") {
			dev_free(p->fac_nat.u);
			p->fac_nat.u = nullptr;
			p->fac_nat.valid_u = false;
		} else if (w == "lcopy") {
			dev_free(p->fac_nat.l);
			p->fac_nat.l = nullptr;
			p->fac_nat.valid_l = false;
		} else if (w == "nat") {
			for (auto *c : {&p->fac_nat, &p->mat_nat}) {
				dev_free(c->l);
				dev_free(c->u);
				c->l = c->u = nullptr;
				c->invalidate();
			}
			free_level_schedule(p->natstore);
		} else
			BHIP_FAIL(BLASTED_HIP_EINVAL, "probe_move: ytemp | iluvals | ucopy | lcopy | nat");
	});
}

// Points one of the operator's buffers at caller-owned device memory (an arena slot; 0 = forget it again, nothing is
// freed).  The operator's own buffer is freed first.  "ucopy" / "lcopy" are re-filled by the next application.
int blasted_hip_probe_place(blasted_hip_prec p, const char *what, void *where)
{
	return guarded([&] {
		use_device(p);
		BHIP_CHECK(hipStreamSynchronize(p->stream));
		const std::string w = what ? what : "";
		double *&buf = w == "ytemp" ? p->ytemp : (w == "ucopy" ? p->fac_nat.u : (w == "iluvals" ? p->iluvals : p->fac_nat.l));
		if (w != "ytemp" && w != "ucopy" && w != "lcopy" && w != "iluvals")
			BHIP_FAIL(BLASTED_HIP_EINVAL, "probe_place: ytemp | ucopy | lcopy | iluvals");
		{
			AllocRegistry &r = alloc_registry();
			bool mine;
			{
				std::lock_guard<std::mutex> lk(r.mu);
				mine = r.recs.count(buf) != 0;
			}
			if (mine)
				dev_free(buf);
		}
		buf = static_cast<double *>(where);
		if (w == "ucopy")
			p->fac_nat.valid_u = false;
		if (w == "lcopy")
			p->fac_nat.valid_l = false;
		if (w == "ytemp" && where)
			BHIP_CHECK(hipMemset(where, 0, sizeof(double) * (size_t)p->n()));
		if (w == "iluvals") {
			p->fac_nat.invalidate();
			p->fac_lvl.invalidate();
			p->factored = false;
		}
	});
}

// Device memory built by hand (placement studies): a virtual range of `bytes` whose start is aligned to `va_align`,
// backed by physical allocations of `chunk` bytes each (the last one smaller), mapped in order.  Never freed.
int blasted_hip_probe_vmm_alloc(unsigned long bytes, unsigned long va_align, unsigned long chunk, void **out)
{
	return guarded([&] {
		int dev = 0;
		BHIP_CHECK(hipGetDevice(&dev));
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned;
		prop.location.type = hipMemLocationTypeDevice;
		prop.location.id = dev;
		size_t gran = 0;
		BHIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
		if (!gran || chunk % gran || bytes % gran)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "probe_vmm_alloc: sizes must be multiples of the allocation granularity");
		void *va = nullptr;
		BHIP_CHECK(hipMemAddressReserve(&va, bytes, va_align, nullptr, 0));
		for (size_t at = 0; at < bytes; at += chunk) {
			const size_t sz = bytes - at < chunk ? bytes - at : chunk;
			hipMemGenericAllocationHandle_t h;
			BHIP_CHECK(hipMemCreate(&h, sz, &prop, 0));
			BHIP_CHECK(hipMemMap(static_cast<char *>(va) + at, sz, 0, h, 0));
			BHIP_CHECK(hipMemRelease(h));  // the mapping keeps the memory
		}
		hipMemAccessDesc acc = {};
		acc.location = prop.location;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		BHIP_CHECK(hipMemSetAccess(va, bytes, &acc, 1));
		BHIP_CHECK(hipMemset(va, 0, bytes));
		*out = va;
	});
}

// Addresses of the buffers the asynchronous ILU application streams: out[0..5] = ytemp, lower copy, upper copy,
// iluvals, lower / upper column indices of the copies (0 where absent).
int blasted_hip_probe_addresses(blasted_hip_prec p, unsigned long *out6)
{
	return guarded([&] {
		use_device(p);
		out6[0] = (unsigned long)(uintptr_t)p->ytemp;
		out6[1] = (unsigned long)(uintptr_t)p->fac_nat.l;
		out6[2] = (unsigned long)(uintptr_t)p->fac_nat.u;
		out6[3] = (unsigned long)(uintptr_t)p->iluvals;
		out6[4] = (unsigned long)(uintptr_t)p->natstore.lcol;
		out6[5] = (unsigned long)(uintptr_t)p->natstore.ucol;
	});
}

// ms of one launch of the read-beside-write probe (kernels_aux.hip), average of `reps` after two warm-up launches
int blasted_hip_probe_rw(const void *rd, unsigned long rd_bytes, void *wr, unsigned long wr_bytes, int reps, double *ms_out)
{
	return guarded([&] {
		double *sink = dev_alloc<double>(1);
		hipEvent_t e0, e1;
		BHIP_CHECK(hipEventCreate(&e0));
		BHIP_CHECK(hipEventCreate(&e1));
		const long passes = reps < 0 ? (long)((((size_t)4 << 30) + rd_bytes - 1) / rd_bytes) : 1;  // reps < 0: the product's form
		reps = reps < 0 ? -reps : reps;
		for (int r = 0; r < 2; r++)
			launch_rw_probe(rd, (long)rd_bytes, wr, (long)wr_bytes, sink, nullptr, passes);
		BHIP_CHECK(hipEventRecord(e0, nullptr));
		for (int r = 0; r < reps; r++)
			launch_rw_probe(rd, (long)rd_bytes, wr, (long)wr_bytes, sink, nullptr, passes);
		BHIP_CHECK(hipEventRecord(e1, nullptr));
		BHIP_CHECK(hipEventSynchronize(e1));
		float ms = 0.f;
		BHIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
		BHIP_CHECK(hipEventDestroy(e0));
		BHIP_CHECK(hipEventDestroy(e1));
		dev_free(sink);
		*ms_out = (double)ms / reps;
	});
}

// GB/s of a read stream of 2 x bytes_each whose pieces alternate between r0 and r1 (kernels_aux.hip)
int blasted_hip_probe_read2(const void *r0, const void *r1, unsigned long bytes_each, unsigned long piece, int reps, double *gbps)
{
	return guarded([&] {
		double *sink = dev_alloc<double>(1);
		hipEvent_t e0, e1;
		BHIP_CHECK(hipEventCreate(&e0));
		BHIP_CHECK(hipEventCreate(&e1));
		for (int r = 0; r < 2; r++)
			launch_read2_probe(r0, r1, (long)bytes_each, (long)piece, sink, nullptr);
		BHIP_CHECK(hipEventRecord(e0, nullptr));
		for (int r = 0; r < reps; r++)
			launch_read2_probe(r0, r1, (long)bytes_each, (long)piece, sink, nullptr);
		BHIP_CHECK(hipEventRecord(e1, nullptr));
		BHIP_CHECK(hipEventSynchronize(e1));
		float ms = 0.f;
		BHIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
		BHIP_CHECK(hipEventDestroy(e0));
		BHIP_CHECK(hipEventDestroy(e1));
		dev_free(sink);
		*gbps = 2.0 * (double)bytes_each * reps / (ms * 1e-3) / 1e9;
	});
}

// ns per dependent load of a one-lane walk over [ptr, ptr + nbytes) with `stride` bytes between loads, starting
// `start` bytes in (a multiple of 8 below the stride, so that repeated walks touch fresh cache lines)
int blasted_hip_probe_page_walk(const void *ptr, unsigned long nbytes, unsigned long stride, unsigned long start, double *ns)
{
	return guarded([&] {
		const long nloads = (long)((nbytes - start) / stride);
		if (!ptr || nloads < 1 || (stride & 7) || (start & 7))
			BHIP_FAIL(BLASTED_HIP_EINVAL, "probe_page_walk");
		double *sink = dev_alloc<double>(1);
		hipEvent_t e0, e1;
		BHIP_CHECK(hipEventCreate(&e0));
		BHIP_CHECK(hipEventCreate(&e1));
		BHIP_CHECK(hipDeviceSynchronize());
		BHIP_CHECK(hipEventRecord(e0, nullptr));
		launch_page_walk(static_cast<const char *>(ptr) + start, nloads, (long)stride, sink, nullptr);
		BHIP_CHECK(hipEventRecord(e1, nullptr));
		BHIP_CHECK(hipEventSynchronize(e1));
		float ms = 0.f;
		BHIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
		BHIP_CHECK(hipEventDestroy(e0));
		BHIP_CHECK(hipEventDestroy(e1));
		dev_free(sink);
		*ns = (double)ms * 1e6 / (double)nloads;
	});
}
#endif  // BHIP_PROBES

/* ---- timing -------------------------------------------------------------------------------- */

int blasted_hip_set_timing(blasted_hip_prec p, int enable)
{
	return guarded([&] {
		use_device(p);
		if (!enable)
			fold_timing(p);
		p->timing.enabled = enable != 0;
	});
}

int blasted_hip_get_timing(blasted_hip_prec p, double *out6, int reset)
{
	return guarded([&] {
		use_device(p);
		if (!out6)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "null output");
		fold_timing(p);
		for (int k = 0; k < 3; k++) {
			out6[2 * k] = p->timing.ms[k];
			out6[2 * k + 1] = p->timing.launches[k];
		}
		if (reset)
			for (int k = 0; k < 3; k++)
				p->timing.ms[k] = p->timing.launches[k] = 0;
	});
}

}  // extern "C"
