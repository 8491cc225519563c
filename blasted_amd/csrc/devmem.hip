// devmem.hip -- device memory of the library: allocation tracking (which operator holds what: blasted_hip_memory_stats)
// and the class-aware placement of the buffers the sweeps stream (round 4).
#include "devmem.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>

namespace bhip {

// BLASTED_HIP_TRACE_ALLOC=1: print every device allocation of 64 MiB and more (placement studies)
void trace_alloc(const void *p, size_t bytes)
{
	static const bool on = std::getenv("BLASTED_HIP_TRACE_ALLOC") != nullptr;
	if (on && bytes >= (64u << 20))
		std::fprintf(stderr, "[blasted_hip] alloc %12zu B at %p\n", bytes, p);
}

// ---- device memory accounting ------------------------------------------------------------------
thread_local blasted_hip_prec tl_owner = nullptr;  // the operator whose entry point runs on this thread

AllocRegistry &alloc_registry()
{
	static AllocRegistry *r = new AllocRegistry;  // never destroyed: operators may be freed at process exit
	return *r;
}

// Placement studies (tuning "allocoff=BYTES" / BLASTED_HIP_ALLOC_OFFSET, a multiple of 256): every device allocation
// of 64 MiB and more starts BYTES into a correspondingly larger hipMalloc block.  Results cannot change.
size_t g_alloc_offset = [] {
	const char *e = std::getenv("BLASTED_HIP_ALLOC_OFFSET");
	return e ? (size_t)std::atol(e) & ~(size_t)255 : (size_t)0;
}();


hipError_t tracked_malloc(void **p, size_t bytes)
{
	const size_t off = bytes >= (64u << 20) ? g_alloc_offset : 0;
	void *base = nullptr;
	hipError_t e = hipMalloc(&base, bytes + off);
	if (e != hipSuccess && release_deferred()) {
		(void)hipGetLastError();
		e = hipMalloc(&base, bytes + off);
	}
	if (e != hipSuccess)
		return e;
	*p = static_cast<char *>(base) + off;
	trace_alloc(*p, bytes);
	AllocRegistry &r = alloc_registry();
	std::lock_guard<std::mutex> lk(r.mu);
	AllocRegistry::Rec rec;
	rec.bytes = bytes;
	rec.owner = tl_owner;
	rec.base = base;
	r.recs[*p] = rec;
	if (tl_owner) {
		tl_owner->bytes_owned += (long)bytes;
		if (tl_owner->bytes_owned > tl_owner->bytes_peak)
			tl_owner->bytes_peak = tl_owner->bytes_owned;
	}
	return e;
}

hipError_t tracked_free(void *p)
{
	if (p) {
		AllocRegistry &r = alloc_registry();
		std::lock_guard<std::mutex> lk(r.mu);
		auto it = r.recs.find(p);
		if (it != r.recs.end()) {
			if (it->second.owner)
				it->second.owner->bytes_owned -= (long)it->second.bytes;
			p = it->second.base;
			if (it->second.va_bytes) {  // a class-aware allocation: unmap its pieces, give the range back
				// (hipFree waits for the device before it releases memory; hipMemUnmap does not, and a sweep that still
				// streams this buffer would fault)
				hipError_t e = hipDeviceSynchronize();
				for (const auto &pc : it->second.pieces) {
					const hipError_t e1 = hipMemUnmap(static_cast<char *>(p) + pc.first, pc.second);
					e = e == hipSuccess ? e1 : e;
				}
				// The address range itself is NOT handed back (hipMemAddressFree): a later reservation that got such a
				// range again and mapped new memory into it took a "Memory access fault" on first touch (ROCm 7.2, round 4,
				// twice); address space is not a scarce resource.
				r.recs.erase(it);
				return e;
			}
			r.recs.erase(it);
		}
	}
	return hipFree(p);
}

// an operator that goes away must not be booked on any more
void forget_owner(blasted_hip_prec p)
{
	AllocRegistry &r = alloc_registry();
	std::lock_guard<std::mutex> lk(r.mu);
	for (auto &kv : r.recs)
		if (kv.second.owner == p)
			kv.second.owner = nullptr;
}

// ---- class-aware placement of the sweeps' large buffers -----------------------------------------------------------
// Measured in round 4 (profiles/r04_placement_{streams,slots,pairs,map,rwprobe,interleave}.txt): the 288 GiB of an
// MI355X fall into three address classes of 96 GiB (presumably the three ranks of the 12-high HBM3E stacks; which class
// a piece of memory is in follows from its PHYSICAL address, which the driver chooses).  A kernel that streams one
// buffer and writes another takes 10-12 % longer when both lie in the same class (256^3 bs=4 upper sweep 1.82 against
// 1.63-1.66 ms, lower 1.45 against 1.37 ms), and a read stream that alternates between classes is 5 % slower than one
// that stays in one.  This is the whole of the "fast / slow mode" that rounds 1-3 saw move from process to process and
// from allocation to allocation: whether a 2 GiB piece of a triangle copy happened to share its class with the vector
// the sweep writes.  So the copies a sweep STREAMS are built piece by piece (hipMemCreate / hipMemMap, 2 GiB pieces)
// and every piece is CHECKED -- the address-class probe of kernels_aux.hip, timed with its writes inside the piece
// (same class by construction) and with its writes in the reference vector -- before it is kept: `avoid` = the vector
// the sweep writes (a piece of the same class is handed back and another one asked for), `same` = a vector the copy
// should share its class with.  Pieces that are turned down stay allocated until the search is over, so that the driver
// cannot offer them again.  tuning "placement=0" / BLASTED_HIP_PLACEMENT=0: plain hipMalloc as before.
int g_placement = [] {
	const char *e = std::getenv("BLASTED_HIP_PLACEMENT");
	return e ? (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1)) : 1;
}();

// Pieces and spacers a search held back are handed back LATER -- at the end of the entry point that made the buffers --
// not between one buffer's search and the next: the driver wipes released memory and makes the next allocation wait for
// it, and inside one application three buffers are placed one after the other (6 s of waiting seen, for searches that
// take 0.1 s when nothing is being wiped).
struct HeldPiece {
	hipMemGenericAllocationHandle_t h;
	size_t bytes;
};
static std::vector<HeldPiece> g_deferred_release;
static std::mutex g_deferred_mu;  // (operators of different threads share the list)

bool release_deferred()  // true: there was something to release
{
	std::vector<HeldPiece> mine;
	{
		std::lock_guard<std::mutex> lk(g_deferred_mu);
		mine.swap(g_deferred_release);
	}
	for (const auto &hp : mine)
		(void)hipMemRelease(hp.h);
	return !mine.empty();
}

PlaceStats g_place_stats;

// ms of one launch of the address-class probe: the fastest of `reps` launches after one warm-up; a piece smaller than
// 4 GiB is read several times per launch, so that every launch moves about 4 GiB (0.8 ms)
double probe_ms(const void *rd, size_t rd_bytes, void *wr, size_t wr_bytes, int reps, double *sink, hipStream_t s)
{
	const long passes = (long)((((size_t)4 << 30) + rd_bytes - 1) / rd_bytes);
	std::vector<hipEvent_t> ev((size_t)reps + 1);
	for (auto &e : ev)
		BHIP_CHECK(hipEventCreate(&e));
	launch_rw_probe(rd, (long)rd_bytes, wr, (long)wr_bytes, sink, s, passes);
	BHIP_CHECK(hipEventRecord(ev[0], s));
	for (int r = 0; r < reps; r++) {
		launch_rw_probe(rd, (long)rd_bytes, wr, (long)wr_bytes, sink, s, passes);
		BHIP_CHECK(hipEventRecord(ev[(size_t)r + 1], s));
	}
	BHIP_CHECK(hipEventSynchronize(ev[(size_t)reps]));
	double best = 1e300;
	for (int r = 0; r < reps; r++) {
		float ms = 0.f;
		BHIP_CHECK(hipEventElapsedTime(&ms, ev[(size_t)r], ev[(size_t)r + 1]));
		best = ms < best ? ms : best;
	}
	for (auto &e : ev)
		BHIP_CHECK(hipEventDestroy(e));
	g_place_stats.probes++;
	return best;
}

// Does `piece` (device memory; its contents are rewritten unchanged) satisfy the hint?  +2: musts and the preference,
// +1: the musts only, -1: a must is violated, 0: cannot tell (too small to time).
// The probe is timed once with its writes inside the piece itself (the "same class" time) and once per reference vector.
int class_fits(void *piece, size_t piece_bytes, const PlaceHint &h, double *sink, hipStream_t s)
{
	size_t rd = piece_bytes < ((size_t)2 << 30) ? piece_bytes : ((size_t)2 << 30);
	rd &= ~(size_t)0xffff;
	size_t wr = (rd >> 4) & ~(size_t)0xfff;
	if (wr > h.ref_bytes)
		wr = h.ref_bytes & ~(size_t)0xfff;
	if (rd < ((size_t)32 << 20) || wr < 4096)
		return 0;
	// same class: ratio 0.99-1.01; another class: about 0.90
	const double thr = 0.955;
	static const bool trace = std::getenv("BLASTED_HIP_TRACE_PLACEMENT") != nullptr;
	int fits = +2;
	const void *refs[4] = {h.same, h.avoid, h.avoid2, h.prefer};
	for (int k = 0; k < 4 && fits > 0; k++) {
		if (!refs[k])
			continue;
		// the two timings ALTERNATE (self, ref, self, ref, ...), the fastest of each: a clock that is still ramping
		// or another process on the device shifts both alike (measured one after the other, a drift of 5 % between the two
		// blocks of launches reads as "another class" -- seen once in a thousand-test run, round 4)
		double t_self = 1e300, t_ref = 1e300;
		for (int rep = 0; rep < 2; rep++) {
			const double a = probe_ms(piece, rd, static_cast<char *>(piece) + rd - wr, wr, rep == 0 ? 2 : 1, sink, s);
			const double b = probe_ms(piece, rd, const_cast<void *>(refs[k]), wr, rep == 0 ? 2 : 1, sink, s);
			t_self = a < t_self ? a : t_self;
			t_ref = b < t_ref ? b : t_ref;
		}
		const bool other_class = t_ref < thr * t_self;
		const bool want_same = k == 0 || k == 3;
		if (trace)
			std::fprintf(stderr, "[blasted_hip] class probe: piece %p (%zu MiB) against %p: self %.4f ms, ref %.4f ms, ratio %.3f -> %s (%s: %s)\n",
			             piece, piece_bytes >> 20, refs[k], t_self, t_ref, t_ref / t_self, other_class ? "another class" : "same class",
			             k == 3 ? "preferred" : "must be", want_same ? "same" : "another");
		if (other_class == want_same)
			fits = k == 3 ? +1 : -1;
	}
	return fits;
}

// How much device memory a search may hold back while it looks for pieces of the right class (they are released when
// the buffer is complete).  Consecutive allocations usually come from one class until the driver's free blocks of that
// class run out, so a search can need tens of GiB -- and the driver's allocation calls take anything from 0.1 ms to
// seconds each (freed memory is wiped asynchronously).  Bounded by what is free (less a reserve for everybody else on the
// device) and by time (place_budget_ms).
// ... and how long: the driver's allocation calls take 0.1 ms when the device is quiet (a search that steps over 64 GiB
// was seen to take 11 ms) and SECONDS while memory that other processes freed a moment ago is still being wiped.  The
// default gives a search about fifty applications' worth of time (6 sweeps over the buffer each, at 5 TB/s: 0.5 s for
// the 8.5 GB upper copy of the 256^3 bs=4 case), then keeps what comes; "placement=2" allows 20 s.
static double place_budget_ms(size_t bytes)
{
	if (g_placement >= 2)
		return 20000.0;
	const double ms = (double)bytes * 6e-8;
	return ms < 15.0 ? 15.0 : ms;
}

// And what a search holds back it must hand back: the driver wipes released memory (about 30 ms per GiB on these
// boxes) and the NEXT allocation of the process waits for that -- a search that had stepped over 200 GiB made the
// allocation after it take 6 s (profiles/r04_placement_ab_unbounded_bytes.txt).  The release is therefore put off to the
// end of the entry point (release_deferred): the searches themselves no longer wait, whoever allocates next in the
// process does, once.  Default (quick) search: at most 32 GiB and a quarter of what is free.
static int g_quick_gib = [] {  // BLASTED_HIP_PLACEMENT_GIB: what the quick search may hold back (tools/probes/placement_ab.sh)
	const char *e = std::getenv("BLASTED_HIP_PLACEMENT_GIB");
	const int v = e ? std::atoi(e) : 32;
	return v < 1 ? 1 : (v > 256 ? 256 : v);
}();

static size_t place_budget(size_t bytes)
{
	size_t want = g_placement >= 2 ? (size_t)256 << 30 : (size_t)g_quick_gib << 30;
	// never more than what is free now, less the buffer itself and a reserve for everybody else on the device
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
		return 0;
	const size_t reserve = bytes + total_b / 8;
	size_t room = free_b > reserve ? free_b - reserve : 0;
	if (g_placement < 2 && room > free_b / 4)
		room = free_b / 4;  // (other ranks may share the device: the default never holds back more than a quarter of what is free)
	return want < room ? want : room;
}

bool trace_placement()
{
	static const bool on = std::getenv("BLASTED_HIP_TRACE_PLACEMENT") != nullptr;
	return on;
}

static double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// device memory of `bytes` whose pieces satisfy the hint (see above); nullptr: not applicable / not available (the
// caller allocates plainly)
void *placed_alloc(size_t bytes, const PlaceHint &h, hipStream_t s)
{
	const void *ref = h.same ? h.same : (h.avoid ? h.avoid : (h.avoid2 ? h.avoid2 : h.prefer));
	if (!g_placement || !ref || bytes < ((size_t)64 << 20) || h.ref_bytes < ((size_t)16 << 20))
		return nullptr;
	static const bool trace = std::getenv("BLASTED_HIP_TRACE_PLACEMENT") != nullptr;
	const double t_start = now_ms();
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return nullptr;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = dev;
	size_t gran = 0;
	if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran)
		return nullptr;
	// EQUAL pieces of 1 GiB (one piece of the buffer's size if that is no more): inside one reserved range ROCm 7.2's
	// hipMemSetAccess turns down pieces of unequal sizes at some offsets (tools/probes/vmm_rules.hip: 768 MiB behind
	// 1 GiB, 512 MiB behind 3 GiB -- "invalid argument"), equal ones never
	const size_t piece = (size_t)1 << 30;
	const size_t unit = bytes <= piece ? ((size_t)2 << 20) : piece;
	if (unit % gran)
		return nullptr;
	const size_t total = (bytes + unit - 1) / unit * unit;
	void *va = nullptr;
	if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess)
		return nullptr;
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	std::vector<std::pair<size_t, size_t>> mapped;
	std::vector<HeldPiece> held;  // turned-down pieces and spacers: allocated until the search is over
	// The pieces earlier searches of this call looked at and turned down are looked at FIRST (they are allocated anyway, and
	// what one buffer must not share is often what the next one should: the pieces the lower copy turned down for lying in
	// ytemp's class are the ones the upper copy prefers): a search then begins with what the others have learnt about the
	// driver's free memory instead of stepping over the same stretch again.
	std::vector<HeldPiece> pool;
	if (bytes > piece) {
		std::lock_guard<std::mutex> lk(g_deferred_mu);
		std::vector<HeldPiece> rest;
		for (const auto &hp : g_deferred_release)
			(hp.bytes == piece ? pool : rest).push_back(hp);
		g_deferred_release.swap(rest);
	}
	size_t held_bytes = 0;
	size_t budget = place_budget(total);
	// What an earlier search of this entry point still holds back (its release is deferred to the end of the call) counts
	// as used: a search for `ytemp` that had to step over most of a class (up to 96 GiB of consecutive allocations come
	// from one class) once left the searches for the two copies behind it no room at all -- every piece of theirs was
	// kept unchecked, and all of them happened to lie in the wrong class (profiles/r04_bench_starved_search.txt:
	// 299 sweeps/s instead of 326-334).  A search that finds itself short therefore hands that memory back FIRST
	// (and waits for the driver's wipe, once) instead of going without.
	{
		const size_t enough = total + (g_placement >= 2 ? (size_t)64 << 30 : (size_t)8 << 30);
		if (budget < enough && release_deferred()) {
			if (trace)
				std::fprintf(stderr, "[blasted_hip] placed_alloc: %zu MiB of room for the search; released what earlier searches held back\n", budget >> 20);
			budget = place_budget(total);
		}
	}
	const double budget_ms = place_budget_ms(bytes);
	long from_pool = 0;
	// Every candidate piece is looked at in an address range of its own (never used again) and only a piece that is kept is
	// mapped into the buffer: no address is ever mapped twice (see tracked_free for what that is about).
	const size_t max_tries = total / piece + 2 + budget / piece + pool.size();
	char *scratch = nullptr;
	if (hipMemAddressReserve(reinterpret_cast<void **>(&scratch), max_tries * piece, 0, nullptr, 0) != hipSuccess) {
		(void)hipGetLastError();
		std::lock_guard<std::mutex> lk(g_deferred_mu);
		g_deferred_release.insert(g_deferred_release.end(), pool.begin(), pool.end());
		return nullptr;  // (the buffer's own range is kept reserved: address space is not scarce)
	}
	size_t tries = 0;
	double *sink = nullptr;
	bool ok = hipMalloc(&sink, sizeof(double)) == hipSuccess;
	int misses = 0;           // consecutive pieces of the wrong class
	long unchecked = 0, turned = 0, second_best = 0;
	// pieces that meet the musts but not the preference: held, and used for the slots that are still empty when the
	// search for better ones ends (full-size pieces only: the last, shorter slot takes what comes)
	std::vector<hipMemGenericAllocationHandle_t> fallback;
	size_t searched_since_enough = 0;
	for (size_t at = 0; ok && at < total;) {
		const size_t sz = total - at < piece ? total - at : piece;
		const size_t slots_left = (total - at + piece - 1) / piece;
		if (!fallback.empty() && sz == piece) {
			// enough second-best pieces for every empty slot, and 32 GiB more looked at since: settle for them
			const bool enough = fallback.size() >= slots_left && searched_since_enough >= ((size_t)32 << 30);
			const bool out_of_budget = held_bytes + sz > budget || tries >= max_tries || now_ms() - t_start >= budget_ms;
			if (enough || out_of_budget) {
				hipMemGenericAllocationHandle_t hf = fallback.back();
				fallback.pop_back();
				char *where = static_cast<char *>(va) + at;
				hipError_t e = hipMemMap(where, sz, 0, hf, 0);
				if (e == hipSuccess)
					e = hipMemSetAccess(where, sz, &acc, 1);
				if (e != hipSuccess) {
					(void)hipMemRelease(hf);
					ok = false;
					break;
				}
				(void)hipMemRelease(hf);
				held_bytes -= sz;
				mapped.emplace_back(at, sz);
				g_place_stats.pieces++;
				second_best++;
				at += sz;
				continue;
			}
		}
		const bool pooled = sz == piece && !pool.empty() && now_ms() - t_start < budget_ms;
		// after two misses in a row: step over a larger stretch of the driver's free memory without looking at it
		if (!pooled && misses >= 2 && held_bytes + ((size_t)2 << 30) <= budget && now_ms() - t_start < budget_ms) {
			size_t sp = (size_t)1 << (30 + (misses < 6 ? misses - 1 : 5));  // 2, 4, 8, 16, 32 GiB
			while (held_bytes + sp > budget && sp > piece)
				sp >>= 1;
			hipMemGenericAllocationHandle_t hs;
			if (hipMemCreate(&hs, sp, &prop, 0) == hipSuccess) {
				held.push_back(HeldPiece{hs, sp});
				held_bytes += sp;
				if (fallback.size() >= slots_left)
					searched_since_enough += sp;
			} else
				(void)hipGetLastError();
		}
		hipMemGenericAllocationHandle_t hd;
		hipError_t e = hipSuccess;
		if (pooled) {
			hd = pool.back().h;
			pool.pop_back();
			held_bytes += sz;  // (booked like a fresh piece; taken off again below if it is kept)
		} else {
			e = hipMemCreate(&hd, sz, &prop, 0);
			if (e != hipSuccess && release_deferred()) {  // what earlier searches still hold may be what is missing
				(void)hipGetLastError();
				e = hipMemCreate(&hd, sz, &prop, 0);
			}
		}
		if (e != hipSuccess) {
			if (trace)
				std::fprintf(stderr, "[blasted_hip] placed_alloc: hipMemCreate(%zu) failed: %s\n", sz, hipGetErrorString(e));
			(void)hipGetLastError();
			ok = false;
			break;
		}
		char *where = static_cast<char *>(va) + at;
		if (pooled)
			held_bytes -= sz;
		const bool may_reject = (pooled || held_bytes + sz <= budget) && tries < max_tries && now_ms() - t_start < budget_ms;
		int rel = 0;
		if (may_reject) {
			char *look = scratch + (tries++) * piece;
			e = hipMemMap(look, sz, 0, hd, 0);
			if (e == hipSuccess)
				e = hipMemSetAccess(look, sz, &acc, 1);
			if (e == hipSuccess) {
				try {
					rel = class_fits(look, sz, h, sink, s);
				} catch (...) {
					if (trace)
						std::fprintf(stderr, "[blasted_hip] placed_alloc: probe failed: %s\n", last_error_text());
					rel = 0;
				}
				(void)hipStreamSynchronize(s);
				(void)hipMemUnmap(look, sz);
			} else
				(void)hipGetLastError();
		}
		if (rel == 1 && sz == piece) {  // the musts only: keep it in reserve, look on
			fallback.push_back(hd);
			held_bytes += sz;
			if (!pooled)
				misses++;
			continue;
		}
		if (rel < 0) {
			held.push_back(HeldPiece{hd, sz});
			if (!pooled) {  // (a pooled piece was allocated before this search began: not part of what IT holds back)
				held_bytes += sz;
				misses++;
			}
			turned++;
			g_place_stats.rejected++;
			if (fallback.size() >= slots_left)
				searched_since_enough += sz;
			continue;
		}
		e = hipMemMap(where, sz, 0, hd, 0);
		if (e == hipSuccess)
			e = hipMemSetAccess(where, sz, &acc, 1);
		if (e != hipSuccess) {
			if (trace)
				std::fprintf(stderr, "[blasted_hip] placed_alloc: mapping %zu bytes at %p failed: %s\n", sz, (void *)where, hipGetErrorString(e));
			(void)hipMemRelease(hd);
			ok = false;
			break;
		}
		if (rel == 0) {
			unchecked++;
			g_place_stats.gave_up++;
		}
		if (pooled)
			from_pool++;
		misses = 0;
		(void)hipMemRelease(hd);  // the mapping keeps the memory
		mapped.emplace_back(at, sz);
		g_place_stats.pieces++;
		at += sz;
	}
	{
		std::lock_guard<std::mutex> lk(g_deferred_mu);
		g_deferred_release.insert(g_deferred_release.end(), held.begin(), held.end());
		g_deferred_release.insert(g_deferred_release.end(), pool.begin(), pool.end());
		for (auto hf : fallback)
			g_deferred_release.push_back(HeldPiece{hf, piece});
	}
	if (sink)
		(void)hipFree(sink);
	if (!ok) {
		(void)hipDeviceSynchronize();
		for (const auto &pc : mapped)
			(void)hipMemUnmap(static_cast<char *>(va) + pc.first, pc.second);
		(void)hipGetLastError();  // (the address range is kept, see tracked_free)
		return nullptr;
	}
	trace_alloc(va, bytes);
	if (trace)
		std::fprintf(stderr, "[blasted_hip] placed %zu MiB at %p (%s %p): %zu pieces (%ld of them second best: the musts without the "
		             "preference, %ld from what earlier searches had turned down), %ld turned down, %ld kept unchecked, %zu MiB held back for the search, %.1f ms\n", total >> 20, va,
		             h.same ? "class of" : (h.prefer ? "preferably the class of" : "not the class of"), h.prefer ? h.prefer : ref,
		             mapped.size(), second_best, from_pool, turned, unchecked, held_bytes >> 20, now_ms() - t_start);
	AllocRegistry &r = alloc_registry();
	std::lock_guard<std::mutex> lk(r.mu);
	AllocRegistry::Rec rec;
	rec.bytes = total;
	rec.owner = tl_owner;
	rec.base = va;
	rec.va_bytes = total;
	rec.pieces = std::move(mapped);
	r.recs[va] = rec;
	if (tl_owner) {
		tl_owner->bytes_owned += (long)total;
		if (tl_owner->bytes_owned > tl_owner->bytes_peak)
			tl_owner->bytes_peak = tl_owner->bytes_owned;
	}
	g_place_stats.placed_buffers++;
	return va;
}

}  // namespace bhip
