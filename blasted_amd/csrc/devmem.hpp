// devmem.hpp -- device memory of the library: per-operator accounting of every allocation (tracked_malloc / tracked_free
// are declared in ctx.hpp) and the class-aware placement of the sweeps' large buffers (devmem.hip).
#pragma once
#include "ctx.hpp"

#include <map>
#include <mutex>
#include <vector>

namespace bhip {

// the operator whose entry point runs on this thread: what it allocates is booked on it (capi.hip sets and clears it)
extern thread_local blasted_hip_prec tl_owner;
void forget_owner(blasted_hip_prec p);  // an operator that goes away must not be booked on any more

struct AllocRegistry {
	struct Rec {
		size_t bytes;
		blasted_hip_prec owner;
		void *base;  // what hipMalloc returned (differs from the key under "allocoff")
		// class-aware allocations (placed_alloc): a reserved address range of va_bytes with physical pieces mapped
		// into it, each (offset, size)
		size_t va_bytes = 0;
		std::vector<std::pair<size_t, size_t>> pieces;
	};
	std::mutex mu;
	std::map<void *, Rec> recs;
};
AllocRegistry &alloc_registry();

extern size_t g_alloc_offset;  // tuning "allocoff=BYTES" (placement studies)
extern int g_placement;        // 0: plain hipMalloc, 1: quick search (default), 2: thorough (BLASTED_HIP_PLACEMENT, tuning "placement=")

template <typename T>
inline T *dev_alloc(size_t count)
{
	T *p = nullptr;
	BHIP_CHECK(tracked_malloc(&p, sizeof(T) * (count ? count : 1)));
	return p;
}

inline void dev_free(void *p)
{
	if (p)
		(void)tracked_free(p);
}

struct PlaceStats {
	long placed_buffers = 0, pieces = 0, rejected = 0, gave_up = 0, probes = 0;
};
extern PlaceStats g_place_stats;

// What a buffer's pieces should be: MUST not share the class of `avoid` / `avoid2` (the 10 % constraint: the vector the
// sweep writes) and SHOULD share the class of `prefer` (the 2-3 % one: the other streams the sweep reads).  `same` is a
// must-share (the probes' and place_ytemp's use).
struct PlaceHint {
	const void *avoid = nullptr, *avoid2 = nullptr, *same = nullptr, *prefer = nullptr;
	size_t ref_bytes = 0;  // length of those vectors
	bool any() const { return avoid || avoid2 || same || prefer; }
};

int class_fits(void *piece, size_t piece_bytes, const PlaceHint &h, double *sink, hipStream_t s);
double probe_ms(const void *rd, size_t rd_bytes, void *wr, size_t wr_bytes, int reps, double *sink, hipStream_t s);
// device memory of `bytes` whose pieces satisfy the hint; nullptr: not applicable / not available (allocate plainly)
void *placed_alloc(size_t bytes, const PlaceHint &h, hipStream_t s);
bool release_deferred();  // hands back what the searches of this call still hold; true: there was something
bool trace_placement();   // BLASTED_HIP_TRACE_PLACEMENT
const char *last_error_text();  // (capi.hip) the calling thread's last error message

template <typename T>
inline T *dev_alloc_placed(size_t count, const PlaceHint &h, hipStream_t s)
{
	if (void *p = placed_alloc(sizeof(T) * count, h, s))
		return static_cast<T *>(p);
	return dev_alloc<T>(count);
}

}  // namespace bhip
