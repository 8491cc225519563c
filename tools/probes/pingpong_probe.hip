// pingpong_probe.hip -- how long does one "publish -> seen -> publish back" hop between two waves take on gfx950,
// (a) on different XCDs, (b) on the same XCD, with agent-scope (sc1) or workgroup-scope (sc0) accesses?
// This is the per-level latency floor of every dependency-polling pass (exact solves, exact factorisation).
// build: hipcc --offload-arch=gfx950 -O3 -o pingpong_probe pingpong_probe.hip ; run: ./pingpong_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
			std::exit(1);                                                             \
		}                                                                             \
	} while (0)

template <int SCOPE>
__global__ void pingpong(unsigned long long *flags, int wa, int wb, int n, int *xcc, int limit)
{
	const int bid = blockIdx.x;
	if (threadIdx.x == 0) {
		unsigned v;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
		xcc[bid] = (int)(v & 0xf);
	}
	if (bid != wa && bid != wb)
		return;
	unsigned long long *mine = flags + (bid == wa ? 0 : 32), *theirs = flags + (bid == wa ? 32 : 0);  // 256 B apart
	if (threadIdx.x != 0)
		return;
	for (int i = 1; i <= n; i++) {
		if (bid == wa)
			__hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
		int spins = 0;
		while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE) < (unsigned long long)i) {
			if (++spins > limit) {
				xcc[64] = 1;  // gave up (the scope does not carry the value across)
				return;
			}
		}
		if (bid == wb)
			__hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
	}
}

template <int SCOPE>
static void run(const char *name, int wa, int wb, unsigned long long *flags, int *xcc)
{
	const int n = 20000;
	CHECK(hipMemset(flags, 0, 512));
	CHECK(hipMemset(xcc, 0, 65 * sizeof(int)));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL(pingpong<SCOPE>, dim3(64), dim3(64), 0, 0, flags, wa, wb, n, xcc, 2000000);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	int h[65];
	CHECK(hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost));
	std::printf("%-28s workgroups %2d (XCD %d) <-> %2d (XCD %d): %s%.3f us per round trip (two hops)\n", name, wa, h[wa], wb,
	            h[wb], h[64] ? "GAVE UP after " : "", 1e3 * ms / n);
}

int main()
{
	unsigned long long *flags;
	int *xcc;
	CHECK(hipMalloc(&flags, 512));
	CHECK(hipMalloc(&xcc, 65 * sizeof(int)));
	run<__HIP_MEMORY_SCOPE_AGENT>("agent scope, other XCD", 0, 1, flags, xcc);
	run<__HIP_MEMORY_SCOPE_AGENT>("agent scope, same XCD", 0, 8, flags, xcc);
	run<__HIP_MEMORY_SCOPE_AGENT>("agent scope, same XCD", 0, 16, flags, xcc);
	run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup scope, same XCD", 0, 8, flags, xcc);
	run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup scope, other XCD", 0, 1, flags, xcc);
	run<__HIP_MEMORY_SCOPE_SYSTEM>("system scope, other XCD", 0, 1, flags, xcc);
	return 0;
}
