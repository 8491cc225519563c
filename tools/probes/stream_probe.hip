// Practical HBM streaming ceiling of the box: a read-only pass (16-byte loads, a sum kept per lane)
// and a read+write pass over a buffer the size of the 256^3 bs=4 factor (15 GB), several launch shapes.
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const v2d *__restrict__ in, double *__restrict__ out, long n16)
{
	const long stride = (long)gridDim.x * 256;
	long i = (long)blockIdx.x * 256 + threadIdx.x;
	double acc = 0;
	for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
		v2d v[UNR];
#pragma unroll
		for (int u = 0; u < UNR; ++u)
			v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNR; ++u)
			acc += v[u].x + v[u].y;
	}
	for (; i < n16; i += stride) {
		const v2d v = in[i];
		acc += v.x + v.y;
	}
	if (acc == 1.2345e300)
		out[0] = acc;
}

// contiguous chunk per workgroup (the shape the sweep kernels use: 128 rows x 512+ bytes)
template <bool NT>
__global__ __launch_bounds__(256) void read_chunk_kernel(const v2d *__restrict__ in, double *__restrict__ out, long n16,
                                                       int per_wg)
{
	const long base = (long)blockIdx.x * per_wg;
	double acc = 0;
	for (int k = threadIdx.x; k < per_wg; k += 256) {
		const long i = base + k;
		if (i < n16) {
			const v2d v = NT ? __builtin_nontemporal_load(in + i) : in[i];
			acc += v.x + v.y;
		}
	}
	if (acc == 1.2345e300)
		out[0] = acc;
}

// cache-policy variants of the chunked read: MODE 0 default, 1 nt, 2 sc0 sc1 (system scope), 3 sc0 sc1 nt, 4 sc1, 5 sc0
template <int MODE>
__global__ __launch_bounds__(256) void read_policy_kernel(const v2d *__restrict__ in, double *__restrict__ out, long n16,
                                                        int per_wg)
{
	const long base = (long)blockIdx.x * per_wg;
	double acc = 0;
	for (int k = threadIdx.x; k < per_wg; k += 256) {
		const long i = base + k;
		if (i < n16) {
			v2d v;
			const v2d *ptr = in + i;
			if (MODE == 0)
				asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
			else if (MODE == 1)
				asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
			else if (MODE == 2)
				asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
			else if (MODE == 3)
				asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
			else if (MODE == 4)
				asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
			else
				asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
			acc += v.x + v.y;
		}
	}
	if (acc == 1.2345e300)
		out[0] = acc;
}

__global__ __launch_bounds__(256) void copy_kernel(const v2d *__restrict__ in, v2d *__restrict__ out, long n16)
{
	const long stride = (long)gridDim.x * 256;
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
		__builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

template <class F>
static double time_ms(F f, int reps = 10)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	f();
	f();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a, 0));
	for (int r = 0; r < reps; ++r)
		f();
	CK(hipEventRecord(b, 0));
	CK(hipEventSynchronize(b));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

// "rounds" mode: is a buffer's streaming rate a property of where it was allocated?
static int rounds_mode()
{
	const long bytes = 8564768768L;  // the diagonal+upper copy of the 256^3 bs=4 factor
	const long n16 = bytes / 16;
	double *sink;
	CK(hipMalloc(&sink, 8));
	const size_t spacers[] = {0, 0, 1u << 20, 3u << 20, 64u << 20, 1100u << 20, 0, 7u << 20, 300u << 20, 0};
	for (size_t sp : spacers) {
		void *spacer = nullptr;
		if (sp)
			CK(hipMalloc(&spacer, sp));
		v2d *in;
		CK(hipMalloc(&in, bytes));
		CK(hipMemset(in, 0, bytes));
		const int per_wg = 4096;
		const unsigned grid = (unsigned)((n16 + per_wg - 1) / per_wg);
		const double t = time_ms([&] { hipLaunchKernelGGL((read_chunk_kernel<true>), dim3(grid), dim3(256), 0, 0, in, sink, n16, per_wg); }, 20);
		std::printf("spacer %10zu B, buffer at %p: %6.3f ms  %.2f TB/s\n", sp, (void *)in, t, bytes / t / 1e9);
		CK(hipFree(in));
		if (spacer)
			CK(hipFree(spacer));
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc > 1)
		return rounds_mode();
	const long bytes = 15032385536L;  // the 256^3 bs=4 factor
	const long n16 = bytes / 16;
	v2d *in, *out;
	double *sink;
	CK(hipMalloc(&in, bytes));
	CK(hipMalloc(&out, bytes));
	CK(hipMalloc(&sink, 8));
	CK(hipMemset(in, 0, bytes));
	CK(hipMemset(out, 0, bytes));
	std::printf("buffer %.2f GB\n", bytes / 1e9);
	for (int wgs : {2048, 8192, 32768, 131072}) {
		double t;
		t = time_ms([&] { hipLaunchKernelGGL((read_kernel<1, false>), dim3(wgs), dim3(256), 0, 0, in, sink, n16); });
		std::printf("read  grid-stride u1 default   %7d WGs: %6.3f ms  %.2f TB/s\n", wgs, t, bytes / t / 1e9);
		t = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(wgs), dim3(256), 0, 0, in, sink, n16); });
		std::printf("read  grid-stride u4 default   %7d WGs: %6.3f ms  %.2f TB/s\n", wgs, t, bytes / t / 1e9);
		t = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, true>), dim3(wgs), dim3(256), 0, 0, in, sink, n16); });
		std::printf("read  grid-stride u4 nontemp   %7d WGs: %6.3f ms  %.2f TB/s\n", wgs, t, bytes / t / 1e9);
	}
	for (int per_wg : {1024, 4096, 16384}) {
		const unsigned grid = (unsigned)((n16 + per_wg - 1) / per_wg);
		double t = time_ms([&] { hipLaunchKernelGGL((read_chunk_kernel<false>), dim3(grid), dim3(256), 0, 0, in, sink, n16, per_wg); });
		std::printf("read  chunk %6d B/WG default  %7u WGs: %6.3f ms  %.2f TB/s\n", per_wg * 16, grid, t, bytes / t / 1e9);
		t = time_ms([&] { hipLaunchKernelGGL((read_chunk_kernel<true>), dim3(grid), dim3(256), 0, 0, in, sink, n16, per_wg); });
		std::printf("read  chunk %6d B/WG nontemp  %7u WGs: %6.3f ms  %.2f TB/s\n", per_wg * 16, grid, t, bytes / t / 1e9);
	}
#define POLICY(M, NAME) { const int per_wg = 4096; const unsigned grid = (unsigned)((n16 + per_wg - 1) / per_wg); \
		const double t = time_ms([&] { hipLaunchKernelGGL((read_policy_kernel<M>), dim3(grid), dim3(256), 0, 0, in, sink, n16, per_wg); }); \
		std::printf("read  chunk 64 KiB, one load in flight per lane, %-12s: %6.3f ms  %.2f TB/s\n", NAME, t, bytes / t / 1e9); }
	POLICY(0, "default") POLICY(1, "nt") POLICY(2, "sc0 sc1") POLICY(3, "sc0 sc1 nt") POLICY(4, "sc1") POLICY(5, "sc0")
	for (int wgs : {8192, 65536}) {
		const double t = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(wgs), dim3(256), 0, 0, in, out, n16); });
		std::printf("copy  grid-stride nontemp      %7d WGs: %6.3f ms  %.2f TB/s (read+write)\n", wgs, t, 2.0 * bytes / t / 1e9);
	}
	const double t = time_ms([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); });
	std::printf("hipMemcpy device-to-device                  : %6.3f ms  %.2f TB/s (read+write)\n", t, 2.0 * bytes / t / 1e9);
	return 0;
}
