// Does the 8-byte alignment of 200-byte (5x5) blocks cost request rate?  A stream of blocks read the way
// kernels_sweepodd.hip / kernels_factorodd.hip read them -- 16 lanes per block, 13 of them active, 16 bytes per lane --
// with the blocks 200 bytes apart (every other block 8 mod 16) and 208 bytes apart (all 16-byte aligned); plus the
// same bytes as a plain dense 16-byte stream.  Cache-resident repeat (a 64 MB window) and HBM stream (8 GB).
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/block_align_probe.hip -o /tmp/bap && /tmp/bap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
typedef v2d v2du __attribute__((aligned(8)));

// one wave = 4 blocks per step; a workgroup walks `per_wg` consecutive blocks
template <int STRIDE_D, bool WRITE>
__global__ __launch_bounds__(256) void block_kernel(double *__restrict__ buf, double *__restrict__ out, long nblocks,
                                                    int per_wg)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane >> 4, t = lane & 15;
	const bool act = t < 13;
	const long b0 = (long)blockIdx.x * per_wg;
	double acc = 0;
	for (int s = 0; s < per_wg; s += 16) {
		const long b = b0 + s + wave * 4 + g;
		if (b < nblocks && act) {
			double *p = buf + b * STRIDE_D + 2 * (t < 12 ? t : 11) + (t == 12 ? 1 : 0);
			if (STRIDE_D % 2 == 0) {
				v2d v = *reinterpret_cast<const v2d *>(buf + b * STRIDE_D + 2 * t);
				acc += v.x + v.y;
				if (WRITE) {
					v.x += 1.0;
					*reinterpret_cast<v2d *>(buf + b * STRIDE_D + 2 * t) = v;
				}
			} else {
				(void)p;
				const long off = b * STRIDE_D + (t < 12 ? 2 * t : STRIDE_D - 2);
				v2d v = *reinterpret_cast<const v2du *>(buf + off);
				acc += v.x + v.y;
				if (WRITE) {
					v.x += 1.0;
					*reinterpret_cast<v2du *>(buf + off) = v;
				}
			}
		}
	}
	if (acc == 1.2345e300)
		out[0] = acc;
}

template <int STRIDE_D, bool WRITE>
static void run(const char *name, double *buf, double *out, long nblocks, int reps)
{
	const int per_wg = 2048;
	const unsigned grid = (unsigned)((nblocks + per_wg - 1) / per_wg);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((block_kernel<STRIDE_D, WRITE>), dim3(grid), dim3(256), 0, 0, buf, out, nblocks, per_wg);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	for (int r = 0; r < reps; r++)
		hipLaunchKernelGGL((block_kernel<STRIDE_D, WRITE>), dim3(grid), dim3(256), 0, 0, buf, out, nblocks, per_wg);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	ms /= reps;
	const double useful = (double)nblocks * 200.0 * (WRITE ? 2 : 1);
	std::printf("%-46s %9ld blocks  %8.3f ms  %6.2f TB/s of block bytes  %7.2f G blocks/s\n", name, nblocks, ms,
	            useful / ms / 1e9, nblocks / ms / 1e6);
}

int main()
{
	const size_t bytes = 9ull << 30;
	double *buf, *out;
	CK(hipMalloc(&buf, bytes));
	CK(hipMalloc(&out, 64));
	CK(hipMemset(buf, 0, bytes));
	for (int pass = 0; pass < 2; pass++) {
		const long nb = pass == 0 ? 300000 : 30000000;  // 60 MB (cache resident) / 6 GB (HBM stream); 30 M x 256 B = 7.7 GB < the 9 GB buffer
		const int reps = pass == 0 ? 200 : 5;
		std::printf(pass == 0 ? "--- 60 MB window, repeated (L2 / Infinity Cache resident)\n" : "--- 6-7.7 GB stream\n");
		run<25, false>("read  200-byte blocks, 8-byte aligned", buf, out, nb, reps);
		run<26, false>("read  208-byte stride, 16-byte aligned", buf, out, nb, reps);
		run<32, false>("read  256-byte stride, 16-byte aligned", buf, out, nb, reps);
		run<25, true>("read+write 200-byte blocks, 8-byte aligned", buf, out, nb, reps);
		run<26, true>("read+write 208-byte stride, 16-byte aligned", buf, out, nb, reps);
		run<32, true>("read+write 256-byte stride, 16-byte aligned", buf, out, nb, reps);
	}
	return 0;
}
