// kernels_aux.hip -- integer-only helpers: ILU(0) position lists (compute_ILU_positions_CSR_CSR,
// src/ilu_pattern.cpp:32-163, with inner_search of src/helper_algorithms.hpp:38-49 and the serial
// inclusive_scan of src/helper_algorithms.cpp), and the pattern validation the reference only assumes
// (ascending columns, src/ilu_pattern.cpp:51,67; diagonal block present in every row).
//
// The reference builds the lists serially in two passes over all stored entries.  Here each block-row
// is independent: one thread per row counts (pass 1) and fills (pass 2); the prefix sum in between is a
// three-phase device scan.  The result is bit-identical to the serial lists: pairs of one entry are
// emitted in ascending position k exactly as the reference's loop does.
#include "ctx.hpp"

namespace bhip {

__device__ __forceinline__ int inner_search_dev(const int *aind, int start, int end, int tofind)
{
	for (int j = start; j < end; j++)
		if (aind[j] == tofind)
			return j;
	return -1;
}

template <bool FILL>
__global__ __launch_bounds__(256) void ilu_positions_kernel(const Pattern pat, int *posptr,
                                                            int *lowerp, int *upperp)
{
	const int irow = blockIdx.x * blockDim.x + threadIdx.x;
	if (irow >= pat.nbrows)
		return;
	const int rbeg = pat.browptr[irow], rend = pat.browptr[irow + 1];
	for (int j = rbeg; j < rend; j++) {
		const int colj = pat.bcolind[j];
		const int klimit = (irow > colj) ? colj : irow;
		int cnt = 0;
		const int base = FILL ? posptr[j] : 0;
		for (int k = rbeg; k < rend && pat.bcolind[k] < klimit; k++) {
			const int krow = pat.bcolind[k];
			const int ipos = inner_search_dev(pat.bcolind, pat.diagind[krow], pat.browptr[krow + 1], colj);
			if (ipos > -1) {
				if (FILL) {
					lowerp[base + cnt] = k;
					upperp[base + cnt] = ipos;
				}
				cnt++;
			}
		}
		if (!FILL)
			posptr[j + 1] = cnt;
	}
}

// ---- inclusive scan of int32 (three-phase, recursive on the block sums)

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

__global__ __launch_bounds__(256) void scan_tile_kernel(int *data, long n, int *tilesums)
{
	__shared__ int wsum[4];
	const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
	int v[SCAN_ITEMS];
	int run = 0;
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; q++) {
		const long idx = base + q;
		run += (idx < n) ? data[idx] : 0;
		v[q] = run;
	}
	// scan of the per-thread totals: inside the wave, then across the 4 waves
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int incl = run;
	for (int off = 1; off < 64; off <<= 1) {
		const int o = __shfl_up(incl, off, 64);
		if (lane >= off)
			incl += o;
	}
	if (lane == 63)
		wsum[wave] = incl;
	__syncthreads();
	int woff = 0;
	for (int w = 0; w < wave; w++)
		woff += wsum[w];
	const int excl = woff + incl - run;
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; q++) {
		const long idx = base + q;
		if (idx < n)
			data[idx] = v[q] + excl;
	}
	if (tilesums && threadIdx.x == 255)
		tilesums[blockIdx.x] = woff + incl;
}

__global__ void scan_add_kernel(int *data, long n, const int *tilesums_incl)
{
	const long idx = (long)blockIdx.x * 256 + threadIdx.x + SCAN_TILE;  // tile 0 needs no offset
	if (idx < n)
		data[idx] += tilesums_incl[idx / SCAN_TILE - 1];
}

static void inclusive_scan_device(int *data, long n, hipStream_t s)
{
	if (n <= 0)
		return;
	const long ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	if (ntiles == 1) {
		hipLaunchKernelGGL(scan_tile_kernel, dim3(1), dim3(256), 0, s, data, n, (int *)nullptr);
		BHIP_CHECK(hipGetLastError());
		return;
	}
	int *sums = nullptr;
	BHIP_CHECK(tracked_malloc(&sums, sizeof(int) * ntiles));
	try {
		hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, data, n, sums);
		BHIP_CHECK(hipGetLastError());
		inclusive_scan_device(sums, ntiles, s);
		const long rest = n - SCAN_TILE;
		hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, s, data, n,
		                   sums);
		BHIP_CHECK(hipGetLastError());
		BHIP_CHECK(hipStreamSynchronize(s));
	} catch (...) {
		(void)tracked_free(sums);
		throw;
	}
	BHIP_CHECK(tracked_free(sums));
}

long run_ilu_positions(const Pattern &pat, int **posptr_out, int **lowerp_out, int **upperp_out,
                       hipStream_t s)
{
	int *posptr = nullptr, *lowerp = nullptr, *upperp = nullptr;
	const long nent = pat.nnzb;
	BHIP_CHECK(tracked_malloc(&posptr, sizeof(int) * (nent + 1)));
	try {
		BHIP_CHECK(hipMemsetAsync(posptr, 0, sizeof(int) * (nent + 1), s));
		const unsigned grid = (unsigned)((pat.nbrows + 255) / 256);
		if (grid) {
			hipLaunchKernelGGL(ilu_positions_kernel<false>, dim3(grid), dim3(256), 0, s, pat, posptr,
			                   (int *)nullptr, (int *)nullptr);
			BHIP_CHECK(hipGetLastError());
		}
		inclusive_scan_device(posptr, nent + 1, s);
		int total = 0;
		BHIP_CHECK(hipMemcpyAsync(&total, posptr + nent, sizeof(int), hipMemcpyDeviceToHost, s));
		BHIP_CHECK(hipStreamSynchronize(s));
		if (total < 0)
			BHIP_FAIL(BLASTED_HIP_EINVAL, "ILU position list overflows the int32 index type");
		BHIP_CHECK(tracked_malloc(&lowerp, sizeof(int) * (total > 0 ? total : 1)));
		BHIP_CHECK(tracked_malloc(&upperp, sizeof(int) * (total > 0 ? total : 1)));
		if (grid && total > 0) {
			hipLaunchKernelGGL(ilu_positions_kernel<true>, dim3(grid), dim3(256), 0, s, pat, posptr,
			                   lowerp, upperp);
			BHIP_CHECK(hipGetLastError());
		}
		*posptr_out = posptr;
		*lowerp_out = lowerp;
		*upperp_out = upperp;
		return total;
	} catch (...) {
		(void)tracked_free(posptr);
		(void)tracked_free(lowerp);
		(void)tracked_free(upperp);
		throw;
	}
}

// flags: 1 = unsorted/duplicate columns, 2 = diagind does not point at the diagonal block,
//        4 = column index out of range, 8 = browptr not monotone or wrong total
__global__ void validate_kernel(const Pattern pat, int *flags)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= pat.nbrows)
		return;
	int f = 0;
	const int b = pat.browptr[i], e = pat.browptr[i + 1];
	if (e > b)
		atomicMax(flags + 1, e - b);
	if (b > e || b < 0 || e > pat.nnzb)
		f |= 8;
	else {
		if (i == 0 && b != 0)
			f |= 8;
		if (i == pat.nbrows - 1 && e != pat.nnzb)
			f |= 8;
		for (int j = b; j < e; j++) {
			const int c = pat.bcolind[j];
			if (c < 0 || c >= pat.nbrows)
				f |= 4;
			if (j > b && pat.bcolind[j - 1] >= c)
				f |= 1;
		}
		const int d = pat.diagind[i];
		if (d < b || d >= e || pat.bcolind[d] != i)
			f |= 2;
	}
	if (f)
		atomicOr(flags, f);
}

int validate_pattern_device(const Pattern &pat, hipStream_t s, int *max_row_len)
{
	int *flags = nullptr;
	BHIP_CHECK(tracked_malloc(&flags, 2 * sizeof(int)));
	int h[2] = {0, 0};
	try {
		BHIP_CHECK(hipMemsetAsync(flags, 0, 2 * sizeof(int), s));
		const unsigned grid = (unsigned)((pat.nbrows + 255) / 256);
		if (grid) {
			hipLaunchKernelGGL(validate_kernel, dim3(grid), dim3(256), 0, s, pat, flags);
			BHIP_CHECK(hipGetLastError());
		}
		BHIP_CHECK(hipMemcpyAsync(h, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
		BHIP_CHECK(hipStreamSynchronize(s));
	} catch (...) {
		(void)tracked_free(flags);
		throw;
	}
	BHIP_CHECK(tracked_free(flags));
	if (max_row_len)
		*max_row_len = h[1];
	return h[0];
}

// dst[i] = vals[diagind[i]] per block-row: the diagonal blocks of a factor as one contiguous array
__global__ __launch_bounds__(256) void gather_diag_blocks_kernel(const Pattern pat, const double *__restrict__ vals,
                                                                double *__restrict__ dst)
{
	const long bs2 = (long)pat.bs * pat.bs;
	const long i = (long)blockIdx.x * 256 + threadIdx.x;
	if (i < (long)pat.nbrows * bs2) {
		const long row = i / bs2;
		dst[i] = vals[(long)pat.diagind[row] * bs2 + (i - row * bs2)];
	}
}

void launch_gather_diag_blocks(const Pattern &pat, const double *vals, double *dst, hipStream_t s)
{
	const long n = (long)pat.nbrows * pat.bs * pat.bs;
	if (n == 0)
		return;
	hipLaunchKernelGGL(gather_diag_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pat, vals, dst);
}

// Read-only stream over a buffer in the shape of the sweeps' value stream: a workgroup takes 64 KiB
// contiguous, 16 bytes per lane, non-temporal (blasted_hip_measure_read_stream).
__global__ __launch_bounds__(256) void read_stream_kernel(const double *__restrict__ in, double *__restrict__ sink,
                                                         const long n16)
{
	typedef double v2d __attribute__((ext_vector_type(2)));
	const long base = (long)blockIdx.x * 4096;
	double acc = 0.0;
#pragma unroll 4
	for (int k = threadIdx.x; k < 4096; k += 256) {
		const long i = base + k;
		if (i < n16) {
			const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(in) + i);
			acc += v.x + v.y;
		}
	}
	if (acc == 1.2345e300)  // never: keeps the loads alive
		sink[0] = acc;
}

typedef double rwp_d2 __attribute__((ext_vector_type(2)));
// Address-class probe (devmem.hip, placed_alloc; profiles/r04_placement_*.txt).  A read stream beside a write stream, the
// shape of a triangular sweep: workgroup chunk c reads 64 KiB of `rd` (16 bytes per lane, non-temporal) and rewrites
// 4 KiB of `wr` with what it holds (values unchanged).  On MI355X the 288 GiB of HBM3E fall into three classes of
// 96 GiB (presumably the three ranks of the 12-high stacks): a launch whose writes go to the class its reads come
// from takes 10-12 % longer than one whose writes go elsewhere -- which is what this kernel is timed for.
__global__ void __launch_bounds__(256) rw_probe_kernel(const rwp_d2 *__restrict__ rd, long nchunks, long passes, rwp_d2 *wr,
                                                       long wr_chunks, double *sink)
{
	double acc = 0;
	for (long c = blockIdx.x; c < nchunks * passes; c += gridDim.x) {
		const rwp_d2 *src = rd + (c % nchunks) * 4096 + threadIdx.x;
		rwp_d2 v[16];
#pragma unroll
		for (int k = 0; k < 16; k++)
			v[k] = __builtin_nontemporal_load(src + k * 256);
		rwp_d2 *dst = wr + (c % wr_chunks) * 256 + threadIdx.x;
		rwp_d2 w = *dst;
#pragma unroll
		for (int k = 0; k < 16; k++)
			acc += v[k].x + v[k].y;
		if (acc == 1.2345e300)  // never: keeps the loads alive and the store data-dependent
			w.x = acc;
		asm volatile("" : "+v"(w));  // the store stays although it writes back what was read
		*dst = w;
	}
	if (acc == 1.2345e300)
		sink[0] = acc;
}

// (`passes` walks over the read piece: a small piece is read several times so that a launch lasts long enough to be timed)
void launch_rw_probe(const void *rd, long rd_bytes, void *wr, long wr_bytes, double *sink, hipStream_t s, long passes)
{
	const long nchunks = rd_bytes >> 16, wr_chunks = wr_bytes >> 12;
	if (nchunks < 1 || wr_chunks < 1 || passes < 1)
		return;
	const unsigned grid = (unsigned)(nchunks * passes < 8192 ? nchunks * passes : 8192);
	hipLaunchKernelGGL(rw_probe_kernel, dim3(grid), dim3(256), 0, s, static_cast<const rwp_d2 *>(rd), nchunks, passes,
	                   static_cast<rwp_d2 *>(wr), wr_chunks, sink);
}

#ifdef BHIP_PROBES
// One lane walks the buffer with one dependent 8-byte load per `stride` bytes (probes build: how much of a
// buffer's address range one translation covers -- a stride of 2 MiB pays a page walk per load where the
// driver mapped the range in 2 MiB fragments and none where it is one large fragment).
__global__ void page_walk_kernel(const double *__restrict__ buf, long nloads, long stride8, double *sink)
{
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	long at = 0;
	double acc = 0;
	for (long k = 0; k < nloads; k++) {
		const double v = __builtin_nontemporal_load(buf + at);
		acc += v;
		at += stride8 + (v == 1.2345e300 ? 1 : 0);  // the next address depends on the loaded value
	}
	sink[0] = acc;
}

// A read stream whose consecutive `piece`-byte pieces come alternately from two buffers (piece a multiple of 64 KiB):
// what a range interleaved from two address classes would look like to a streaming kernel.
__global__ void __launch_bounds__(256) read2_probe_kernel(const rwp_d2 *__restrict__ r0, const rwp_d2 *__restrict__ r1,
                                                          long nchunks, long chunks_per_piece, double *sink)
{
	double acc = 0;
	for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
		const long piece = c / chunks_per_piece;
		const long within = (piece >> 1) * chunks_per_piece + c % chunks_per_piece;
		const rwp_d2 *src = ((piece & 1) ? r1 : r0) + within * 4096 + threadIdx.x;
		rwp_d2 v[16];
#pragma unroll
		for (int k = 0; k < 16; k++)
			v[k] = __builtin_nontemporal_load(src + k * 256);
#pragma unroll
		for (int k = 0; k < 16; k++)
			acc += v[k].x + v[k].y;
	}
	if (acc == 1.2345e300)
		sink[0] = acc;
}

void launch_read2_probe(const void *r0, const void *r1, long bytes_each, long piece_bytes, double *sink, hipStream_t s)
{
	const long nchunks = 2 * (bytes_each >> 16);
	hipLaunchKernelGGL(read2_probe_kernel, dim3(8192), dim3(256), 0, s, static_cast<const rwp_d2 *>(r0),
	                   static_cast<const rwp_d2 *>(r1), nchunks, piece_bytes >> 16, sink);
}

void launch_page_walk(const void *buf, long nloads, long stride_bytes, double *sink, hipStream_t s)
{
	hipLaunchKernelGGL(page_walk_kernel, dim3(1), dim3(64), 0, s, static_cast<const double *>(buf), nloads, stride_bytes / 8, sink);
}
#endif

void launch_read_stream(const void *buf, unsigned long nbytes, double *sink, hipStream_t s)
{
	const long n16 = (long)(nbytes / 16);
	if (n16 == 0)
		return;
	const unsigned grid = (unsigned)((n16 + 4095) / 4096);
	hipLaunchKernelGGL(read_stream_kernel, dim3(grid), dim3(256), 0, s, static_cast<const double *>(buf), sink, n16);
}

}  // namespace bhip
