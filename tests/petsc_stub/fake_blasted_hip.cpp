// fake_blasted_hip.cpp -- TEST INFRASTRUCTURE ONLY: a CPU stand-in for the C ABI of include/blasted_hip.h with NO
// numerical content, so that everything ABOVE the ABI -- the host C++ layer (blasted_amd/host/src), the PCSHELL
// glue, the mini-PETSc and the drivers under tests/cpp -- can run under AddressSanitizer / UBSan in the CPU
// container (GPU sanitizers are not available on the pool).  It is linked only into tests/cpp/build/*_asan by
// tests/test_host_sanitizers.py; nothing under blasted_amd/ knows it exists, and it is not a CPU fallback: every
// "result" is a fixed function of the input (z = 0.5 r) that no parity test would accept.
//
// What it does do, deliberately: every entry point reads each input array and writes each output array over
// the FULL extent the header documents (memcpy / loops the sanitizer instruments), keeps "device" memory on the
// host heap, and checks call order the way the real library does -- so a caller that passes a short array, a
// dangling pointer, or frees something twice is caught.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "blasted_hip.h"

namespace {

std::string g_err;
std::set<void *> g_buffers;
std::map<void *, unsigned long> g_pins;
std::mutex g_mu;

int fail(int code, const char *msg)
{
	g_err = msg;
	return code;
}

}  // namespace

struct blasted_hip_prec_s {
	int nbrows = 0, nnzb = 0, bs = 0, layout = 0;
	bool pattern = false, values = false, factored = false, jacobi = false, positions = false;
	std::vector<int> browptr, bcolind, diagind;
	std::vector<double> vals, ilu, dblocks, scale, ytemp;
	const double *borrowed = nullptr;
	long n() const { return (long)nbrows * bs; }
	long nv() const { return (long)nnzb * bs * bs; }
};

static double sink;  // keeps the reads alive
static void touch(const double *p, long n)
{
	double s = 0;
	for (long i = 0; i < n; i++)
		s += p[i];
	sink = s;
}
static void half(const double *r, double *z, long n)
{
	std::vector<double> t(r, r + n);  // r and z may alias in the relaxations
	for (long i = 0; i < n; i++)
		z[i] = 0.5 * t[i];
}

extern "C" {

const char *blasted_hip_last_error(void) { return g_err.c_str(); }
int blasted_hip_device_count(void) { return 1; }

int blasted_hip_create(blasted_hip_prec *out, int device, void *, int)
{
	if (!out || device != 0)
		return fail(BLASTED_HIP_EINVAL, "create: bad arguments");
	*out = new blasted_hip_prec_s;
	return BLASTED_HIP_OK;
}
int blasted_hip_destroy(blasted_hip_prec p)
{
	delete p;
	return BLASTED_HIP_OK;
}
int blasted_hip_synchronize(blasted_hip_prec) { return BLASTED_HIP_OK; }
int blasted_hip_device_synchronize(int) { return BLASTED_HIP_OK; }

int blasted_hip_set_pattern(blasted_hip_prec p, int nbrows, int nnzb, int bs, int layout, const int *browptr,
                            const int *bcolind, const int *diagind, int)
{
	if (p->pattern)
		return fail(BLASTED_HIP_ESTATE, "set_pattern: the pattern of an operator is set once");
	p->nbrows = nbrows;
	p->nnzb = nnzb;
	p->bs = bs;
	p->layout = layout;
	p->browptr.assign(browptr, browptr + nbrows + 1);
	p->bcolind.assign(bcolind, bcolind + nnzb);
	p->diagind.assign(diagind, diagind + nbrows);
	if (nbrows > 0 && p->browptr[nbrows] != nnzb)
		return fail(BLASTED_HIP_EINVAL, "set_pattern: browptr[nbrows] != nnzb");
	for (int i = 0; i < nbrows; i++)
		if (p->diagind[i] < p->browptr[i] || p->diagind[i] >= p->browptr[i + 1] || p->bcolind[p->diagind[i]] != i)
			return fail(BLASTED_HIP_EINVAL, "set_pattern: diagind does not point at the diagonal block");
	p->pattern = true;
	return BLASTED_HIP_OK;
}

int blasted_hip_set_values(blasted_hip_prec p, const double *vals, int loc)
{
	if (!p->pattern)
		return fail(BLASTED_HIP_ESTATE, "set_values before set_pattern");
	if (loc == BLASTED_HIP_DEVICE) {
		p->borrowed = vals;
		touch(vals, p->nv());
	} else {
		p->borrowed = nullptr;
		p->vals.assign(vals, vals + p->nv());
	}
	p->values = true;
	return BLASTED_HIP_OK;
}

int blasted_hip_ilu0_positions(blasted_hip_prec p)
{
	p->positions = true;
	return BLASTED_HIP_OK;
}
int blasted_hip_ilu0_positions_size(blasted_hip_prec, long *npairs)
{
	*npairs = 0;
	return BLASTED_HIP_OK;
}
int blasted_hip_ilu0_get_positions(blasted_hip_prec p, int *posptr, int *, int *)
{
	std::memset(posptr, 0, sizeof(int) * ((size_t)p->nnzb + 1));
	return BLASTED_HIP_OK;
}

int blasted_hip_ilu0_factorize(blasted_hip_prec p, int, int, int use_scaling, int, double *precinfo)
{
	if (!p->values)
		return fail(BLASTED_HIP_ESTATE, "ilu0_factorize before set_values");
	if (p->borrowed)
		touch(p->borrowed, p->nv());
	p->ilu = p->borrowed ? std::vector<double>(p->borrowed, p->borrowed + p->nv()) : p->vals;
	p->scale.assign((size_t)p->n(), use_scaling ? 1.0 : 0.0);
	p->ytemp.assign((size_t)p->n(), 0.0);
	if (precinfo)
		for (int i = 0; i < 6; i++)
			precinfo[i] = 0.25 * (i + 1);
	p->factored = true;
	return BLASTED_HIP_OK;
}

int blasted_hip_ilu0_apply(blasted_hip_prec p, const double *r, double *z, int, int apply_init, int, int)
{
	if (!p->factored)
		return fail(BLASTED_HIP_ESTATE, "ilu0_apply before ilu0_factorize");
	if (apply_init != BLASTED_HIP_INIT_A_ZERO && apply_init != BLASTED_HIP_INIT_A_JACOBI)
		return fail(BLASTED_HIP_EINVAL, " scalar_ilu0_apply: Invalid init type!");
	half(r, z, p->n());
	return BLASTED_HIP_OK;
}

int blasted_hip_jacobi_compute(blasted_hip_prec p)
{
	if (!p->values)
		return fail(BLASTED_HIP_ESTATE, "jacobi_compute before set_values");
	p->dblocks.assign((size_t)p->nbrows * p->bs * p->bs, 1.0);
	p->ytemp.assign((size_t)p->n(), 0.0);
	p->jacobi = true;
	return BLASTED_HIP_OK;
}
int blasted_hip_jacobi_apply(blasted_hip_prec p, const double *r, double *z, int)
{
	if (!p->jacobi)
		return fail(BLASTED_HIP_ESTATE, "jacobi_apply before jacobi_compute");
	half(r, z, p->n());
	return BLASTED_HIP_OK;
}
int blasted_hip_jacobi_relax(blasted_hip_prec p, const double *b, double *x, int maxits, int, double, double, double,
                             int *steps_done, int)
{
	if (!p->jacobi)
		return fail(BLASTED_HIP_ESTATE, "jacobi_relax before jacobi_compute");
	touch(x, p->n());
	half(b, x, p->n());
	if (steps_done)
		*steps_done = maxits;
	return BLASTED_HIP_OK;
}
int blasted_hip_sgs_apply(blasted_hip_prec p, const double *r, double *z, int, int, int, int)
{
	if (!p->jacobi)
		return fail(BLASTED_HIP_ESTATE, "sgs_apply before jacobi_compute");
	half(r, z, p->n());
	return BLASTED_HIP_OK;
}
int blasted_hip_sgs_relax(blasted_hip_prec p, const double *b, double *x, int, int, int)
{
	if (!p->jacobi)
		return fail(BLASTED_HIP_ESTATE, "sgs_relax before jacobi_compute");
	touch(x, p->n());
	half(b, x, p->n());
	return BLASTED_HIP_OK;
}
int blasted_hip_gs_relax(blasted_hip_prec p, const double *b, double *x, int, int, int)
{
	if (!p->jacobi)
		return fail(BLASTED_HIP_ESTATE, "gs_relax before jacobi_compute");
	touch(x, p->n());
	half(b, x, p->n());
	return BLASTED_HIP_OK;
}

int blasted_hip_level_schedule(blasted_hip_prec) { return BLASTED_HIP_OK; }
int blasted_hip_level_count(blasted_hip_prec, int *nlevels)
{
	*nlevels = 1;
	return BLASTED_HIP_OK;
}
int blasted_hip_level_stats(blasted_hip_prec, long *out4)
{
	out4[0] = 1;
	out4[1] = out4[2] = out4[3] = 0;
	return BLASTED_HIP_OK;
}
int blasted_hip_memory_stats(blasted_hip_prec, long *out4)
{
	out4[0] = out4[1] = out4[2] = 0;
	out4[3] = 0;
	std::lock_guard<std::mutex> lk(g_mu);
	for (auto &kv : g_pins)
		out4[3] += (long)kv.second;
	return BLASTED_HIP_OK;
}
int blasted_hip_get_levels(blasted_hip_prec p, int *level_of_row, int *rows_by_level, int *level_ptr)
{
	for (int i = 0; i < p->nbrows; i++) {
		if (level_of_row)
			level_of_row[i] = 0;
		if (rows_by_level)
			rows_by_level[i] = i;
	}
	if (level_ptr) {
		level_ptr[0] = 0;
		level_ptr[1] = p->nbrows;
	}
	return BLASTED_HIP_OK;
}

int blasted_hip_spmv(blasted_hip_prec p, const double *x, double *y, int)
{
	if (!p->values)
		return fail(BLASTED_HIP_ESTATE, "spmv before set_values");
	half(x, y, p->n());
	return BLASTED_HIP_OK;
}
int blasted_hip_gemv3(blasted_hip_prec p, double a, const double *x, double b, const double *y, double *z, int)
{
	if (!p->values)
		return fail(BLASTED_HIP_ESTATE, "gemv3 before set_values");
	for (long i = 0; i < p->n(); i++)
		z[i] = 0.5 * a * x[i] + b * y[i];
	return BLASTED_HIP_OK;
}

int blasted_hip_get_iluvals(blasted_hip_prec p, double *out)
{
	if (!p->factored)
		return fail(BLASTED_HIP_ESTATE, "iluvals is not available");
	std::memcpy(out, p->ilu.data(), sizeof(double) * (size_t)p->nv());
	return BLASTED_HIP_OK;
}
int blasted_hip_get_dblocks(blasted_hip_prec p, double *out)
{
	if (!p->jacobi)
		return fail(BLASTED_HIP_ESTATE, "dblocks is not available");
	std::memcpy(out, p->dblocks.data(), sizeof(double) * p->dblocks.size());
	return BLASTED_HIP_OK;
}
int blasted_hip_get_scale(blasted_hip_prec p, double *out)
{
	if (p->scale.empty())
		return fail(BLASTED_HIP_ESTATE, "scale is not available");
	std::memcpy(out, p->scale.data(), sizeof(double) * p->scale.size());
	return BLASTED_HIP_OK;
}
int blasted_hip_get_ytemp(blasted_hip_prec p, double *out)
{
	if (p->ytemp.empty())
		return fail(BLASTED_HIP_ESTATE, "ytemp is not available");
	std::memcpy(out, p->ytemp.data(), sizeof(double) * p->ytemp.size());
	return BLASTED_HIP_OK;
}
int blasted_hip_iluvals_device(blasted_hip_prec p, double **dev_ptr)
{
	*dev_ptr = p->ilu.data();
	return BLASTED_HIP_OK;
}

int blasted_hip_buffer_alloc(void **dev_ptr, unsigned long nbytes, int)
{
	*dev_ptr = std::malloc(nbytes ? nbytes : 1);
	std::lock_guard<std::mutex> lk(g_mu);
	g_buffers.insert(*dev_ptr);
	return BLASTED_HIP_OK;
}
int blasted_hip_buffer_free(void *dev_ptr)
{
	if (!dev_ptr)
		return BLASTED_HIP_OK;
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_buffers.erase(dev_ptr))
		return fail(BLASTED_HIP_EINVAL, "buffer_free: not a buffer of this library (or freed twice)");
	std::free(dev_ptr);
	return BLASTED_HIP_OK;
}
int blasted_hip_buffer_upload(void *dev_ptr, const void *host_ptr, unsigned long nbytes)
{
	std::memcpy(dev_ptr, host_ptr, nbytes);
	return BLASTED_HIP_OK;
}
int blasted_hip_buffer_download(void *host_ptr, const void *dev_ptr, unsigned long nbytes)
{
	std::memcpy(host_ptr, dev_ptr, nbytes);
	return BLASTED_HIP_OK;
}
int blasted_hip_host_register(void *host_ptr, unsigned long nbytes)
{
	if (!host_ptr || !nbytes)
		return fail(BLASTED_HIP_EINVAL, "host_register: null pointer or empty range");
	std::lock_guard<std::mutex> lk(g_mu);
	if (g_pins.count(host_ptr))
		return fail(BLASTED_HIP_ESTATE, "host_register: this address is registered already");
	// a registered range must be live memory over its whole extent
	volatile const char *c = static_cast<const char *>(host_ptr);
	char s = 0;
	for (unsigned long i = 0; i < nbytes; i += 512)
		s += c[i];
	s += c[nbytes - 1];
	(void)s;
	g_pins[host_ptr] = nbytes;
	return BLASTED_HIP_OK;
}
int blasted_hip_host_unregister(void *host_ptr)
{
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_pins.find(host_ptr);
	if (it == g_pins.end())
		return fail(BLASTED_HIP_ESTATE, "host_unregister: this address is not registered");
	// ... and must still be live when it is released (the hazard: memory freed while registered)
	volatile const char *c = static_cast<const char *>(host_ptr);
	char s = c[0] + c[it->second - 1];
	(void)s;
	g_pins.erase(it);
	return BLASTED_HIP_OK;
}
int blasted_hip_measure_read_stream(const void *, unsigned long, int, double *gbps)
{
	*gbps = 1.0;
	return BLASTED_HIP_OK;
}
int blasted_hip_set_tuning(const char *) { return BLASTED_HIP_OK; }
int blasted_hip_set_timing(blasted_hip_prec, int) { return BLASTED_HIP_OK; }
int blasted_hip_get_timing(blasted_hip_prec, double *out6, int)
{
	for (int i = 0; i < 6; i++)
		out6[i] = 0;
	return BLASTED_HIP_OK;
}

}  // extern "C"
