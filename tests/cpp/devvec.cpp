// devvec.cpp -- device_vector<double> as the HIP buffer holder (the reference's
// include/device_container.hpp:19-20 "becomes the HIP buffer holder"): a device_vector is filled on the host,
// mirrored into HBM with to_device(), handed to the operators' device entry points (SRPreconditioner::apply_device,
// SRMatrixView::apply_device) through device_data(), and brought back with to_host().  The results go to
// <out>_z.bin / <out>_y.bin (float64) for tests/test_gpu_host_api.py to compare with the CPU oracle; the report on
// stdout says whether the host-vector members of the same operators gave the same bits and how the mirror behaves
// under copy, move and resize.
// usage: devvec --mat_file F --mat_type csr|bsr [--block_size N] --preconditioner_type T [--build_sweeps N]
//               [--apply_sweeps N] --out PREFIX
#undef NDEBUG
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <utility>

#include "blockmatrices.hpp"
#include "coomatrix.hpp"
#include "device_container.hpp"
#include "solverfactory.hpp"

using namespace blasted;

static long petsc_classid(const std::string &path)
{
	std::ifstream f(path, std::ios::binary);
	unsigned char h[4] = {0, 0, 0, 0};
	f.read(reinterpret_cast<char *>(h), 4);
	return f ? (long)(((unsigned)h[0] << 24) | ((unsigned)h[1] << 16) | ((unsigned)h[2] << 8) | h[3]) : -1;
}

template <int bs>
static SRMatrixStorage<double, int> read_bsr(const std::string &path)
{
	COOMatrix<double, int> coo;
	if (petsc_classid(path) == 1211216)
		coo.readPetscBinary(path);
	else
		coo.readMatrixMarket(path);
	return getSRMatrixFromCOO<double, int, bs>(coo, "colmajor");
}

static void write_bin(const std::string &path, const device_vector<double> &v)
{
	std::ofstream f(path, std::ios::binary);
	f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(double)));
}

template <int bs>
static int run(std::map<std::string, std::string> &kv)
{
	const std::string file = kv["--mat_file"], out = kv["--out"];
	SRMatrixView<double, int> *mat =
	    bs == 1 ? static_cast<SRMatrixView<double, int> *>(new CSRMatrixView<double, int>(move_to_const<double, int>(read_bsr<1>(file))))
	            : new BSRMatrixView<double, int, bs, ColMajor>(move_to_const<double, int>(read_bsr<bs>(file)));
	SRFactory<double, int> fctry;
	AsyncSolverSettings s;
	s.scale = false;
	s.nbuildsweeps = kv.count("--build_sweeps") ? std::atoi(kv["--build_sweeps"].c_str()) : 1;
	s.napplysweeps = kv.count("--apply_sweeps") ? std::atoi(kv["--apply_sweeps"].c_str()) : 1;
	s.thread_chunk_size = 128;
	s.bs = bs;
	s.prectype = fctry.solverTypeFromString(kv["--preconditioner_type"]);
	s.fact_inittype = INIT_F_ORIGINAL;
	s.apply_inittype = INIT_A_ZERO;
	s.blockstorage = ColMajor;
	s.relax = false;
	s.compute_precinfo = false;
	SRPreconditioner<double, int> *prec = fctry.create_preconditioner(move_to_const<double, int>(read_bsr<bs>(file)), s);
	prec->compute();

	const int n = mat->dim();
	device_vector<double> r(n), z(n, -7.0), y(n, -7.0);
	for (int i = 0; i < n; i++)
		r[i] = std::sin(0.37 * i) + 1.1;
	std::printf("n = %d\nmirror_before_upload = %d\n", n, r.device_data() != nullptr);

	// host -> HBM, the operators on the mirrors, HBM -> host
	const double *rd = r.to_device();
	double *zd = z.to_device(), *yd = y.to_device();
	std::printf("mirror_after_upload = %d\n", rd && zd && yd && rd == r.device_data() && zd == z.device_data());
	prec->apply_device(rd, zd);
	mat->apply_device(rd, yd);
	blasted::detail::device_synchronize();  // the operators run on their own streams
	std::printf("host_untouched_before_download = %d\n", z[0] == -7.0 && y[n - 1] == -7.0);
	z.to_host();
	y.to_host();
	write_bin(out + "_z.bin", z);
	write_bin(out + "_y.bin", y);

	// the host-vector members of the same operators: same bits for the deterministic operator types
	device_vector<double> zh(n, 0.0), yh(n, 0.0);
	prec->apply(r.data(), zh.data());
	mat->apply(r.data(), yh.data());
	double dz = 0, dy = 0;
	for (int i = 0; i < n; i++) {
		dz = std::fmax(dz, std::fabs(zh[i] - z[i]));
		dy = std::fmax(dy, std::fabs(yh[i] - y[i]));
	}
	std::printf("max_abs_diff_host_vs_device_apply = %.3e\nmax_abs_diff_host_vs_device_spmv = %.3e\n", dz, dy);

	// the mirror under copy / move / resize / a second upload
	device_vector<double> copy(z);
	std::printf("copy_has_no_mirror = %d\ncopy_equals_host = %d\n", copy.device_data() == nullptr, copy == z);
	const double *before = z.device_data();
	device_vector<double> moved(std::move(z));
	std::printf("move_keeps_mirror = %d\nmoved_from_has_none = %d\n", moved.device_data() == before, z.device_data() == nullptr);
	moved[0] = 123.5;          // a changed host array, uploaded again and read back into another vector's host side
	moved.to_device();
	device_vector<double> probe(n, 0.0);
	blasted::detail::device_buffer_download(probe.data(), moved.device_data(), sizeof(double) * (size_t)n);
	std::printf("second_upload_seen = %d\n", probe[0] == 123.5 && probe[n - 1] == moved[n - 1]);
	moved.resize(n / 2);
	const double *half = moved.to_device();  // a new size gets a new mirror
	blasted::detail::device_buffer_download(probe.data(), half, sizeof(double) * (size_t)(n / 2));
	std::printf("resized_upload_seen = %d\n", probe[0] == 123.5 && probe[n / 2 - 1] == moved[n / 2 - 1]);
	moved.release_device();
	std::printf("released = %d\ndone = 1\n", moved.device_data() == nullptr);
	delete prec;
	delete mat;
	return 0;
}

int main(int argc, char **argv)
{
	std::map<std::string, std::string> kv;
	for (int i = 1; i + 1 < argc; i += 2)
		kv[argv[i]] = argv[i + 1];
	try {
		if (kv["--mat_type"] == "csr")
			return run<1>(kv);
		int bs = kv.count("--block_size") ? std::atoi(kv["--block_size"].c_str()) : 4;
		switch (bs) {
		case 3: return run<3>(kv);
		case 4: return run<4>(kv);
		case 7: return run<7>(kv);
		default: std::cerr << "block size not built into the driver\n"; return 2;
		}
	} catch (const std::exception &e) {
		std::cerr << "exception: " << e.what() << "\n";
		return 3;
	}
}
