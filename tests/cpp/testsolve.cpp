// testsolve.cpp -- native (no PETSc) end-to-end driver through the C++ operator API, the counterpart
// of the reference's tests/testsolve.cpp + tests/solvers.cpp: read a Matrix-Market system, build the
// preconditioner through SRFactory, solve with BiCGSTAB / Richardson / GCR whose matrix-vector products and
// preconditioner applications run on the GPU, and compare with the known solution.
// Same command-line option names as the reference driver (tests/testsolve.cpp:133-187).
#undef NDEBUG
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "blockmatrices.hpp"
#include "coomatrix.hpp"
#include "solverfactory.hpp"

using namespace blasted;

struct Params {
	std::string solvertype = "bcgs", precontype = "jacobi", factinittype = "init_original",
	            applyinittype = "init_zero", mattype = "csr", storageorder = "colmajor", mat_file, b_file,
	            x_file = "NONE";
	int blocksize = 4, maxiter = 1000, nbuildsweeps = 1, napplysweeps = 1, threadchunksize = 256, restart = 30;
	double testtol = 1e-4, tol = 1e-6;
};

static Params parse(int argc, char **argv)
{
	Params p;
	std::map<std::string, std::string> kv;
	for (int i = 1; i + 1 < argc; i += 2) {
		std::string k = argv[i];
		if (k.rfind("--", 0) != 0) {
			std::cerr << "bad option " << k << "\n";
			std::exit(2);
		}
		kv[k.substr(2)] = argv[i + 1];
	}
	auto S = [&](const char *k, std::string &v) { if (kv.count(k)) v = kv[k]; };
	auto I = [&](const char *k, int &v) { if (kv.count(k)) v = std::atoi(kv[k].c_str()); };
	auto D = [&](const char *k, double &v) { if (kv.count(k)) v = std::atof(kv[k].c_str()); };
	S("solver_type", p.solvertype); S("preconditioner_type", p.precontype);
	S("fact_init_type", p.factinittype); S("apply_init_type", p.applyinittype);
	S("mat_type", p.mattype); S("storage_order", p.storageorder);
	S("mat_file", p.mat_file); S("b_file", p.b_file); S("x_file", p.x_file);
	I("block_size", p.blocksize); I("max_iter", p.maxiter); I("build_sweeps", p.nbuildsweeps);
	I("apply_sweeps", p.napplysweeps); I("thread_chunk_size", p.threadchunksize); I("solver_restart", p.restart);
	D("test_tol", p.testtol); D("solver_tol", p.tol);
	return p;
}

// PETSc binary files (the *.pmat fixtures the reference's PETSc drivers MatLoad / VecLoad) start with a
// big-endian class id; anything else is taken as Matrix Market
static long petsc_classid(const std::string &path)
{
	std::ifstream f(path, std::ios::binary);
	unsigned char h[4] = {0, 0, 0, 0};
	f.read(reinterpret_cast<char *>(h), 4);
	return f ? (long)(((unsigned)h[0] << 24) | ((unsigned)h[1] << 16) | ((unsigned)h[2] << 8) | h[3]) : -1;
}

// Input is the host library's (coomatrix.hpp), as in the reference's driver (tests/testsolve.cpp:60-107)
static std::vector<double> read_dense(const std::string &path)
{
	const device_vector<double> v =
	    petsc_classid(path) == 1211214 ? readPetscBinaryVector<double>(path) : readDenseMatrixMarket<double>(path);
	return std::vector<double>(v.begin(), v.end());
}

template <int bs>
static SRMatrixStorage<double, int> read_bsr(const std::string &path, const bool rowmajor)
{
	COOMatrix<double, int> coo;
	if (petsc_classid(path) == 1211216)
		coo.readPetscBinary(path);
	else
		coo.readMatrixMarket(path);
	return getSRMatrixFromCOO<double, int, bs>(coo, rowmajor ? "rowmajor" : "colmajor");
}

static double dot(const std::vector<double> &a, const std::vector<double> &b)
{
	double s = 0;
	for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
	return s;
}

struct SolveInfo { int iters; double relres; };

// right-preconditioned BiCGSTAB, the algorithm of the reference's test solver (tests/solvers.cpp:140-239)
static SolveInfo bicgstab(const SRMatrixView<double, int> &A, const Preconditioner<double, int> &M,
                          const std::vector<double> &rhs, std::vector<double> &x, double tol, int maxiter)
{
	const size_t n = rhs.size();
	std::vector<double> r(n), rhat(n), p(n, 0.0), v(n, 0.0), y(n), z(n), t(n);
	double omega = 1, rhoold = 1, alpha = 1, resnorm = 100;
	A.gemv3(-1.0, x.data(), 1.0, rhs.data(), r.data());
	rhat = r;
	const double bnorm = std::sqrt(dot(rhs, rhs));
	int step = 0;
	while (step < maxiter) {
		const double rho = dot(rhat, r);
		const double beta = rho * alpha / (rhoold * omega);
		for (size_t i = 0; i < n; i++) p[i] = r[i] + beta * p[i] - beta * omega * v[i];
		M.apply(p.data(), y.data());
		A.apply(y.data(), v.data());
		alpha = rho / dot(rhat, v);
		for (size_t i = 0; i < n; i++) r[i] -= alpha * v[i];
		M.apply(r.data(), z.data());
		A.apply(z.data(), t.data());
		omega = dot(t, r) / dot(t, t);
		for (size_t i = 0; i < n; i++) {
			x[i] += alpha * y[i] + omega * z[i];
			r[i] -= omega * t[i];
		}
		resnorm = std::sqrt(dot(r, r));
		if (resnorm / bnorm < tol) break;
		rhoold = rho;
		step++;
	}
	return {step + 1, resnorm / bnorm};
}

static SolveInfo richardson(const SRMatrixView<double, int> &A, const Preconditioner<double, int> &M,
                            const std::vector<double> &rhs, std::vector<double> &x, double tol, int maxiter)
{
	const size_t n = rhs.size();
	std::vector<double> s(n), d(n);
	const double bnorm = std::sqrt(dot(rhs, rhs));
	double rel = 1e300;
	int step = 0;
	while (step < maxiter) {
		A.gemv3(-1.0, x.data(), 1.0, rhs.data(), s.data());
		rel = std::sqrt(dot(s, s)) / bnorm;
		if (rel < tol) break;
		std::fill(d.begin(), d.end(), 0.0);  // relaxations (gs) read the output vector as their initial guess
		M.apply(s.data(), d.data());
		for (size_t i = 0; i < n; i++) x[i] += d[i];
		step++;
	}
	return {step, rel};
}

// right-preconditioned restarted GCR, the reference's flexible test solver (tests/solvers.cpp:247-352): the
// direction p_k = M r_k is kept beside q_k = A p_k, so M may differ from one application to the next
static SolveInfo gcr(const SRMatrixView<double, int> &A, const Preconditioner<double, int> &M,
                     const std::vector<double> &rhs, std::vector<double> &x, double tol, int maxiter, int nrestart)
{
	const size_t n = rhs.size();
	std::vector<double> res(n), z(n), beta(nrestart + 1), qq(nrestart);
	std::vector<std::vector<double>> p(nrestart, std::vector<double>(n)), q(nrestart, std::vector<double>(n));
	const double bnorm = std::sqrt(dot(rhs, rhs));
	double rel = 1.0;
	int step = 0;
	while (step < maxiter) {
		A.gemv3(-1.0, x.data(), 1.0, rhs.data(), res.data());
		M.apply(res.data(), p[0].data());
		A.apply(p[0].data(), q[0].data());
		qq[0] = dot(q[0], q[0]);
		for (int k = 0; k < nrestart; k++) {
			const double alpha = dot(res, q[k]) / qq[k];
			for (size_t i = 0; i < n; i++) {
				x[i] += alpha * p[k][i];
				res[i] -= alpha * q[k][i];
			}
			rel = std::sqrt(dot(res, res)) / bnorm;
			step++;
			if (rel < tol || k == nrestart - 1 || step >= maxiter) break;
			M.apply(res.data(), z.data());
			A.apply(z.data(), q[k + 1].data());
			p[k + 1] = z;
			for (int i = 0; i <= k; i++) beta[i] = -dot(q[k + 1], q[i]) / qq[i];
			for (int l = 0; l <= k; l++)
				for (size_t i = 0; i < n; i++) {
					p[k + 1][i] += beta[l] * p[l][i];
					q[k + 1][i] += beta[l] * q[l][i];
				}
			qq[k + 1] = dot(q[k + 1], q[k + 1]);
		}
		if (rel < tol) break;
	}
	return {step, rel};
}

template <int bs>
static int test_solve(const Params &params)
{
	const bool rm = params.storageorder == "rowmajor";
	SRMatrixView<double, int> *mat = nullptr;
	if (bs == 1)
		mat = new CSRMatrixView<double, int>(move_to_const<double, int>(read_bsr<1>(params.mat_file, false)));
	else if (rm)
		mat = new BSRMatrixView<double, int, bs, RowMajor>(move_to_const<double, int>(read_bsr<bs>(params.mat_file, true)));
	else
		mat = new BSRMatrixView<double, int, bs, ColMajor>(move_to_const<double, int>(read_bsr<bs>(params.mat_file, false)));
	SRMatrixStorage<const double, const int> cmat = move_to_const<double, int>(read_bsr<bs>(params.mat_file, rm));
	const std::vector<double> b = read_dense(params.b_file);
	std::printf("Read matrix with %d (block-)rows, %d nonzero blocks, block size %d\n", cmat.nbrows, cmat.nnzb, bs);

	SRFactory<double, int> fctry;
	AsyncSolverSettings aparams;
	aparams.scale = false;
	aparams.nbuildsweeps = params.nbuildsweeps;
	aparams.napplysweeps = params.napplysweeps;
	aparams.thread_chunk_size = params.threadchunksize;
	aparams.bs = bs;
	aparams.prectype = fctry.solverTypeFromString(params.precontype);
	aparams.fact_inittype = getFactInitFromString(params.factinittype);
	aparams.apply_inittype = getApplyInitFromString(params.applyinittype);
	aparams.blockstorage = rm ? RowMajor : ColMajor;
	aparams.relax = false;
	aparams.compute_precinfo = false;

	SRPreconditioner<double, int> *prec = fctry.create_preconditioner(std::move(cmat), aparams);
	prec->compute();

	std::vector<double> x(mat->dim(), 0.0);
	SolveInfo info;
	if (params.solvertype == "richardson")
		info = richardson(*mat, *prec, b, x, params.tol, params.maxiter);
	else if (params.solvertype == "bcgs")
		info = bicgstab(*mat, *prec, b, x, params.tol, params.maxiter);
	else if (params.solvertype == "gcr")
		info = gcr(*mat, *prec, b, x, params.tol, params.maxiter, params.restart);
	else {
		std::cerr << " ! Invalid solver option!\n";
		std::abort();
	}
	std::printf(" Num iters = %d, final rel res = %g\n", info.iters, info.relres);
	int rc = info.relres < params.tol ? 0 : 1;
	if (params.x_file != "NONE") {
		const std::vector<double> ans = read_dense(params.x_file);
		double l2 = 0;
		for (int i = 0; i < mat->dim(); i++) l2 += (x[i] - ans[i]) * (x[i] - ans[i]);
		l2 = std::sqrt(l2);
		std::printf(" L2 norm of error = %g\n", l2);
		if (!(l2 < params.testtol)) rc = 1;
	}
	delete prec;
	delete mat;
	return rc;
}

int main(int argc, char **argv)
{
	const Params p = parse(argc, argv);
	try {
		if (p.mattype == "csr") return test_solve<1>(p);
		int blocksize = p.blocksize;
		if (blocksize <= 0)  // --block_size 0: as MatLoad does, from <mat_file>.info (-matload_block_size)
			blocksize = petscBinaryBlockSize(p.mat_file);
		switch (blocksize) {
		case 3: return test_solve<3>(p);
		case 4: return test_solve<4>(p);
		case 5: return test_solve<5>(p);
		case 7: return test_solve<7>(p);
		default: std::cerr << "block size not built into the driver\n"; return 2;
		}
	} catch (const std::exception &e) {
		std::cerr << "exception: " << e.what() << "\n";
		return 3;
	}
}
