// Dumps what the host library's Matrix-Market / PETSc-binary (*.pmat) route (coomatrix.hpp) makes of a file, for
// tests/test_host_build.py: "coo_dump <file.mtx> <bs> <rowmajor|colmajor> <out.bin>" writes
// int32 nbrows, nnzb, then browptr[nbrows+1], bcolind[nnzb], diagind[nbrows] (int32) and vals (float64);
// "coo_dump <file.mtx> dense <out.bin>" writes int64 count and the values.  Exit code 3 with the message on
// stderr when the library throws MatrixReadException.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "coomatrix.hpp"

using namespace blasted;

template <int bs>
static int dump(const COOMatrix<double, int> &c, const std::string &order, const char *out)
{
	const SRMatrixStorage<double, int> m = getSRMatrixFromCOO<double, int, bs>(c, order);
	FILE *f = std::fopen(out, "wb");
	const int hdr[2] = {m.nbrows, m.nnzb};
	std::fwrite(hdr, 4, 2, f);
	std::fwrite(&m.browptr[0], 4, (size_t)m.nbrows + 1, f);
	std::fwrite(&m.bcolind[0], 4, (size_t)m.nnzb, f);
	std::fwrite(&m.diagind[0], 4, (size_t)m.nbrows, f);
	std::fwrite(&m.vals[0], 8, (size_t)m.nnzb * bs * bs, f);
	std::fclose(f);
	if (m.browendptr.size() != m.nbrows || (m.nbrows > 0 && &m.browendptr[0] != &m.browptr[1]) || m.nbstored != m.nnzb) {
		std::cerr << "browendptr / nbstored inconsistent\n";
		return 4;
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 4) {
		std::cerr << "usage: coo_dump file.mtx <bs> <rowmajor|colmajor> out.bin | coo_dump file.mtx dense out.bin\n";
		return 2;
	}
	try {
		if (std::strcmp(argv[2], "dense") == 0) {
			const std::string path = argv[1];
			const bool petsc = path.size() > 5 && path.compare(path.size() - 5, 5, ".pmat") == 0;
			const device_vector<double> v = petsc ? readPetscBinaryVector<double>(path) : readDenseMatrixMarket<double>(path);
			FILE *f = std::fopen(argv[3], "wb");
			const long long n = (long long)v.size();
			std::fwrite(&n, 8, 1, f);
			std::fwrite(v.data(), 8, v.size(), f);
			std::fclose(f);
			return 0;
		}
		const std::string path = argv[1];
		const bool petsc = path.size() > 5 && path.compare(path.size() - 5, 5, ".pmat") == 0;
		// "info": the block size MatLoad would use, from <file>.info
		const int bs = std::strcmp(argv[2], "info") == 0 ? petscBinaryBlockSize(path) : std::atoi(argv[2]);
		COOMatrix<double, int> c;
		if (petsc)
			c.readPetscBinary(path);
		else
			c.readMatrixMarket(path);
		switch (bs) {
		case 1: return dump<1>(c, argv[3], argv[4]);
		case 2: return dump<2>(c, argv[3], argv[4]);
		case 3: return dump<3>(c, argv[3], argv[4]);
		case 4: return dump<4>(c, argv[3], argv[4]);
		case 5: return dump<5>(c, argv[3], argv[4]);
		case 7: return dump<7>(c, argv[3], argv[4]);
		case 8: return dump<8>(c, argv[3], argv[4]);
		default: std::cerr << "block size not instantiated\n"; return 2;
		}
	} catch (const MatrixReadException &e) {
		std::cerr << "MatrixReadException: " << e.what() << "\n";
		return 3;
	}
}
