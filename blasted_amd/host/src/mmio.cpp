// mmio.cpp -- Matrix-Market / PETSc-binary input and COO -> CSR/BSR conversion (see blasted/mmio.hpp for the reference
// surface this provides and the deliberate differences).
#include "blasted/mmio.hpp"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>

namespace blasted {

MatrixReadException::MatrixReadException(const std::string &msg) : std::runtime_error(msg) {}

namespace {

std::vector<std::string> words(const std::string &line)
{
	std::istringstream iss(line);
	std::vector<std::string> w;
	for (std::string t; iss >> t;)
		w.push_back(t);
	return w;
}

std::string lower(std::string s)
{
	std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
	return s;
}

// banner: "%%MatrixMarket matrix <coordinate|array> <real|complex|integer|pattern> <general|...>"
MMDescription read_banner(std::ifstream &fin, const std::string &file)
{
	std::string line;
	if (!std::getline(fin, line))
		throw MatrixReadException(file + ": empty file");
	const std::vector<std::string> w = words(line);
	if (w.size() != 5)
		throw MatrixReadException(file + ": the Matrix Market banner needs five terms");
	if (w[0] != "%%MatrixMarket" || lower(w[1]) != "matrix")
		throw MatrixReadException(file + ": not a Matrix Market matrix file");
	MMDescription d;
	const std::string st = lower(w[2]), sc = lower(w[3]), mt = lower(w[4]);
	if (st == "coordinate")
		d.storagetype = COORDINATE;
	else if (st == "array")
		d.storagetype = ARRAY;
	else
		throw MatrixReadException(file + ": invalid storage type '" + w[2] + "'");
	if (sc == "real")
		d.scalartype = REAL;
	else if (sc == "complex")
		d.scalartype = COMPLEX;
	else if (sc == "integer")
		d.scalartype = INTEGER;
	else if (sc == "pattern")
		d.scalartype = PATTERN;
	else
		throw MatrixReadException(file + ": invalid scalar type '" + w[3] + "'");
	if (mt == "general")
		d.matrixtype = GENERAL;
	else if (mt == "symmetric")
		d.matrixtype = SYMMETRIC;
	else if (mt == "skew-symmetric" || mt == "skewsymmetric")
		d.matrixtype = SKEWSYMMETRIC;
	else if (mt == "hermitian")
		d.matrixtype = HERMITIAN;
	else
		throw MatrixReadException(file + ": invalid matrix type '" + w[4] + "'");
	return d;
}

// the size line: first line that is neither a comment nor blank
std::vector<long> read_sizes(std::ifstream &fin, const std::string &file, const size_t need)
{
	std::string line;
	while (std::getline(fin, line)) {
		const std::vector<std::string> w = words(line);
		if (w.empty() || w[0][0] == '%')
			continue;
		if (w.size() < need)
			throw MatrixReadException(file + ": not enough size information");
		std::vector<long> sizes;
		for (const std::string &t : w) {
			try {
				size_t used = 0;
				const long v = std::stol(t, &used);
				if (used != t.size() || v < 0)
					throw std::invalid_argument(t);
				sizes.push_back(v);
			} catch (const std::exception &) {
				throw MatrixReadException(file + ": invalid size '" + t + "'");
			}
		}
		return sizes;
	}
	throw MatrixReadException(file + ": no size line");
}

template <typename index>
index checked_index(const long v, const std::string &file)
{
	if (v > (long)std::numeric_limits<index>::max())
		throw MatrixReadException(file + ": size is too large for the index type");
	return (index)v;
}

}  // namespace

template <typename scalar>
device_vector<scalar> readDenseMatrixMarket(const std::string file)
{
	std::ifstream fin(file);
	if (!fin)
		throw MatrixReadException(file + ": could not be opened to read");
	const MMDescription d = read_banner(fin, file);
	if (d.matrixtype != GENERAL)
		throw MatrixReadException(file + ": dense matrix should be general");
	if (d.storagetype != ARRAY)
		throw MatrixReadException(file + ": matrix should be stored as dense (array)");
	const std::vector<long> sizes = read_sizes(fin, file, 2);
	const long total = sizes[0] * sizes[1];
	device_vector<scalar> vals((size_t)total);
	for (long i = 0; i < total; i++)
		if (!(fin >> vals[(size_t)i]))
			throw MatrixReadException(file + ": fewer values than the size line promises");
	return vals;
}

template device_vector<double> readDenseMatrixMarket<double>(const std::string file);
template device_vector<float> readDenseMatrixMarket<float>(const std::string file);

template <typename scalar, typename index>
COOMatrix<scalar, index>::COOMatrix()
{
}

template <typename scalar, typename index>
COOMatrix<scalar, index>::~COOMatrix()
{
}

template <typename scalar, typename index>
index COOMatrix<scalar, index>::numrows() const
{
	return nrows;
}

template <typename scalar, typename index>
index COOMatrix<scalar, index>::numcols() const
{
	return ncols;
}

template <typename scalar, typename index>
index COOMatrix<scalar, index>::numnonzeros() const
{
	return nnz;
}

template <typename scalar, typename index>
void COOMatrix<scalar, index>::readMatrixMarket(const std::string file)
{
	std::ifstream fin(file);
	if (!fin)
		throw MatrixReadException(file + ": could not be opened to read");
	const MMDescription d = read_banner(fin, file);
	if (d.storagetype != COORDINATE)
		throw MatrixReadException(file + ": COOMatrix can only read coordinate storage");
	if (d.scalartype == PATTERN || d.scalartype == COMPLEX)
		throw MatrixReadException(file + ": COOMatrix cannot read pattern or complex matrices");
	if (d.matrixtype != GENERAL)
		throw MatrixReadException(file + ": COOMatrix can only read general matrices");
	const std::vector<long> sizes = read_sizes(fin, file, 3);
	nrows = checked_index<index>(sizes[0], file);
	ncols = checked_index<index>(sizes[1], file);
	nnz = checked_index<index>(sizes[2], file);

	entries.resize((size_t)nnz);
	for (index k = 0; k < nnz; k++) {
		long ri, ci;
		scalar v;
		if (!(fin >> ri >> ci >> v))
			throw MatrixReadException(file + ": fewer entries than the size line promises");
		if (ri < 1 || ri > nrows || ci < 1 || ci > ncols)
			throw MatrixReadException(file + ": entry outside the matrix");
		entries[(size_t)k] = {(index)(ri - 1), (index)(ci - 1), v};  // the file is 1-based
	}

	sortEntries();
}

template <typename scalar, typename index>
void COOMatrix<scalar, index>::sortEntries()
{
	// by row, then by column; equal positions keep their file order
	std::stable_sort(entries.begin(), entries.end(), [](const Entry<scalar, index> &a, const Entry<scalar, index> &b) {
		return a.rowind != b.rowind ? a.rowind < b.rowind : a.colind < b.colind;
	});
	rowptr.assign((size_t)nrows + 1, 0);
	for (const Entry<scalar, index> &e : entries)
		rowptr[(size_t)e.rowind + 1]++;
	for (index i = 0; i < nrows; i++)
		rowptr[(size_t)i + 1] += rowptr[(size_t)i];
}

namespace {

constexpr long PETSC_MAT_CLASSID = 1211216, PETSC_VEC_CLASSID = 1211214;

std::vector<unsigned char> read_whole_file(const std::string &file)
{
	std::ifstream fin(file, std::ios::binary | std::ios::ate);
	if (!fin)
		throw MatrixReadException(file + ": could not be opened to read");
	const std::streamsize size = fin.tellg();
	fin.seekg(0);
	std::vector<unsigned char> raw((size_t)size);
	if (size > 0 && !fin.read(reinterpret_cast<char *>(raw.data()), size))
		throw MatrixReadException(file + ": read error");
	return raw;
}

// PETSc binary files are big-endian whatever the host
long be_int32(const unsigned char *p)
{
	const uint32_t u = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
	return (long)(int32_t)u;
}

double be_float64(const unsigned char *p)
{
	uint64_t u = 0;
	for (int i = 0; i < 8; i++)
		u = (u << 8) | p[i];
	double d;
	std::memcpy(&d, &u, sizeof d);
	return d;
}

}  // namespace

template <typename scalar, typename index>
void COOMatrix<scalar, index>::readPetscBinary(const std::string file)
{
	const std::vector<unsigned char> raw = read_whole_file(file);
	if (raw.size() < 16 || be_int32(&raw[0]) != PETSC_MAT_CLASSID)
		throw MatrixReadException(file + ": not a PETSc binary matrix");
	const long nr = be_int32(&raw[4]), nc = be_int32(&raw[8]), nz = be_int32(&raw[12]);
	if (nz < 0)
		throw MatrixReadException(file + ": dense PETSc binary matrices are not supported");
	if (nr < 0 || nc < 0)
		throw MatrixReadException(file + ": negative size");
	if (raw.size() != 16 + 4 * (size_t)nr + 12 * (size_t)nz)
		throw MatrixReadException(file + ": inconsistent PETSc binary matrix (file size)");
	nrows = checked_index<index>(nr, file);
	ncols = checked_index<index>(nc, file);
	nnz = checked_index<index>(nz, file);
	const unsigned char *const lens = &raw[16], *const cols = lens + 4 * (size_t)nr,
	                           *const vals = cols + 4 * (size_t)nz;
	entries.resize((size_t)nnz);
	size_t k = 0;
	for (long i = 0; i < nr; i++) {
		const long len = be_int32(lens + 4 * i);
		if (len < 0 || k + (size_t)len > (size_t)nz)
			throw MatrixReadException(file + ": inconsistent PETSc binary matrix (row lengths)");
		for (long q = 0; q < len; q++, k++) {
			const long ci = be_int32(cols + 4 * k);
			if (ci < 0 || ci >= nc)
				throw MatrixReadException(file + ": entry outside the matrix");
			entries[k] = {(index)i, (index)ci, (scalar)be_float64(vals + 8 * k)};
		}
	}
	if (k != (size_t)nz)
		throw MatrixReadException(file + ": inconsistent PETSc binary matrix (row lengths)");
	sortEntries();
}

template <typename scalar>
device_vector<scalar> readPetscBinaryVector(const std::string file)
{
	const std::vector<unsigned char> raw = read_whole_file(file);
	if (raw.size() < 8 || be_int32(&raw[0]) != PETSC_VEC_CLASSID)
		throw MatrixReadException(file + ": not a PETSc binary vector");
	const long n = be_int32(&raw[4]);
	if (n < 0 || raw.size() != 8 + 8 * (size_t)n)
		throw MatrixReadException(file + ": inconsistent PETSc binary vector");
	device_vector<scalar> v((size_t)n);
	for (long i = 0; i < n; i++)
		v[(size_t)i] = (scalar)be_float64(&raw[8 + 8 * (size_t)i]);
	return v;
}

template device_vector<double> readPetscBinaryVector<double>(const std::string file);
template device_vector<float> readPetscBinaryVector<float>(const std::string file);

int petscBinaryBlockSize(const std::string file)
{
	std::ifstream fin(file + ".info");
	int bs = 1;
	for (std::string line; fin && std::getline(fin, line);) {
		const std::vector<std::string> w = words(line);
		if (w.size() == 2 && w[0] == "-matload_block_size") {
			try {
				bs = std::stoi(w[1]);
			} catch (const std::exception &) {
				throw MatrixReadException(file + ".info: invalid -matload_block_size");
			}
		}
	}
	if (bs < 1)
		throw MatrixReadException(file + ".info: invalid -matload_block_size");
	return bs;
}

template <typename scalar, typename index>
SRMatrixStorage<scalar, index> COOMatrix<scalar, index>::convertToCSR() const
{
	SRMatrixStorage<scalar, index> m;
	m.nbrows = nrows;
	m.browptr.resize(nrows + 1);
	m.bcolind.resize(nnz);
	m.vals.resize(nnz);
	m.diagind.resize(nrows);
	for (index k = 0; k < nnz; k++) {
		m.bcolind[k] = entries[(size_t)k].colind;
		m.vals[k] = entries[(size_t)k].value;
	}
	for (index i = 0; i <= nrows; i++)
		m.browptr[i] = rowptr[(size_t)i];
	for (index i = 0; i < nrows; i++) {
		m.diagind[i] = -1;
		for (index k = rowptr[(size_t)i]; k < rowptr[(size_t)i + 1]; k++)
			if (entries[(size_t)k].colind == i)
				m.diagind[i] = k;
	}
	if (nrows > 0)
		m.browendptr.wrap(&m.browptr[1], nrows);
	m.nnzb = nnz;
	m.nbstored = nnz;
	return m;
}

template <typename scalar, typename index>
template <int bs, StorageOptions stor>
SRMatrixStorage<scalar, index> COOMatrix<scalar, index>::convertToBSR() const
{
	static_assert(bs > 0, "Block size must be positive!");
	static_assert(stor == RowMajor || stor == ColMajor, "Invalid storage option!");
	if (nrows != ncols)
		throw std::invalid_argument("convertToBSR: the matrix must be square");
	if (nrows % bs != 0)
		throw std::invalid_argument("convertToBSR: the dimension must be a multiple of the block size");
	const index nb = nrows / bs;

	SRMatrixStorage<scalar, index> m;
	m.nbrows = nb;
	m.browptr.resize(nb + 1);
	m.diagind.resize(nb);
	m.browptr[0] = 0;

	// pass 1: the sorted, distinct block columns of every block-row (merge of its bs sorted scalar rows)
	std::vector<index> bcols, rowcols;
	for (index ib = 0; ib < nb; ib++) {
		rowcols.clear();
		for (index k = rowptr[(size_t)ib * bs]; k < rowptr[(size_t)(ib + 1) * bs]; k++)
			rowcols.push_back(entries[(size_t)k].colind / bs);
		std::sort(rowcols.begin(), rowcols.end());
		rowcols.erase(std::unique(rowcols.begin(), rowcols.end()), rowcols.end());
		m.diagind[ib] = -1;
		for (size_t q = 0; q < rowcols.size(); q++) {
			if (rowcols[q] == ib)
				m.diagind[ib] = (index)(bcols.size() + q);
		}
		bcols.insert(bcols.end(), rowcols.begin(), rowcols.end());
		if (bcols.size() > (size_t)std::numeric_limits<index>::max())
			throw std::overflow_error("convertToBSR: too many blocks for the index type");
		m.browptr[ib + 1] = (index)bcols.size();
	}
	const index nnzb = (index)bcols.size();
	m.bcolind.resize(nnzb);
	m.vals.resize(nnzb * bs * bs);
	for (index j = 0; j < nnzb; j++)
		m.bcolind[j] = bcols[(size_t)j];
	for (index q = 0; q < nnzb * bs * bs; q++)
		m.vals[q] = 0;

	// pass 2: scatter the entries into their blocks
	for (index i = 0; i < nrows; i++) {
		const index ib = i / bs, r = i % bs;
		const index *const rb = &m.bcolind[0] + m.browptr[ib];
		const index *const re = &m.bcolind[0] + m.browptr[ib + 1];
		for (index k = rowptr[(size_t)i]; k < rowptr[(size_t)i + 1]; k++) {
			const index j = entries[(size_t)k].colind, jb = j / bs, c = j % bs;
			const index pos = (index)(std::lower_bound(rb, re, jb) - &m.bcolind[0]);
			const index off = stor == RowMajor ? r * bs + c : c * bs + r;
			m.vals[pos * bs * bs + off] = entries[(size_t)k].value;
		}
	}
	if (nb > 0)
		m.browendptr.wrap(&m.browptr[1], nb);
	m.nnzb = nnzb;
	m.nbstored = nnzb;
	return m;
}

template <typename scalar, typename index>
const std::vector<Entry<scalar, index>> &COOMatrix<scalar, index>::getEntries() const
{
	return entries;
}

template <typename scalar, typename index>
const std::vector<index> &COOMatrix<scalar, index>::getRowPtrs() const
{
	return rowptr;
}

template <typename scalar, typename index, int bs>
SRMatrixStorage<scalar, index> getSRMatrixFromCOO(const COOMatrix<scalar, index> &coom, const std::string storageorder)
{
	if (bs == 1)
		return coom.convertToCSR();
	if (storageorder == "rowmajor")
		return coom.template convertToBSR<bs, RowMajor>();
	if (storageorder == "colmajor")
		return coom.template convertToBSR<bs, ColMajor>();
	throw std::runtime_error("getSRMatrixFromCOO: invalid storage order!");
}

template class COOMatrix<double, int>;

#define BLASTED_COO_INST(BS)                                                                        \
	template SRMatrixStorage<double, int> getSRMatrixFromCOO<double, int, BS>(const COOMatrix<double, int> &, \
	                                                                         const std::string);
BLASTED_COO_INST(1)
BLASTED_COO_INST(2)
BLASTED_COO_INST(3)
BLASTED_COO_INST(4)
BLASTED_COO_INST(5)
BLASTED_COO_INST(7)
BLASTED_COO_INST(8)
#undef BLASTED_COO_INST

}  // namespace blasted
