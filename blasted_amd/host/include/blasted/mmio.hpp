// mmio.hpp -- Matrix Market and PETSc-binary input for the native (non-PETSc) route.  It provides what a driver written against
// the reference's include/coomatrix.hpp uses (tests/testsolve.cpp:60-107): COOMatrix, getSRMatrixFromCOO,
// readDenseMatrixMarket, MatrixReadException and the MM* header enums; `coomatrix.hpp` forwards here.
// Deliberate differences:
//  * convertToBSR orders the blocks of a block-row by ascending block column, whatever the order of the
//    entries in the file (the reference appends blocks by first appearance, src/coomatrix.cpp:329-353, which
//    breaks the "lower blocks come before diagind" assumption of the ILU/SGS kernels for unsorted files);
//  * rows without entries are allowed (the reference asserts that there are none, :236-247);
//  * every malformed-file condition throws MatrixReadException (the reference aborts on some, :52-83);
//  * constructBSRMatrixFromMatrixMarketFile is not provided: it returns the reference's owning BSRMatrix
//    assembly class, which is outside this backend -- use getSRMatrixFromCOO and a
//    BSRMatrixView / CSRMatrixView.
#pragma once

#include <limits>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "blasted/storage.hpp"
#include "blasted/types.hpp"

namespace blasted {

/// Thrown when a matrix or vector file is malformed or of an unsupported kind
class MatrixReadException : public std::runtime_error {
public:
	explicit MatrixReadException(const std::string &msg);
};

/// One stored entry, zero-based position
template <typename scalar, typename index>
struct Entry {
	index rowind;
	index colind;
	scalar value;
};

/// Coordinate-format sparse matrix as read from a "coordinate real|integer general" file
template <typename scalar, typename index>
class COOMatrix {
	static_assert(!std::is_const<scalar>::value && !std::is_const<index>::value, "mutable scalar and index types");
	static_assert(std::numeric_limits<index>::is_integer && std::numeric_limits<index>::is_signed,
	              "signed integer index type");

public:
	COOMatrix();
	virtual ~COOMatrix();

	/// Reads the file (1-based there, 0-based here); afterwards the entries are sorted by (row, column) and
	/// getRowPtrs() delimits the rows
	void readMatrixMarket(const std::string file);

	/// Reads a PETSc binary AIJ matrix (what the reference's PETSc drivers load with MatLoad: the *.pmat
	/// fixtures of tests/input): big-endian {int32 classid 1211216, rows, cols, nnz, int32 rowlengths[rows],
	/// int32 colidx[nnz], float64 values[nnz]}.  Same post-conditions as readMatrixMarket.
	void readPetscBinary(const std::string file);

	index numrows() const;
	index numcols() const;
	index numnonzeros() const;
	const std::vector<Entry<scalar, index>> &getEntries() const;
	const std::vector<index> &getRowPtrs() const;

	/// New CSR matrix owning its arrays; nbrows = number of rows; diagind = -1 where a row has no diagonal
	SRMatrixStorage<scalar, index> convertToCSR() const;

	/// New BSR matrix owning its arrays (square matrix whose dimension is a multiple of bs); the blocks of a
	/// block-row ascend in block column, entries a stored block lacks are zero
	template <int bs, StorageOptions stor>
	SRMatrixStorage<scalar, index> convertToBSR() const;

protected:
	void sortEntries();  // by (row, column), stable; fills rowptr

	index nrows = 0, ncols = 0, nnz = 0;
	std::vector<Entry<scalar, index>> entries;  // sorted by (row, column)
	std::vector<index> rowptr;                  // nrows + 1 offsets into entries
};

/// CSR (bs == 1) or BSR with "rowmajor" / "colmajor" blocks out of a COO matrix
template <typename scalar, typename index, int bs>
SRMatrixStorage<scalar, index> getSRMatrixFromCOO(const COOMatrix<scalar, index> &coo_mat,
                                                 const std::string block_storage_order);

/// The values of a dense "array ... general" file, in file order
template <typename scalar>
device_vector<scalar> readDenseMatrixMarket(const std::string file);

/// The values of a PETSc binary vector (VecLoad's format: big-endian {int32 classid 1211214, n, float64 values[n]})
template <typename scalar>
device_vector<scalar> readPetscBinaryVector(const std::string file);

/// The block size MatLoad would give the matrix in `file`: `-matload_block_size N` in `file`.info, 1 without it
int petscBinaryBlockSize(const std::string file);

/// What the banner line of a Matrix Market file says (names as in the reference, include/coomatrix.hpp:32-46)
enum MMStorageType { COORDINATE, ARRAY };
enum MMScalarType { REAL, COMPLEX, INTEGER, PATTERN };
enum MMMatrixType { GENERAL, SYMMETRIC, SKEWSYMMETRIC, HERMITIAN };
struct MMDescription {
	MMStorageType storagetype;
	MMScalarType scalartype;
	MMMatrixType matrixtype;
};

}  // namespace blasted
