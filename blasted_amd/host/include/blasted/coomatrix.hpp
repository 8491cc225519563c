// coomatrix.hpp -- Matrix-Market file input for the native (non-PETSc) route: the reference's
// include/coomatrix.hpp surface (COOMatrix, readDenseMatrixMarket, getSRMatrixFromCOO,
// MatrixReadException, the MM* enums), so that a driver written against tests/testsolve.cpp:60-107
// recompiles unchanged.  Differences, all deliberate:
//  * convertToBSR orders the blocks of a block-row by ascending block column, whatever the order of the
//    entries inside the scalar rows (the reference appends blocks by first appearance,
//    src/coomatrix.cpp:329-353, which breaks the "lower blocks come before diagind" assumption of the
//    ILU/SGS kernels for files whose rows are not sorted by column);
//  * rows without entries are allowed (the reference asserts that there are none, :236-247);
//  * every malformed-file condition throws MatrixReadException (the reference aborts on some, :52-83);
//  * constructBSRMatrixFromMatrixMarketFile is not provided: it returns the reference's owning
//    BSRMatrix assembly class, which is outside this backend -- use getSRMatrixFromCOO and a
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

/// Matrix types, storage types and scalar types of a Matrix Market header (include/coomatrix.hpp:32-37)
enum MMMatrixType { GENERAL, SYMMETRIC, SKEWSYMMETRIC, HERMITIAN };
enum MMStorageType { COORDINATE, ARRAY };
enum MMScalarType { REAL, COMPLEX, INTEGER, PATTERN };

/// The banner line of a Matrix Market file
struct MMDescription {
	MMStorageType storagetype;
	MMScalarType scalartype;
	MMMatrixType matrixtype;
};

/// Thrown when a matrix or vector file is malformed or of an unsupported kind
class MatrixReadException : public std::runtime_error {
public:
	explicit MatrixReadException(const std::string &msg);
};

/// A dense "array general" Matrix Market file, entries in file order (include/coomatrix.hpp:48-50)
template <typename scalar>
device_vector<scalar> readDenseMatrixMarket(const std::string file);

/// One entry of a coordinate matrix (zero-based indices)
template <typename scalar, typename index>
struct Entry {
	index rowind;
	index colind;
	scalar value;
};

/// Coordinate-format sparse matrix read from a "coordinate real|integer general" Matrix Market file
template <typename scalar, typename index>
class COOMatrix {
	static_assert(!std::is_const<index>::value, "Index type should be mutable.");
	static_assert(!std::is_const<scalar>::value, "Scalar type should be mutable.");
	static_assert(std::numeric_limits<index>::is_signed, "Signed index type required!");
	static_assert(std::numeric_limits<index>::is_integer, "Integer index type required!");

public:
	COOMatrix();
	virtual ~COOMatrix();

	index numrows() const;
	index numcols() const;
	index numnonzeros() const;

	/// Reads the file; afterwards the entries are sorted by (row, column) and getRowPtrs() delimits rows
	void readMatrixMarket(const std::string file);

	/// New CSR matrix owning its arrays; nbrows = number of rows; diagind = -1 where a row has no diagonal
	SRMatrixStorage<scalar, index> convertToCSR() const;

	/// New BSR matrix owning its arrays (square matrix whose dimension is a multiple of bs); blocks of a
	/// block-row in ascending block-column order, absent entries of a stored block are zero
	template <int bs, StorageOptions stor>
	SRMatrixStorage<scalar, index> convertToBSR() const;

	const std::vector<Entry<scalar, index>> &getEntries() const;
	const std::vector<index> &getRowPtrs() const;

protected:
	std::vector<Entry<scalar, index>> entries;
	index nnz = 0;
	index nrows = 0;
	index ncols = 0;
	std::vector<index> rowptr;
};

/// CSR (bs == 1) or BSR with "rowmajor" / "colmajor" blocks from a COO matrix (include/coomatrix.hpp:139-141)
template <typename scalar, typename index, int bs>
SRMatrixStorage<scalar, index> getSRMatrixFromCOO(const COOMatrix<scalar, index> &coo_mat,
                                                 const std::string block_storage_order);

}  // namespace blasted
