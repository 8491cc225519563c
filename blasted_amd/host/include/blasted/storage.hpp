// storage.hpp -- sparse-row storage of the BLASTed API, restated without Boost/Eigen.
//   ArrayView<T>                      include/arrayview.hpp:31-144   (wrap-or-own array, move only)
//   device_vector<T>                  include/device_container.hpp:19-20
//   SRMatrixStorage<mscalar,mindex>   include/srmatrixdefs.hpp:37-79
//   CRawBSRMatrix / RawBSRMatrix      include/srmatrixdefs.hpp:98-157
//   move_to_const, share_with_const, createRawView   src/rawsrmatrixutils.cpp:20-80,330-341
//
// On this backend device_vector<T> is the HIP buffer holder the north-star asks for: a 64-byte aligned
// host array (what the reference's container is) plus an optional mirror in HBM that is filled and
// read back explicitly; the mirror is managed through the C ABI (blasted_hip_buffer_*).
#pragma once

#include <cassert>
#include <cstddef>
#include <cstdlib>
#include <limits>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "types.hpp"

#ifndef CACHE_LINE_LEN
#define CACHE_LINE_LEN 64
#endif

namespace blasted {

/// 64-byte aligned raw allocation (the reference uses boost::alignment::aligned_alloc)
inline void *aligned_alloc(const std::size_t alignment, const std::size_t nbytes)
{
	const std::size_t padded = ((nbytes ? nbytes : 1) + alignment - 1) / alignment * alignment;
	void *p = std::aligned_alloc(alignment, padded);
	if (!p)
		throw std::bad_alloc();
	return p;
}
inline void aligned_free(const void *p)
{
	std::free(const_cast<void *>(p));
}

template <typename T, std::size_t Align>
struct aligned_allocator {
	using value_type = T;
	template <typename U>
	struct rebind {
		using other = aligned_allocator<U, Align>;
	};
	aligned_allocator() noexcept = default;
	template <typename U>
	aligned_allocator(const aligned_allocator<U, Align> &) noexcept {}
	T *allocate(std::size_t n) { return static_cast<T *>(blasted::aligned_alloc(Align, n * sizeof(T))); }
	void deallocate(T *p, std::size_t) noexcept { aligned_free(p); }
	bool operator==(const aligned_allocator &) const noexcept { return true; }
	bool operator!=(const aligned_allocator &) const noexcept { return false; }
};

namespace detail {
void *device_buffer_alloc(std::size_t nbytes);
void device_buffer_free(void *dev);
void device_buffer_upload(void *dev, const void *host, std::size_t nbytes);
void device_buffer_download(void *host, const void *dev, std::size_t nbytes);
/// waits for everything enqueued on the operators' device, on any stream (callers that mix the operators'
/// private streams with another library's, e.g. the PCSHELL glue on PETSc's HIP vectors)
void device_synchronize();
/// page-lock / release a host range the caller owns (blasted_hip_host_register): false when it cannot be pinned
bool host_register(const void *host, std::size_t nbytes);
void host_unregister(const void *host);
}  // namespace detail

/// Aligned host array with an explicit mirror in HBM.  T must be plain old data.
template <typename T>
class device_vector : public std::vector<T, aligned_allocator<T, CACHE_LINE_LEN>> {
	using base = std::vector<T, aligned_allocator<T, CACHE_LINE_LEN>>;

public:
	using base::base;
	device_vector() = default;
	device_vector(const device_vector &o) : base(o), dev_{nullptr}, devcount_{0} {}
	device_vector(device_vector &&o) noexcept : base(std::move(o)), dev_{o.dev_}, devcount_{o.devcount_}
	{
		o.dev_ = nullptr;
		o.devcount_ = 0;
	}
	device_vector &operator=(const device_vector &o)
	{
		base::operator=(o);
		return *this;
	}
	device_vector &operator=(device_vector &&o) noexcept
	{
		release_device();
		base::operator=(std::move(o));
		dev_ = o.dev_;
		devcount_ = o.devcount_;
		o.dev_ = nullptr;
		o.devcount_ = 0;
		return *this;
	}
	~device_vector() { release_device(); }

	/// Copies the host contents into the HBM mirror (allocated on first use) and returns it
	T *to_device()
	{
		if (devcount_ != this->size()) {
			release_device();
			dev_ = static_cast<T *>(detail::device_buffer_alloc(this->size() * sizeof(T)));
			devcount_ = this->size();
		}
		detail::device_buffer_upload(dev_, this->data(), this->size() * sizeof(T));
		return dev_;
	}
	/// Copies the HBM mirror back into the host array
	void to_host()
	{
		if (dev_)
			detail::device_buffer_download(this->data(), dev_, devcount_ * sizeof(T));
	}
	T *device_data() { return dev_; }
	const T *device_data() const { return dev_; }
	void release_device()
	{
		if (dev_)
			detail::device_buffer_free(dev_);
		dev_ = nullptr;
		devcount_ = 0;
	}

private:
	T *dev_ = nullptr;
	std::size_t devcount_ = 0;
};

template <typename T>
class ArrayView;
template <typename T>
ArrayView<typename std::add_const<T>::type> move_to_const(ArrayView<T> &&array);

/// A contiguous array that either borrows its memory or owns it; move-only by design
template <typename T>
class ArrayView {
	using mutable_t = typename std::remove_const<T>::type;

public:
	ArrayView() = default;
	explicit ArrayView(const int size)
	    : data{static_cast<T *>(blasted::aligned_alloc(CACHE_LINE_LEN, sizeof(T) * (std::size_t)size))},
	      len{size}, owner{true}
	{
		assert(size >= 0);
	}
	ArrayView(T *const arr, const int length) : data{arr}, len{length}, owner{false} { assert(length >= 0); }
	ArrayView(T *arr, const int length, const bool make_owner) : data{arr}, len{length}, owner{make_owner}
	{
		assert(length >= 0);
	}
	ArrayView(const ArrayView &) = delete;
	ArrayView &operator=(const ArrayView &) = delete;
	ArrayView(ArrayView<T> &&other) noexcept : data{other.data}, len{other.len}, owner{other.owner}
	{
		other.forget();
	}
	~ArrayView() { drop(); }

	int size() const { return len; }

	/// Frees owned contents and allocates `size` fresh elements
	void resize(const int size)
	{
		assert(size >= 0);
		drop();
		data = static_cast<T *>(blasted::aligned_alloc(CACHE_LINE_LEN, sizeof(T) * (std::size_t)size));
		len = size;
		owner = true;
	}
	/// Borrow external memory
	void wrap(T *const arr, const int length)
	{
		assert(length >= 0);
		drop();
		data = arr;
		len = length;
		owner = false;
	}
	/// Adopt external memory (it must have come from blasted::aligned_alloc)
	void take_control(T *const arr, const int length)
	{
		assert(length >= 0);
		drop();
		data = arr;
		len = length;
		owner = true;
	}

	const T &operator[](const int i) const
	{
		assert(i < len);
		return data[i];
	}
	T &operator[](const int i)
	{
		assert(i < len);
		return data[i];
	}

	friend ArrayView<typename std::add_const<T>::type> move_to_const<>(ArrayView<T> &&array);

private:
	T *data = nullptr;
	int len = 0;
	bool owner = false;

	void forget()
	{
		data = nullptr;
		len = 0;
		owner = false;
	}
	void drop()
	{
		if (owner)
			aligned_free(const_cast<mutable_t *>(data));
		forget();
	}
};

template <typename T>
ArrayView<typename std::add_const<T>::type> move_to_const(ArrayView<T> &&array)
{
	ArrayView<typename std::add_const<T>::type> out(array.data, array.len, array.owner);
	array.forget();
	return out;
}

template <typename mscalar, typename mindex>
struct SRMatrixStorage;

template <typename scalar, typename index>
SRMatrixStorage<typename std::add_const<scalar>::type, typename std::add_const<index>::type>
move_to_const(SRMatrixStorage<scalar, index> &&smat);

/// Sparse (block-)row matrix: browptr / bcolind / vals / diagind / browendptr + counts
template <typename mscalar, typename mindex>
struct SRMatrixStorage {
	typedef typename std::remove_cv<mscalar>::type scalar;
	typedef typename std::remove_cv<mindex>::type index;
	static_assert(std::numeric_limits<index>::is_integer && std::numeric_limits<index>::is_signed,
	              "Signed integer index type required!");

	ArrayView<mindex> browptr;
	ArrayView<mindex> bcolind;
	ArrayView<mscalar> vals;
	ArrayView<mindex> diagind;
	ArrayView<mindex> browendptr;
	index nbrows = 0;
	index nnzb = 0;
	index nbstored = 0;

	SRMatrixStorage() = default;

	/// Borrow caller-owned arrays (block_size only determines the length of vals)
	SRMatrixStorage(mindex *const brptrs, mindex *const bcinds, mscalar *const values,
	                mindex *const diag_inds, mindex *const brendptrs, const index n_brows,
	                const index n_nzb, const index n_bstored, const int block_size)
	    : browptr(brptrs, n_brows + 1), bcolind(bcinds, n_nzb), vals(values, n_nzb * block_size * block_size),
	      diagind(diag_inds, n_brows), browendptr(brendptrs, n_brows), nbrows{n_brows}, nnzb{n_nzb},
	      nbstored{n_bstored}
	{
	}

	SRMatrixStorage(ArrayView<mindex> &&brptrs, ArrayView<mindex> &&bcinds, ArrayView<mscalar> &&values,
	                ArrayView<mindex> &&diag_inds, ArrayView<mindex> &&brendptrs, const index n_brows,
	                const index n_nzb, const index n_bstored)
	    : browptr(std::move(brptrs)), bcolind(std::move(bcinds)), vals(std::move(values)),
	      diagind(std::move(diag_inds)), browendptr(std::move(brendptrs)), nbrows{n_brows}, nnzb{n_nzb},
	      nbstored{n_bstored}
	{
	}

	SRMatrixStorage(SRMatrixStorage<mscalar, mindex> &&other)
	    : browptr(std::move(other.browptr)), bcolind(std::move(other.bcolind)), vals(std::move(other.vals)),
	      diagind(std::move(other.diagind)), browendptr(std::move(other.browendptr)), nbrows{other.nbrows},
	      nnzb{other.nnzb}, nbstored{other.nbstored}
	{
		other.nbrows = other.nnzb = other.nbstored = 0;
	}
};

template <typename scalar, typename index>
SRMatrixStorage<typename std::add_const<scalar>::type, typename std::add_const<index>::type>
move_to_const(SRMatrixStorage<scalar, index> &&smat)
{
	SRMatrixStorage<typename std::add_const<scalar>::type, typename std::add_const<index>::type> out(
	    move_to_const<index>(std::move(smat.browptr)), move_to_const<index>(std::move(smat.bcolind)),
	    move_to_const<scalar>(std::move(smat.vals)), move_to_const<index>(std::move(smat.diagind)),
	    move_to_const<index>(std::move(smat.browendptr)), smat.nbrows, smat.nnzb, smat.nbstored);
	smat.nbrows = smat.nnzb = smat.nbstored = 0;
	return out;
}

/// A second, non-owning immutable view of the same arrays
template <typename scalar, typename index>
SRMatrixStorage<typename std::add_const<scalar>::type, typename std::add_const<index>::type>
share_with_const(const SRMatrixStorage<scalar, index> &smat, const int block_size)
{
	return SRMatrixStorage<typename std::add_const<scalar>::type, typename std::add_const<index>::type>(
	    &smat.browptr[0], &smat.bcolind[0], &smat.vals[0], &smat.diagind[0], &smat.browendptr[0],
	    smat.nbrows, smat.nnzb, smat.nbstored, block_size);
}

/// Plain-pointer immutable view (what the kernels of the reference take)
template <typename scalar, typename index>
struct CRawBSRMatrix {
	const index *browptr = nullptr;
	const index *bcolind = nullptr;
	const scalar *vals = nullptr;
	const index *diagind = nullptr;
	const index *browendptr = nullptr;
	index nbrows = 0;
	index nnzb = 0;
	index nbstored = 0;

	CRawBSRMatrix() = default;
	CRawBSRMatrix(const index *const brptrs, const index *const bcinds, const scalar *const values,
	              const index *const diag_inds, const index *const brendptrs, const index n_brows,
	              const index n_nzb, const index n_bstored)
	    : browptr{brptrs}, bcolind{bcinds}, vals{values}, diagind{diag_inds}, browendptr{brendptrs},
	      nbrows{n_brows}, nnzb{n_nzb}, nbstored{n_bstored}
	{
	}
};

/// Mutable twin of CRawBSRMatrix (same layout)
template <typename scalar, typename index>
struct RawBSRMatrix {
	index *browptr = nullptr;
	index *bcolind = nullptr;
	scalar *vals = nullptr;
	index *diagind = nullptr;
	index *browendptr = nullptr;
	index nbrows = 0;
	index nnzb = 0;
	index nbstored = 0;

	RawBSRMatrix() = default;
	RawBSRMatrix(index *const brptrs, index *const bcinds, scalar *const values, index *const diag_inds,
	             index *const brendptrs, const index n_brows, const index n_nzb, const index n_bstored)
	    : browptr{brptrs}, bcolind{bcinds}, vals{values}, diagind{diag_inds}, browendptr{brendptrs},
	      nbrows{n_brows}, nnzb{n_nzb}, nbstored{n_bstored}
	{
	}
};

template <typename scalar, typename index>
CRawBSRMatrix<scalar, index> createRawView(const SRMatrixStorage<const scalar, const index> &&smat)
{
	return CRawBSRMatrix<scalar, index>(&smat.browptr[0], &smat.bcolind[0], &smat.vals[0], &smat.diagind[0],
	                                    &smat.browendptr[0], smat.nbrows, smat.nnzb, smat.nbstored);
}

}  // namespace blasted
