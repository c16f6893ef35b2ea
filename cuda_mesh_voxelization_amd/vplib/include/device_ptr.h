// device_ptr.h -- RAII device buffer over the C ABI: the MI355X counterpart of the reference's CudaPtr<T>
// (/root/reference/vplib/src/cuda_ptr.h:15-93) with the same value semantics: copying allocates and copies device to
// device (:42-46, :52-63 -- vp_memcpy_d2d), CopyFromHost re-allocates on a size change (:79-86), CopyToHost throws
// std::out_of_range when asked for more than it holds (:88-92).  `CudaPtr<T>` is an alias, so code written against the
// reference's name compiles.  The buffer lives on the device of the process-wide context (vp_runtime.h); there is no
// host-side element access (the reference's operator[] is only meaningful inside its kernels).
#ifndef VPLIB_DEVICE_PTR_H
#define VPLIB_DEVICE_PTR_H

#include <cstddef>
#include <stdexcept>
#include <utility>

#include "debug_utils.h"
#include "vp_runtime.h"

template <typename T>
class DevicePtr {
    T* mPtr = nullptr;
    size_t mSize = 0;

    void Allocate(size_t size)
    {
        mSize = size;
        void* p = nullptr;
        gpuAssert(vp_malloc(vplib::Context(), mSize * sizeof(T), &p));
        mPtr = static_cast<T*>(p);
    }
    void Release()
    {
        if (mPtr) vp_free(vplib::Context(), mPtr);
        mPtr = nullptr;
        mSize = 0;
    }

public:
    DevicePtr() = default;
    DevicePtr(size_t size) { Allocate(size); }                      // implicit, like CudaPtr(size_t) (SURVEY A-2 relies on it)
    DevicePtr(const T* host, size_t size) { Allocate(size); gpuAssert(vp_upload(vplib::Context(), mPtr, host, mSize * sizeof(T))); }
    DevicePtr(const DevicePtr& o)
    {
        Allocate(o.mSize);
        gpuAssert(vp_memcpy_d2d(vplib::Context(), mPtr, o.mPtr, mSize * sizeof(T)));
    }
    DevicePtr(DevicePtr&& o) noexcept { swap(o); }
    ~DevicePtr() { Release(); }

    DevicePtr& operator=(const DevicePtr& o)
    {
        if (this == &o) return *this;
        if (mSize != o.mSize) { Release(); Allocate(o.mSize); }
        gpuAssert(vp_memcpy_d2d(vplib::Context(), mPtr, o.mPtr, mSize * sizeof(T)));
        return *this;
    }
    DevicePtr& operator=(DevicePtr&& o) noexcept { swap(o); return *this; }

    void swap(DevicePtr& o) noexcept { std::swap(mPtr, o.mPtr); std::swap(mSize, o.mSize); }
    friend void swap(DevicePtr& a, DevicePtr& b) noexcept { a.swap(b); }

    T* get() { return mPtr; }
    const T* get() const { return mPtr; }
    size_t Size() const { return mSize; }

    void CopyFromHost(const T* src, size_t size)
    {
        if (mSize != size) { Release(); Allocate(size); }
        gpuAssert(vp_upload(vplib::Context(), mPtr, src, mSize * sizeof(T)));
    }
    void CopyToHost(T* dst, size_t size) const
    {
        if (size > mSize) throw std::out_of_range("copyToHost: size too large");
        gpuAssert(vp_download(vplib::Context(), dst, mPtr, size * sizeof(T)));
    }
    void SetMemoryToZero() { gpuAssert(vp_memset(vplib::Context(), mPtr, 0, mSize * sizeof(T))); }
};

template <typename T>
using CudaPtr = DevicePtr<T>;

#endif
