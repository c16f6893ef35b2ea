// device_ptr.h -- RAII device buffer over the C ABI (vp_malloc / vp_free / vp_upload / vp_download):
// the MI355X counterpart of the reference's CudaPtr<T> (/root/reference/vplib/src/cuda_ptr.h:15-93).
// Copying deep-copies through the host-visible API like CudaPtr does device-to-device.
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

public:
    DevicePtr() = default;
    explicit DevicePtr(size_t size) : mSize(size)
    {
        void* p = nullptr;
        gpuAssert(vp_malloc(vplib::Context(), mSize * sizeof(T), &p));
        mPtr = static_cast<T*>(p);
    }
    DevicePtr(const T* host, size_t size) : DevicePtr(size) { CopyFromHost(host, size); }
    DevicePtr(const DevicePtr&) = delete;            // device-to-device copies are explicit in this design
    DevicePtr& operator=(const DevicePtr&) = delete;
    DevicePtr(DevicePtr&& o) noexcept { swap(o); }
    DevicePtr& operator=(DevicePtr&& o) noexcept { swap(o); return *this; }
    ~DevicePtr() { if (mPtr) vp_free(vplib::Context(), mPtr); }

    void swap(DevicePtr& o) noexcept { std::swap(mPtr, o.mPtr); std::swap(mSize, o.mSize); }
    T* get() { return mPtr; }
    const T* get() const { return mPtr; }
    size_t Size() const { return mSize; }

    void CopyFromHost(const T* src, size_t size)
    {
        if (size > mSize) throw std::out_of_range("CopyFromHost: size too large");
        gpuAssert(vp_upload(vplib::Context(), mPtr, src, size * sizeof(T)));
    }
    void CopyToHost(T* dst, size_t size) const
    {
        if (size > mSize) throw std::out_of_range("CopyToHost: size too large");   // cuda_ptr.h:86-90
        gpuAssert(vp_download(vplib::Context(), dst, mPtr, size * sizeof(T)));
    }
    void SetMemoryToZero() { gpuAssert(vp_memset(vplib::Context(), mPtr, 0, mSize * sizeof(T))); }
};

#endif
