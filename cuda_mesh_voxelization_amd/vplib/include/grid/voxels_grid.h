// voxels_grid.h -- bit-packed occupancy grid (/root/reference/vplib/src/grid/voxels_grid.h:32-284).
// Layout contract: voxel (x,y,z) -> linear bit i = x + y*nx + z*nx*ny, word i / wordbits, bit
// i % wordbits, LSB first (:116-129); storage ceil(n^3 / wordbits) words (:189-192).  A uint64_t
// grid and a uint32_t grid of the same shape hold identical bytes (little endian), which is what
// lets both word types share the uint32 device kernels.
// VoxelsGrid<T, device>: `device = true` is the view DeviceVoxelsGrid<T> hands out -- same frame, DEVICE pointer.  In the
// reference that flag switches Bit to atomics inside its kernels (:52-76); here the kernels live behind the C ABI, so the
// device view offers the frame, Data() and StorageSize() on the host and no per-voxel access.
#ifndef VPLIB_VOXELS_GRID_H
#define VPLIB_VOXELS_GRID_H

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <span>
#include <type_traits>

#include "grid/grid.h"

template <typename T>
concept VGType = std::is_same_v<T, uint32_t> || std::is_same_v<T, uint64_t>;

template <VGType T>
class HostVoxelsGrid;
template <VGType T>
class DeviceVoxelsGrid;

template <VGType T, bool device = false>
class VoxelsGrid {
    size_t mSizeX = 1, mSizeY = 1, mSizeZ = 1;
    std::span<T> mGrid;
    float mVoxelSize = 1;
    float mOriginX = 0, mOriginY = 0, mOriginZ = 0;

public:
    class Bit {
        T* mWord;
        T mMask;

    public:
        Bit(T* word, T mask) : mWord(word), mMask(mask) {}
        Bit& operator=(bool v) { if (v) *mWord |= mMask; else *mWord &= ~mMask; return *this; }
        Bit& operator^=(bool v) { if (v) *mWord ^= mMask; return *this; }
        operator bool() const { return (*mWord & mMask) != 0; }
    };

    VoxelsGrid() = default;
    VoxelsGrid(T* data, size_t nx, size_t ny, size_t nz, float voxelSize = 1.0f)
        : mSizeX(nx), mSizeY(ny), mSizeZ(nz), mGrid(data, CalculateStorageSize(nx, ny, nz)), mVoxelSize(voxelSize) {}
    VoxelsGrid(T* data, size_t n, float voxelSize = 1.0f) : VoxelsGrid(data, n, n, n, voxelSize) {}

    size_t Index(size_t x, size_t y, size_t z) const { return x + (y * mSizeX) + (z * mSizeX * mSizeY); }
    Bit Voxel(size_t x, size_t y, size_t z) requires (!device)
    {
        assert(x < mSizeX && y < mSizeY && z < mSizeZ);
        const size_t i = Index(x, y, z);
        return Bit(&mGrid[i / WordSize()], T(1) << (i % WordSize()));
    }
    bool Voxel(size_t x, size_t y, size_t z) const requires (!device)
    {
        assert(x < mSizeX && y < mSizeY && z < mSizeZ);
        const size_t i = Index(x, y, z);
        return (mGrid[i / WordSize()] & (T(1) << (i % WordSize()))) != 0;
    }
    T& Word(size_t x, size_t y, size_t z) requires (!device) { return mGrid[Index(x, y, z) / WordSize()]; }
    T Word(size_t x, size_t y, size_t z) const requires (!device) { return mGrid[Index(x, y, z) / WordSize()]; }

    size_t Size() const { return mSizeX * mSizeY * mSizeZ; }
    size_t SizeX() const { return mSizeX; }
    size_t SizeY() const { return mSizeY; }
    size_t SizeZ() const { return mSizeZ; }
    size_t VoxelsPerSide() const { assert(mSizeX == mSizeY && mSizeY == mSizeZ); return mSizeX; }
    float VoxelSize() const { return mVoxelSize; }
    void SetOrigin(float x, float y, float z) { mOriginX = x; mOriginY = y; mOriginZ = z; }
    float OriginX() const { return mOriginX; }
    float OriginY() const { return mOriginY; }
    float OriginZ() const { return mOriginZ; }

    // debug dump, one row of voxels per line, a blank line between planes (voxels_grid.h:171-183 of the reference)
    void Print() const requires (!device)
    {
        for (size_t z = 0; z < mSizeZ; ++z) {
            for (size_t y = 0; y < mSizeY; ++y) {
                for (size_t x = 0; x < mSizeX; ++x) std::printf("%d ", static_cast<int>(Voxel(x, y, z)));
                std::printf("\n");
            }
            std::printf("\n");
        }
    }

    T* Data() { return mGrid.data(); }
    const T* Data() const { return mGrid.data(); }
    size_t StorageSize() const { return mGrid.size(); }

    static constexpr size_t WordSize() { return sizeof(T) * 8; }
    static size_t CalculateStorageSize(size_t n) { return CalculateStorageSize(n, n, n); }
    static size_t CalculateStorageSize(size_t nx, size_t ny, size_t nz) { return (nx * ny * nz + WordSize() - 1) / WordSize(); }

    friend class HostVoxelsGrid<T>;
    friend class DeviceVoxelsGrid<T>;
};

template <VGType T>
class HostVoxelsGrid {
    std::unique_ptr<T[]> mData;
    VoxelsGrid<T, false> mView;

public:
    HostVoxelsGrid() = default;
    HostVoxelsGrid(size_t nx, size_t ny, size_t nz, float voxelSize = 1.0f)
        : mData(std::make_unique<T[]>(VoxelsGrid<T, false>::CalculateStorageSize(nx, ny, nz))),     // zero-filled (voxels_grid.cu:16,24)
          mView(mData.get(), nx, ny, nz, voxelSize) {}
    explicit HostVoxelsGrid(size_t n, float voxelSize = 1.0f) : HostVoxelsGrid(n, n, n, voxelSize) {}
    HostVoxelsGrid(const HostVoxelsGrid& o)
        : mData(std::make_unique<T[]>(o.mView.StorageSize())),
          mView(mData.get(), o.mView.SizeX(), o.mView.SizeY(), o.mView.SizeZ(), o.mView.VoxelSize())
    {
        std::copy_n(o.mData.get(), o.mView.StorageSize(), mData.get());
        mView.SetOrigin(o.mView.OriginX(), o.mView.OriginY(), o.mView.OriginZ());
    }
    HostVoxelsGrid(const DeviceVoxelsGrid<T>& device);             // download, frame included (voxels_grid.cu:28-37)
    HostVoxelsGrid(HostVoxelsGrid&& o) noexcept { swap(o); }
    HostVoxelsGrid& operator=(HostVoxelsGrid o) noexcept { swap(o); return *this; }

    void swap(HostVoxelsGrid& o) noexcept { std::swap(mData, o.mData); std::swap(mView, o.mView); }
    friend void swap(HostVoxelsGrid& a, HostVoxelsGrid& b) noexcept { a.swap(b); }

    VoxelsGrid<T, false>& View() { return mView; }
    const VoxelsGrid<T, false>& View() const { return mView; }

    friend class DeviceVoxelsGrid<T>;
};

// Owning device grid (voxels_grid.h:244-278, voxels_grid.cu:66-133): zero-filled on construction, deep copies device to
// device, converts from / to HostVoxelsGrid by upload / download; the frame (voxel size, origin) travels with every copy.
template <VGType T>
class DeviceVoxelsGrid {
    DevicePtr<T> mData;
    VoxelsGrid<T, true> mView;

    template <bool D>
    void Adopt(const VoxelsGrid<T, D>& v)
    {
        mView = VoxelsGrid<T, true>(mData.get(), v.SizeX(), v.SizeY(), v.SizeZ(), v.VoxelSize());
        mView.SetOrigin(v.OriginX(), v.OriginY(), v.OriginZ());
    }

public:
    DeviceVoxelsGrid() = default;
    DeviceVoxelsGrid(size_t nx, size_t ny, size_t nz, float voxelSize = 1.0f)
        : mData(VoxelsGrid<T, true>::CalculateStorageSize(nx, ny, nz)), mView(mData.get(), nx, ny, nz, voxelSize)
    {
        mData.SetMemoryToZero();
    }
    explicit DeviceVoxelsGrid(size_t n, float voxelSize = 1.0f) : DeviceVoxelsGrid(n, n, n, voxelSize) {}
    DeviceVoxelsGrid(const HostVoxelsGrid<T>& host) : mData(host.View().Data(), host.View().StorageSize()) { Adopt(host.View()); }
    DeviceVoxelsGrid(const DeviceVoxelsGrid& o) : mData(o.mData) { Adopt(o.mView); }
    DeviceVoxelsGrid(DeviceVoxelsGrid&& o) noexcept { swap(o); }
    DeviceVoxelsGrid& operator=(const DeviceVoxelsGrid& o)
    {
        if (this == &o) return *this;
        mData = o.mData;
        Adopt(o.mView);
        return *this;
    }
    DeviceVoxelsGrid& operator=(DeviceVoxelsGrid&& o) noexcept { swap(o); return *this; }

    void swap(DeviceVoxelsGrid& o) noexcept { mData.swap(o.mData); std::swap(mView, o.mView); }
    friend void swap(DeviceVoxelsGrid& a, DeviceVoxelsGrid& b) noexcept { a.swap(b); }

    VoxelsGrid<T, true>& View() { return mView; }
    const VoxelsGrid<T, true>& View() const { return mView; }

    friend class HostVoxelsGrid<T>;
};

template <VGType T>
HostVoxelsGrid<T>::HostVoxelsGrid(const DeviceVoxelsGrid<T>& device)
    : mData(std::make_unique<T[]>(device.View().StorageSize())),
      mView(mData.get(), device.View().SizeX(), device.View().SizeY(), device.View().SizeZ(), device.View().VoxelSize())
{
    device.mData.CopyToHost(mData.get(), mView.StorageSize());
    mView.SetOrigin(device.View().OriginX(), device.View().OriginY(), device.View().OriginZ());
}

using HostVoxelsGrid32bit = HostVoxelsGrid<uint32_t>;
using HostVoxelsGrid64bit = HostVoxelsGrid<uint64_t>;
using DeviceVoxelsGrid32bit = DeviceVoxelsGrid<uint32_t>;
using DeviceVoxelsGrid64bit = DeviceVoxelsGrid<uint64_t>;

#endif
