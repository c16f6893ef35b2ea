// grid.h -- dense x-fastest 3-D array view + owning host and device wrappers
// (/root/reference/vplib/src/grid/grid.h:22-230).  Index is 64-bit here (reference: 32-bit, :89-92).
// DeviceGrid<T> owns its storage through DevicePtr<T> (device_ptr.h): copying is a device-to-device deep copy, Host <->
// Device construction uploads / downloads, exactly the reference's conversions (:129-135, :184-201).  A Grid<T> obtained from
// DeviceGrid::View() carries a DEVICE pointer: its shape accessors and Data() are usable on the host, element access is not.
#ifndef VPLIB_GRID_H
#define VPLIB_GRID_H

#include <algorithm>
#include <cassert>
#include <type_traits>
#include <cstddef>
#include <cstdio>
#include <memory>
#include <span>

#include "device_ptr.h"
#include "mesh/mesh.h"

template <typename T>
class HostGrid;
template <typename T>
class DeviceGrid;

template <typename T>
class Grid {
protected:
    size_t mSizeX = 1, mSizeY = 1, mSizeZ = 1;
    std::span<T> mGrid;

public:
    Grid() = default;
    Grid(T* data, size_t sizeX, size_t sizeY, size_t sizeZ)
        : mSizeX(sizeX), mSizeY(sizeY), mSizeZ(sizeZ), mGrid(data, sizeX * sizeY * sizeZ) {}
    Grid(T* data, size_t size) : Grid(data, size, size, size) {}

    T operator()(size_t x, size_t y, size_t z) const { assert(x < mSizeX && y < mSizeY && z < mSizeZ); return mGrid[Index(x, y, z)]; }
    T& operator()(size_t x, size_t y, size_t z) { assert(x < mSizeX && y < mSizeY && z < mSizeZ); return mGrid[Index(x, y, z)]; }

    size_t Size() const { return mSizeX * mSizeY * mSizeZ; }
    size_t SizeX() const { return mSizeX; }
    size_t SizeY() const { return mSizeY; }
    size_t SizeZ() const { return mSizeZ; }
    size_t Index(size_t x, size_t y, size_t z) const { return x + (y * mSizeX) + (z * mSizeX * mSizeY); }
    T* Data() { return mGrid.data(); }
    const T* Data() const { return mGrid.data(); }

    // debug dump (grid.h:74-86,98-109 of the reference): "%.2f " per float, "%d " per int, "(x, y, z) " per position, "error" for any other type
    void PrintValue(T value) const
    {
        if constexpr (std::is_same_v<T, float>) std::printf("%.2f ", value);
        else if constexpr (std::is_same_v<T, int>) std::printf("%d ", value);
        else if constexpr (requires { value.X; value.Y; value.Z; }) std::printf("(%.2f, %.2f, %.2f) ", value.X, value.Y, value.Z);
        else std::printf("error");
    }
    void Print() const
    {
        for (size_t z = 0; z < mSizeZ; ++z) {
            for (size_t y = 0; y < mSizeY; ++y) {
                for (size_t x = 0; x < mSizeX; ++x) PrintValue((*this)(x, y, z));
                std::printf("\n");
            }
            std::printf("\n");
        }
    }

    friend class HostGrid<T>;
    friend class DeviceGrid<T>;
};

template <typename T>
class HostGrid {
    std::unique_ptr<T[]> mData;
    Grid<T> mView;

public:
    HostGrid() = default;
    HostGrid(size_t size, const T init) : HostGrid(size, size, size, init) {}
    explicit HostGrid(size_t size) : HostGrid(size, size, size, T{}) {}   // reference reaches this through DeviceGrid (SURVEY A-2)
    HostGrid(size_t sx, size_t sy, size_t sz, const T init)
        : mData(std::make_unique<T[]>(sx * sy * sz)), mView(mData.get(), sx, sy, sz)
    {
        std::fill_n(mData.get(), sx * sy * sz, init);
    }
    HostGrid(const HostGrid& o)
        : mData(std::make_unique<T[]>(o.mView.Size())), mView(mData.get(), o.mView.SizeX(), o.mView.SizeY(), o.mView.SizeZ())
    {
        std::copy_n(o.mData.get(), o.mView.Size(), mData.get());
    }
    HostGrid(const DeviceGrid<T>& device);                          // download (grid.h:129-135)
    HostGrid(HostGrid&& o) noexcept { swap(o); }
    HostGrid& operator=(HostGrid o) noexcept { swap(o); return *this; }

    void swap(HostGrid& o) noexcept { std::swap(mData, o.mData); std::swap(mView, o.mView); }
    friend void swap(HostGrid& a, HostGrid& b) noexcept { a.swap(b); }

    Grid<T>& View() { return mView; }
    const Grid<T>& View() const { return mView; }

    friend class DeviceGrid<T>;
};

template <typename T>
class DeviceGrid {
    DevicePtr<T> mData;
    Grid<T> mView;

public:
    DeviceGrid() = default;
    DeviceGrid(size_t size) : DeviceGrid(size, size, size) {}
    DeviceGrid(size_t sx, size_t sy, size_t sz) : mData(sx * sy * sz), mView(mData.get(), sx, sy, sz) {}
    DeviceGrid(const HostGrid<T>& host)                              // upload (grid.h:184-191)
        : mData(host.View().Data(), host.View().Size()), mView(mData.get(), host.View().SizeX(), host.View().SizeY(), host.View().SizeZ()) {}
    DeviceGrid(const DeviceGrid& o) : mData(o.mData), mView(mData.get(), o.mView.SizeX(), o.mView.SizeY(), o.mView.SizeZ()) {}
    DeviceGrid(DeviceGrid&& o) noexcept { swap(o); }
    DeviceGrid& operator=(const DeviceGrid& o)
    {
        if (this == &o) return *this;
        mData = o.mData;                                            // device-to-device
        mView = Grid<T>(mData.get(), o.mView.SizeX(), o.mView.SizeY(), o.mView.SizeZ());
        return *this;
    }
    DeviceGrid& operator=(DeviceGrid&& o) noexcept { swap(o); return *this; }

    void swap(DeviceGrid& o) noexcept { mData.swap(o.mData); std::swap(mView, o.mView); }
    friend void swap(DeviceGrid& a, DeviceGrid& b) noexcept { a.swap(b); }

    Grid<T>& View() { return mView; }
    const Grid<T>& View() const { return mView; }

    friend class HostGrid<T>;
};

template <typename T>
HostGrid<T>::HostGrid(const DeviceGrid<T>& device)
    : mData(std::make_unique<T[]>(device.View().Size())),
      mView(mData.get(), device.View().SizeX(), device.View().SizeY(), device.View().SizeZ())
{
    device.mData.CopyToHost(mData.get(), mView.Size());
}

#endif
