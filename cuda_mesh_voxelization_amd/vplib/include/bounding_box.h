// bounding_box.h -- AABB of a span of positions, returns the longest side
// (/root/reference/vplib/src/bounding_box.h:22-61; the else-if chain is kept as written).
#ifndef VPLIB_BOUNDING_BOX_H
#define VPLIB_BOUNDING_BOX_H

#include <algorithm>
#include <functional>
#include <optional>
#include <span>
#include <utility>
#include <vector>

#include "debug_utils.h"
#include "mesh/mesh.h"
#include "vp_runtime.h"

using MinMax = std::pair<float, float>;
using MinMaxRef = std::optional<std::reference_wrapper<MinMax>>;

// device = true (the reference declares the flag and leaves that branch empty, bounding_box.h:31-33): `pts` spans DEVICE
// memory of the process-wide context; the points are brought to the host and reduced there -- the frame of a mesh is
// computed once per run and is not on the hot path.
template <bool device = false>
inline float CalculateBoundingBox(std::span<const Position> pts, MinMaxRef outX = std::nullopt,
                                  MinMaxRef outY = std::nullopt, MinMaxRef outZ = std::nullopt)
{
    if constexpr (device) {
        std::vector<Position> host(pts.size());
        gpuAssert(vp_download(vplib::Context(), host.data(), pts.data(), pts.size() * sizeof(Position)));
        return CalculateBoundingBox<false>(std::span<const Position>(host.data(), host.size()), outX, outY, outZ);
    }
    MinMax x{pts[0].X, pts[0].X}, y{pts[0].Y, pts[0].Y}, z{pts[0].Z, pts[0].Z};
    for (size_t i = 1; i < pts.size(); ++i) {
        const Position& p = pts[i];
        if (p.X < x.first) x.first = p.X; else if (p.X > x.second) x.second = p.X;
        if (p.Y < y.first) y.first = p.Y; else if (p.Y > y.second) y.second = p.Y;
        if (p.Z < z.first) z.first = p.Z; else if (p.Z > z.second) z.second = p.Z;
    }
    if (outX) outX->get() = x;
    if (outY) outY->get() = y;
    if (outZ) outZ->get() = z;
    return std::max({x.second - x.first, y.second - y.first, z.second - z.first});
}

// the reference's exact parameter type (bounding_box.h:23: std::span<Position>)
template <bool device = false>
inline float CalculateBoundingBox(std::span<Position> pts, MinMaxRef outX = std::nullopt,
                                  MinMaxRef outY = std::nullopt, MinMaxRef outZ = std::nullopt)
{
    return CalculateBoundingBox<device>(std::span<const Position>(pts.data(), pts.size()), outX, outY, outZ);
}

#endif
