// bounding_box.h -- AABB of a span of positions, returns the longest side
// (/root/reference/vplib/src/bounding_box.h:22-61; the else-if chain is kept as written).
#ifndef VPLIB_BOUNDING_BOX_H
#define VPLIB_BOUNDING_BOX_H

#include <algorithm>
#include <functional>
#include <optional>
#include <span>
#include <utility>

#include "mesh/mesh.h"

using MinMax = std::pair<float, float>;
using MinMaxRef = std::optional<std::reference_wrapper<MinMax>>;

inline float CalculateBoundingBox(std::span<const Position> pts, MinMaxRef outX = std::nullopt,
                                  MinMaxRef outY = std::nullopt, MinMaxRef outZ = std::nullopt)
{
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

#endif
