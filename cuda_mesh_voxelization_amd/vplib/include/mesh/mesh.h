// mesh.h -- Vec3 / Position / Normal / Color / Mesh with the member names of the reference
// (/root/reference/vplib/src/mesh/mesh.h:44-170).  Cross and Dot keep the reference's operation
// order (mesh.h:114-126): it is part of the bit-exact contract of the voxelizer.
#ifndef VPLIB_MESH_H
#define VPLIB_MESH_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct Color {
    Color() = default;
    Color(float r, float g, float b, float a) { SetColor(r, g, b, a); }
    explicit Color(uint32_t rgba) : mColor(rgba) {}

    void SetColor(float r, float g, float b, float a)
    {
        auto q = [](float v) { return static_cast<uint32_t>(std::round(v * 255)); };
        mColor = (q(r) << 24) | (q(g) << 16) | (q(b) << 8) | q(a);
    }
    uint8_t R() const { return (mColor >> 24) & 0xFF; }
    uint8_t G() const { return (mColor >> 16) & 0xFF; }
    uint8_t B() const { return (mColor >> 8) & 0xFF; }
    uint8_t A() const { return mColor & 0xFF; }
    // the reference's one-channel setters (mesh.h:27-36) re-quantise the OTHER channels from their bytes as if they were fractions -- i.e.
    // SetColor(r, G(), B(), A()) with G() = 0 .. 255: kept as written there (such a channel spills into the lanes above it)
    void R(float r) { SetColor(r, G(), B(), A()); }
    void G(float g) { SetColor(R(), g, B(), A()); }
    void B(float b) { SetColor(R(), G(), b, A()); }
    void A(float a) { SetColor(R(), G(), B(), a); }

private:
    uint32_t mColor = 0xFFFFFFFFu;
};

template <typename T = float>
struct Vec3 {
    T X = 0, Y = 0, Z = 0;

    Vec3() = default;
    Vec3(T x, T y, T z) : X(x), Y(y), Z(z) {}

    Vec3 operator+(const Vec3& v) const { return {X + v.X, Y + v.Y, Z + v.Z}; }
    Vec3 operator-(const Vec3& v) const { return {X - v.X, Y - v.Y, Z - v.Z}; }
    Vec3 operator+(T s) const { return {X + s, Y + s, Z + s}; }
    Vec3 operator-(T s) const { return {X - s, Y - s, Z - s}; }
    Vec3 operator*(T s) const { return {X * s, Y * s, Z * s}; }
    Vec3 operator/(T s) const { return {X / s, Y / s, Z / s}; }
    Vec3 operator-() const { return {-X, -Y, -Z}; }
    Vec3& operator+=(const Vec3& v) { X += v.X; Y += v.Y; Z += v.Z; return *this; }
    Vec3& operator-=(const Vec3& v) { X -= v.X; Y -= v.Y; Z -= v.Z; return *this; }
    Vec3& operator*=(T s) { X *= s; Y *= s; Z *= s; return *this; }
    Vec3& operator/=(T s) { X /= s; Y /= s; Z /= s; return *this; }
    bool operator==(const Vec3& v) const { return X == v.X && Y == v.Y && Z == v.Z; }
    bool operator!=(const Vec3& v) const { return !(*this == v); }

    static T Dot(const Vec3& a, const Vec3& b) { return (a.X * b.X + a.Y * b.Y + a.Z * b.Z); }

    static Vec3 Cross(const Vec3& a, const Vec3& b)
    {
        const T x = (a.Y * b.Z) - (a.Z * b.Y);
        const T y = (a.Z * b.X) - (a.X * b.Z);
        const T z = (a.X * b.Y) - (a.Y * b.X);
        return {x, y, z};
    }
};

using Position = Vec3<float>;
using Normal = Vec3<float>;
static_assert(sizeof(Position) == 12, "Position must be three packed floats (device layout)");

struct Mesh {
    std::string Name = "mesh_default";
    std::vector<uint32_t> FacesCoords;    // 3 vertex indices per triangle
    std::vector<uint32_t> FacesNormals;
    std::vector<Position> Coords;
    std::vector<Normal> Normals;
    std::vector<Color> Colors;

    Mesh() = default;
    explicit Mesh(std::string name) : Name(std::move(name)) {}

    void VerticesReserve(size_t n) { Coords.reserve(n); Normals.reserve(n); Colors.reserve(n); }
    void FacesReserve(size_t n) { FacesCoords.reserve(n * 6); FacesNormals.reserve(n * 6); }
    void ShrinkToFit()
    {
        FacesCoords.shrink_to_fit(); FacesNormals.shrink_to_fit();
        Coords.shrink_to_fit(); Normals.shrink_to_fit(); Colors.shrink_to_fit();
    }
    void Clear() { FacesCoords.clear(); FacesNormals.clear(); Coords.clear(); Normals.clear(); Colors.clear(); }

    size_t VerticesSize() const { return Coords.size(); }
    size_t NormalsSize() const { return Normals.size(); }
    size_t FacesSize() const { return FacesCoords.size() / 6; }     // reference quirk kept (mesh.h:169)
    size_t TrianglesSize() const { return FacesCoords.size() / 3; } // what the voxelizers iterate (sequential.cpp:16)
};

#endif
