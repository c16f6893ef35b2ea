/*
 * vp_oracle.c -- plain-C restatement of the reference's sequential CPU path.
 * TEST INFRASTRUCTURE ONLY (see vp_oracle.h).  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off [-fopenmp]); FMA contraction changes results.
 *
 * Intentional divergences from the reference -- only where the reference has undefined
 * behaviour (out-of-bounds writes):
 *   - (y,z) columns outside [0,n) are skipped (reference: out-of-bounds / wrapped write);
 *   - a non-finite plane solve (A == 0) is skipped (reference: (int)inf, out-of-bounds loop);
 *   - startX is clamped to [0,n] (reference: negative index writes).
 * None of these is reachable on the reference's assets.
 * Indices are 64-bit (reference: 32-bit, valid to n = 1024, vplib/src/grid/grid.h:89-92).
 */
#include "vp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float X, Y, Z; } P3;

int vpo_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* vplib/src/bounding_box.h:31-57 (note the else-if, kept as written) */
float vpo_bounding_box(const float* xyz, size_t nverts, float out[6])
{
    float minX = xyz[0], maxX = xyz[0];
    float minY = xyz[1], maxY = xyz[1];
    float minZ = xyz[2], maxZ = xyz[2];
    for (size_t i = 1; i < nverts; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (x < minX) minX = x; else if (x > maxX) maxX = x;
        if (y < minY) minY = y; else if (y > maxY) maxY = y;
        if (z < minZ) minZ = z; else if (z > maxZ) maxZ = z;
    }
    if (out) {
        out[0] = minX; out[1] = maxX; out[2] = minY; out[3] = maxY; out[4] = minZ; out[5] = maxZ;
    }
    float side = maxX - minX;
    if (maxY - minY > side) side = maxY - minY;
    if (maxZ - minZ > side) side = maxZ - minZ;
    return side;
}

/* apps/cli/main.cpp:77-86 */
void vpo_frame(const float* xyz, size_t nverts, unsigned n, float origin[3], float* voxel_size)
{
    float bb[6];
    const float side = vpo_bounding_box(xyz, nverts, bb);
    origin[0] = bb[0]; origin[1] = bb[2]; origin[2] = bb[4];
    *voxel_size = side / n;                       /* float / unsigned -> float division */
}

/* toggle linear bits [b0, b1) */
static void toggle_range(uint32_t* words, uint64_t b0, uint64_t b1)
{
    if (b0 >= b1) return;
    uint64_t w0 = b0 >> 5, w1 = (b1 - 1) >> 5;
    const uint32_t m0 = 0xFFFFFFFFu << (b0 & 31);
    const uint32_t m1 = 0xFFFFFFFFu >> (31 - ((b1 - 1) & 31));
    if (w0 == w1) { words[w0] ^= (m0 & m1); return; }
    words[w0] ^= m0;
    for (uint64_t w = w0 + 1; w < w1; ++w) words[w] ^= 0xFFFFFFFFu;
    words[w1] ^= m1;
}

/* vplib/src/vox/vox.h:22-24 */
static inline float edge_zy(P3 a, P3 b, float y, float z)
{
    return ((z - a.Z) * (b.Y - a.Y)) - ((y - a.Y) * (b.Z - a.Z));
}

/* vplib/src/vox/sequential.cpp:18-61 */
void vpo_voxelize(uint32_t* words, unsigned n, float vs, const float origin[3],
                  const float* xyz, const uint32_t* tri, size_t ntris)
{
    const float ox = origin[0], oy = origin[1], oz = origin[2];
    const int N = (int)n;
    for (size_t i = 0; i < ntris; ++i) {
        const float* p0 = xyz + 3 * (size_t)tri[3 * i];
        const float* p1 = xyz + 3 * (size_t)tri[3 * i + 1];
        const float* p2 = xyz + 3 * (size_t)tri[3 * i + 2];
        const P3 V0 = { p0[0], p0[1], p0[2] }, V1 = { p1[0], p1[1], p1[2] }, V2 = { p2[0], p2[1], p2[2] };

        /* :23-24  normal = Cross(V1-V0, V2-V1); only .X is used (mesh.h:119-126) */
        const P3 a = { V1.X - V0.X, V1.Y - V0.Y, V1.Z - V0.Z };
        const P3 b = { V2.X - V1.X, V2.Y - V1.Y, V2.Z - V1.Z };
        const float normalX = (a.Y * b.Z) - (a.Z * b.Y);
        const int sign = 2 * (normalX >= 0) - 1;

        /* :26-28  per-triangle bbox (bounding_box.h:31-44 semantics) */
        float minY = V0.Y, maxY = V0.Y, minZ = V0.Z, maxZ = V0.Z;
        if (V1.Y < minY) minY = V1.Y; else if (V1.Y > maxY) maxY = V1.Y;
        if (V1.Z < minZ) minZ = V1.Z; else if (V1.Z > maxZ) maxZ = V1.Z;
        if (V2.Y < minY) minY = V2.Y; else if (V2.Y > maxY) maxY = V2.Y;
        if (V2.Z < minZ) minZ = V2.Z; else if (V2.Z > maxZ) maxZ = V2.Z;

        /* :30-33 */
        int startY = (int)floorf((minY - oy) / vs);
        int endY   = (int)ceilf((maxY - oy) / vs);
        int startZ = (int)floorf((minZ - oz) / vs);
        int endZ   = (int)ceilf((maxZ - oz) / vs);

        /* :35-38  plane through the triangle */
        const P3 e0 = { V1.X - V0.X, V1.Y - V0.Y, V1.Z - V0.Z };
        const P3 e1 = { V2.X - V0.X, V2.Y - V0.Y, V2.Z - V0.Z };
        const float A = (e0.Y * e1.Z) - (e0.Z * e1.Y);
        const float B = (e0.Z * e1.X) - (e0.X * e1.Z);
        const float C = (e0.X * e1.Y) - (e0.Y * e1.X);
        const float D = (A * V0.X + B * V0.Y) + C * V0.Z;

        if (startY < 0) startY = 0;             /* divergence from UB, see header */
        if (startZ < 0) startZ = 0;
        if (endY > N) endY = N;
        if (endZ > N) endZ = N;

        for (int y = startY; y < endY; ++y) {
            for (int z = startZ; z < endZ; ++z) {
                /* :44-45 */
                const float centerY = oy + ((y * vs) + (vs / 2));
                const float centerZ = oz + ((z * vs) + (vs / 2));
                /* :47-49 */
                const float E0 = edge_zy(V0, V1, centerY, centerZ) * sign;
                const float E1 = edge_zy(V1, V2, centerY, centerZ) * sign;
                const float E2 = edge_zy(V2, V0, centerY, centerZ) * sign;
                if (E0 >= 0 && E1 >= 0 && E2 >= 0) {
                    /* :52-57 */
                    const float intersection = (D - (B * centerY) - (C * centerZ)) / A;
                    const float fx = (intersection - ox) / vs;
                    if (!(fx > -2147483648.0f && fx < 2147483648.0f)) continue;   /* NaN / inf */
                    int startX = (int)fx;
                    if (startX < 0) startX = 0;
                    if (startX >= N) continue;
                    const uint64_t row = ((uint64_t)z * n + (uint64_t)y) * n;
                    toggle_range(words, row + (uint64_t)startX, row + n);
                }
            }
        }
    }
}

/* vplib/src/csg/sequential.cpp:18-28, vplib/src/csg/csg.h:14-30 */
void vpo_csg(uint32_t* a, const uint32_t* b, size_t nwords, int op)
{
    if (op == 1)      { for (size_t i = 0; i < nwords; ++i) a[i] |= b[i]; }
    else if (op == 2) { for (size_t i = 0; i < nwords; ++i) a[i] &= b[i]; }
    else if (op == 3) { for (size_t i = 0; i < nwords; ++i) a[i] &= ~b[i]; }
}

static inline int voxel_bit(const uint32_t* words, uint64_t n, int x, int y, int z)
{
    const uint64_t i = (uint64_t)x + ((uint64_t)y + (uint64_t)z * n) * n;
    return (words[i >> 5] >> (i & 31)) & 1u;
}

/* vplib/src/jfa/jfa.h:19-20 */
static inline float calc_distance(P3 p0, P3 p1)
{
    return ((p1.X - p0.X) * (p1.X - p0.X)) + ((p1.Y - p0.Y) * (p1.Y - p0.Y)) + ((p1.Z - p0.Z) * (p1.Z - p0.Z));
}

/* vplib/src/jfa/sequential.cpp:7-127 */
int vpo_jfa(const uint32_t* words, unsigned n, float vs, const float origin[3], float* sdf, int max_passes)
{
    const float ox = origin[0], oy = origin[1], oz = origin[2];
    const int N = (int)n;
    const uint64_t total = (uint64_t)n * n * n;

    /* :13-17 positions (reference leaves them uninitialised; they are only read where a seed exists) */
    P3* pos = (P3*)malloc(total * sizeof(P3));
    float* sdfApp = (float*)malloc(total * sizeof(float));
    P3* posApp = (P3*)malloc(total * sizeof(P3));
    if (!pos || !sdfApp || !posApp) { free(pos); free(sdfApp); free(posApp); return -1; }

    /* :24-64 initialisation */
#pragma omp parallel for collapse(2) schedule(static)
    for (int vz = 0; vz < N; ++vz) {
        for (int vy = 0; vy < N; ++vy) {
            for (int vx = 0; vx < N; ++vx) {
                const uint64_t idx = (uint64_t)vx + ((uint64_t)vy + (uint64_t)vz * n) * n;
                pos[idx].X = pos[idx].Y = pos[idx].Z = 0.0f;
                if (!voxel_bit(words, n, vx, vy, vz)) continue;
                int found = 0;
                for (int z = -1; z <= 1; z++)
                    for (int y = -1; y <= 1; y++)
                        for (int x = -1; x <= 1; x++) {
                            if (x == 0 && y == 0 && z == 0) continue;
                            const int nx = vx + x, ny = vy + y, nz = vz + z;
                            const int isBorder = nx < 0 || nx >= N || ny < 0 || ny >= N || nz < 0 || nz >= N;
                            if (isBorder || !voxel_bit(words, n, nx, ny, nz)) found = 1;
                        }
                if (found) {
                    sdf[idx] = 0.0f;
                    pos[idx].X = ox + (vx * vs);
                    pos[idx].Y = oy + (vy * vs);
                    pos[idx].Z = oz + (vz * vs);
                } else {
                    sdf[idx] = INFINITY;
                }
            }
        }
    }

    /* :68-125 passes.  The reference writes improved voxels into the *App copies and then deep-copies
     * them back (:123-124); writing every voxel into the other buffer and swapping is the same map. */
    float* sIn = sdf;   P3* pIn = pos;
    float* sOut = sdfApp; P3* pOut = posApp;
    int passes = 0;
    for (int k = N / 2; k >= 1; k /= 2) {
        if (max_passes >= 0 && passes >= max_passes) break;
#pragma omp parallel for collapse(2) schedule(static)
        for (int vz = 0; vz < N; ++vz) {
            for (int vy = 0; vy < N; ++vy) {
                for (int vx = 0; vx < N; ++vx) {
                    const uint64_t idx = (uint64_t)vx + ((uint64_t)vy + (uint64_t)vz * n) * n;
                    const P3 voxelPos = { ox + (vx * vs), oy + (vy * vs), oz + (vz * vs) };
                    float bestDistance = sIn[idx];
                    P3 bestPosition = pIn[idx];
                    for (int z = -1; z <= 1; z++) {
                        const int nz = vz + (z * k);
                        if (nz < 0 || nz >= N) continue;
                        for (int y = -1; y <= 1; y++) {
                            const int ny = vy + (y * k);
                            if (ny < 0 || ny >= N) continue;
                            for (int x = -1; x <= 1; x++) {
                                if (x == 0 && y == 0 && z == 0) continue;
                                const int nx = vx + (x * k);
                                if (nx < 0 || nx >= N) continue;
                                const uint64_t nidx = (uint64_t)nx + ((uint64_t)ny + (uint64_t)nz * n) * n;
                                const float seed = sIn[nidx];
                                if (fabsf(seed) < INFINITY) {
                                    const P3 seedPos = pIn[nidx];
                                    const float distance = calc_distance(voxelPos, seedPos);
                                    if (distance < fabsf(bestDistance)) {
                                        bestDistance = copysignf(distance, bestDistance);
                                        bestPosition = seedPos;
                                    }
                                }
                            }
                        }
                    }
                    sOut[idx] = bestDistance;
                    pOut[idx] = bestPosition;
                }
            }
        }
        { float* t = sIn; sIn = sOut; sOut = t; }
        { P3* t = pIn; pIn = pOut; pOut = t; }
        ++passes;
    }
    if (sIn != sdf) memcpy(sdf, sIn, total * sizeof(float));
    free(pos); free(sdfApp); free(posApp);
    return 0;
}

uint64_t vpo_fnv1a64(const void* p, size_t nbytes)
{
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < nbytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

uint64_t vpo_popcount(const uint32_t* words, size_t nwords)
{
    uint64_t pc = 0;
    for (size_t i = 0; i < nwords; ++i) pc += (uint64_t)__builtin_popcount(words[i]);
    return pc;
}

void vpo_sdf_stats(const float* sdf, size_t n, uint64_t counts[3], double sums[2], float minmax[2])
{
    uint64_t zero = 0, pinf = 0, ninf = 0;
    double sp = 0, sn = 0;
    float mx = 0, mn = 0;
    for (size_t i = 0; i < n; ++i) {
        const float v = sdf[i];
        if (v == 0) zero++;
        else if (isinf(v)) { if (v > 0) pinf++; else ninf++; }
        else if (v > 0) { sp += v; if (v > mx) mx = v; }
        else { sn += v; if (v < mn) mn = v; }
    }
    counts[0] = zero; counts[1] = pinf; counts[2] = ninf;
    sums[0] = sp; sums[1] = sn;
    minmax[0] = mn; minmax[1] = mx;
}
