/*
 * oracle_export.c -- CPU restatement of the reference's three grid exporters (the `-e` outputs): VoxelsGridToMeshCompressed,
 * VoxelsGridToMesh (sdf-coloured cubes) and VoxelsGridToPointCloud, with SDFToRGB and the Color quantisation ExportMesh prints.
 *
 * TEST INFRASTRUCTURE ONLY (see vp_oracle.h): loaded by tests/test_export.py as the checker of
 * cuda_mesh_voxelization_amd/vplib/src/grid_to_mesh.cpp and of the vp_extract front end; never by the product.
 * Parity status: PINNED (round 6) -- the reference's own grid_to_mesh.cpp and mesh_io.cpp build in this image (oracle/_ref, oracle/Makefile) and
 * tests/test_reference_build.py compares the three restatements below with the files the reference's exporters write, line by line.  The
 * file follows the reference line by line instead of restructuring it:
 *   /root/reference/vplib/src/mesh/grid_to_mesh.cpp:10-60   the z, y, x walk, six AddFacesVertex* calls per set voxel (:37-44)
 *   /root/reference/vplib/src/mesh/grid_to_mesh.h:25-92     AddFacesVertex: plane_index (:31), face_index and the faces_marked test
 *                                                           (:34-43), the four vertices in (v, u) order through vertices_marked
 *                                                           (:46-65), the two triangles per plane / side (:67-85), normals (:87)
 *   /root/reference/vplib/src/mesh/grid_to_mesh.cpp:65-173  VoxelsGridToMesh: 8 vertices (dz, dy, dx order, :92-105) and 12 triangles (:107-163) per set
 *                                                           voxel with a finite sdf (:89), the normal slots of :76-82
 *   /root/reference/vplib/src/mesh/grid_to_mesh.cpp:175-201 VoxelsGridToPointCloud: one vertex at the centre of every set voxel
 *   /root/reference/vplib/src/mesh/grid_to_mesh.h:15-22     SDFToRGB;  mesh/mesh.h:19-33 Color::SetColor / R() G() B();  mesh/mesh_io.cpp:99-104
 *                                                           ExportMesh prints R() / 255.0f: the colour arrays below hold R(), G(), B()
 * The reference keeps its marks in three std::vector<bool> and an unordered_map keyed by the lattice-point index; here they are three
 * byte arrays and one direct-index array -- the same decisions in the same order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    unsigned n;
    float vs, ox, oy, oz;
    const uint32_t* words;
    unsigned char* faces_marked[3];
    uint32_t* vertex_of;            /* lattice point -> vertex index + 1 (0 = not yet), vertices_marked */
    float* coords; size_t nverts, cap_verts;
    uint32_t* faces; uint32_t* normals; size_t nidx, cap_idx;
    int failed;
} Ctx;

static int voxel(const Ctx* c, unsigned x, unsigned y, unsigned z)        /* voxels_grid.h:116-129 */
{
    const size_t i = (size_t)x + (size_t)c->n * ((size_t)y + (size_t)c->n * z);
    return (c->words[i >> 5] >> (i & 31)) & 1u;
}

static void push_index(Ctx* c, uint32_t a, uint32_t b, uint32_t d, uint32_t normal)
{
    if (c->nidx + 3 > c->cap_idx) {
        c->cap_idx = c->cap_idx ? c->cap_idx * 2 : 1 << 16;
        c->faces = (uint32_t*)realloc(c->faces, c->cap_idx * 4); c->normals = (uint32_t*)realloc(c->normals, c->cap_idx * 4);
        if (!c->faces || !c->normals) { c->failed = 1; return; }
    }
    c->faces[c->nidx] = a; c->faces[c->nidx + 1] = b; c->faces[c->nidx + 2] = d;
    c->normals[c->nidx] = c->normals[c->nidx + 1] = c->normals[c->nidx + 2] = normal;
    c->nidx += 3;
}

/* grid_to_mesh.h:25-92 with the template parameters as arguments */
static void add_faces_vertex(Ctx* c, unsigned voxelX, unsigned voxelY, unsigned voxelZ, int X, int Y, int Z, int front)
{
    const unsigned plane_index = (unsigned)((!Y * 2) + !X);                                  /* :31 */
    const unsigned N = c->n, VERTEX_SIZE = N + 1;
    const unsigned XX = X ? voxelX : voxelZ;
    const unsigned YY = Y ? voxelY : voxelZ;
    const unsigned ZZ = X ? (Y ? (voxelZ + front) : (voxelY + front)) : voxelX + front;
    const size_t face_index = ((size_t)ZZ * N * N) + ((size_t)YY * N) + XX;
    if (c->faces_marked[plane_index][face_index]) return;                                    /* :40-41 */
    c->faces_marked[plane_index][face_index] = 1;
    uint32_t fv[4];
    for (unsigned v = 0; v < 2; ++v)
        for (unsigned u = 0; u < 2; ++u) {
            const unsigned Vx = voxelX + (!X * front) + (X * u);
            const unsigned Vy = voxelY + (!Y * front) + (Y * v);
            const unsigned Vz = voxelZ + (!Z * front) + (Z * ((X * v) + (Y * u)));
            const size_t vertex_index = ((size_t)Vz * VERTEX_SIZE * VERTEX_SIZE) + ((size_t)Vy * VERTEX_SIZE) + Vx;
            if (!c->vertex_of[vertex_index]) {                                               /* try_emplace: new vertex (:53-62) */
                if (c->nverts == c->cap_verts) {
                    c->cap_verts = c->cap_verts ? c->cap_verts * 2 : 1 << 14;
                    c->coords = (float*)realloc(c->coords, c->cap_verts * 12);
                    if (!c->coords) { c->failed = 1; return; }
                }
                c->coords[c->nverts * 3 + 0] = c->ox + (Vx * c->vs);
                c->coords[c->nverts * 3 + 1] = c->oy + (Vy * c->vs);
                c->coords[c->nverts * 3 + 2] = c->oz + (Vz * c->vs);
                c->vertex_of[vertex_index] = (uint32_t)(++c->nverts);
            }
            fv[u + (v * 2)] = c->vertex_of[vertex_index] - 1;
        }
    const uint32_t nrm = (uint32_t)((front * 3) + plane_index);                               /* :87 */
    if (front) {
        if (plane_index != 0) { push_index(c, fv[0], fv[2], fv[1], nrm); push_index(c, fv[1], fv[2], fv[3], nrm); }
        else                  { push_index(c, fv[0], fv[1], fv[2], nrm); push_index(c, fv[1], fv[3], fv[2], nrm); }
    } else {
        if (plane_index != 0) { push_index(c, fv[0], fv[1], fv[2], nrm); push_index(c, fv[1], fv[3], fv[2], nrm); }
        else                  { push_index(c, fv[0], fv[2], fv[1], nrm); push_index(c, fv[1], fv[2], fv[3], nrm); }
    }
}

/* grid_to_mesh.cpp:10-60.  Returns 0 and malloc'd arrays (caller frees with vpo_free): coords [nverts * 3], faces / face normals
 * [nindices] (three per triangle, six per quad); -1 if an allocation failed. */
int vpo_grid_to_mesh_compressed(const uint32_t* words, unsigned n, float voxel_size, const float origin[3],
                                float** coords, size_t* nverts, uint32_t** faces, uint32_t** face_normals, size_t* nindices)
{
    Ctx c; memset(&c, 0, sizeof c);
    c.n = n; c.vs = voxel_size; c.ox = origin[0]; c.oy = origin[1]; c.oz = origin[2]; c.words = words;
    const size_t max_faces = (size_t)n * n * (n + 1), max_verts = (size_t)(n + 1) * (n + 1) * (n + 1);
    for (int i = 0; i < 3; ++i) c.faces_marked[i] = (unsigned char*)calloc(max_faces, 1);
    c.vertex_of = (uint32_t*)calloc(max_verts, 4);
    if (!c.faces_marked[0] || !c.faces_marked[1] || !c.faces_marked[2] || !c.vertex_of) c.failed = 1;
    for (unsigned z = 0; z < n && !c.failed; ++z)
        for (unsigned y = 0; y < n; ++y)
            for (unsigned x = 0; x < n; ++x) {
                if (!voxel(&c, x, y, z)) continue;
                add_faces_vertex(&c, x, y, z, 1, 1, 0, 0);       /* AddFacesVertexXY<T, false>  (:37) */
                add_faces_vertex(&c, x, y, z, 1, 1, 0, 1);       /* AddFacesVertexXY<T, true>          */
                add_faces_vertex(&c, x, y, z, 1, 0, 1, 0);       /* AddFacesVertexXZ<T, false>  (:40)  */
                add_faces_vertex(&c, x, y, z, 1, 0, 1, 1);
                add_faces_vertex(&c, x, y, z, 0, 1, 1, 0);       /* AddFacesVertexYZ<T, false>  (:43)  */
                add_faces_vertex(&c, x, y, z, 0, 1, 1, 1);
            }
    for (int i = 0; i < 3; ++i) free(c.faces_marked[i]);
    free(c.vertex_of);
    if (c.failed) { free(c.coords); free(c.faces); free(c.normals); return -1; }
    *coords = c.coords; *nverts = c.nverts; *faces = c.faces; *face_normals = c.normals; *nindices = c.nidx;
    return 0;
}

/* grid_to_mesh.h:15-22  SDFToRGB(float v, float max).  std::min / std::max return their FIRST argument when the
 * comparison is false: a NaN v (the sqrt of a negative sdf -- never a set voxel's) passes std::min and is replaced by 0 in std::max;
 * fminf / fmaxf decide differently, so both are spelled out */
static float min_ref(float a, float b) { return (b < a) ? b : a; }
static float max_ref(float a, float b) { return (a < b) ? b : a; }
static void sdf_to_rgb_ref(float v, float max, float* r, float* g, float* b)
{
    float t = max_ref(0.0f, min_ref(v, max)) / max;
    t = cbrtf(t);
    *r = t; *g = 0.0f; *b = 1.0f - t;
}
/* mesh.h:19-24 (Color::SetColor: static_cast<uint32_t>(std::round(c * 255)) into one byte lane) and :26-33 (R(), G(), B(): & 0xFF) */
static unsigned char quantise(float c) { return (unsigned char)(((uint32_t)roundf(c * 255)) & 0xFFu); }
/* grid_to_mesh.cpp:84 / :181  float max = std::sqrt(std::pow(grid.VoxelsPerSide() * grid.VoxelSize(), 2) * 3): the product is a float,
 * std::pow(float, int) computes in double, so does the sqrt; the result is narrowed to float */
static float colour_range(unsigned n, float vs) { return (float)sqrt(pow((double)((float)n * vs), 2.0) * 3.0); }

/* grid_to_mesh.cpp:65-173.  coords [nverts * 3], rgb [nverts * 3] (R(), G(), B() of each vertex colour), faces / face normals [nindices].
 * Returns 0 and malloc'd arrays (vpo_free), -1 if an allocation failed. */
int vpo_grid_to_mesh_cubes(const uint32_t* words, const float* sdf, unsigned n, float voxel_size, const float origin[3],
                           float** coords, unsigned char** rgb, size_t* nverts, uint32_t** faces, uint32_t** face_normals, size_t* nindices)
{
    Ctx c; memset(&c, 0, sizeof c);
    c.n = n; c.words = words;
    const float vs = voxel_size, ox = origin[0], oy = origin[1], oz = origin[2];
    const float max = colour_range(n, vs);                                                    /* :84 */
    size_t cubes = 0;
    for (unsigned z = 0; z < n; ++z) for (unsigned y = 0; y < n; ++y) for (unsigned x = 0; x < n; ++x) {
        const size_t i = (size_t)x + (size_t)n * ((size_t)y + (size_t)n * z);
        if (voxel(&c, x, y, z) && !(fabsf(sdf[i]) == INFINITY)) ++cubes;
    }
    float* co = (float*)malloc((cubes ? cubes : 1) * 8 * 3 * sizeof(float));
    unsigned char* col = (unsigned char*)malloc((cubes ? cubes : 1) * 8 * 3);
    uint32_t* fa = (uint32_t*)malloc((cubes ? cubes : 1) * 36 * 4);
    uint32_t* fn = (uint32_t*)malloc((cubes ? cubes : 1) * 36 * 4);
    if (!co || !col || !fa || !fn) { free(co); free(col); free(fa); free(fn); return -1; }
    /* the twelve triangles of a cube and the normal slot of each face, in the order of :107-163 (BACK, FRONT, TOP, BOTTOM, RIGHT, LEFT) */
    static const uint32_t tri[12][3] = {{0, 2, 1}, {1, 2, 3}, {4, 5, 6}, {5, 7, 6}, {6, 3, 2}, {3, 6, 7}, {0, 1, 4}, {1, 5, 4}, {1, 3, 5}, {3, 7, 5}, {0, 4, 2}, {2, 4, 6}};
    static const uint32_t slot[6] = {0, 3, 1, 4, 2, 5};
    uint32_t numberVoxelInsert = 0;
    size_t v = 0, k = 0;
    for (unsigned z = 0; z < n; ++z)
        for (unsigned y = 0; y < n; ++y)
            for (unsigned x = 0; x < n; ++x) {
                const size_t i = (size_t)x + (size_t)n * ((size_t)y + (size_t)n * z);
                if (!voxel(&c, x, y, z) || fabsf(sdf[i]) == INFINITY) continue;               /* :89-90 */
                for (int dz = 0; dz <= 1; ++dz)
                    for (int dy = 0; dy <= 1; ++dy)
                        for (int dx = 0; dx <= 1; ++dx) {
                            co[v * 3 + 0] = ox + (x * vs) + (vs * dx);                         /* :95-99 */
                            co[v * 3 + 1] = oy + (y * vs) + (vs * dy);
                            co[v * 3 + 2] = oz + (z * vs) + (vs * dz);
                            float r, g, b;
                            sdf_to_rgb_ref(sqrtf(sdf[i]), max, &r, &g, &b);                   /* :100 */
                            col[v * 3 + 0] = quantise(r); col[v * 3 + 1] = quantise(g); col[v * 3 + 2] = quantise(b);   /* :102, alpha 1.0f */
                            ++v;
                        }
                for (int t = 0; t < 12; ++t) {
                    for (int j = 0; j < 3; ++j) { fa[k + j] = (numberVoxelInsert * 8) + tri[t][j]; fn[k + j] = slot[t / 2]; }
                    k += 3;
                }
                numberVoxelInsert++;
            }
    *coords = co; *rgb = col; *nverts = v; *faces = fa; *face_normals = fn; *nindices = k;
    return 0;
}

/* grid_to_mesh.cpp:175-201: every set voxel (an infinite sdf included: no test), its centre, the same colouring */
int vpo_grid_to_point_cloud(const uint32_t* words, const float* sdf, unsigned n, float voxel_size, const float origin[3],
                            float** coords, unsigned char** rgb, size_t* nverts)
{
    Ctx c; memset(&c, 0, sizeof c);
    c.n = n; c.words = words;
    const float vs = voxel_size, ox = origin[0], oy = origin[1], oz = origin[2];
    const float max = colour_range(n, vs);                                                    /* :182 */
    size_t count = 0;
    for (unsigned z = 0; z < n; ++z) for (unsigned y = 0; y < n; ++y) for (unsigned x = 0; x < n; ++x) count += (size_t)voxel(&c, x, y, z);
    float* co = (float*)malloc((count ? count : 1) * 3 * sizeof(float));
    unsigned char* col = (unsigned char*)malloc((count ? count : 1) * 3);
    if (!co || !col) { free(co); free(col); return -1; }
    size_t v = 0;
    for (unsigned z = 0; z < n; ++z)
        for (unsigned y = 0; y < n; ++y)
            for (unsigned x = 0; x < n; ++x) {
                if (!voxel(&c, x, y, z)) continue;                                             /* :186-187 */
                const size_t i = (size_t)x + (size_t)n * ((size_t)y + (size_t)n * z);
                co[v * 3 + 0] = ox + (x * vs) + (vs / 2);                                      /* :189-193 */
                co[v * 3 + 1] = oy + (y * vs) + (vs / 2);
                co[v * 3 + 2] = oz + (z * vs) + (vs / 2);
                float r, g, b;
                sdf_to_rgb_ref(sqrtf(sdf[i]), max, &r, &g, &b);                               /* :194-195 */
                col[v * 3 + 0] = quantise(r); col[v * 3 + 1] = quantise(g); col[v * 3 + 2] = quantise(b);
                ++v;
            }
    *coords = co; *rgb = col; *nverts = v;
    return 0;
}

void vpo_free(void* p) { free(p); }
