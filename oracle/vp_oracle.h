/*
 * vp_oracle.h -- CPU restatement of the reference's SEQUENTIAL voxelize / CSG / JFA path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / the timed CPU baseline.  The product path (libvphip.so) never
 * links, loads or calls it.
 *
 * Parity status, per function (round 6):
 *   PINNED by the reference itself (oracle/_ref: its own mesh_io.cpp, grid_to_mesh.cpp, csg/sequential.cpp, voxels_grid.cu and
 *   bounding_box.h compiled from /root/reference where they lie, oracle/Makefile; tests/test_reference_build.py):
 *       vpo_bounding_box / vpo_frame, vpo_csg, the bit layout, and the three exporters of oracle_export.c.
 *   PARITY UNPINNED in the formal sense: vpo_voxelize and vpo_jfa.  They are checked in tests/test_oracle_golden.py against
 *   tests/golden/survey_table.json -- the popcount / FNV-1a-64 / SDF-sum table the survey captured by running the reference's own
 *   sequential sources in this container -- but that run needed stand-in CUDA headers and has no committed
 *   recipe, the reference itself holds no golden vectors, and those two units cannot be part of oracle/_ref (vox/vox.h needs
 *   <cub/cub.cuh>, which the image lacks; jfa/sequential.cpp allocates through cudaMalloc at run time).  What _ref adds for them: the
 *   table's CSG column is reproduced by the reference's own CSG on the oracle's grids, hash for hash.
 *   tests/golden/own_oracle_runs.json holds outputs of THIS oracle for sizes the reference cannot run; they pin the GPU path to the
 *   oracle, not the oracle.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef VP_ORACLE_H
#define VP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vplib/src/bounding_box.h:22-61.  out = {minX,maxX,minY,maxY,minZ,maxZ}; returns longest side. */
float vpo_bounding_box(const float* xyz, size_t nverts, float out[6]);

/* apps/cli/main.cpp:65-87: frame shared by all meshes.  origin = bbox min, voxel = side / N. */
void vpo_frame(const float* xyz, size_t nverts, unsigned n, float origin[3], float* voxel_size);

/* vplib/src/vox/sequential.cpp:6-63.  XOR-accumulates into `words` (caller zero-fills for a
 * fresh grid).  words = ceil(n^3/32) uint32, voxel (x,y,z) -> bit (x + y*n + z*n*n),
 * LSB-first (vplib/src/grid/voxels_grid.h:116-129). */
void vpo_voxelize(uint32_t* words, unsigned n, float voxel_size, const float origin[3],
                  const float* xyz, const uint32_t* tri, size_t ntris);

/* vplib/src/csg/sequential.cpp:7-30 with the functors of vplib/src/csg/csg.h:14-30.
 * op: 1 union (a |= b), 2 intersection (a &= b), 3 difference (a &= ~b), 0 no-op. */
void vpo_csg(uint32_t* a, const uint32_t* b, size_t nwords, int op);

/* vplib/src/jfa/sequential.cpp:7-127.  `sdf` (n^3 floats) must be pre-filled by the caller
 * (the CLI uses -INFINITY, apps/cli/main.cpp:200) and is overwritten with the signed SQUARED
 * distance.  Uses the reference's own state layout (float sdf + float3 position, two copies).
 * max_passes < 0 runs all log2 passes; otherwise stops after that many (bench sampling).
 * Returns 0, or -1 if allocation failed. */
int vpo_jfa(const uint32_t* words, unsigned n, float voxel_size, const float origin[3],
            float* sdf, int max_passes);

/* Number of threads the JFA / CSG loops use (1 when built without OpenMP). */
int vpo_threads(void);

/* FNV-1a-64 over raw bytes -- the hash used for the golden table. */
uint64_t vpo_fnv1a64(const void* p, size_t nbytes);

/* popcount over words */
uint64_t vpo_popcount(const uint32_t* words, size_t nwords);

/* SDF summary used by the golden table: zeros, +inf count, -inf count, sum of positive finite,
 * sum of negative finite (double accumulation in index order), max, min. */
void vpo_sdf_stats(const float* sdf, size_t n, uint64_t counts[3], double sums[2], float minmax[2]);

#ifdef __cplusplus
}
#endif
#endif
