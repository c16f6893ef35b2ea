/*
 * vphip.h -- C ABI of libvphip.so, the MI355X (gfx950) implementation of the reference's
 * voxelize -> CSG -> JFA hot path.
 *
 * The reference (bigmat18/cuda-mesh-voxelization) has no FFI layer: its boundary is the vplib
 * C++ template API.  Each entry point below names the reference interface it stands behind;
 * the C++ mirror of that API (cuda_mesh_voxelization_amd/vplib/) and the Python harness call
 * nothing but these functions.  All citations are file:line under the reference repo.
 *
 * Conventions
 *   - Every function returns 0 on success, otherwise a non-zero code (hipError_t value, or
 *     VP_ERR_* below) and stores a message retrievable with vp_last_error().  Nothing throws,
 *     nothing calls exit(): the C++ mirror turns a non-zero code into the reference's
 *     print-and-exit behaviour (vplib/src/debug_utils.h:43-50).
 *   - Pointers named d_* are device pointers (any hipMalloc'd / torch-owned memory of the
 *     context's device); h_* are host pointers.  The caller owns every buffer.
 *   - Work is enqueued on the context's stream and is asynchronous unless stated otherwise.
 *   - Grid layout is the reference's (vplib/src/grid/voxels_grid.h:116-129,
 *     vplib/src/grid/grid.h:89-92): voxel (x,y,z) is bit (x + y*n + z*n*n) of a little-endian
 *     uint32 word array, LSB first; dense fields (sdf) are x-fastest float arrays.
 *   - A vp_frame may describe a Z-slab [z0,z1) of the global n^3 grid: buffers then hold only
 *     those planes (plane z0 first).  z0 = 0, z1 = n is the whole grid.
 */
#ifndef VPHIP_H
#define VPHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP_ABI_VERSION 6

enum {
    VP_OK = 0,
    VP_ERR_INVALID = 10001,     /* bad argument (null pointer, unsupported n, finite fill, ...) */
    VP_ERR_UNSUPPORTED = 10002, /* valid request this build cannot serve */
    VP_ERR_NOMEM = 10003
};

/* vplib/src/proc_utils.h:7-9  enum class Types {SEQUENTIAL, NAIVE, TILED, OPENMP}: the GPU values */
enum { VP_ALGO_NAIVE = 1, VP_ALGO_TILED = 2 };

/* vplib/src/csg/csg.h:10-12  enum class Op {VOID, UNION, INTERSECTION, DIFFERENCE} */
enum { VP_OP_VOID = 0, VP_OP_UNION = 1, VP_OP_INTERSECTION = 2, VP_OP_DIFFERENCE = 3 };

/* Grid frame: what VoxelsGrid carries besides its words (voxels_grid.h:39-43,160-169),
 * plus the Z-slab this buffer holds. */
typedef struct vp_frame {
    uint32_t n;           /* voxels per side of the GLOBAL grid (n % 32 == 0, 32 <= n <= 2048) */
    float    voxel_size;
    float    origin[3];
    uint32_t z0, z1;      /* planes held: z0 <= z < z1; multiples of 8 */
} vp_frame;

typedef struct vp_ctx vp_ctx;

/* ---- context ------------------------------------------------------------------------------
 * Replaces `cudaSetDevice(0)` + default stream + per-call cudaMalloc/cudaFree of the reference
 * (apps/cli/main.cpp:22-23, vplib/src/cuda_ptr.h:15-93).  The context owns a stream and a
 * grow-only workspace so that steady-state calls allocate nothing. */
int vp_device_count(int* out);                                    /* devices visible to the process */
int vp_ctx_create(int device, vp_ctx** out);
int vp_ctx_destroy(vp_ctx* ctx);
/* external != 0: enqueue on the caller's hipStream_t `hip_stream` (NULL = the device's null stream,
 * which is what torch's default stream is); external == 0: back to the context's own stream. */
int vp_ctx_set_stream(vp_ctx* ctx, void* hip_stream, int external);
int vp_ctx_sync(vp_ctx* ctx);
const char* vp_last_error(void);
int vp_abi_version(void);

/* ---- device memory (CudaPtr<T> equivalent, vplib/src/cuda_ptr.h:24-93) ---------------------
 * Alignment contract: every grid, id, sdf and workspace buffer handed to the entry points below must be 16-byte aligned -- what
 * hipMalloc, vp_malloc and the usual framework allocators return; the kernels move these buffers as 16-byte vectors.  A pointer
 * that is not is refused with VP_ERR_INVALID (slab planes inside such a buffer are aligned by construction: n % 32 == 0). */
int vp_malloc(vp_ctx* ctx, size_t bytes, void** d_out);
int vp_free(vp_ctx* ctx, void* d_ptr);
int vp_memset(vp_ctx* ctx, void* d_ptr, int byte_value, size_t bytes);           /* async */
/* CudaPtr's copy constructor / assignment: device-to-device deep copy (cuda_ptr.h:42-53).  async */
int vp_memcpy_d2d(vp_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* Measurement aid (no reference counterpart; SURVEY.md 8(d) asks for a stream-copy peak measured on the box): a plain 16-bytes-
 * per-lane grid-stride copy kernel on the context's stream -- the access shape of the CSG and prefix-XOR kernels with nothing
 * computed.  bench.py times it over 1 GiB and reports the HBM-bound kernels against that rate beside the 8 TB/s specification.
 * Buffers 16-byte aligned, bytes a multiple of 16.  async */
int vp_stream_copy(vp_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* Context-owned scratch: slot in [0, VP_WORKSPACE_SLOTS), grow-only, valid until the next call for the same slot
 * with a larger size, vp_ctx_release or vp_ctx_destroy.  What the Compute() wrappers use instead of the reference's
 * per-call cudaMalloc/cudaFree (vox/tiled.cu:496-575 allocates ~15 buffers per call). */
#define VP_WORKSPACE_SLOTS 8
int vp_ctx_workspace(vp_ctx* ctx, int slot, size_t bytes, void** d_out);
/* Frees every workspace slot and the JFA workspace (synchronises). */
int vp_ctx_release(vp_ctx* ctx);
int vp_upload(vp_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);        /* blocking */
int vp_download(vp_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);      /* blocking */

/* words in the buffer of a frame: n*n*(z1-z0)/32 */
size_t vp_grid_words(const vp_frame* f);
/* voxels in the buffer of a frame: n*n*(z1-z0) */
size_t vp_grid_voxels(const vp_frame* f);

/* ---- voxelize -----------------------------------------------------------------------------
 * Stands behind VOX::Compute<Types::NAIVE|TILED,T>(HostVoxelsGrid<T>&, const Mesh&)
 * (vplib/src/vox/vox.h:107-111, vox/naive.cu:86-122, vox/tiled.cu:488-576); the result is the
 * bitmask of VOX::Compute<Types::SEQUENTIAL> (vox/sequential.cpp:6-63).
 *   d_xyz   nverts x 3 float  (Mesh::Coords, mesh.h:133-170)
 *   d_tri   ntris x 3 uint32  (Mesh::FacesCoords; ntris = indices/3 as in sequential.cpp:16)
 *   accumulate = 0: d_words is overwritten (GPU variants of the reference replace the grid,
 *                   vox/tiled.cu:572-575); 1: XOR into the existing words (sequential semantics).
 * Asynchronous in steady state: list sizes stay on the device (no read-back, no stream synchronisation).  The FIRST call, and
 * any call that needs a larger internal buffer than the context has (grow-only), synchronises the stream and allocates:
 * the record list of large triangles (80 B each; 5 MiB, then what earlier calls counted + 25 %, at most one per triangle -- a
 * large triangle that finds it full is rasterised in place, so any size is correct), the tile work queue (>= 4 MiB, grows
 * the same way) and the 8 x 8-column tile tables. */
int vp_voxelize(vp_ctx* ctx, const vp_frame* f, uint32_t* d_words,
                const float* d_xyz, size_t nverts, const uint32_t* d_tri, size_t ntris,
                int algo, int accumulate);

/* ---- CSG ----------------------------------------------------------------------------------
 * Stands behind CSG::Compute<Types::NAIVE,T,func>(grid1, grid2, Op) (vplib/src/csg/csg.h:35-36,
 * csg/naive.cu:26-64): d_a[i] = d_a[i] op d_b[i] with the functors of csg.h:14-30. */
int vp_csg(vp_ctx* ctx, uint32_t* d_a, const uint32_t* d_b, size_t nwords, int op);

/* ---- JFA ----------------------------------------------------------------------------------
 * Stands behind JFA::Compute<Types::NAIVE|TILED,T>(HostVoxelsGrid<T>&, HostGrid<float>&)
 * (vplib/src/jfa/jfa.h:42-43, jfa/naive.cu:121-180, jfa/tiled.cu:244-337); results are those of
 * the sequential path (jfa/sequential.cpp:7-127): signed SQUARED distance, +inside, -outside.
 *
 * State between passes is one packed id per voxel -- the coordinates of the nearest seed found so far -- instead of the
 * reference's float sdf + float3 position (jfa/sequential.cpp:69-70); distances are recomputed from it with the reference's
 * expressions.  Two layouts:
 *   PLAIN ids   vp_jfa_id_bytes(f) = 4 bytes per voxel for n <= 1024, 8 for n <= 2048, x-fastest, in planes the CALLER addresses
 *               (vp_jfa_init / vp_jfa_pass / vp_jfa_finalize below): the form of the direct kernel (VP_ALGO_NAIVE).
 *   WINDOWS     buffers in the library's own layout (vp_jfa_window_*): what the tile kernels (VP_ALGO_TILED, n >= 96) run on.
 * Id buffers are opaque either way.
 *
 * vp_jfa runs seeding + all passes + the id -> sdf conversion on one device for a whole-grid frame.
 *   fill_unset  value the caller pre-filled the sdf with (apps/cli/main.cpp:200 uses -INFINITY);
 *               must be +-infinity (a finite fill is undefined behaviour in the reference).
 *   d_work      scratch of vp_jfa_workspace_bytes(f) bytes (two id volumes + border mask), or NULL: the
 *               context then keeps a grow-only workspace of its own (work_bytes ignored).
 *   algo        VP_ALGO_NAIVE: direct kernel; VP_ALGO_TILED: tile kernels.  Same results. */
size_t vp_jfa_workspace_bytes(const vp_frame* f);
size_t vp_jfa_id_bytes(const vp_frame* f);
/* Bytes of id state per voxel that vp_jfa really streams per pass -- the S of SURVEY.md 8(d) "as implemented": 4 for n <= 1024; above
 * that 5 with VP_ALGO_TILED (windows: a 32-bit word plane + a byte plane), 8 with VP_ALGO_NAIVE.  Measurement only. */
size_t vp_jfa_state_bytes(const vp_frame* f, int algo);
int vp_jfa(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, float fill_unset,
           float* d_sdf, void* d_work, size_t work_bytes, int algo);
/* vp_jfa in the two parts the reference times separately ("::Initialization" = seeding, jfa/tiled.cu:265-290;
 * "::Processing" = the passes, jfa/tiled.cu:292-334): vp_jfa == vp_jfa_start + vp_jfa_run on the same workspace.
 * This is the sequence JFA::Compute<NAIVE|TILED> and the CLI run -- the same kernels the benchmark times.
 * The context records what vp_jfa_start left in the workspace (grid pointer, n, algo, workspace, border mask or init ids);
 * vp_jfa_run returns VP_ERR_INVALID unless exactly that start preceded it -- one start serves one run, and the record is
 * dropped when the workspace is released, regrown or freed, and when the grid buffer or the workspace is written through this ABI
 * in between -- vp_voxelize, vp_csg, vp_upload, vp_memset, vp_memcpy_d2d, vp_stream_copy into ANY part of it (a slab of the grid, a
 * sub-range of the workspace: the byte ranges are compared), vp_free, or the buffer handed out again by vp_ctx_workspace: "the same
 * grid" means the same CONTENTS -- the border mask of the start no longer describes them. */
int vp_jfa_start(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, void* d_work, size_t work_bytes, int algo);
int vp_jfa_run(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, float fill_unset,
               float* d_sdf, void* d_work, size_t work_bytes, int algo);

/* The stages separately on PLAIN ids in caller-addressed planes, for Z-slab sharding (halo exchange happens between calls).
 * Halo pointers may be NULL where the slab touches the global boundary.
 *   init:  d_plane_below / d_plane_above = bitmask plane z0-1 / z1 (n*n/32 words each).
 *   pass:  step k; d_minus holds id planes [z0-k, min(z0, z1-k)), d_plus holds
 *          [max(z1, z0+k), z1+k), each clipped to the global grid but indexed from the unclipped
 *          start (plane p of d_minus is global plane z0-k+p).
 *   finalize: ids -> float sdf for the slab.
 * VP_ALGO_NAIVE serves any such buffers (the direct kernel: one thread per voxel -- the independent form the tile kernels are tested
 * against, pass by pass).  VP_ALGO_TILED: any buffers as well -- the tile kernel where the plain ids ARE a window (4-byte ids, n <= 1024,
 * and the three buffers ONE run of consecutive planes: d_minus + k planes == d_in, d_plus == d_in + the slab -- whole grids, and slabs
 * with their halo planes right below / above them), the table kernel below n = 96, the direct kernel for everything else (8-byte ids,
 * halo buffers of their own): same ids bit for bit, but not the tile kernel's speed -- hand such state over as a window for that. */
int vp_jfa_init(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words,
                const uint32_t* d_plane_below, const uint32_t* d_plane_above, void* d_ids);
int vp_jfa_pass(vp_ctx* ctx, const vp_frame* f, uint32_t k, const void* d_in,
                const void* d_minus, const void* d_plus, void* d_out, int algo);
int vp_jfa_finalize(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, const void* d_ids,
                    float fill_unset, float* d_sdf);
/* The last pass (step k = 1) and the finalize in one call: where the kernel supports it the pass writes
 * the sdf directly and the id volume is neither written nor re-read; d_scratch (one id volume) is used
 * only when it does not.  Same result as vp_jfa_pass(k = 1) + vp_jfa_finalize. */
int vp_jfa_last_pass(vp_ctx* ctx, const vp_frame* f, const void* d_in, const void* d_minus,
                     const void* d_plus, void* d_scratch, const uint32_t* d_words, float fill_unset,
                     float* d_sdf, int algo);

/* Id WINDOWS: the slab form of the tile kernels (VP_ALGO_TILED, n >= 96) -- what the multi-GPU pipelines run (cuda_mesh_voxelization_amd/slab.py,
 * vp_multi_* below), and what vp_jfa itself runs on its workspace.  A window is a buffer of `planes` id planes in the library's layout:
 * vp_jfa_window_bytes(f, planes) bytes -- 4-byte ids up to n = 1024; above that `planes` planes of 32-bit words followed by `planes`
 * planes of bytes (5 instead of 8 bytes per voxel in memory, per pass and on the wire).  A call names a frame f (the planes [z0, z1) it
 * produces) and says where plane z0 sits in the window (`at`); the planes around it are the halo the pass reads.  Nothing else is
 * assumed about which global planes a window holds: a ghost-plane pipeline keeps whole volumes (planes = n, at = z0), a hybrid one the
 * planes a rank touches, a halo pipeline its slab with room for the halos on both sides.
 *   stride  where the planes z - k and z + k of a plane z are found: `stride` planes below / above it.  stride = k for a window of
 *           consecutive planes.  A pass whose step spans whole slabs (k >= z1 - z0: the wide passes of a halo pipeline) keeps
 *           [slab holding z - k | own slab | slab holding z + k] and passes stride = the slab height: the tile kernel then runs on the
 *           received slabs where they landed.
 *   vp_jfa_window_span     where the planes [p0, p1) of a window live, for whoever moves them (halo exchange): one or two byte ranges
 *                          relative to d_ids (bytes[1] = 0 for 4-byte ids)
 *   vp_jfa_window_clear    every id := "none" (a pipeline whose regions are rounded outwards to whole tiles reads planes nobody
 *                          produced: cleared once, they hold ids of the window's own layout -- never stale bytes of another one)
 *   vp_jfa_window_init     seeding: ids of the planes of f from its bitmask (halo planes as in vp_jfa_init)
 *   vp_jfa_window_first_pass   the pass k = n/2 of the planes of f straight from the border bitmask of the WHOLE grid (vp_surface on
 *                          a whole-grid frame): no init ids are written or read (jfa/sequential.cpp:55-60: before any pass a border
 *                          voxel's seed is itself and nothing else has one).  n % 128 == 0 (vp_jfa_can_start_from_mask).
 *   vp_jfa_window_first_two    the passes k = n/2 AND k = n/4 of a WHOLE grid in one launch from its border bitmask; whole-grid frame,
 *                          a window of n planes, at = 0 (vp_jfa_can_fuse_first_two: any n >= 96)
 *   vp_jfa_window_pass     one pass with step k over the planes of f: reads `in` (the planes of f and `stride` planes on each side of
 *                          them, as far as the grid goes), writes the planes of f in `out` (same planes / at as `in`)
 *   vp_jfa_window_last_pass    step 1 fused with the id -> sdf conversion; d_words_region / d_sdf_region hold the planes [z0, z1) only
 * Same results as the plain-id calls, bit for bit. */
typedef struct vp_window {
    void*    d_ids;       /* the buffer, 16-byte aligned */
    size_t   bytes;       /* its size: at least vp_jfa_window_bytes(f, planes) -- checked by every call (VP_ERR_INVALID) */
    uint32_t planes;      /* id planes the buffer holds (the layout depends on it: above n = 1024 the byte planes follow `planes` word planes) */
    uint32_t at;          /* index inside the buffer of plane z0 of the frame given with it */
} vp_window;
size_t vp_jfa_window_bytes(const vp_frame* f, uint32_t planes);
int vp_jfa_window_span(const vp_frame* f, uint32_t planes, uint32_t p0, uint32_t p1, size_t offset[2], size_t bytes[2]);
int vp_jfa_window_clear(vp_ctx* ctx, const vp_frame* f, const vp_window* w);
int vp_jfa_window_init(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, const uint32_t* d_plane_below,
                       const uint32_t* d_plane_above, const vp_window* out);
int vp_jfa_can_start_from_mask(const vp_frame* f, int algo);
int vp_jfa_window_first_pass(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_border_grid, const vp_window* out);
int vp_jfa_can_fuse_first_two(const vp_frame* f, int algo);
int vp_jfa_window_first_two(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_border_grid, const vp_window* out);
int vp_jfa_window_pass(vp_ctx* ctx, const vp_frame* f, uint32_t k, const vp_window* in, const vp_window* out, uint32_t stride);
int vp_jfa_window_last_pass(vp_ctx* ctx, const vp_frame* f, const vp_window* in, const vp_window* scratch, uint32_t stride,
                            const uint32_t* d_words_region, float fill_unset, float* d_sdf_region);

/* CYCLIC plane distribution -- the first phase of the transposed multi-GPU pipeline (VP_MULTI_TRANSPOSE below).
 * The reference's pass with step k reads, for a voxel of plane z, the planes z - k, z, z + k and nothing else (jfa/sequential.cpp:72: k = n/2
 * .. 1; :86-94: neighbours at -+k).  With the planes dealt cyclically over G ranks -- plane z on rank z mod G, kept at index z / G of that
 * rank's window of n / G planes -- every pass whose step is a multiple of G finds all three planes on the rank that owns z: no exchange at
 * all, chains of full length, and each rank does exactly 1/G of the pass.  G a power of two, n / G a multiple of 8 planes.
 *   vp_jfa_cyclic_passes            how many passes of the sequence n/2, n/4, ... have such a step, counted from the first (0: none, or fewer
 *                                   than the two the fused start produces -- use another pipeline)
 *   vp_jfa_window_first_two_cyclic  the passes k = n/2 and k = n/4 of rank `rank`'s planes from the border bitmask of the WHOLE grid
 *                                   (vp_jfa_window_first_two restricted to the tiles of the rank's z residues); whole-grid frame, a window of
 *                                   n / ranks planes, at = 0
 *   vp_jfa_window_pass_cyclic       one pass with step k (a multiple of `ranks`, one of the first vp_jfa_cyclic_passes steps) over rank `rank`'s
 *                                   planes; whole-grid frame, two windows of n / ranks planes, at = 0.  Ids hold global coordinates, so the
 *                                   planes can be handed to the slab form of the calls above as they are
 *   vp_jfa_window_interleave        the re-deal into slabs: `in` holds `ranks` chunks of `count` planes, chunk s = the planes b0 + s, b0 + s +
 *                                   ranks, ... of some run of ranks * count consecutive planes as rank s kept them (what an all-to-all of
 *                                   contiguous plane ranges delivers); plane out->at + j * ranks + s of `out` := plane s * count + j of `in`.
 *                                   `in` is a window of exactly ranks * count planes (in->at ignored)
 * Same results as the consecutive-plane calls, id for id. */
int vp_jfa_cyclic_passes(const vp_frame* f, uint32_t ranks);
int vp_jfa_window_first_two_cyclic(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_border_grid, const vp_window* out, uint32_t ranks, uint32_t rank);
int vp_jfa_window_pass_cyclic(vp_ctx* ctx, const vp_frame* f, uint32_t k, const vp_window* in, const vp_window* out, uint32_t ranks, uint32_t rank);
int vp_jfa_window_interleave(vp_ctx* ctx, const vp_frame* f, const vp_window* in, const vp_window* out, uint32_t ranks, uint32_t count);

/* "Surface" output (README.md:9): the border-voxel mask that JFA seeds
 * from (jfa/sequential.cpp:24-64), as a bitmask with the grid's layout. */
int vp_surface(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words,
               const uint32_t* d_plane_below, const uint32_t* d_plane_above, uint32_t* d_border_words);

/* ---- export: ordered compaction of the grid into voxel records --------------------------------
 * The reference's exporters walk all n^3 voxels on the CPU (vplib/src/mesh/grid_to_mesh.cpp:10-201); this is the
 * accelerated front end: the walk is a GPU stream and the host only sees the voxels it will emit, in the exporter's own
 * scan order (z, y, x), so the files it writes are byte-identical.  Whole-grid frames only.
 *   mode VP_EXTRACT_SET      every set voxel (point cloud, sdf cubes: grid_to_mesh.cpp:133-201)
 *   mode VP_EXTRACT_EXPOSED  set voxels with a face towards an unset voxel or the outside of the grid, with the mask of
 *                            those faces (visible-surface mesh)
 *   mode VP_EXTRACT_FACES    every set voxel, with that mask: VoxelsGridToMeshCompressed (grid_to_mesh.cpp:10-60, grid_to_mesh.h:25-92)
 *                            emits every face of every set voxel ONCE, interior faces included -- the three faces on a voxel's plus sides
 *                            always, a face on a minus side iff the voxel behind it is unset (else that voxel emitted it already)
 *   record = linear voxel index x + n (y + n z) in bits 0..39 | face mask << 40 (bit = axis * 2 + side; X, Y, Z; 0 = minus)
 * vp_extract_count runs the counting pass and returns the number of records (blocking); vp_extract then writes up to
 * `capacity` records (and, when d_sdf and d_values are given, the sdf value of each voxel) -- it must follow a count call for
 * the same grid and mode, else VP_ERR_INVALID.  "Same grid" means same contents: the count is forgotten as soon as d_words is
 * written through this ABI (vp_voxelize, vp_csg, vp_upload, vp_memset, vp_memcpy_d2d), freed, or handed out again by
 * vp_ctx_workspace; a caller that writes the buffer with its own kernels must count again itself. */
enum { VP_EXTRACT_SET = 0, VP_EXTRACT_EXPOSED = 1, VP_EXTRACT_FACES = 2 };
int vp_extract_count(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, int mode, uint64_t* h_count);
int vp_extract(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, int mode, const float* d_sdf,
               uint64_t* d_records, float* d_values, size_t capacity);

/* ---- several GPUs of one node: Z-slabs -------------------------------------------------------
 * Replaces the reference's hard-wired device 0 (apps/cli/main.cpp:22-23) when more than one device is given: ONE process,
 * one context per device, rank r owns the planes [r n/G, (r+1) n/G) of the grid and of the sdf; G must divide n
 * into slabs of a multiple of 8 planes.  The stages are the slab forms of the calls above (vp_frame.z0 / z1); the JFA state of a rank
 * lives in two id windows (vp_jfa_window_*):
 *   voxelize / CSG   no exchange (columns and words are independent)
 *   JFA              VP_MULTI_HALO : windows of 3 n/G planes, [slab holding z - k | own slab | slab holding z + k].  Bitmask planes z0-1 /
 *                                    z1 before the seeding and, before the pass with step k, the id planes [z0-k, min(z0, z1-k)),
 *                                    [max(z1, z0+k), z1+k) travel device to device (hipMemcpyPeerAsync behind stream events; no host
 *                                    synchronisation inside the JFA): next to the slab for k <= n/2G, a slab height away for the steps
 *                                    that span whole slabs (stride = n/G, see vp_jfa_window_pass)
 *                    VP_MULTI_GHOST: the bitmask slabs are all-gathered once (n^3/8 bytes) and every device recomputes the
 *                                    ghost planes its later passes reach: no exchange between passes, two windows of the whole grid
 *                                    per device
 *                    VP_MULTI_HYBRID: ghost planes for the passes with k > nz/2 (as far as the later such passes reach), halo
 *                                    planes from the two adjacent devices for the passes with k <= nz/2; the windows hold only
 *                                    the planes a device touches (vp_multi_window): capacity, not speed
 *                    VP_MULTI_TRANSPOSE: the planes are dealt cyclically (plane z on device z mod G) for every pass whose step is a multiple
 *                                    of G -- no exchange, no ghost planes, 1/G of each pass per device (vp_jfa_window_*_cyclic) -- then ONE
 *                                    re-deal into slabs widened by the reach of the remaining log2 G passes: one peer copy per pair of
 *                                    devices, n^3 S / G + margins bytes into each device per job (0.49 GiB at n = 1024, G = 8, against
 *                                    3.5 GiB of halos), compute per device ~ 1 / G of one device's.  G a power of two; where no step is a
 *                                    multiple of G it is the ghost mode
 * Results are bit-identical to the single-device calls for any G.  The sharded forms run the tile kernels whatever `algo` says (both
 * algorithms give the same sdf); grids below their range (n < 96) are not sharded: every device computes the whole grid with `algo`
 * and keeps its slab.  `devices` may name one device several times (several
 * contexts on it): that is how the parity tests run on a one-GPU box.  Grid, sdf and mesh stay resident on the devices
 * between calls; host arrays are whole-grid arrays in the reference's layout. */
typedef struct vp_multi vp_multi;
enum { VP_MULTI_HALO = 0, VP_MULTI_GHOST = 1, VP_MULTI_HYBRID = 2, VP_MULTI_TRANSPOSE = 3 };
int vp_multi_create(const int* devices, int ndev, vp_multi** out);
int vp_multi_destroy(vp_multi* m);
int vp_multi_count(const vp_multi* m);
vp_ctx* vp_multi_ctx(vp_multi* m, int rank);                      /* the context of a rank (timers, stream) */
int vp_multi_sync(vp_multi* m);
/* broadcast of the mesh to every device (blocking) -- Mesh::Coords / Mesh::FacesCoords as in vp_voxelize */
int vp_multi_set_mesh(vp_multi* m, const float* h_xyz, size_t nverts, const uint32_t* h_tri, size_t ntris);
/* every device rasterises the resident mesh into its slab of a fresh grid with frame f (async) */
int vp_multi_voxelize(vp_multi* m, const vp_frame* f, int algo);
/* scatter a host grid into the slabs / gather the slabs (blocking) */
int vp_multi_set_grid(vp_multi* m, const vp_frame* f, const uint32_t* h_words);
int vp_multi_get_grid(vp_multi* m, uint32_t* h_words);
/* resident grid = resident grid op h_other (CSG::Compute's "result in the first grid", csg/naive.cu:62); blocking.
 * nwords = words of h_other; must equal the resident grid's (the reference requires equal grids, csg/naive.cu:30-33). */
int vp_multi_csg(vp_multi* m, const uint32_t* h_other, size_t nwords, int op);
/* JFA of the resident grid into the resident slab sdfs (async); vp_multi_get_sdf gathers them (blocking) */
int vp_multi_jfa(vp_multi* m, float fill_unset, int algo, int mode);
int vp_multi_get_sdf(vp_multi* m, float* h_sdf);
/* device-to-device bytes the last vp_multi_jfa enqueued (halo planes / the bitmask all-gather) */
uint64_t vp_multi_bytes_moved(const vp_multi* m);
/* JFA state a rank held during the last vp_multi_jfa: the global planes [lo, hi) its two id windows are addressed by -- the whole grid with
 * VP_MULTI_GHOST, the slab with VP_MULTI_HALO (its windows also hold the two slabs received from z -+ k), the planes the rank touches
 * with VP_MULTI_HYBRID, the widened slab of the second phase with VP_MULTI_TRANSPOSE -- and the bytes of device memory in the id windows
 * that job used.  Any out pointer may be NULL. */
int vp_multi_window(const vp_multi* m, int rank, uint32_t* lo, uint32_t* hi, uint64_t* id_bytes);

/* ---- host-in / host-out conveniences (the reference's Compute() calling convention) -------
 * Upload, run, download, synchronise -- what every reference Compute<NAIVE|TILED> does
 * (vox/tiled.cu:504-575, csg/naive.cu:38-63, jfa/tiled.cu:254-336).  Whole-grid frames only. */
int vp_voxelize_host(vp_ctx* ctx, const vp_frame* f, uint32_t* h_words,
                     const float* h_xyz, size_t nverts, const uint32_t* h_tri, size_t ntris, int algo);
int vp_csg_host(vp_ctx* ctx, uint32_t* h_a, const uint32_t* h_b, size_t nwords, int op);
int vp_jfa_host(vp_ctx* ctx, const vp_frame* f, const uint32_t* h_words, float fill_unset,
                float* h_sdf, int algo);

/* ---- per-kernel timing (PROFILING_SCOPE equivalent for device time, vplib/src/profiling.h:8-33)
 * When enabled, every kernel launch is bracketed by hipEvents on the context's stream. */
enum {
    VP_K_VOX_SETUP = 0, VP_K_VOX_SCAN, VP_K_VOX_SCATTER, VP_K_VOX_TILE, VP_K_VOX_NAIVE,
    VP_K_VOX_FILL, VP_K_CSG, VP_K_JFA_INIT,
    VP_K_JFA_PASS,      /* direct kernel (VP_ALGO_NAIVE) and the small-grid table kernel (n < 96) */
    VP_K_JFA_FINAL, VP_K_SURFACE,
    /* the tile kernels of VP_ALGO_TILED (n >= 96), one key per variant -- their algorithmic bytes differ: */
    VP_K_JFA_FIRST,     /* k = n/2 straight from the border mask: 4 n^3 + n^3/8 bytes (S = 4) */
    VP_K_JFA_SPARSE,    /* k >= n/4: 2 S n^3 */
    VP_K_JFA_DENSE,     /* k <  n/4: 2 S n^3 */
    VP_K_JFA_LAST,      /* k = 1 fused with the id -> sdf conversion: S n^3 + 4 n^3 + n^3/8 */
    VP_K_EXTRACT,       /* vp_extract_count / vp_extract: 2 n^3/8 + records */
    VP_K_VOX_ZERO,      /* the voxelizer's zero-fill of the toggle grid (+ the tile histogram): n^3/8 */
    VP_K_JFA_REDEAL,    /* vp_jfa_window_interleave: 2 S x the planes woven */
    VP_K_COUNT
};
int vp_prof_enable(vp_ctx* ctx, int on);
/* Restricts the bracketing to the keys whose bit is set (bit i = key i; default: all).  An event pair costs ~3 us of stream
 * time (0.11 ms over the 18 launches of a 512^3 step, 3 %): a caller that times ONE kernel inside a wall-clock region
 * selects just that key. */
int vp_prof_select(vp_ctx* ctx, uint64_t kernel_mask);
int vp_prof_reset(vp_ctx* ctx);
/* Synchronises the stream, folds pending events in, returns total ms and launch count. */
int vp_prof_get(vp_ctx* ctx, int kernel, double* total_ms, uint64_t* launches);
const char* vp_prof_name(int kernel);

#ifdef __cplusplus
}
#endif
#endif /* VPHIP_H */
