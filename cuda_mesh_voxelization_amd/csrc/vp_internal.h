// vp_internal.h -- shared declarations of libvphip.so (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/vphip.h"

namespace vp {

// JFA state "no seed yet" of the two 32-bit id formats (jfa_common.h, IdU<9> / IdU<10>): y and z fields all ones, x = 2^BITS
constexpr uint32_t kNone9 = 0xFF9FF200u;    // n <= 512
constexpr uint32_t kNone10 = 0xFFDFFC00u;   // 512 < n <= 1024
constexpr int kTile = 8;                  // voxelizer tile: 8x8 (y,z) columns = one wave64
constexpr int kRecDwords = 20;            // per-triangle record, see vox.hip

// Device-side copy of vp_frame plus derived constants.
struct Frame {
    uint32_t n;        // voxels per side (global)
    uint32_t w;        // words per x-row = n / 32
    uint32_t z0, z1;   // slab
    float vs, ox, oy, oz;
};

inline Frame make_frame(const vp_frame* f)
{
    Frame r;
    r.n = f->n; r.w = f->n / 32; r.z0 = f->z0; r.z1 = f->z1;
    r.vs = f->voxel_size; r.ox = f->origin[0]; r.oy = f->origin[1]; r.oz = f->origin[2];
    return r;
}

struct ProfSpan { int kernel; hipEvent_t a, b; };

struct Buffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace vp

struct vp_ctx {
    int device = 0;
    int cus = 256;                             // compute units of the device (workgroup slots = cus x workgroups per CU)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // grow-only workspaces
    vp::Buffer rec, tile_cnt, tile_off, tile_cur, pairs, scratch, none_row;
    vp::Buffer jfa_work;                       // vp_jfa* with d_work = NULL: two id volumes + border mask
    // what the last vp_jfa_start left in its workspace, so that vp_jfa_run can refuse anything else (a run on a workspace that
    // holds init ids where it expects a border mask, or nothing at all, would produce an sdf from stale memory)
    struct JfaStarted {
        bool valid = false;
        bool mask = false;                     // border mask behind the two id volumes (fast sequence) / init ids in the first volume
        uint32_t n = 0;
        int algo = 0;
        const void* work = nullptr;
        size_t work_bytes = 0;
        const uint32_t* words = nullptr;
    } jfa_started;
    vp::Buffer slots[VP_WORKSPACE_SLOTS];      // vp_ctx_workspace
    // vp_extract_*: block counts / offsets of the last count call and what it was for
    vp::Buffer ext_cnt, ext_off;
    const uint32_t* ext_words = nullptr;
    int ext_mode = -1;
    uint32_t ext_n = 0;
    uint64_t ext_total = 0;
    // voxelizer: work-queue size of an earlier call, copied back lazily (never waited for) to size the next call's queue
    uint32_t* vox_total_host = nullptr;
    hipEvent_t vox_total_event = nullptr;
    bool vox_total_pending = false;
    uint64_t vox_total_seen = 0;
    uint64_t vox_nbig_seen = 0;                // large triangles an earlier call counted (sizes the record list)
    bool vox_counts_known = false;             // at least one such count has come back
    // the job (triangle buffer, count, grid side, slab) whose large-triangle count last came back as ZERO: only that very job leaves the
    // list kernels out (a coarse mesh that follows a fine one on the same context must not be walked triangle by triangle)
    struct VoxJob { const void* tri = nullptr; size_t ntris = 0; uint32_t n = 0, z0 = 0, z1 = 0;
                    bool operator==(const VoxJob& o) const { return tri == o.tri && ntris == o.ntris && n == o.n && z0 == o.z0 && z1 == o.z1; } };
    VoxJob vox_pending_job, vox_nolist_job;
    // profiling
    bool prof_on = false;
    uint64_t prof_mask = ~0ull;                                    // timing keys that get events (vp_prof_select)
    std::vector<vp::ProfSpan> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[VP_K_COUNT] = {};
    uint64_t prof_n[VP_K_COUNT] = {};
};

namespace vp {

int set_error(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define VP_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) return vp::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define VP_TRY(call)                 \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != 0) return rc_;    \
    } while (0)

int reserve(vp_ctx* ctx, Buffer& b, size_t bytes, bool headroom = true);
void release(Buffer& b);

// RAII-less profiling bracket: begin() before the launch, end() after.
struct ProfScope {
    vp_ctx* ctx; int kernel; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(vp_ctx* c, int k);
    ~ProfScope();
};

// ---- id windows (include/vphip.h, vp_jfa_window_*) ----
// A buffer of `planes` id planes in the library's layout -- 4-byte ids (jfa_common.h: IdU<9> / IdU<10>) up to n = 1024; above that `planes`
// planes of 32-bit words followed by `planes` planes of bytes (IdC: 5 bytes per voxel).  `at` = index, inside the buffer, of plane z0 of
// the frame a call is made with.
struct IdWin {
    char* base = nullptr;
    uint32_t planes = 0, at = 0;
};
constexpr uint32_t kTileMinN = 96;   // VP_ALGO_TILED runs the tile kernels from this side on (n % 32 == 0: 96, 128, ...); the table kernel below it
                                     // (64 / 32: 0.063 / 0.061 ms per step against 0.117 / 0.102: launch-bound, profiles/r04/ab_tilemin.txt)
inline bool win_compact(uint32_t n) { return n > 1024; }
inline size_t win_plane_bytes(uint32_t n) { return (size_t)n * n * 4; }                     // of the (word) plane
inline size_t win_bytes(uint32_t n, uint32_t planes) { return (size_t)n * n * planes * (win_compact(n) ? 5 : 4); }
inline char* win_words(const IdWin& w, uint32_t n, int64_t plane)
{
    return reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(w.base) + (uintptr_t)(plane * (int64_t)win_plane_bytes(n)));
}
inline char* win_bytes_plane(const IdWin& w, uint32_t n, int64_t plane)
{
    return reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(w.base) + (uintptr_t)((int64_t)w.planes * (int64_t)win_plane_bytes(n) + plane * (int64_t)n * n));
}

// ---- stage launchers (each enqueues on ctx->stream) ----
int launch_voxelize(vp_ctx* ctx, const Frame& f, uint32_t* d_words, const float* d_xyz, size_t nverts,
                    const uint32_t* d_tri, size_t ntris, int algo, int accumulate);
int launch_csg(vp_ctx* ctx, uint32_t* d_a, const uint32_t* d_b, size_t nwords, int op);
int launch_stream_copy(vp_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);   // 16 B per lane: the measured HBM copy rate
// jfa_seed.hip
size_t jfa_id_bytes(const Frame& f);                              // plain ids: 4 (n <= 1024) or 8 bytes
int launch_jfa_init(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* below,
                    const uint32_t* above, void* d_ids, uint32_t* d_border_words);
int launch_jfa_pass(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                    const void* d_plus, void* d_out, int algo);
bool jfa_pass_can_fuse_final(const Frame& f, uint32_t k, int algo);
int launch_jfa_pass_ex(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                       const void* d_plus, void* d_out, int algo, const uint32_t* d_words, float fill, float* d_sdf);
int launch_jfa_final(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const void* d_ids,
                     float fill, float* d_sdf);
bool jfa_can_start_from_mask(const Frame& f, int algo);         // n % 128 == 0: the pass k = n/2 straight from the border mask
bool jfa_can_fuse_first_two(const Frame& f, int algo);          // passes n/2 and n/4 in one launch from the border mask (whole grids)
int launch_win_init(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* below, const uint32_t* above, const IdWin& out);
int launch_win_first_pass(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, const IdWin& out);
int launch_win_clear(vp_ctx* ctx, uint32_t n, const IdWin& w);
// One pass with step k over the planes of f (tile kernel, jfa_dense.hip); the planes z -+ k of a plane z are found `stride` planes below /
// above it in the window (stride = k for a window of consecutive planes).  d_sdf != nullptr: the last pass, fused with the id -> sdf conversion.
int launch_win_pass(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t stride,
                    const uint32_t* d_words, float fill, float* d_sdf);
#define VP_DENSE_DECL(NAME) int NAME(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t stride, const uint32_t* d_words, float fill, float* d_sdf)
VP_DENSE_DECL(launch_dense_id9_pass);  VP_DENSE_DECL(launch_dense_id9_last);      // per id format; _last = fused with the id -> sdf conversion
VP_DENSE_DECL(launch_dense_id10_pass); VP_DENSE_DECL(launch_dense_id10_last);
VP_DENSE_DECL(launch_dense_idc_pass);  VP_DENSE_DECL(launch_dense_idc_last);
#undef VP_DENSE_DECL
// jfa_first_two.hip: passes n/2 and n/4 of a whole grid from its border mask into a window of n planes
// ranks > 1: the planes of rank `rank` of a cyclic distribution of the grid over `ranks` ranks into a window of n / ranks planes
int launch_win_first_two(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, const IdWin& out, uint32_t ranks = 1, uint32_t rank = 0);
// ---- cyclic plane distribution (the first phase of the transposed multi-GPU pipeline; include/vphip.h, vp_jfa_window_*_cyclic) ----
// passes of the sequence n/2, n/4, ... that can run on planes dealt cyclically to `ranks` ranks (their steps are multiples of `ranks`): 0 or >= 2
uint32_t jfa_cyclic_passes(uint32_t n, uint32_t ranks);
// one pass with step k (a multiple of `ranks`) over the n / ranks planes of rank `rank`: jfa_dense.hip, jfa_pass_dense<CYC>
int launch_win_pass_cyclic(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t ranks, uint32_t rank);
#define VP_CYCLIC_DECL(NAME) int NAME(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t ranks, uint32_t rank)
VP_CYCLIC_DECL(launch_cyclic_id9_pass); VP_CYCLIC_DECL(launch_cyclic_id10_pass); VP_CYCLIC_DECL(launch_cyclic_idc_pass);
#undef VP_CYCLIC_DECL
// plane at + j * ranks + s of `out` := plane s * count + j of `in` (s < ranks, j < count): `ranks` chunks of `count` planes each, chunk s
// holding every ranks-th plane from s on, woven into consecutive planes (jfa_seed.hip)
int launch_win_interleave(vp_ctx* ctx, uint32_t n, const IdWin& in, const IdWin& out, uint32_t ranks, uint32_t count);
int launch_extract_count(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, int mode, uint64_t* h_count);
int launch_extract_write(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, int mode, const float* d_sdf,
                         uint64_t* d_records, float* d_values, size_t capacity);

}  // namespace vp
