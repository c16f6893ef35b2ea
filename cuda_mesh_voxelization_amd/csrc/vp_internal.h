// vp_internal.h -- shared declarations of libvphip.so (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/vphip.h"

namespace vp {

// JFA state "no seed yet" of the two 32-bit id formats (jfa.hip, IdU<9> / IdU<10>): y and z fields all ones, x = 2^BITS
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
    uint32_t compact;  // JFA id volumes of this call are in the compact layout (jfa.hip: IdC); set by vp_jfa_run only
};

inline Frame make_frame(const vp_frame* f)
{
    Frame r;
    r.n = f->n; r.w = f->n / 32; r.z0 = f->z0; r.z1 = f->z1;
    r.vs = f->voxel_size; r.ox = f->origin[0]; r.oy = f->origin[1]; r.oz = f->origin[2];
    r.compact = 0;
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

// ---- stage launchers (each enqueues on ctx->stream) ----
int launch_voxelize(vp_ctx* ctx, const Frame& f, uint32_t* d_words, const float* d_xyz, size_t nverts,
                    const uint32_t* d_tri, size_t ntris, int algo, int accumulate);
int launch_csg(vp_ctx* ctx, uint32_t* d_a, const uint32_t* d_b, size_t nwords, int op);
int launch_stream_copy(vp_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);   // 16 B per lane: the measured HBM copy rate
size_t jfa_id_bytes(const Frame& f);                              // 4 (n <= 1024) or 8
int launch_jfa_init(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const uint32_t* below,
                    const uint32_t* above, void* d_ids, uint32_t* d_border_words);
int launch_jfa_pass(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                    const void* d_plus, void* d_out, int algo);
bool jfa_pass_can_fuse_final(const Frame& f, uint32_t k, int algo);
int launch_jfa_pass_ex(vp_ctx* ctx, const Frame& f, uint32_t k, const void* d_in, const void* d_minus,
                       const void* d_plus, void* d_out, int algo, const uint32_t* d_words, float fill, float* d_sdf);
bool jfa_can_start_from_mask(const Frame& f, int algo);
int launch_jfa_first_pass(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, void* d_out);
bool jfa_can_fuse_first_two(const Frame& f, int algo);          // passes n/2 and n/4 in one launch from the border mask
bool jfa_compact_applies(const Frame& f, int algo);             // whole-grid vp_jfa at n > 1024: 5-byte id state (jfa.hip: IdC)
int launch_jfa_first_two(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, void* d_out);
int launch_jfa_final(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, const void* d_ids,
                     float fill, float* d_sdf);
int launch_extract_count(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, int mode, uint64_t* h_count);
int launch_extract_write(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, int mode, const float* d_sdf,
                         uint64_t* d_records, float* d_values, size_t capacity);

}  // namespace vp
