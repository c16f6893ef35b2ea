// capi.hip -- the extern "C" surface of libvphip.so (declared in include/vphip.h):
// context / stream / workspace management, argument validation, per-kernel hipEvent timing and
// the host-in/host-out conveniences that reproduce the reference's Compute() calling convention.
#include "vp_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <initializer_list>

namespace vp {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    // same information as the reference's gpuAssert line (vplib/src/debug_utils.h:43-50)
    snprintf(g_err, sizeof(g_err), "[%s:%d] HIP Assert: %s (%s)", file, line, hipGetErrorString(e), what);
    return (int)e;
}

void release(Buffer& b)
{
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr; b.bytes = 0;
}

int reserve(vp_ctx* ctx, Buffer& b, size_t bytes, bool headroom)
{
    if (bytes <= b.bytes) return 0;
    VP_HIP(hipStreamSynchronize(ctx->stream));
    if (b.ptr) VP_HIP(hipFree(b.ptr));
    b.ptr = nullptr; b.bytes = 0;
    const size_t want = headroom ? bytes + bytes / 4 : bytes;     // head-room: fewer regrows (not for the id volumes of the slab driver: GiBs)
    VP_HIP(hipMalloc(&b.ptr, want));
    b.bytes = want;
    return 0;
}

ProfScope::ProfScope(vp_ctx* c, int k) : ctx(c), kernel(k)
{
    if (!ctx->prof_on || !((ctx->prof_mask >> k) & 1u)) return;
    auto take = [&]() -> hipEvent_t {
        if (!ctx->prof_pool.empty()) { hipEvent_t e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    };
    a = take(); b = take();
    if (a) (void)hipEventRecord(a, ctx->stream);
}

ProfScope::~ProfScope()
{
    if (!ctx->prof_on || !a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof_pending.push_back({kernel, a, b});
}

static int prof_fold(vp_ctx* ctx)
{
    if (ctx->prof_pending.empty()) return 0;
    VP_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& s : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            ctx->prof_ms[s.kernel] += ms;
            ctx->prof_n[s.kernel] += 1;
        }
        ctx->prof_pool.push_back(s.a);
        ctx->prof_pool.push_back(s.b);
    }
    ctx->prof_pending.clear();
    return 0;
}

static int check_frame(const vp_frame* f, const char* who, bool whole)
{
    if (!f) return set_error(VP_ERR_INVALID, "%s: null frame", who);
    if (f->n < 32 || f->n > 2048 || (f->n % 32) != 0)
        return set_error(VP_ERR_UNSUPPORTED, "%s: n=%u unsupported (need 32 <= n <= 2048, n %% 32 == 0)", who, f->n);
    if (!(f->z0 < f->z1) || f->z1 > f->n || (f->z0 % 8) != 0 || (f->z1 % 8) != 0)
        return set_error(VP_ERR_INVALID, "%s: bad slab [%u,%u) for n=%u (multiples of 8 required)", who, f->z0, f->z1, f->n);
    if (whole && !(f->z0 == 0 && f->z1 == f->n))
        return set_error(VP_ERR_INVALID, "%s: whole-grid frame required", who);
    if (!(f->voxel_size > 0.0f) || !std::isfinite(f->voxel_size))
        return set_error(VP_ERR_INVALID, "%s: voxel_size must be positive and finite", who);
    return 0;
}

// Work must land on the context's device even if the calling thread's current HIP device changed in between
// (the null stream in particular belongs to the CURRENT device).
static int bind_device(vp_ctx* ctx)
{
    VP_HIP(hipSetDevice(ctx->device));
    return 0;
}

// vp_extract must follow a vp_extract_count of the same grid CONTENTS; the record of that count is dropped as soon as the
// buffer it was taken from is written through this ABI or handed out again as a workspace slot.
// The same holds for the record of vp_jfa_start: its border mask (or init ids) was computed from the grid CONTENTS, so a write to
// that grid -- or to the workspace itself -- between start and run invalidates it ("same grid" in the header means same contents).
// Byte RANGES are compared: a write into the interior of the grid -- vp_voxelize / vp_csg on a slab frame, vp_memcpy_d2d
// into a sub-range -- invalidates the records like a write to its first byte does.
static bool overlaps(const void* a, size_t an, const void* b, size_t bn)
{
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a && b && a0 < b0 + (bn ? bn : 1) && b0 < a0 + (an ? an : 1);
}
static void grid_written(vp_ctx* ctx, const void* d_ptr, size_t bytes)
{
    if (!d_ptr) return;
    if (overlaps(d_ptr, bytes, ctx->ext_words, (size_t)ctx->ext_n * ctx->ext_n * ctx->ext_n / 8)) ctx->ext_words = nullptr;
    const vp_ctx::JfaStarted& st = ctx->jfa_started;
    if (st.valid && (overlaps(d_ptr, bytes, st.words, (size_t)st.n * st.n * st.n / 8) || overlaps(d_ptr, bytes, st.work, st.work_bytes)))
        ctx->jfa_started.valid = false;
}

static const char* kNames[VP_K_COUNT] = {
    "vox_setup", "vox_scan", "vox_scatter", "vox_tile", "vox_naive", "vox_fill",
    "csg_words", "jfa_init", "jfa_pass", "jfa_final", "surface", "jfa_first", "jfa_sparse", "jfa_dense", "jfa_last", "extract", "vox_zero", "jfa_redeal"
};

}  // namespace vp

using namespace vp;

extern "C" {

int vp_abi_version(void) { return VP_ABI_VERSION; }

const char* vp_last_error(void) { return g_err; }

int vp_device_count(int* out)
{
    if (!out) return set_error(VP_ERR_INVALID, "vp_device_count: null out");
    *out = 0;
    VP_HIP(hipGetDeviceCount(out));
    return 0;
}

int vp_ctx_create(int device, vp_ctx** out)
{
    if (!out) return set_error(VP_ERR_INVALID, "vp_ctx_create: null out");
    *out = nullptr;
    int count = 0;
    VP_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return set_error(VP_ERR_INVALID, "vp_ctx_create: device %d not present (%d visible)", device, count);
    VP_HIP(hipSetDevice(device));
    vp_ctx* c = new (std::nothrow) vp_ctx();
    if (!c) return set_error(VP_ERR_NOMEM, "vp_ctx_create: out of host memory");
    c->device = device;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->cus = v; }
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    c->stream = c->own_stream;
    *out = c;
    return 0;
}

int vp_ctx_destroy(vp_ctx* ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    Buffer* bufs[] = { &ctx->rec, &ctx->tile_cnt, &ctx->tile_off, &ctx->tile_cur, &ctx->pairs, &ctx->scratch, &ctx->none_row, &ctx->jfa_work,
                       &ctx->ext_cnt, &ctx->ext_off };
    for (Buffer* b : bufs) release(*b);
    for (int i = 0; i < VP_WORKSPACE_SLOTS; ++i) release(ctx->slots[i]);
    for (auto& s : ctx->prof_pending) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
    if (ctx->vox_total_event) (void)hipEventDestroy(ctx->vox_total_event);
    if (ctx->vox_total_host) (void)hipHostFree(ctx->vox_total_host);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return 0;
}

int vp_ctx_set_stream(vp_ctx* ctx, void* hip_stream, int external)
{
    if (!ctx) return set_error(VP_ERR_INVALID, "vp_ctx_set_stream: null ctx");
    VP_TRY(bind_device(ctx));
    VP_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = external ? (hipStream_t)hip_stream : ctx->own_stream;
    return 0;
}

int vp_ctx_sync(vp_ctx* ctx)
{
    if (!ctx) return set_error(VP_ERR_INVALID, "vp_ctx_sync: null ctx");
    VP_TRY(bind_device(ctx));
    VP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int vp_malloc(vp_ctx* ctx, size_t bytes, void** d_out)
{
    if (!ctx || !d_out) return set_error(VP_ERR_INVALID, "vp_malloc: null argument");
    VP_HIP(hipSetDevice(ctx->device));
    VP_HIP(hipMalloc(d_out, bytes ? bytes : 1));
    return 0;
}

int vp_free(vp_ctx* ctx, void* d_ptr)
{
    if (!ctx) return set_error(VP_ERR_INVALID, "vp_free: null ctx");
    VP_TRY(bind_device(ctx));
    grid_written(ctx, d_ptr, 1);
    if (d_ptr) { VP_HIP(hipStreamSynchronize(ctx->stream)); VP_HIP(hipFree(d_ptr)); }
    return 0;
}

int vp_memcpy_d2d(vp_ctx* ctx, void* d_dst, const void* d_src, size_t bytes)
{
    if (!ctx || ((!d_dst || !d_src) && bytes)) return set_error(VP_ERR_INVALID, "vp_memcpy_d2d: null argument");
    VP_TRY(bind_device(ctx));
    grid_written(ctx, d_dst, bytes);
    if (bytes) VP_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int vp_ctx_workspace(vp_ctx* ctx, int slot, size_t bytes, void** d_out)
{
    if (!ctx || !d_out || slot < 0 || slot >= VP_WORKSPACE_SLOTS) return set_error(VP_ERR_INVALID, "vp_ctx_workspace: bad argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(reserve(ctx, ctx->slots[slot], bytes ? bytes : 1));
    *d_out = ctx->slots[slot].ptr;
    grid_written(ctx, *d_out, ctx->slots[slot].bytes);             // whoever asks for the slot is about to fill it
    return 0;
}

int vp_ctx_release(vp_ctx* ctx)
{
    if (!ctx) return set_error(VP_ERR_INVALID, "vp_ctx_release: null ctx");
    VP_TRY(bind_device(ctx));
    VP_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < VP_WORKSPACE_SLOTS; ++i) release(ctx->slots[i]);
    release(ctx->jfa_work);
    ctx->jfa_started.valid = false;
    ctx->ext_words = nullptr;
    return 0;
}

int vp_memset(vp_ctx* ctx, void* d_ptr, int byte_value, size_t bytes)
{
    if (!ctx || (!d_ptr && bytes)) return set_error(VP_ERR_INVALID, "vp_memset: null argument");
    VP_TRY(bind_device(ctx));
    grid_written(ctx, d_ptr, bytes);
    if (bytes) VP_HIP(hipMemsetAsync(d_ptr, byte_value, bytes, ctx->stream));
    return 0;
}

int vp_upload(vp_ctx* ctx, void* d_dst, const void* h_src, size_t bytes)
{
    if (!ctx || ((!d_dst || !h_src) && bytes)) return set_error(VP_ERR_INVALID, "vp_upload: null argument");
    VP_TRY(bind_device(ctx));
    grid_written(ctx, d_dst, bytes);
    if (bytes) {
        VP_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        VP_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int vp_download(vp_ctx* ctx, void* h_dst, const void* d_src, size_t bytes)
{
    if (!ctx || ((!h_dst || !d_src) && bytes)) return set_error(VP_ERR_INVALID, "vp_download: null argument");
    VP_TRY(bind_device(ctx));
    if (bytes) {
        VP_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        VP_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

size_t vp_grid_words(const vp_frame* f) { return f ? (size_t)f->n * f->n * (f->z1 - f->z0) / 32 : 0; }
size_t vp_grid_voxels(const vp_frame* f) { return f ? (size_t)f->n * f->n * (f->z1 - f->z0) : 0; }

// Every grid / id / sdf buffer handed in must be 16-byte aligned (include/vphip.h): the kernels move them as 16-byte vectors.
static int check_aligned(const char* who, std::initializer_list<const void*> ptrs)
{
    for (const void* p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return set_error(VP_ERR_INVALID, "%s: device buffer %p is not 16-byte aligned", who, p);
    return 0;
}

int vp_voxelize(vp_ctx* ctx, const vp_frame* f, uint32_t* d_words, const float* d_xyz, size_t nverts,
                const uint32_t* d_tri, size_t ntris, int algo, int accumulate)
{
    if (!ctx || !d_words) return set_error(VP_ERR_INVALID, "vp_voxelize: null argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_voxelize", false));
    VP_TRY(check_aligned("vp_voxelize", {d_words}));
    if (ntris && (!d_xyz || !d_tri || !nverts)) return set_error(VP_ERR_INVALID, "vp_voxelize: null mesh arrays");
    if (algo != VP_ALGO_NAIVE && algo != VP_ALGO_TILED) return set_error(VP_ERR_INVALID, "vp_voxelize: algo %d", algo);
    if (ntris > 0xFFFFFFFFull / 3) return set_error(VP_ERR_UNSUPPORTED, "vp_voxelize: too many triangles");
    grid_written(ctx, d_words, vp_grid_words(f) * 4);
    return launch_voxelize(ctx, make_frame(f), d_words, d_xyz, nverts, d_tri, ntris, algo, accumulate ? 1 : 0);
}

int vp_csg(vp_ctx* ctx, uint32_t* d_a, const uint32_t* d_b, size_t nwords, int op)
{
    if (!ctx || ((!d_a || !d_b) && nwords)) return set_error(VP_ERR_INVALID, "vp_csg: null argument");
    VP_TRY(bind_device(ctx));
    if (op < VP_OP_VOID || op > VP_OP_DIFFERENCE) return set_error(VP_ERR_INVALID, "vp_csg: unknown op %d", op);
    VP_TRY(check_aligned("vp_csg", {d_a, d_b}));
    grid_written(ctx, d_a, nwords * 4);
    return launch_csg(ctx, d_a, d_b, nwords, op);
}

int vp_stream_copy(vp_ctx* ctx, void* d_dst, const void* d_src, size_t bytes)
{
    if (!ctx || !d_dst || !d_src || bytes == 0 || (bytes % 16) != 0 || ((uintptr_t)d_dst % 16) != 0 || ((uintptr_t)d_src % 16) != 0)
        return set_error(VP_ERR_INVALID, "vp_stream_copy: 16-byte aligned buffers of a multiple of 16 bytes required");
    VP_TRY(bind_device(ctx));
    grid_written(ctx, d_dst, bytes);
    return launch_stream_copy(ctx, d_dst, d_src, bytes);
}

size_t vp_jfa_workspace_bytes(const vp_frame* f)
{
    // two id volumes + border mask
    return f ? 2 * vp_grid_voxels(f) * vp_jfa_id_bytes(f) + vp_grid_words(f) * 4 : 0;
}

size_t vp_jfa_id_bytes(const vp_frame* f) { return (f && f->n > 1024) ? 8 : 4; }

static int check_fill(float fill, const char* who)
{
    if (!std::isinf(fill)) return set_error(VP_ERR_INVALID, "%s: fill_unset must be +-infinity", who);
    return 0;
}

int vp_jfa_init(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, const uint32_t* d_plane_below,
                const uint32_t* d_plane_above, void* d_ids)
{
    if (!ctx || !d_words || !d_ids) return set_error(VP_ERR_INVALID, "vp_jfa_init: null argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_jfa_init", false));
    VP_TRY(check_aligned("vp_jfa_init", {d_words, d_plane_below, d_plane_above, d_ids}));
    return launch_jfa_init(ctx, make_frame(f), d_words, d_plane_below, d_plane_above, d_ids, nullptr);
}

int vp_surface(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, const uint32_t* d_plane_below,
               const uint32_t* d_plane_above, uint32_t* d_border_words)
{
    if (!ctx || !d_words || !d_border_words) return set_error(VP_ERR_INVALID, "vp_surface: null argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_surface", false));
    VP_TRY(check_aligned("vp_surface", {d_words, d_plane_below, d_plane_above, d_border_words}));
    return launch_jfa_init(ctx, make_frame(f), d_words, d_plane_below, d_plane_above, nullptr, d_border_words);
}

int vp_jfa_pass(vp_ctx* ctx, const vp_frame* f, uint32_t k, const void* d_in, const void* d_minus,
                const void* d_plus, void* d_out, int algo)
{
    if (!ctx || !d_in || !d_out || d_in == d_out) return set_error(VP_ERR_INVALID, "vp_jfa_pass: bad buffers");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_jfa_pass", false));
    VP_TRY(check_aligned("vp_jfa_pass", {d_in, d_minus, d_plus, d_out}));
    if (k == 0 || k >= f->n) return set_error(VP_ERR_INVALID, "vp_jfa_pass: step %u out of range", k);
    if (algo != VP_ALGO_NAIVE && algo != VP_ALGO_TILED) return set_error(VP_ERR_INVALID, "vp_jfa_pass: algo %d", algo);
    // halos are mandatory wherever a neighbour plane exists outside the slab
    if (f->z0 > 0 && !d_minus) return set_error(VP_ERR_INVALID, "vp_jfa_pass: slab needs d_minus");
    if (f->z1 < f->n && !d_plus) return set_error(VP_ERR_INVALID, "vp_jfa_pass: slab needs d_plus");
    return launch_jfa_pass(ctx, make_frame(f), k, d_in, d_minus, d_plus, d_out, algo);
}

int vp_jfa_finalize(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, const void* d_ids,
                    float fill_unset, float* d_sdf)
{
    if (!ctx || !d_words || !d_ids || !d_sdf) return set_error(VP_ERR_INVALID, "vp_jfa_finalize: null argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_jfa_finalize", false));
    VP_TRY(check_fill(fill_unset, "vp_jfa_finalize"));
    return launch_jfa_final(ctx, make_frame(f), d_words, d_ids, fill_unset, d_sdf);
}

// The two halves of vp_jfa (the reference times them separately: "::Initialization" / "::Processing",
// jfa/tiled.cu:265-334).  start: border mask (tile kernels) or init ids; run: every pass + the id -> sdf conversion.
static int jfa_check(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, void*& d_work, size_t& work_bytes, int algo, const char* who)
{
    if (!ctx || !d_words) return set_error(VP_ERR_INVALID, "%s: null argument", who);
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, who, true));
    VP_TRY(check_aligned(who, {d_words, d_work}));
    if (algo != VP_ALGO_NAIVE && algo != VP_ALGO_TILED) return set_error(VP_ERR_INVALID, "%s: algo %d", who, algo);
    if (!d_work) {                                                 // context-owned workspace (grow-only, reused by later calls)
        const void* before = ctx->jfa_work.ptr;
        VP_TRY(reserve(ctx, ctx->jfa_work, vp_jfa_workspace_bytes(f)));
        if (ctx->jfa_work.ptr != before) ctx->jfa_started.valid = false;   // regrown: what a vp_jfa_start left there is gone
        d_work = ctx->jfa_work.ptr;
        work_bytes = ctx->jfa_work.bytes;
    } else if (work_bytes < vp_jfa_workspace_bytes(f)) {
        return set_error(VP_ERR_INVALID, "%s: workspace too small", who);
    }
    return 0;
}

// The tile-kernel sequence of a whole grid: border mask -> passes n/2 + n/4 in one launch -> tile passes on two windows of n planes
// inside the workspace -> last pass fused with the id -> sdf conversion.
static bool jfa_tile_sequence(const Frame& fr, int algo) { return jfa_can_fuse_first_two(fr, algo) && fr.n / 4 >= 1; }

int vp_jfa_start(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, void* d_work, size_t work_bytes, int algo)
{
    VP_TRY(jfa_check(ctx, f, d_words, d_work, work_bytes, algo, "vp_jfa_start"));
    const Frame fr = make_frame(f);
    const size_t volBytes = vp_grid_voxels(f) * vp_jfa_id_bytes(f);
    char* a = (char*)d_work;
    const bool mask = jfa_tile_sequence(fr, algo);
    ctx->jfa_started.valid = false;
    if (mask) VP_TRY(launch_jfa_init(ctx, fr, d_words, nullptr, nullptr, nullptr, (uint32_t*)(a + 2 * volBytes)));   // border mask only
    else      VP_TRY(launch_jfa_init(ctx, fr, d_words, nullptr, nullptr, a, nullptr));
    ctx->jfa_started = {true, mask, f->n, algo, d_work, work_bytes, d_words};
    return 0;
}

int vp_jfa_run(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, float fill_unset, float* d_sdf,
               void* d_work, size_t work_bytes, int algo)
{
    if (!d_sdf) return set_error(VP_ERR_INVALID, "vp_jfa_run: null argument");
    VP_TRY(check_aligned("vp_jfa_run", {d_sdf}));
    VP_TRY(jfa_check(ctx, f, d_words, d_work, work_bytes, algo, "vp_jfa_run"));
    VP_TRY(check_fill(fill_unset, "vp_jfa_run"));
    const Frame fr = make_frame(f);
    const size_t volBytes = vp_grid_voxels(f) * vp_jfa_id_bytes(f);
    char* a = (char*)d_work;
    char* b = a + volBytes;
    uint32_t k = f->n / 2;                                         // jfa/sequential.cpp:72
    // The workspace must hold what THIS sequence starts from: the record of the matching vp_jfa_start (same grid, frame size,
    // algo and workspace).  One start serves one run: the passes overwrite the volumes.
    const bool wantMask = jfa_tile_sequence(fr, algo);
    const vp_ctx::JfaStarted st = ctx->jfa_started;
    ctx->jfa_started.valid = false;
    if (!st.valid || st.n != f->n || st.algo != algo || st.work != d_work || st.words != d_words || st.mask != wantMask)
        return set_error(VP_ERR_INVALID, "vp_jfa_run: call vp_jfa_start with the same grid, frame, algo and workspace first");
    if (wantMask) {
        // two windows of n planes (5 bytes per voxel above n = 1024, inside the 8-byte volumes of the workspace: nobody outside this
        // function sees them)
        IdWin wa, wb;
        wa.base = a; wb.base = b; wa.planes = wb.planes = f->n; wa.at = wb.at = 0;
        VP_TRY(launch_win_first_two(ctx, fr, (const uint32_t*)(b + volBytes), wa));    // passes n/2 and n/4 straight from the border mask
        for (k /= 4; k >= 1; k /= 2) {
            if (k == 1) return launch_win_pass(ctx, fr, 1, wa, wb, 1, d_words, fill_unset, d_sdf);     // last pass writes the sdf itself
            VP_TRY(launch_win_pass(ctx, fr, k, wa, wb, k, nullptr, 0.0f, nullptr));
            std::swap(wa, wb);
        }
        return set_error(VP_ERR_INVALID, "vp_jfa_run: internal: no last pass");       // n >= 96: the loop always ends in its k = 1 branch
    }
    for (; k >= 1; k /= 2) {
        if (k == 1 && jfa_pass_can_fuse_final(fr, k, algo))        // last pass writes the sdf itself
            return launch_jfa_pass_ex(ctx, fr, k, a, nullptr, nullptr, b, algo, d_words, fill_unset, d_sdf);
        VP_TRY(launch_jfa_pass(ctx, fr, k, a, nullptr, nullptr, b, algo));
        char* t = a; a = b; b = t;
    }
    return launch_jfa_final(ctx, fr, d_words, a, fill_unset, d_sdf);
}

int vp_jfa(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, float fill_unset, float* d_sdf,
           void* d_work, size_t work_bytes, int algo)
{
    if (!d_sdf) return set_error(VP_ERR_INVALID, "vp_jfa: null argument");
    VP_TRY(check_aligned("vp_jfa", {d_sdf}));
    VP_TRY(check_fill(fill_unset, "vp_jfa"));
    VP_TRY(vp_jfa_start(ctx, f, d_words, d_work, work_bytes, algo));
    return vp_jfa_run(ctx, f, d_words, fill_unset, d_sdf, d_work, work_bytes, algo);
}

size_t vp_jfa_state_bytes(const vp_frame* f, int algo)
{
    if (!f || check_frame(f, "vp_jfa_state_bytes", false) != 0) return 0;
    return (algo == VP_ALGO_TILED && win_compact(f->n)) ? 5 : vp_jfa_id_bytes(f);
}

int vp_jfa_can_start_from_mask(const vp_frame* f, int algo)
{
    return (f && check_frame(f, "vp_jfa_can_start_from_mask", false) == 0 && jfa_can_start_from_mask(make_frame(f), algo)) ? 1 : 0;
}

int vp_jfa_can_fuse_first_two(const vp_frame* f, int algo)
{
    return (f && check_frame(f, "vp_jfa_can_fuse_first_two", false) == 0 && jfa_can_fuse_first_two(make_frame(f), algo)) ? 1 : 0;
}

int vp_jfa_last_pass(vp_ctx* ctx, const vp_frame* f, const void* d_in, const void* d_minus, const void* d_plus,
                     void* d_scratch, const uint32_t* d_words, float fill_unset, float* d_sdf, int algo)
{
    if (!ctx || !d_in || !d_scratch || !d_words || !d_sdf || d_in == d_scratch)
        return set_error(VP_ERR_INVALID, "vp_jfa_last_pass: bad buffers");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_jfa_last_pass", false));
    VP_TRY(check_aligned("vp_jfa_last_pass", {d_in, d_minus, d_plus, d_words, d_sdf}));
    VP_TRY(check_fill(fill_unset, "vp_jfa_last_pass"));
    if (algo != VP_ALGO_NAIVE && algo != VP_ALGO_TILED) return set_error(VP_ERR_INVALID, "vp_jfa_last_pass: algo %d", algo);
    if (f->z0 > 0 && !d_minus) return set_error(VP_ERR_INVALID, "vp_jfa_last_pass: slab needs d_minus");
    if (f->z1 < f->n && !d_plus) return set_error(VP_ERR_INVALID, "vp_jfa_last_pass: slab needs d_plus");
    const Frame fr = make_frame(f);
    if (jfa_pass_can_fuse_final(fr, 1, algo))
        return launch_jfa_pass_ex(ctx, fr, 1, d_in, d_minus, d_plus, d_scratch, algo, d_words, fill_unset, d_sdf);
    VP_TRY(launch_jfa_pass(ctx, fr, 1, d_in, d_minus, d_plus, d_scratch, algo));
    return launch_jfa_final(ctx, fr, d_words, d_scratch, fill_unset, d_sdf);
}

// ---- id windows: the slab form of the tile kernels ------------------------------------------------
size_t vp_jfa_window_bytes(const vp_frame* f, uint32_t planes)
{
    if (!f || check_frame(f, "vp_jfa_window_bytes", false) != 0) return 0;
    return win_bytes(f->n, planes);
}

int vp_jfa_window_span(const vp_frame* f, uint32_t planes, uint32_t p0, uint32_t p1, size_t offset[2], size_t bytes[2])
{
    if (!offset || !bytes) return set_error(VP_ERR_INVALID, "vp_jfa_window_span: null argument");
    VP_TRY(check_frame(f, "vp_jfa_window_span", false));
    if (p0 > p1 || p1 > planes) return set_error(VP_ERR_INVALID, "vp_jfa_window_span: planes [%u, %u) of %u", p0, p1, planes);
    const size_t wp = win_plane_bytes(f->n), bp = (size_t)f->n * f->n;
    offset[0] = (size_t)p0 * wp; bytes[0] = (size_t)(p1 - p0) * wp;
    offset[1] = (size_t)planes * wp + (size_t)p0 * bp; bytes[1] = win_compact(f->n) ? (size_t)(p1 - p0) * bp : 0;
    return 0;
}

// A window and the frame it is used with: the planes of f must lie inside it; `below` / `above` = planes the call reads beyond them.
static int window_check(vp_ctx* ctx, const vp_frame* f, const vp_window* w, const char* who, uint32_t below, uint32_t above, IdWin& out)
{
    if (!ctx || !w || !w->d_ids) return set_error(VP_ERR_INVALID, "%s: null argument", who);
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, who, false));
    VP_TRY(check_aligned(who, {w->d_ids}));
    if (f->n < kTileMinN) return set_error(VP_ERR_UNSUPPORTED, "%s: windows are the layout of the tile kernels (n >= %u)", who, kTileMinN);
    const uint64_t nz = f->z1 - f->z0;
    if (w->bytes < win_bytes(f->n, w->planes))
        return set_error(VP_ERR_INVALID, "%s: a window of %u planes needs %zu bytes, %zu given", who, w->planes, win_bytes(f->n, w->planes), w->bytes);
    if (w->planes == 0 || (uint64_t)w->at + nz + above > w->planes || w->at < below)
        return set_error(VP_ERR_INVALID, "%s: planes [%u, %u) at index %u (+%u below, +%u above) do not fit a window of %u planes", who, f->z0, f->z1, w->at, below, above, w->planes);
    out.base = (char*)w->d_ids; out.planes = w->planes; out.at = w->at;
    return 0;
}

// Every vp_jfa_window_* call that writes ids: the output window is a write through the ABI like any other (it may lie inside the workspace
// a vp_jfa_start left its state in, or be a grid a vp_extract_count counted): the records are dropped.
static void window_written(vp_ctx* ctx, const vp_window* w) { grid_written(ctx, w->d_ids, w->bytes); }

int vp_jfa_window_clear(vp_ctx* ctx, const vp_frame* f, const vp_window* w)
{
    if (!ctx || !w || !w->d_ids || !w->planes) return set_error(VP_ERR_INVALID, "vp_jfa_window_clear: null argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_jfa_window_clear", false));
    VP_TRY(check_aligned("vp_jfa_window_clear", {w->d_ids}));
    if (w->bytes < win_bytes(f->n, w->planes)) return set_error(VP_ERR_INVALID, "vp_jfa_window_clear: a window of %u planes needs %zu bytes, %zu given", w->planes, win_bytes(f->n, w->planes), w->bytes);
    IdWin iw; iw.base = (char*)w->d_ids; iw.planes = w->planes; iw.at = 0;
    window_written(ctx, w);
    return launch_win_clear(ctx, f->n, iw);
}

int vp_jfa_window_init(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, const uint32_t* d_plane_below,
                       const uint32_t* d_plane_above, const vp_window* out)
{
    if (!d_words) return set_error(VP_ERR_INVALID, "vp_jfa_window_init: null argument");
    IdWin w;
    VP_TRY(window_check(ctx, f, out, "vp_jfa_window_init", 0, 0, w));
    VP_TRY(check_aligned("vp_jfa_window_init", {d_words, d_plane_below, d_plane_above}));
    window_written(ctx, out);
    return launch_win_init(ctx, make_frame(f), d_words, d_plane_below, d_plane_above, w);
}

int vp_jfa_window_first_pass(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_border_grid, const vp_window* out)
{
    if (!d_border_grid) return set_error(VP_ERR_INVALID, "vp_jfa_window_first_pass: null argument");
    IdWin w;
    VP_TRY(window_check(ctx, f, out, "vp_jfa_window_first_pass", 0, 0, w));
    VP_TRY(check_aligned("vp_jfa_window_first_pass", {d_border_grid}));
    const Frame fr = make_frame(f);
    if (!jfa_can_start_from_mask(fr, VP_ALGO_TILED)) return set_error(VP_ERR_UNSUPPORTED, "vp_jfa_window_first_pass: needs n %% 128 == 0");
    window_written(ctx, out);
    return launch_win_first_pass(ctx, fr, d_border_grid, w);
}

int vp_jfa_window_first_two(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_border_grid, const vp_window* out)
{
    if (!d_border_grid) return set_error(VP_ERR_INVALID, "vp_jfa_window_first_two: null argument");
    IdWin w;
    VP_TRY(window_check(ctx, f, out, "vp_jfa_window_first_two", 0, 0, w));
    VP_TRY(check_aligned("vp_jfa_window_first_two", {d_border_grid}));
    const Frame fr = make_frame(f);
    if (!jfa_can_fuse_first_two(fr, VP_ALGO_TILED) || w.planes != f->n || w.at != 0)
        return set_error(VP_ERR_INVALID, "vp_jfa_window_first_two: whole-grid frame and a window of n planes (at = 0) required");
    window_written(ctx, out);
    return launch_win_first_two(ctx, fr, d_border_grid, w);
}

// ---- cyclic plane distribution (the first phase of the transposed multi-GPU pipeline) ----
int vp_jfa_cyclic_passes(const vp_frame* f, uint32_t ranks)
{
    if (!f || check_frame(f, "vp_jfa_cyclic_passes", true) != 0) return 0;
    return (int)jfa_cyclic_passes(f->n, ranks);
}

// a whole-grid frame and a window of the n / ranks planes of one rank, at = 0
static int cyclic_check(vp_ctx* ctx, const vp_frame* f, const vp_window* w, uint32_t ranks, uint32_t rank, const char* who, IdWin& out)
{
    if (!ctx || !w || !w->d_ids) return set_error(VP_ERR_INVALID, "%s: null argument", who);
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, who, true));
    VP_TRY(check_aligned(who, {w->d_ids}));
    if (jfa_cyclic_passes(f->n, ranks) == 0 || rank >= ranks)
        return set_error(VP_ERR_UNSUPPORTED, "%s: n = %u cannot be dealt cyclically to %u ranks (rank %u): a power of two, n / ranks a multiple of 8, n/4 a multiple of it", who, f->n, ranks, rank);
    if (w->planes != f->n / ranks || w->at != 0 || w->bytes < win_bytes(f->n, w->planes))
        return set_error(VP_ERR_INVALID, "%s: a window of n / ranks = %u planes (at = 0, %zu bytes) required", who, f->n / ranks, win_bytes(f->n, f->n / ranks));
    out.base = (char*)w->d_ids; out.planes = w->planes; out.at = 0;
    return 0;
}

int vp_jfa_window_first_two_cyclic(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_border_grid, const vp_window* out, uint32_t ranks, uint32_t rank)
{
    if (!d_border_grid) return set_error(VP_ERR_INVALID, "vp_jfa_window_first_two_cyclic: null argument");
    IdWin w;
    VP_TRY(cyclic_check(ctx, f, out, ranks, rank, "vp_jfa_window_first_two_cyclic", w));
    VP_TRY(check_aligned("vp_jfa_window_first_two_cyclic", {d_border_grid}));
    window_written(ctx, out);
    return launch_win_first_two(ctx, make_frame(f), d_border_grid, w, ranks, rank);
}

int vp_jfa_window_pass_cyclic(vp_ctx* ctx, const vp_frame* f, uint32_t k, const vp_window* in, const vp_window* out, uint32_t ranks, uint32_t rank)
{
    const char* who = "vp_jfa_window_pass_cyclic";
    IdWin wi, wo;
    VP_TRY(cyclic_check(ctx, f, in, ranks, rank, who, wi));
    VP_TRY(cyclic_check(ctx, f, out, ranks, rank, who, wo));
    if (overlaps(in->d_ids, win_bytes(f->n, in->planes), out->d_ids, win_bytes(f->n, out->planes))) return set_error(VP_ERR_INVALID, "%s: the two windows overlap", who);
    // one of the leading steps of the halving sequence, beyond the two of the fused start
    bool found = false;
    uint32_t c = jfa_cyclic_passes(f->n, ranks), kk = f->n / 2;
    for (uint32_t i = 0; i < c; ++i, kk /= 2) if (kk == k && i >= 2) found = true;
    if (!found) return set_error(VP_ERR_INVALID, "%s: step %u is not one of the steps n/8 .. of n = %u that are multiples of %u ranks", who, k, f->n, ranks);
    window_written(ctx, out);
    return launch_win_pass_cyclic(ctx, make_frame(f), k, wi, wo, ranks, rank);
}

int vp_jfa_window_interleave(vp_ctx* ctx, const vp_frame* f, const vp_window* in, const vp_window* out, uint32_t ranks, uint32_t count)
{
    const char* who = "vp_jfa_window_interleave";
    if (!ctx || !in || !out || !in->d_ids || !out->d_ids || !ranks || !count) return set_error(VP_ERR_INVALID, "%s: null argument", who);
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, who, false));
    VP_TRY(check_aligned(who, {in->d_ids, out->d_ids}));
    if (f->n < kTileMinN) return set_error(VP_ERR_UNSUPPORTED, "%s: windows are the layout of the tile kernels (n >= %u)", who, kTileMinN);
    const uint64_t planes = (uint64_t)ranks * count;
    if (in->planes != planes || in->bytes < win_bytes(f->n, in->planes))
        return set_error(VP_ERR_INVALID, "%s: the source must be a window of exactly ranks x count = %llu planes", who, (unsigned long long)planes);
    if ((uint64_t)out->at + planes > out->planes || out->bytes < win_bytes(f->n, out->planes))
        return set_error(VP_ERR_INVALID, "%s: %llu planes at index %u do not fit a window of %u planes", who, (unsigned long long)planes, out->at, out->planes);
    if (overlaps(in->d_ids, win_bytes(f->n, in->planes), out->d_ids, win_bytes(f->n, out->planes))) return set_error(VP_ERR_INVALID, "%s: the two windows overlap", who);
    IdWin wi, wo;
    wi.base = (char*)in->d_ids; wi.planes = in->planes; wi.at = 0;
    wo.base = (char*)out->d_ids; wo.planes = out->planes; wo.at = out->at;
    window_written(ctx, out);
    return launch_win_interleave(ctx, f->n, wi, wo, ranks, count);
}

// planes a pass with step k reads below / above the planes of f, in window planes
static void pass_reach(const vp_frame* f, uint32_t k, uint32_t stride, uint32_t& below, uint32_t& above)
{
    if (stride == k) {                                              // consecutive planes: as far as the grid goes
        below = std::min(k, f->z0);
        above = std::min(k, f->n - f->z1);
    } else {                                                       // whole slabs `stride` planes away
        below = f->z1 > k ? stride : 0;                            // some plane z of f has z - k >= 0
        above = f->z0 + k < f->n ? stride : 0;
    }
}

static int window_pass(vp_ctx* ctx, const vp_frame* f, uint32_t k, const vp_window* in, const vp_window* out, uint32_t stride,
                       const uint32_t* d_words, float fill, float* d_sdf, const char* who)
{
    if (!in || !out || !in->d_ids || !out->d_ids || in->d_ids == out->d_ids) return set_error(VP_ERR_INVALID, "%s: bad windows", who);
    if (in->planes != out->planes || in->at != out->at) return set_error(VP_ERR_INVALID, "%s: the two windows must have the same planes / at", who);
    // two windows carved from one allocation that partly overlap would have the pass read planes it is overwriting
    if (f && f->n && overlaps(in->d_ids, win_bytes(f->n, in->planes), out->d_ids, win_bytes(f->n, out->planes)))
        return set_error(VP_ERR_INVALID, "%s: the two windows overlap", who);
    if (!f || k == 0 || k >= f->n) return set_error(VP_ERR_INVALID, "%s: step %u out of range", who, k);
    if (stride == 0 || (stride != k && (k < f->z1 - f->z0 || stride < f->z1 - f->z0)))
        return set_error(VP_ERR_INVALID, "%s: stride %u: either the step itself or, for a step of at least the slab height, the distance of the slabs in the window", who, stride);
    uint32_t below = 0, above = 0;
    pass_reach(f, k, stride, below, above);
    IdWin wi, wo;
    VP_TRY(window_check(ctx, f, in, who, below, above, wi));
    VP_TRY(window_check(ctx, f, out, who, 0, 0, wo));
    window_written(ctx, out);
    return launch_win_pass(ctx, make_frame(f), k, wi, wo, stride, d_words, fill, d_sdf);
}

int vp_jfa_window_pass(vp_ctx* ctx, const vp_frame* f, uint32_t k, const vp_window* in, const vp_window* out, uint32_t stride)
{
    return window_pass(ctx, f, k, in, out, stride, nullptr, 0.0f, nullptr, "vp_jfa_window_pass");
}

int vp_jfa_window_last_pass(vp_ctx* ctx, const vp_frame* f, const vp_window* in, const vp_window* scratch, uint32_t stride,
                            const uint32_t* d_words_region, float fill_unset, float* d_sdf_region)
{
    if (!d_words_region || !d_sdf_region) return set_error(VP_ERR_INVALID, "vp_jfa_window_last_pass: null argument");
    VP_TRY(check_aligned("vp_jfa_window_last_pass", {d_words_region, d_sdf_region}));
    VP_TRY(check_fill(fill_unset, "vp_jfa_window_last_pass"));
    return window_pass(ctx, f, 1, in, scratch, stride, d_words_region, fill_unset, d_sdf_region, "vp_jfa_window_last_pass");
}

// ---- export front end -------------------------------------------------------------------------
int vp_extract_count(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, int mode, uint64_t* h_count)
{
    if (!ctx || !d_words) return set_error(VP_ERR_INVALID, "vp_extract_count: null argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_extract_count", true));
    if (mode < VP_EXTRACT_SET || mode > VP_EXTRACT_FACES) return set_error(VP_ERR_INVALID, "vp_extract_count: mode %d", mode);
    return launch_extract_count(ctx, make_frame(f), d_words, mode, h_count);
}

int vp_extract(vp_ctx* ctx, const vp_frame* f, const uint32_t* d_words, int mode, const float* d_sdf,
               uint64_t* d_records, float* d_values, size_t capacity)
{
    if (!ctx || !d_words || (!d_records && capacity)) return set_error(VP_ERR_INVALID, "vp_extract: null argument");
    if ((d_values != nullptr) != (d_sdf != nullptr)) return set_error(VP_ERR_INVALID, "vp_extract: d_sdf and d_values go together");
    VP_TRY(bind_device(ctx));
    VP_TRY(check_frame(f, "vp_extract", true));
    if (mode < VP_EXTRACT_SET || mode > VP_EXTRACT_FACES) return set_error(VP_ERR_INVALID, "vp_extract: mode %d", mode);
    return launch_extract_write(ctx, make_frame(f), d_words, mode, d_sdf, d_records, d_values, capacity);
}

// ---- host-in / host-out ----------------------------------------------------------------------
// Device buffers come from the context's workspace slots (grow-only): steady-state calls allocate nothing, where the
// reference's Compute() does ~15 cudaMalloc/cudaFree per call.
enum { SLOT_GRID_A = 0, SLOT_GRID_B = 1, SLOT_XYZ = 2, SLOT_TRI = 3, SLOT_SDF = 4 };

int vp_voxelize_host(vp_ctx* ctx, const vp_frame* f, uint32_t* h_words, const float* h_xyz, size_t nverts,
                     const uint32_t* h_tri, size_t ntris, int algo)
{
    if (!ctx || !h_words) return set_error(VP_ERR_INVALID, "vp_voxelize_host: null argument");
    VP_TRY(check_frame(f, "vp_voxelize_host", true));
    void *dw = nullptr, *dx = nullptr, *dt = nullptr;
    const size_t wb = vp_grid_words(f) * 4;
    VP_TRY(vp_ctx_workspace(ctx, SLOT_GRID_A, wb, &dw));
    VP_TRY(vp_ctx_workspace(ctx, SLOT_XYZ, nverts * 12, &dx));
    VP_TRY(vp_ctx_workspace(ctx, SLOT_TRI, ntris * 12, &dt));
    VP_TRY(vp_upload(ctx, dx, h_xyz, nverts * 12));
    VP_TRY(vp_upload(ctx, dt, h_tri, ntris * 12));
    VP_TRY(vp_voxelize(ctx, f, (uint32_t*)dw, (const float*)dx, nverts, (const uint32_t*)dt, ntris, algo, 0));
    return vp_download(ctx, h_words, dw, wb);
}

int vp_csg_host(vp_ctx* ctx, uint32_t* h_a, const uint32_t* h_b, size_t nwords, int op)
{
    if (!ctx || ((!h_a || !h_b) && nwords)) return set_error(VP_ERR_INVALID, "vp_csg_host: null argument");
    void *da = nullptr, *db = nullptr;
    VP_TRY(vp_ctx_workspace(ctx, SLOT_GRID_A, nwords * 4, &da));
    VP_TRY(vp_ctx_workspace(ctx, SLOT_GRID_B, nwords * 4, &db));
    VP_TRY(vp_upload(ctx, da, h_a, nwords * 4));
    VP_TRY(vp_upload(ctx, db, h_b, nwords * 4));
    VP_TRY(vp_csg(ctx, (uint32_t*)da, (const uint32_t*)db, nwords, op));
    return vp_download(ctx, h_a, da, nwords * 4);
}

int vp_jfa_host(vp_ctx* ctx, const vp_frame* f, const uint32_t* h_words, float fill_unset, float* h_sdf, int algo)
{
    if (!ctx || !h_words || !h_sdf) return set_error(VP_ERR_INVALID, "vp_jfa_host: null argument");
    VP_TRY(check_frame(f, "vp_jfa_host", true));
    void *dw = nullptr, *ds = nullptr;
    const size_t wb = vp_grid_words(f) * 4, sb = vp_grid_voxels(f) * 4;
    VP_TRY(vp_ctx_workspace(ctx, SLOT_GRID_A, wb, &dw));
    VP_TRY(vp_ctx_workspace(ctx, SLOT_SDF, sb, &ds));
    VP_TRY(vp_upload(ctx, dw, h_words, wb));
    VP_TRY(vp_jfa(ctx, f, (const uint32_t*)dw, fill_unset, (float*)ds, nullptr, 0, algo));
    return vp_download(ctx, h_sdf, ds, sb);
}

// ---- profiling -------------------------------------------------------------------------------
int vp_prof_enable(vp_ctx* ctx, int on)
{
    if (!ctx) return set_error(VP_ERR_INVALID, "vp_prof_enable: null ctx");
    VP_TRY(bind_device(ctx));
    if (!on) VP_TRY(prof_fold(ctx));
    ctx->prof_on = on != 0;
    return 0;
}

int vp_prof_select(vp_ctx* ctx, uint64_t kernel_mask)
{
    if (!ctx) return set_error(VP_ERR_INVALID, "vp_prof_select: null ctx");
    ctx->prof_mask = kernel_mask;
    return 0;
}

int vp_prof_reset(vp_ctx* ctx)
{
    if (!ctx) return set_error(VP_ERR_INVALID, "vp_prof_reset: null ctx");
    VP_TRY(bind_device(ctx));
    VP_TRY(prof_fold(ctx));
    for (int i = 0; i < VP_K_COUNT; ++i) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; }
    return 0;
}

int vp_prof_get(vp_ctx* ctx, int kernel, double* total_ms, uint64_t* launches)
{
    if (!ctx || kernel < 0 || kernel >= VP_K_COUNT) return set_error(VP_ERR_INVALID, "vp_prof_get: bad argument");
    VP_TRY(bind_device(ctx));
    VP_TRY(prof_fold(ctx));
    if (total_ms) *total_ms = ctx->prof_ms[kernel];
    if (launches) *launches = ctx->prof_n[kernel];
    return 0;
}

const char* vp_prof_name(int kernel) { return (kernel >= 0 && kernel < VP_K_COUNT) ? kNames[kernel] : "?"; }

}  // extern "C"
