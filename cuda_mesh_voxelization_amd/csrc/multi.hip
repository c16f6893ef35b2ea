// multi.hip -- vp_multi_*: one process driving several GPUs of a node, the grid cut into Z-slabs (include/vphip.h).
//
// The reference is single-GPU: apps/cli/main.cpp:22-23 pins device 0 and every Compute() uses it.  This file is what replaces
// that line when more than one device is given: one vp_ctx (stream + workspace) per device, rank r owns the planes
// [r n/G, (r+1) n/G) of every buffer, and the stages are the C-ABI stages of a slab frame (vp_frame.z0 / z1):
//
//   voxelize   every (y, z) column is independent (vox/sequential.cpp:40-57): each device rasterises the mesh into its slab.
//   CSG        word-wise: each device combines its slabs.
//   JFA        on id windows (include/vphip.h, vp_jfa_window_*), three ways of feeding a pass its planes z -+ k:
//                  VP_MULTI_HALO   seeding needs one bitmask plane from each Z-neighbour; the pass with step k needs the id planes
//                              [z0-k, min(z0, z1-k)) and [max(z1, z0+k), z1+k) from whoever owns them.  They are moved with
//                              hipMemcpyPeerAsync on the RECEIVER's stream behind an event of the sender's stream, and every
//                              stream waits for the copies that read its buffers before it overwrites them two passes later:
//                              the whole JFA is enqueued without a host synchronisation (jfa_halo below).
//                  VP_MULTI_HYBRID ghost planes for the passes with k > nz/2, halo copies from the two adjacent ranks for the
//                              others; the windows hold only the planes a rank touches (jfa_hybrid below).
//                  VP_MULTI_TRANSPOSE the planes are dealt CYCLICALLY (plane z on rank z mod G) for every pass whose step is a multiple of G --
//                              such a pass finds the planes z - k, z, z + k of each plane a rank owns on that rank: no exchange, no ghost
//                              planes, 1/G of the pass per device -- then ONE re-deal (peer copies of contiguous plane ranges) into slabs
//                              widened by the reach of the remaining steps, which run like the last ghost regions (jfa_transpose below).
//                  VP_MULTI_GHOST  no exchange between passes: the bitmask slabs are all-gathered once (n^3/8 bytes), every
//                              device runs pass i on its slab widened by the reach of the later passes and the regions shrink
//                              to the bare slab at k = 1.  Costs two windows of the whole grid per device.
//
// Every stage is a pure function of the previous buffers, so the concatenated slabs are bit-identical to the single-device
// result for any number of devices -- including several contexts on ONE device, which is how the tests run it.
// Host code only; compiled with the kernels because it uses the context internals (stream, device).
#include "vp_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

using namespace vp;

namespace {

struct Rank {
    vp_ctx* ctx = nullptr;
    int device = 0;
    uint32_t z0 = 0, z1 = 0;
    // resident buffers (grow-only)
    Buffer mesh_xyz, mesh_tri;
    Buffer words;                              // bitmask of the slab (halo) / of the whole grid (ghost, after the all-gather)
    Buffer other;                              // second operand of a CSG
    Buffer below, above;                       // bitmask planes z0-1 / z1
    Buffer ids[2];                             // the two id windows of the last vp_multi_jfa (ensure_windows)
    uint32_t win_n = 0, win_planes = 0;        // ... and their geometry
    Buffer cyc[2], staging;                    // VP_MULTI_TRANSPOSE: the two windows of the cyclic phase, the chunks the re-deal delivers
    uint32_t cyc_n = 0, cyc_planes = 0, stg_n = 0, stg_planes = 0;
    Buffer border;                             // ghost / hybrid: border mask of the whole grid
    Buffer whole_sdf;                          // n < 96 (not sharded): sdf of the whole grid
    Buffer sdf;                                // slab
    hipEvent_t ready = nullptr;                // "my buffers hold what the peers may read"
    hipEvent_t copied = nullptr;               // "the copies INTO my buffers of this step are done"
};

}  // namespace

struct vp_multi {
    std::vector<Rank> ranks;
    vp_frame frame{};                          // global frame of the resident grid
    bool have_grid = false, have_sdf = false;
    size_t nverts = 0, ntris = 0;
    uint64_t bytes_moved = 0;                  // device-to-device bytes of the last vp_multi_jfa
    int last_mode = -1;
    std::vector<uint32_t> window_lo, window_hi; // id planes each rank held during the last vp_multi_jfa
};

namespace {

int bind(const Rank& r)
{
    VP_HIP(hipSetDevice(r.device));
    return 0;
}

int grow(Rank& r, Buffer& b, size_t bytes)
{
    VP_TRY(bind(r));
    return reserve(r.ctx, b, bytes ? bytes : 1);
}

// The id windows of a rank (two buffers of `planes` id planes in the library's layout, include/vphip.h: vp_jfa_window_*).  They are
// cleared -- every id := "none" -- whenever their geometry changes: the ghost regions below are rounded OUTWARDS to the 8-plane tile, and
// the excess planes of a pass read planes the pass before it never produced (see ghost_regions).  What they read is then "none" or ids an
// earlier job left in the SAME layout -- never memory nobody wrote, never bytes of another layout.
// Test builds (-DVP_TEST_HOOKS, libvphip_hooks.so): VP_MULTI_POISON=<byte> refills the word planes with that byte before every job to
// show that the results do not depend on what those planes hold.
int ensure_window_set(Rank& r, Buffer* bufs, int nbufs, uint32_t& gn, uint32_t& gplanes, const vp_frame& G, uint32_t planes)
{
    VP_TRY(bind(r));
    const size_t bytes = vp_jfa_window_bytes(&G, planes);
    bool fresh = gn != G.n || gplanes != planes;
    for (int i = 0; i < nbufs; ++i) {
        Buffer& b = bufs[i];
        const void* before = b.ptr;
        VP_TRY(reserve(r.ctx, b, bytes, /*headroom=*/false));       // exact: a window must not cost more than it saves
        fresh = fresh || b.ptr != before;
    }
    gn = G.n; gplanes = planes;
#ifdef VP_TEST_HOOKS
    const char* poison = getenv("VP_MULTI_POISON");
    fresh = fresh || poison != nullptr;
#endif
    if (fresh)
        for (int i = 0; i < nbufs; ++i) { const vp_window w{bufs[i].ptr, bufs[i].bytes, planes, 0}; VP_TRY(vp_jfa_window_clear(r.ctx, &G, &w)); }
#ifdef VP_TEST_HOOKS
    if (poison)
        for (int i = 0; i < nbufs; ++i) VP_HIP(hipMemsetAsync(bufs[i].ptr, (int)strtol(poison, nullptr, 0) & 0xFF, (size_t)planes * G.n * G.n * 4, r.ctx->stream));
#endif
    return 0;
}

int ensure_windows(Rank& r, const vp_frame& G, uint32_t planes) { return ensure_window_set(r, r.ids, 2, r.win_n, r.win_planes, G, planes); }

vp_frame slab_frame(const vp_frame& g, uint32_t z0, uint32_t z1)
{
    vp_frame f = g;
    f.z0 = z0; f.z1 = z1;
    return f;
}

int check_split(const vp_multi* m, const vp_frame* f, const char* who)
{
    if (!m || !f) return set_error(VP_ERR_INVALID, "%s: null argument", who);
    const uint32_t g = (uint32_t)m->ranks.size();
    if (f->n < 32 || f->n > 2048 || f->n % 32 != 0) return set_error(VP_ERR_UNSUPPORTED, "%s: n=%u unsupported (need 32 <= n <= 2048, n %% 32 == 0)", who, f->n);
    if (f->z0 != 0 || f->z1 != f->n) return set_error(VP_ERR_INVALID, "%s: whole-grid frame required", who);
    if (f->n % g != 0 || (f->n / g) % 8 != 0)
        return set_error(VP_ERR_INVALID, "%s: n=%u cannot be cut into %u Z-slabs of a multiple of 8 planes", who, f->n, g);
    return 0;
}

void assign_slabs(vp_multi* m, const vp_frame* f)
{
    const uint32_t g = (uint32_t)m->ranks.size(), nz = f->n / g;
    for (uint32_t r = 0; r < g; ++r) { m->ranks[r].z0 = r * nz; m->ranks[r].z1 = (r + 1) * nz; }
    m->frame = *f;
}

// dst's stream waits for src's `ready` event, then copies.  (hipMemcpyPeerAsync between two contexts of one device is an
// ordinary device-to-device copy.)
int peer_copy(vp_multi* m, Rank& dst, void* d_dst, const Rank& src, const void* d_src, size_t bytes)
{
    if (!bytes) return 0;
    VP_TRY(bind(dst));
    VP_HIP(hipStreamWaitEvent(dst.ctx->stream, src.ready, 0));
    if (dst.device == src.device) VP_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, dst.ctx->stream));
    else VP_HIP(hipMemcpyPeerAsync(d_dst, dst.device, d_src, src.device, bytes, dst.ctx->stream));
    m->bytes_moved += bytes;
    return 0;
}

int mark_ready(Rank& r)
{
    VP_TRY(bind(r));
    VP_HIP(hipEventRecord(r.ready, r.ctx->stream));
    return 0;
}

// After the copies of a step: every receiver records `copied`; every stream waits for all of them before it goes on (its next
// kernels may overwrite what a peer has just read -- the ping-pong volume written by the pass after this one).
int fence_copies(vp_multi* m)
{
    for (Rank& r : m->ranks) { VP_TRY(bind(r)); VP_HIP(hipEventRecord(r.copied, r.ctx->stream)); }
    for (Rank& r : m->ranks) {
        VP_TRY(bind(r));
        for (Rank& o : m->ranks) if (&o != &r) VP_HIP(hipStreamWaitEvent(r.ctx->stream, o.copied, 0));
    }
    return 0;
}

// Every rank keeps its slab at its GLOBAL position inside a buffer of the whole grid (n^3/8 bytes, small next to the id
// volumes): the halo mode uses the slab alone, the ghost mode fills in the rest.
char* slab_words(const vp_multi* m, const Rank& r)
{
    return (char*)r.words.ptr + (size_t)r.z0 * ((size_t)m->frame.n * m->frame.n / 8);
}

struct HaloMove { uint32_t src, dst; bool minusSide; uint32_t g0, g1; };

// For step k: dst needs the global planes [g0, g1) owned by src for its minus / plus buffer (one entry per owner).
std::vector<HaloMove> halo_plan(uint32_t n, uint32_t world, uint32_t k)
{
    const uint32_t nz = n / world;
    std::vector<HaloMove> plan;
    for (uint32_t dst = 0; dst < world; ++dst) {
        const int64_t z0 = (int64_t)dst * nz, z1 = z0 + nz;
        const int64_t lo[2] = {std::max<int64_t>(z0 - k, 0), std::max<int64_t>(z1, z0 + k)};
        const int64_t hi[2] = {std::min<int64_t>(z0, z1 - (int64_t)k), std::min<int64_t>(z1 + k, n)};
        for (int side = 0; side < 2; ++side)
            for (int64_t g = lo[side]; g < hi[side];) {
                const uint32_t src = (uint32_t)(g / nz);
                const int64_t e = std::min<int64_t>(hi[side], (int64_t)(src + 1) * nz);
                plan.push_back({src, dst, side == 0, (uint32_t)g, (uint32_t)e});
                g = e;
            }
    }
    return plan;
}

struct Region { uint32_t k, b0, b1; };

// Planes each pass must produce on a rank so that no exchange is needed: the slab widened by the sum of the later steps
// (its REACH g_i), rounded outwards to the 8-plane tile and clipped to the grid.
// Invariant (tests/test_multi_gpu.py::test_multi_ghost_ignores_unproduced_planes): a plane of region i is NEEDED iff it lies within
// g_i of the slab; needed planes of pass i read only planes within g_i + k_i = g_(i-1) of the slab, all of which pass i - 1 produced.
// The planes the rounding adds are computed too (whole tiles) but from planes pass i - 1 may not have produced -- their values are
// never read by a needed plane of any later pass, so the slab is exact whatever those planes held (grow_ids gives them defined bytes).
// Rounding the regions so that each contains the next one widened by its step instead would cost up to 16 more planes per side
// and pass (k = 8: 24 instead of 8), i.e. time, for values nobody reads.
std::vector<Region> ghost_regions(uint32_t n, uint32_t z0, uint32_t z1)
{
    std::vector<uint32_t> ks;
    for (uint32_t k = n / 2; k >= 1; k /= 2) ks.push_back(k);
    std::vector<Region> out;
    for (size_t i = 0; i < ks.size(); ++i) {
        uint32_t g = 0;
        for (size_t j = i + 1; j < ks.size(); ++j) g += ks[j];
        const uint32_t b0 = z0 > g ? (z0 - g) / 8 * 8 : 0;
        const uint32_t b1 = std::min(n, (z1 + g + 7) / 8 * 8);
        out.push_back({ks[i], b0, b1});
    }
    return out;
}

// `count` id planes from index `sp` of a window of src (splanes planes) to index `dp` of a window of dst (dplanes): one or two byte ranges
// (above n = 1024 the word planes and the byte planes: 5 bytes per voxel on the wire)
int copy_planes_buf(vp_multi* m, uint32_t dplanes, Rank& dst, const Buffer& dbuf, uint32_t dp, uint32_t splanes, const Rank& src, const Buffer& sbuf, uint32_t sp, uint32_t count)
{
    size_t so[2], sb[2], dof[2], db[2];
    VP_TRY(vp_jfa_window_span(&m->frame, splanes, sp, sp + count, so, sb));
    VP_TRY(vp_jfa_window_span(&m->frame, dplanes, dp, dp + count, dof, db));
    for (int i = 0; i < 2; ++i)
        VP_TRY(peer_copy(m, dst, (char*)dbuf.ptr + dof[i], src, (const char*)sbuf.ptr + so[i], sb[i]));
    return 0;
}
int copy_planes(vp_multi* m, uint32_t dplanes, Rank& dst, int dwhich, uint32_t dp, uint32_t splanes, const Rank& src, int swhich, uint32_t sp, uint32_t count)
{
    return copy_planes_buf(m, dplanes, dst, dst.ids[dwhich], dp, splanes, src, src.ids[swhich], sp, count);
}

// all-gather of the bitmask slabs: every device ends up with the whole grid (its own slab stays where it is: plane z0)
int gather_words(vp_multi* m)
{
    const uint32_t world = (uint32_t)m->ranks.size();
    const size_t slabWords = (size_t)(m->frame.n / world) * ((size_t)m->frame.n * m->frame.n / 8);
    for (Rank& r : m->ranks) VP_TRY(mark_ready(r));
    for (uint32_t r = 0; r < world; ++r)
        for (uint32_t o = 0; o < world; ++o)
            if (o != r) VP_TRY(peer_copy(m, m->ranks[r], (char*)m->ranks[r].words.ptr + (size_t)o * slabWords, m->ranks[o],
                                         (const char*)m->ranks[o].words.ptr + (size_t)o * slabWords, slabWords));
    if (world > 1) VP_TRY(fence_copies(m));
    return 0;
}

// Grids below the tile kernels' range (n < 96: at most 64^3 voxels, 0.06 ms per JFA) are not sharded: every device computes the whole
// grid and keeps its slab.
int jfa_small(vp_multi* m, float fill, int algo)
{
    const vp_frame& G = m->frame;
    const size_t plane = (size_t)G.n * G.n;
    VP_TRY(gather_words(m));
    for (Rank& r : m->ranks) {
        VP_TRY(grow(r, r.whole_sdf, vp_grid_voxels(&G) * 4));
        VP_TRY(grow(r, r.sdf, (size_t)(r.z1 - r.z0) * plane * 4));
        VP_TRY(vp_jfa(r.ctx, &G, (const uint32_t*)r.words.ptr, fill, (float*)r.whole_sdf.ptr, nullptr, 0, algo));
        VP_TRY(vp_memcpy_d2d(r.ctx, r.sdf.ptr, (const char*)r.whole_sdf.ptr + (size_t)r.z0 * plane * 4, (size_t)(r.z1 - r.z0) * plane * 4));
    }
    return 0;
}

// VP_MULTI_HALO.  A rank's windows hold [slab of z - k | own slab | slab of z + k] = 3 nz planes, the own slab in the middle.  The halos
// of the narrow passes (k <= nz/2) land in the k planes right below / above the slab (stride = k: consecutive planes); for the wide
// passes (k >= nz: whole slabs of distant ranks) the received slabs land nz planes below / above the own planes and the tile kernel runs
// with stride = nz -- no separate whole-slab buffers, no second kernel for them.
int jfa_halo(vp_multi* m, float fill)
{
    const vp_frame& G = m->frame;
    const uint32_t n = G.n, world = (uint32_t)m->ranks.size(), nz = n / world;
    const size_t planeWords = (size_t)n * n / 8;                   // bytes
    const uint32_t planes = world > 1 ? 3 * nz : nz, at = world > 1 ? nz : 0;
    for (Rank& r : m->ranks) {
        VP_TRY(ensure_windows(r, G, planes));
        VP_TRY(grow(r, r.sdf, (size_t)nz * n * n * 4));
        if (world > 1) { VP_TRY(grow(r, r.below, planeWords)); VP_TRY(grow(r, r.above, planeWords)); }
    }
    // bitmask planes for the 26-neighbourhood of the seeding (jfa/sequential.cpp:24-64)
    for (Rank& r : m->ranks) VP_TRY(mark_ready(r));
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        if (r > 0) VP_TRY(peer_copy(m, me, me.below.ptr, m->ranks[r - 1], slab_words(m, m->ranks[r - 1]) + (size_t)(nz - 1) * planeWords, planeWords));
        if (r + 1 < world) VP_TRY(peer_copy(m, me, me.above.ptr, m->ranks[r + 1], slab_words(m, m->ranks[r + 1]), planeWords));
    }
    if (world > 1) VP_TRY(fence_copies(m));
    int cur = 0;
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        const vp_frame f = slab_frame(G, me.z0, me.z1);
        const vp_window w{me.ids[cur].ptr, me.ids[cur].bytes, planes, at};
        VP_TRY(vp_jfa_window_init(me.ctx, &f, (const uint32_t*)slab_words(m, me), r > 0 ? (const uint32_t*)me.below.ptr : nullptr,
                                  r + 1 < world ? (const uint32_t*)me.above.ptr : nullptr, &w));
    }
    for (uint32_t k = n / 2; k >= 1; k /= 2) {
        const uint32_t stride = (world > 1 && k >= nz) ? nz : k;
        if (world > 1) {
            for (Rank& r : m->ranks) VP_TRY(mark_ready(r));
            for (const HaloMove& h : halo_plan(n, world, k)) {
                if (h.src == h.dst) continue;
                Rank& dst = m->ranks[h.dst];
                const Rank& src = m->ranks[h.src];
                // global plane g feeds the plane z = g + k (minus side) / g - k (plus side) of the receiver, which sits `stride` planes away from it
                const int64_t dp = h.minusSide ? (int64_t)at + ((int64_t)h.g0 + k - dst.z0) - stride : (int64_t)at + ((int64_t)h.g0 - k - dst.z0) + stride;
                VP_TRY(copy_planes(m, planes, dst, cur, (uint32_t)dp, planes, src, cur, at + (h.g0 - src.z0), h.g1 - h.g0));
            }
            VP_TRY(fence_copies(m));
        }
        for (uint32_t r = 0; r < world; ++r) {
            Rank& me = m->ranks[r];
            const vp_frame f = slab_frame(G, me.z0, me.z1);
            const vp_window in{me.ids[cur].ptr, me.ids[cur].bytes, planes, at}, out{me.ids[cur ^ 1].ptr, me.ids[cur ^ 1].bytes, planes, at};
            if (k == 1) VP_TRY(vp_jfa_window_last_pass(me.ctx, &f, &in, &out, stride, (const uint32_t*)slab_words(m, me), fill, (float*)me.sdf.ptr));
            else        VP_TRY(vp_jfa_window_pass(me.ctx, &f, k, &in, &out, stride));
        }
        cur ^= 1;
    }
    return 0;
}

// VP_MULTI_GHOST: windows of the whole grid (n planes, a plane sits at its global index); the first two passes are the one whole-grid
// launch of the single-device path on every rank (its second pass would cover 35 % of the grid or more on any rank of 2 .. 8 slabs: the
// break-even of the measured kernel times, profiles/r03/slab_scaling_*.txt), every later pass runs on the rank's region.
int jfa_ghost(vp_multi* m, float fill)
{
    const vp_frame& G = m->frame;
    const uint32_t n = G.n, world = (uint32_t)m->ranks.size(), nz = n / world;
    const size_t planeWords = (size_t)n * n / 8;
    m->last_mode = VP_MULTI_GHOST;                                  // (also what VP_MULTI_TRANSPOSE runs where no step is a multiple of the device count)
    for (Rank& r : m->ranks) {
        VP_TRY(grow(r, r.border, (size_t)n * planeWords));
        VP_TRY(ensure_windows(r, G, n));
        VP_TRY(grow(r, r.sdf, (size_t)nz * n * n * 4));
    }
    // words buffers hold the whole grid with the rank's own slab at its global position (vp_multi_set_grid / vp_multi_voxelize place it
    // there); here the slabs of the peers are filled in
    VP_TRY(gather_words(m));
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        const std::vector<Region> regs = ghost_regions(n, me.z0, me.z1);
        const uint32_t* words = (const uint32_t*)me.words.ptr;
        int cur = 0;
        VP_TRY(vp_surface(me.ctx, &G, words, nullptr, nullptr, (uint32_t*)me.border.ptr));
        { const vp_window w{me.ids[cur].ptr, me.ids[cur].bytes, n, 0}; VP_TRY(vp_jfa_window_first_two(me.ctx, &G, (const uint32_t*)me.border.ptr, &w)); }
        for (size_t i = 2; i < regs.size(); ++i) {
            const Region& g = regs[i];
            const vp_frame f = slab_frame(G, g.b0, g.b1);
            const vp_window in{me.ids[cur].ptr, me.ids[cur].bytes, n, g.b0}, out{me.ids[cur ^ 1].ptr, me.ids[cur ^ 1].bytes, n, g.b0};
            if (i + 1 == regs.size()) VP_TRY(vp_jfa_window_last_pass(me.ctx, &f, &in, &out, 1, words + (size_t)g.b0 * (planeWords / 4), fill, (float*)me.sdf.ptr));
            else                      VP_TRY(vp_jfa_window_pass(me.ctx, &f, g.k, &in, &out, g.k));
            cur ^= 1;
        }
    }
    return 0;
}

// ---- VP_MULTI_HYBRID (round 4): ghost planes where planes are cheap to recompute and dear to move, halos where it is the other way
// round -- the one-process form of slab.py's HybridSlabPipeline.  Passes with k > nz/2 ("wide": whole slabs of distant ranks would have
// to travel) run on the slab widened by the reach of the LATER WIDE passes only; passes with k <= nz/2 ("narrow") run on the bare slab
// behind k halo planes copied from the two adjacent ranks.  Id buffers hold the planes [lo, hi) a rank touches -- its WINDOW -- instead of
// the whole volume (vp_multi_window reports it): G = 8 at n = 1024: 896 of 1024 planes; G = 8 at n = 2048: 1792 of 2048.
struct HybridPlan { std::vector<Region> wide; std::vector<uint32_t> narrow; uint32_t lo, hi; };

HybridPlan hybrid_plan(uint32_t n, uint32_t world, uint32_t z0, uint32_t z1, bool maskStart)
{
    HybridPlan p;
    const uint32_t nz = z1 - z0, H = world > 1 ? nz / 2 : 0;
    std::vector<uint32_t> wideK;
    for (uint32_t k = n / 2; k >= 1; k /= 2) { if (k > H) wideK.push_back(k); else p.narrow.push_back(k); }
    for (size_t i = 0; i < wideK.size(); ++i) {
        uint32_t g = 0;
        for (size_t j = i + 1; j < wideK.size(); ++j) g += wideK[j];
        p.wide.push_back({wideK[i], z0 > g ? (z0 - g) / 8 * 8 : 0, std::min(n, (z1 + g + 7) / 8 * 8)});
    }
    // window: the slab with room for the narrow halos, every wide region, and what a wide pass that READS ids reads beyond its region
    p.lo = z0 > H ? z0 - H : 0; p.hi = std::min(n, z1 + H);
    for (size_t i = 0; i < p.wide.size(); ++i) {
        const Region& r = p.wide[i];
        p.lo = std::min(p.lo, r.b0); p.hi = std::max(p.hi, r.b1);
        if (i > 0 || !maskStart) { p.lo = std::min(p.lo, r.b0 > r.k ? r.b0 - r.k : 0); p.hi = std::max(p.hi, std::min(n, r.b1 + r.k)); }
    }
    return p;
}

int jfa_hybrid(vp_multi* m, float fill)
{
    const vp_frame& G = m->frame;
    const uint32_t n = G.n, world = (uint32_t)m->ranks.size(), nz = n / world;
    const size_t planeWords = (size_t)n * n / 8;
    const bool maskStart = vp_jfa_can_start_from_mask(&G, VP_ALGO_TILED) != 0;
    std::vector<HybridPlan> plans;
    m->window_lo.assign(world, 0); m->window_hi.assign(world, 0);
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        plans.push_back(hybrid_plan(n, world, me.z0, me.z1, maskStart));
        const HybridPlan& p = plans.back();
        m->window_lo[r] = p.lo; m->window_hi[r] = p.hi;
        VP_TRY(grow(me, me.border, (size_t)n * planeWords));
        VP_TRY(ensure_windows(me, G, p.hi - p.lo));
        VP_TRY(grow(me, me.sdf, (size_t)nz * n * n * 4));
    }
    // the wide passes need the bitmask of the whole grid on every device (as in the ghost mode)
    VP_TRY(gather_words(m));
    const size_t nwide = plans[0].wide.size(), nnarrow = plans[0].narrow.size();     // the same on every rank (they depend on nz only)
    std::vector<int> cur(world, 0);
    // window `which` of rank r, positioned for a frame that starts at global plane g
    auto win = [&](uint32_t r, int which, uint32_t g) { return vp_window{m->ranks[r].ids[which].ptr, m->ranks[r].ids[which].bytes, plans[r].hi - plans[r].lo, g - plans[r].lo}; };
    // ---- wide passes: ghost planes inside the window, no exchange
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        const HybridPlan& p = plans[r];
        const uint32_t* words = (const uint32_t*)me.words.ptr;
        size_t start = 0;
        if (maskStart && nwide > 0) {
            VP_TRY(vp_surface(me.ctx, &G, words, nullptr, nullptr, (uint32_t*)me.border.ptr));
            const vp_frame f = slab_frame(G, p.wide[0].b0, p.wide[0].b1);
            const vp_window w = win(r, 1, p.wide[0].b0);
            VP_TRY(vp_jfa_window_first_pass(me.ctx, &f, (const uint32_t*)me.border.ptr, &w));
            cur[r] = 1; start = 1;
        } else {
            const vp_frame f = slab_frame(G, p.lo, p.hi);
            const vp_window w = win(r, 0, p.lo);
            VP_TRY(vp_jfa_window_init(me.ctx, &f, words + (size_t)p.lo * (planeWords / 4), p.lo > 0 ? words + (size_t)(p.lo - 1) * (planeWords / 4) : nullptr,
                                      p.hi < n ? words + (size_t)p.hi * (planeWords / 4) : nullptr, &w));
        }
        for (size_t i = start; i < nwide; ++i) {
            const Region& g = p.wide[i];
            const vp_frame f = slab_frame(G, g.b0, g.b1);
            const vp_window in = win(r, cur[r], g.b0), out = win(r, cur[r] ^ 1, g.b0);
            if (i + 1 == nwide && nnarrow == 0)                     // one rank: the last pass is a wide one
                VP_TRY(vp_jfa_window_last_pass(me.ctx, &f, &in, &out, 1, words + (size_t)g.b0 * (planeWords / 4), fill, (float*)me.sdf.ptr));
            else
                VP_TRY(vp_jfa_window_pass(me.ctx, &f, g.k, &in, &out, g.k));
            cur[r] ^= 1;
        }
    }
    // ---- narrow passes: k halo planes from each adjacent rank, device to device, then the bare slab
    for (size_t j = 0; j < nnarrow; ++j) {
        const uint32_t k = plans[0].narrow[j];
        if (world > 1) {
            for (Rank& r : m->ranks) VP_TRY(mark_ready(r));
            for (uint32_t r = 0; r < world; ++r) {
                Rank& me = m->ranks[r];
                const uint32_t mine = plans[r].hi - plans[r].lo;
                if (r > 0)         VP_TRY(copy_planes(m, mine, me, cur[r], me.z0 - k - plans[r].lo, plans[r - 1].hi - plans[r - 1].lo, m->ranks[r - 1], cur[r - 1], me.z0 - k - plans[r - 1].lo, k));
                if (r + 1 < world) VP_TRY(copy_planes(m, mine, me, cur[r], me.z1 - plans[r].lo, plans[r + 1].hi - plans[r + 1].lo, m->ranks[r + 1], cur[r + 1], me.z1 - plans[r + 1].lo, k));
            }
            VP_TRY(fence_copies(m));
        }
        for (uint32_t r = 0; r < world; ++r) {
            Rank& me = m->ranks[r];
            const vp_frame f = slab_frame(G, me.z0, me.z1);
            const vp_window in = win(r, cur[r], me.z0), out = win(r, cur[r] ^ 1, me.z0);
            const uint32_t* slabW = (const uint32_t*)me.words.ptr + (size_t)me.z0 * (planeWords / 4);
            if (j + 1 == nnarrow) VP_TRY(vp_jfa_window_last_pass(me.ctx, &f, &in, &out, 1, slabW, fill, (float*)me.sdf.ptr));
            else                  VP_TRY(vp_jfa_window_pass(me.ctx, &f, k, &in, &out, k));
            cur[r] ^= 1;
        }
    }
    return 0;
}

// ---- VP_MULTI_TRANSPOSE (round 6).  The reference's pass with step k reads the planes z - k, z, z + k of a voxel's plane and nothing else
// (jfa/sequential.cpp:72, :92-94).  Phase A: device r keeps the planes z = r (mod G) in two windows of n / G planes; every pass whose step is
// a multiple of G (vp_jfa_cyclic_passes: all of them down to k = G on a power-of-two grid) runs there with no exchange and no ghost plane.
// The re-deal: the planes of device t's widened slab [t0, t1) that device s holds are the CONTIGUOUS planes [t0 / G, t1 / G) of s's window:
// one peer copy per pair of devices (two above n = 1024) into chunk s of t's staging window, straight from the window the last cyclic pass
// wrote -- no pack.  Phase B: vp_jfa_window_interleave weaves the chunks into consecutive planes and the remaining steps run on the slab
// widened by their reach, as the last regions of the ghost mode do.  The one-process form of slab.py's TransposeSlabPipeline.
struct TransposePlan { uint32_t c; std::vector<Region> regs; uint32_t t0, t1, lo, hi; };

TransposePlan transpose_plan(uint32_t n, uint32_t world, uint32_t z0, uint32_t z1, uint32_t c)
{
    TransposePlan p;
    p.c = c;
    const std::vector<Region> all = ghost_regions(n, z0, z1);
    uint32_t g = 0;
    for (size_t i = c; i < all.size(); ++i) { p.regs.push_back(all[i]); g += all[i].k; }
    p.t0 = z0 > g ? (z0 - g) / world * world : 0;
    p.t1 = std::min(n, (z1 + g + world - 1) / world * world);
    p.lo = p.t0; p.hi = p.t1;
    for (const Region& r : p.regs) { p.lo = std::min(p.lo, r.b0 > r.k ? r.b0 - r.k : 0); p.hi = std::max(p.hi, std::min(n, r.b1 + r.k)); }
    return p;
}

int jfa_ghost(vp_multi* m, float fill);

int jfa_transpose(vp_multi* m, float fill)
{
    const vp_frame& G = m->frame;
    const uint32_t n = G.n, world = (uint32_t)m->ranks.size(), nz = n / world;
    const uint32_t c = (uint32_t)vp_jfa_cyclic_passes(&G, world);
    if (c == 0) return jfa_ghost(m, fill);                          // nothing to deal cyclically (one device, a count that is not a power of two)
    const size_t planeWords = (size_t)n * n / 8;
    std::vector<TransposePlan> plans;
    m->window_lo.assign(world, 0); m->window_hi.assign(world, 0);
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        plans.push_back(transpose_plan(n, world, me.z0, me.z1, c));
        const TransposePlan& p = plans.back();
        m->window_lo[r] = p.lo; m->window_hi[r] = p.hi;
        VP_TRY(grow(me, me.border, (size_t)n * planeWords));
        VP_TRY(ensure_window_set(me, me.cyc, 2, me.cyc_n, me.cyc_planes, G, nz));
        VP_TRY(ensure_window_set(me, &me.staging, 1, me.stg_n, me.stg_planes, G, p.t1 - p.t0));
        VP_TRY(ensure_windows(me, G, p.hi - p.lo));
        VP_TRY(grow(me, me.sdf, (size_t)nz * n * n * 4));
    }
    // every device needs the bitmask of the whole grid: the border bits of its planes z = r (mod G) depend on the planes z -+ 1
    VP_TRY(gather_words(m));
    // ---- phase A: cyclic planes, no exchange
    std::vector<int> cur(world, 0);
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        VP_TRY(vp_surface(me.ctx, &G, (const uint32_t*)me.words.ptr, nullptr, nullptr, (uint32_t*)me.border.ptr));
        { const vp_window w{me.cyc[0].ptr, me.cyc[0].bytes, nz, 0}; VP_TRY(vp_jfa_window_first_two_cyclic(me.ctx, &G, (const uint32_t*)me.border.ptr, &w, world, r)); }
        uint32_t k = n / 8;
        for (uint32_t i = 2; i < c; ++i, k /= 2) {
            const vp_window in{me.cyc[cur[r]].ptr, me.cyc[cur[r]].bytes, nz, 0}, out{me.cyc[cur[r] ^ 1].ptr, me.cyc[cur[r] ^ 1].bytes, nz, 0};
            VP_TRY(vp_jfa_window_pass_cyclic(me.ctx, &G, k, &in, &out, world, r));
            cur[r] ^= 1;
        }
    }
    // ---- the re-deal: chunk s of t's staging window := the planes [t0 / G, t1 / G) of s's cyclic window
    for (Rank& r : m->ranks) VP_TRY(mark_ready(r));
    for (uint32_t t = 0; t < world; ++t) {
        const TransposePlan& p = plans[t];
        const uint32_t count = (p.t1 - p.t0) / world;
        for (uint32_t s_ = 0; s_ < world; ++s_) {
            const uint64_t before = m->bytes_moved;
            VP_TRY(copy_planes_buf(m, p.t1 - p.t0, m->ranks[t], m->ranks[t].staging, s_ * count, nz, m->ranks[s_], m->ranks[s_].cyc[cur[s_]], p.t0 / world, count));
            if (s_ == t) m->bytes_moved = before;                   // a device's own planes do not travel
        }
    }
    if (world > 1) VP_TRY(fence_copies(m));
    // ---- phase B: weave, then the remaining steps on the widened slab
    for (uint32_t r = 0; r < world; ++r) {
        Rank& me = m->ranks[r];
        const TransposePlan& p = plans[r];
        const uint32_t planes = p.hi - p.lo, count = (p.t1 - p.t0) / world;
        const uint32_t* words = (const uint32_t*)me.words.ptr;
        int w = 0;
        {
            const vp_window in{me.staging.ptr, me.staging.bytes, p.t1 - p.t0, 0}, out{me.ids[0].ptr, me.ids[0].bytes, planes, p.t0 - p.lo};
            VP_TRY(vp_jfa_window_interleave(me.ctx, &G, &in, &out, world, count));
        }
        for (size_t i = 0; i < p.regs.size(); ++i) {
            const Region& g = p.regs[i];
            const vp_frame f = slab_frame(G, g.b0, g.b1);
            const vp_window in{me.ids[w].ptr, me.ids[w].bytes, planes, g.b0 - p.lo}, out{me.ids[w ^ 1].ptr, me.ids[w ^ 1].bytes, planes, g.b0 - p.lo};
            if (i + 1 == p.regs.size()) VP_TRY(vp_jfa_window_last_pass(me.ctx, &f, &in, &out, 1, words + (size_t)g.b0 * (planeWords / 4), fill, (float*)me.sdf.ptr));
            else                        VP_TRY(vp_jfa_window_pass(me.ctx, &f, g.k, &in, &out, g.k));
            w ^= 1;
        }
    }
    return 0;
}

}  // namespace

extern "C" {

int vp_multi_create(const int* devices, int ndev, vp_multi** out)
{
    if (!out || !devices || ndev < 1 || ndev > 64) return set_error(VP_ERR_INVALID, "vp_multi_create: bad argument");
    *out = nullptr;
    vp_multi* m = new (std::nothrow) vp_multi();
    if (!m) return set_error(VP_ERR_NOMEM, "vp_multi_create: out of host memory");
    m->ranks.resize((size_t)ndev);
    for (int i = 0; i < ndev; ++i) {
        Rank& r = m->ranks[(size_t)i];
        r.device = devices[i];
        int rc = vp_ctx_create(devices[i], &r.ctx);
        if (rc == 0 && hipEventCreateWithFlags(&r.ready, hipEventDisableTiming) != hipSuccess) rc = set_error(VP_ERR_NOMEM, "vp_multi_create: event");
        if (rc == 0 && hipEventCreateWithFlags(&r.copied, hipEventDisableTiming) != hipSuccess) rc = set_error(VP_ERR_NOMEM, "vp_multi_create: event");
        if (rc != 0) { vp_multi_destroy(m); return rc; }
    }
    // peer access between distinct devices (a failure only means the copies are staged by the runtime)
    for (Rank& a : m->ranks)
        for (Rank& b : m->ranks)
            if (a.device != b.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can) {
                    (void)hipSetDevice(a.device);
                    (void)hipDeviceEnablePeerAccess(b.device, 0);   // hipErrorPeerAccessAlreadyEnabled is fine
                    (void)hipGetLastError();
                }
            }
    *out = m;
    return 0;
}

int vp_multi_destroy(vp_multi* m)
{
    if (!m) return 0;
    for (Rank& r : m->ranks) {
        if (!r.ctx) continue;
        (void)hipSetDevice(r.device);
        (void)hipStreamSynchronize(r.ctx->stream);
        Buffer* bufs[] = {&r.mesh_xyz, &r.mesh_tri, &r.words, &r.other, &r.below, &r.above, &r.ids[0], &r.ids[1], &r.cyc[0], &r.cyc[1], &r.staging, &r.border, &r.sdf, &r.whole_sdf};
        for (Buffer* b : bufs) release(*b);
        if (r.ready) (void)hipEventDestroy(r.ready);
        if (r.copied) (void)hipEventDestroy(r.copied);
        vp_ctx_destroy(r.ctx);
    }
    delete m;
    return 0;
}

int vp_multi_count(const vp_multi* m) { return m ? (int)m->ranks.size() : 0; }

vp_ctx* vp_multi_ctx(vp_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->ranks.size()) ? m->ranks[(size_t)rank].ctx : nullptr; }

int vp_multi_sync(vp_multi* m)
{
    if (!m) return set_error(VP_ERR_INVALID, "vp_multi_sync: null argument");
    for (Rank& r : m->ranks) { VP_TRY(bind(r)); VP_HIP(hipStreamSynchronize(r.ctx->stream)); }
    return 0;
}

int vp_multi_set_mesh(vp_multi* m, const float* h_xyz, size_t nverts, const uint32_t* h_tri, size_t ntris)
{
    if (!m || ((!h_xyz || !h_tri) && ntris)) return set_error(VP_ERR_INVALID, "vp_multi_set_mesh: null argument");
    for (Rank& r : m->ranks) {
        VP_TRY(grow(r, r.mesh_xyz, nverts * 12)); VP_TRY(grow(r, r.mesh_tri, ntris * 12));
        if (nverts) VP_HIP(hipMemcpyAsync(r.mesh_xyz.ptr, h_xyz, nverts * 12, hipMemcpyHostToDevice, r.ctx->stream));
        if (ntris) VP_HIP(hipMemcpyAsync(r.mesh_tri.ptr, h_tri, ntris * 12, hipMemcpyHostToDevice, r.ctx->stream));
    }
    m->nverts = nverts; m->ntris = ntris;
    return vp_multi_sync(m);                                       // the host arrays may go away
}

int vp_multi_voxelize(vp_multi* m, const vp_frame* f, int algo)
{
    VP_TRY(check_split(m, f, "vp_multi_voxelize"));
    // from here on the resident grid is being replaced: a failure below must not leave the OLD grid flagged as resident under the
    // NEW frame and slab bounds
    m->have_grid = false; m->have_sdf = false;
    assign_slabs(m, f);
    for (Rank& r : m->ranks) {
        VP_TRY(grow(r, r.words, vp_grid_words(f) * 4));
        const vp_frame sf = slab_frame(*f, r.z0, r.z1);
        VP_TRY(vp_voxelize(r.ctx, &sf, (uint32_t*)slab_words(m, r), (const float*)r.mesh_xyz.ptr, m->nverts, (const uint32_t*)r.mesh_tri.ptr, m->ntris, algo, 0));
    }
    m->have_grid = true;
    return 0;
}

int vp_multi_set_grid(vp_multi* m, const vp_frame* f, const uint32_t* h_words)
{
    VP_TRY(check_split(m, f, "vp_multi_set_grid"));
    if (!h_words) return set_error(VP_ERR_INVALID, "vp_multi_set_grid: null argument");
    m->have_grid = false; m->have_sdf = false;                     // see vp_multi_voxelize
    assign_slabs(m, f);
    const size_t planeWords = (size_t)f->n * f->n / 8;
    for (Rank& r : m->ranks) {
        VP_TRY(grow(r, r.words, vp_grid_words(f) * 4));
        VP_HIP(hipMemcpyAsync(slab_words(m, r), (const char*)h_words + (size_t)r.z0 * planeWords, (size_t)(r.z1 - r.z0) * planeWords, hipMemcpyHostToDevice, r.ctx->stream));
    }
    VP_TRY(vp_multi_sync(m));
    m->have_grid = true;
    return 0;
}

int vp_multi_get_grid(vp_multi* m, uint32_t* h_words)
{
    if (!m || !h_words || !m->have_grid) return set_error(VP_ERR_INVALID, "vp_multi_get_grid: no resident grid");
    const size_t planeWords = (size_t)m->frame.n * m->frame.n / 8;
    for (Rank& r : m->ranks) {
        VP_TRY(bind(r));
        VP_HIP(hipMemcpyAsync((char*)h_words + (size_t)r.z0 * planeWords, slab_words(m, r), (size_t)(r.z1 - r.z0) * planeWords, hipMemcpyDeviceToHost, r.ctx->stream));
    }
    return vp_multi_sync(m);
}

int vp_multi_csg(vp_multi* m, const uint32_t* h_other, size_t nwords, int op)
{
    if (!m || !h_other || !m->have_grid) return set_error(VP_ERR_INVALID, "vp_multi_csg: no resident grid");
    if (nwords != vp_grid_words(&m->frame))
        return set_error(VP_ERR_INVALID, "vp_multi_csg: %zu words given, the resident grid has %zu (csg/naive.cu:30-33 requires equal grids)", nwords, vp_grid_words(&m->frame));
    const size_t planeWords = (size_t)m->frame.n * m->frame.n / 8;
    for (Rank& r : m->ranks) {
        const size_t bytes = (size_t)(r.z1 - r.z0) * planeWords;
        VP_TRY(grow(r, r.other, bytes));
        VP_HIP(hipMemcpyAsync(r.other.ptr, (const char*)h_other + (size_t)r.z0 * planeWords, bytes, hipMemcpyHostToDevice, r.ctx->stream));
        VP_TRY(vp_csg(r.ctx, (uint32_t*)slab_words(m, r), (const uint32_t*)r.other.ptr, bytes / 4, op));
    }
    m->have_sdf = false;
    return vp_multi_sync(m);
}

int vp_multi_jfa(vp_multi* m, float fill_unset, int algo, int mode)
{
    if (!m || !m->have_grid) return set_error(VP_ERR_INVALID, "vp_multi_jfa: no resident grid (vp_multi_voxelize / vp_multi_set_grid first)");
    if (!std::isinf(fill_unset)) return set_error(VP_ERR_INVALID, "vp_multi_jfa: fill_unset must be +-infinity");
    if (algo != VP_ALGO_NAIVE && algo != VP_ALGO_TILED) return set_error(VP_ERR_INVALID, "vp_multi_jfa: algo %d", algo);
    if (mode != VP_MULTI_HALO && mode != VP_MULTI_GHOST && mode != VP_MULTI_HYBRID && mode != VP_MULTI_TRANSPOSE) return set_error(VP_ERR_INVALID, "vp_multi_jfa: mode %d", mode);
    m->bytes_moved = 0;
    m->last_mode = mode;
    // The sharded forms run the tile kernels on id windows whatever `algo` says (both give the same sdf); below their range the grid is
    // not sharded and `algo` picks the kernel of the whole-grid JFA every device runs.
    if (m->frame.n < 96) VP_TRY(jfa_small(m, fill_unset, algo));
    else if (mode == VP_MULTI_GHOST) VP_TRY(jfa_ghost(m, fill_unset));
    else if (mode == VP_MULTI_HYBRID) VP_TRY(jfa_hybrid(m, fill_unset));
    else if (mode == VP_MULTI_TRANSPOSE) VP_TRY(jfa_transpose(m, fill_unset));
    else VP_TRY(jfa_halo(m, fill_unset));
    m->have_sdf = true;
    return 0;
}

int vp_multi_get_sdf(vp_multi* m, float* h_sdf)
{
    if (!m || !h_sdf || !m->have_sdf) return set_error(VP_ERR_INVALID, "vp_multi_get_sdf: no sdf (vp_multi_jfa first)");
    const size_t plane = (size_t)m->frame.n * m->frame.n * 4;
    for (Rank& r : m->ranks) {
        VP_TRY(bind(r));
        VP_HIP(hipMemcpyAsync((char*)h_sdf + (size_t)r.z0 * plane, r.sdf.ptr, (size_t)(r.z1 - r.z0) * plane, hipMemcpyDeviceToHost, r.ctx->stream));
    }
    return vp_multi_sync(m);
}

uint64_t vp_multi_bytes_moved(const vp_multi* m) { return m ? m->bytes_moved : 0; }

int vp_multi_window(const vp_multi* m, int rank, uint32_t* lo, uint32_t* hi, uint64_t* id_bytes)
{
    if (!m || rank < 0 || rank >= (int)m->ranks.size() || m->last_mode < 0) return set_error(VP_ERR_INVALID, "vp_multi_window: no JFA has run");
    const uint32_t n = m->frame.n, world = (uint32_t)m->ranks.size(), nz = n / world;
    const Rank& r = m->ranks[(size_t)rank];
    uint32_t a = 0, b = n;                                          // ghost (and n < 96): whole volumes
    if (n >= 96 && m->last_mode == VP_MULTI_HALO) { a = r.z0; b = r.z1; (void)nz; }       // the slab; its windows also hold the slabs received from z -+ k
    else if (n >= 96 && (m->last_mode == VP_MULTI_HYBRID || m->last_mode == VP_MULTI_TRANSPOSE)) { a = m->window_lo[(size_t)rank]; b = m->window_hi[(size_t)rank]; }
    if (lo) *lo = a;
    if (hi) *hi = b;
    // the id windows the mode of the last job used (buffers are grow-only and kept: those of other modes are not this job's state)
    if (id_bytes) {
        *id_bytes = (uint64_t)vp_jfa_window_bytes(&m->frame, r.win_planes) * 2;
        if (n >= 96 && m->last_mode == VP_MULTI_TRANSPOSE) *id_bytes += (uint64_t)vp_jfa_window_bytes(&m->frame, r.cyc_planes) * 2 + vp_jfa_window_bytes(&m->frame, r.stg_planes);
    }
    return 0;
}

}  // extern "C"
