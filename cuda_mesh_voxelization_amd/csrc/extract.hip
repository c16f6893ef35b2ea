// extract.hip -- ordered compaction of a bit-packed grid into voxel records, for the export stage
// (/root/reference/vplib/src/mesh/grid_to_mesh.cpp:10-201 walks all n^3 voxels on the CPU; here the walk is a GPU
// stream and the CPU only sees the voxels it will emit).
//
//   mode VP_EXTRACT_SET      every set voxel                                   (point cloud / sdf cubes, :133-201)
//   mode VP_EXTRACT_EXPOSED  set voxels with at least one face towards an unset voxel or the outside of the grid,
//                            with the 6-bit mask of those faces: bit = axis * 2 + side, axes X, Y, Z, side 0 = minus
//                            (visible-surface mesh)
//   mode VP_EXTRACT_FACES    every set voxel WITH that mask: what VoxelsGridToMeshCompressed needs (:10-131 emits every face of every set
//                            voxel once: the three "front" faces always, a "back" face iff the voxel behind it is unset -- bit side 0)
// Record = linear voxel index (x + n (y + n z), bits 0..39) | face mask << 40.  Records come out in ascending index
// order = the exporter's z, y, x scan order, so the host builds byte-identical files from them.
//
// Three launches: per-block counts (one lane = one 32-voxel word, 8 words per lane), a one-workgroup exclusive scan of
// the block counts (64-bit totals: n = 2048 has 8.6e9 voxels), and the write pass, which redoes the word tests, scans
// the lane counts inside the workgroup and stores each lane's records behind its block's offset.
#include "vp_internal.h"

namespace vp {

namespace {

constexpr int kWordsPerLane = 8;
constexpr int kBlockWords = 256 * kWordsPerLane;

__device__ __forceinline__ uint32_t word_at(const Frame& f, const uint32_t* __restrict__ words, int xw, int y, int z)
{
    if (xw < 0 || xw >= (int)f.w || y < 0 || y >= (int)f.n || z < 0 || z >= (int)f.n) return 0u;   // outside = unset
    return words[((size_t)z * f.n + y) * f.w + xw];
}

// selected voxels of word wi, and (EXPOSED) the six per-face masks
template <int MODE>
__device__ __forceinline__ uint32_t select(const Frame& f, const uint32_t* __restrict__ words, size_t wi, uint32_t (&face)[6])
{
    const uint32_t c = words[wi];
    for (int q = 0; q < 6; ++q) face[q] = 0u;
    if (MODE == VP_EXTRACT_SET || c == 0u) return c;
    const int xw = (int)(wi % f.w);
    const size_t row = wi / f.w;
    const int y = (int)(row % f.n), z = (int)(row / f.n);
    const uint32_t xm = (c << 1) | (word_at(f, words, xw - 1, y, z) >> 31);      // bit i = voxel x-1
    const uint32_t xp = (c >> 1) | (word_at(f, words, xw + 1, y, z) << 31);      // bit i = voxel x+1
    face[0] = c & ~xm; face[1] = c & ~xp;
    face[2] = c & ~word_at(f, words, xw, y - 1, z); face[3] = c & ~word_at(f, words, xw, y + 1, z);
    face[4] = c & ~word_at(f, words, xw, y, z - 1); face[5] = c & ~word_at(f, words, xw, y, z + 1);
    return MODE == VP_EXTRACT_FACES ? c : (face[0] | face[1] | face[2] | face[3] | face[4] | face[5]);
}

template <int MODE>
__global__ void __launch_bounds__(256)
extract_count(Frame f, const uint32_t* __restrict__ words, size_t nwords, uint32_t* __restrict__ block_count)
{
    __shared__ uint32_t part[4];
    const size_t w0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * kWordsPerLane;
    uint32_t cnt = 0;
    uint32_t face[6];
    for (int j = 0; j < kWordsPerLane; ++j)
        if (w0 + j < nwords) cnt += __popc(select<MODE>(f, words, w0 + j, face));
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// one workgroup: off[i] = sum of cnt[0..i), off[m] = total
__global__ void __launch_bounds__(1024)
extract_scan(const uint32_t* __restrict__ cnt, size_t m, unsigned long long* __restrict__ off)
{
    __shared__ unsigned long long part[1024];
    const size_t tid = threadIdx.x;
    const size_t per = (m + 1023) / 1024;
    const size_t b = min(tid * per, m), e = min(b + per, m);
    unsigned long long s = 0;
    for (size_t i = b; i < e; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = (tid >= (size_t)d) ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = part[tid] - s;
    for (size_t i = b; i < e; ++i) { off[i] = run; run += cnt[i]; }
    if (tid == 1023) off[m] = part[1023];
}

template <int MODE>
__global__ void __launch_bounds__(256)
extract_write(Frame f, const uint32_t* __restrict__ words, size_t nwords, const unsigned long long* __restrict__ block_off,
              const float* __restrict__ sdf, unsigned long long* __restrict__ records, float* __restrict__ values, size_t capacity)
{
    __shared__ uint32_t wave_sum[4];
    const size_t w0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * kWordsPerLane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t sel[kWordsPerLane];
    uint32_t face[kWordsPerLane][6];
    uint32_t cnt = 0;
    for (int j = 0; j < kWordsPerLane; ++j) {
        sel[j] = (w0 + j < nwords) ? select<MODE>(f, words, w0 + j, face[j]) : 0u;
        cnt += __popc(sel[j]);
    }
    // exclusive scan of the lane counts inside the workgroup
    uint32_t incl = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - cnt;
    for (int w = 0; w < wave; ++w) before += wave_sum[w];
    unsigned long long pos = block_off[blockIdx.x] + before;
    for (int j = 0; j < kWordsPerLane; ++j) {
        uint32_t m = sel[j];
        while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            const unsigned long long idx = (unsigned long long)(w0 + j) * 32ull + (unsigned)b;
            unsigned long long rec = idx;
            if (MODE != VP_EXTRACT_SET) {
                unsigned fm = 0;
                for (int q = 0; q < 6; ++q) fm |= ((face[j][q] >> b) & 1u) << q;
                rec |= (unsigned long long)fm << 40;
            }
            if (pos < capacity) {
                records[pos] = rec;
                if (values) values[pos] = sdf[idx];
            }
            ++pos;
        }
    }
}

}  // namespace

// Counts (blocking: the total is read back) and leaves the block offsets in the context for launch_extract_write.
int launch_extract_count(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, int mode, uint64_t* h_count)
{
    const size_t nwords = (size_t)f.n * f.n * f.n / 32;
    const size_t blocks = (nwords + kBlockWords - 1) / kBlockWords;
    VP_TRY(reserve(ctx, ctx->ext_cnt, blocks * 4));
    VP_TRY(reserve(ctx, ctx->ext_off, (blocks + 1) * 8));
    uint32_t* cnt = (uint32_t*)ctx->ext_cnt.ptr;
    unsigned long long* off = (unsigned long long*)ctx->ext_off.ptr;
    {
        ProfScope p(ctx, VP_K_EXTRACT);
        if (mode == VP_EXTRACT_EXPOSED) hipLaunchKernelGGL(extract_count<VP_EXTRACT_EXPOSED>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, f, d_words, nwords, cnt);
        else                            hipLaunchKernelGGL(extract_count<VP_EXTRACT_SET>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, f, d_words, nwords, cnt);   // SET and FACES: every set voxel
        hipLaunchKernelGGL(extract_scan, dim3(1), dim3(1024), 0, ctx->stream, cnt, blocks, off);
    }
    VP_HIP(hipGetLastError());
    unsigned long long total = 0;
    VP_HIP(hipMemcpyAsync(&total, off + blocks, 8, hipMemcpyDeviceToHost, ctx->stream));
    VP_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ext_words = d_words; ctx->ext_mode = mode; ctx->ext_n = f.n; ctx->ext_total = total;
    if (h_count) *h_count = total;
    return 0;
}

int launch_extract_write(vp_ctx* ctx, const Frame& f, const uint32_t* d_words, int mode, const float* d_sdf,
                         uint64_t* d_records, float* d_values, size_t capacity)
{
    if (ctx->ext_words != d_words || ctx->ext_mode != mode || ctx->ext_n != f.n)
        return set_error(VP_ERR_INVALID, "vp_extract: call vp_extract_count with the same grid and mode first");
    const size_t nwords = (size_t)f.n * f.n * f.n / 32;
    const size_t blocks = (nwords + kBlockWords - 1) / kBlockWords;
    const unsigned long long* off = (const unsigned long long*)ctx->ext_off.ptr;
    ProfScope p(ctx, VP_K_EXTRACT);
    if (mode == VP_EXTRACT_SET)
        hipLaunchKernelGGL(extract_write<VP_EXTRACT_SET>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, f, d_words, nwords, off, d_sdf,
                           (unsigned long long*)d_records, d_values, capacity);
    else if (mode == VP_EXTRACT_FACES)
        hipLaunchKernelGGL(extract_write<VP_EXTRACT_FACES>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, f, d_words, nwords, off, d_sdf,
                           (unsigned long long*)d_records, d_values, capacity);
    else
        hipLaunchKernelGGL(extract_write<VP_EXTRACT_EXPOSED>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, f, d_words, nwords, off, d_sdf,
                           (unsigned long long*)d_records, d_values, capacity);
    VP_HIP(hipGetLastError());
    return 0;
}

}  // namespace vp
