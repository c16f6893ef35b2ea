// Dev probe (round 2): facts about gfx950 the f64-min JFA update relies on.
//   1. issue rate of v_min_f64 alone and in the candidate-step sequence (v_add_f32 into the high half + v_min_f64),
//      next to the round-1 sequence (v_add_f32 + v_cmpx + 2 v_mov + s_mov exec);
//   2. v_min_f64 on (hi = bits of a non-negative float, lo = payload) IS the unsigned 64-bit minimum, bit for bit
//      (also for hi = 0 / tiny: f64 denormals must not be flushed; hi = 0x7F800000 is a finite double);
//   3. raw-buffer range check: does soffset take part?  structured (idxen) loads: range check by index;
//   4. wide LDS reads (b64 / b128) against b32.
// hipcc --offload-arch=gfx950 -O3 -o probe probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x

template <int OP>
__global__ void __launch_bounds__(256) rate(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b = seed * 0.5f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a0 * 3, d5 = a1 * 3, d6 = a2 * 3, d7 = a3 * 3, db = b;
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a0 + i;
    __syncthreads();
    unsigned addr4 = (threadIdx.x * 4u) & 4095u, addr8 = (threadIdx.x * 8u) & 4095u, addr16 = (threadIdx.x * 16u) & 4095u;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));) }
        // candidate step, new: d -> high half of the candidate pair, then one 64-bit minimum  (4 steps = 8 VALU)
        if (OP == 1) { REP8(asm volatile("v_add_f32 v101, %4, %5\n v_min_f64 %0, %0, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %1, %1, v[102:103]\n"
                                         "v_add_f32 v101, %4, %5\n v_min_f64 %2, %2, v[100:101]\n v_add_f32 v103, %4, %5\n v_min_f64 %3, %3, v[102:103]"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "v100", "v101", "v102", "v103");) }
        // candidate step, round 1: add + cmpx + 2 moves + exec restore (4 steps = 16 VALU + 4 SALU)
        if (OP == 2) { REP8(asm volatile("s_mov_b64 s[20:21], exec\n"
                                         "v_add_f32 v100, %4, %5\n v_cmpx_lt_f32 exec, v100, %0\n v_mov_b32 %0, v100\n v_mov_b32 %1, %5\n s_mov_b64 exec, s[20:21]\n"
                                         "v_add_f32 v101, %4, %5\n v_cmpx_lt_f32 exec, v101, %2\n v_mov_b32 %2, v101\n v_mov_b32 %3, %5\n s_mov_b64 exec, s[20:21]\n"
                                         "v_add_f32 v100, %4, %5\n v_cmpx_lt_f32 exec, v100, %0\n v_mov_b32 %0, v100\n v_mov_b32 %1, %5\n s_mov_b64 exec, s[20:21]\n"
                                         "v_add_f32 v101, %4, %5\n v_cmpx_lt_f32 exec, v101, %2\n v_mov_b32 %2, v101\n v_mov_b32 %3, %5\n s_mov_b64 exec, s[20:21]"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(seed) : "v100", "v101", "s20", "s21", "vcc");) }
        if (OP == 3) { REP8(asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));) }
        if (OP == 4) { REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr4));) }
        if (OP == 5) { REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                                         : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(addr8));) }
        if (OP == 6) { REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                                         : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(addr16));) }
        if (OP == 7) { REP8(asm volatile("v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        // new step with the payload rebuilt per step (lo = id27 | seq << 27): add + add_u32 + min_f64
        if (OP == 8) { REP8(asm volatile("v_add_f32 v101, %4, %5\n v_add_u32 v100, %6, %7\n v_min_f64 %0, %0, v[100:101]\n v_add_f32 v103, %4, %5\n v_add_u32 v102, %6, %7\n v_min_f64 %1, %1, v[102:103]\n"
                                         "v_add_f32 v101, %4, %5\n v_add_u32 v100, %6, %7\n v_min_f64 %2, %2, v[100:101]\n v_add_f32 v103, %4, %5\n v_add_u32 v102, %6, %7\n v_min_f64 %3, %3, v[102:103]"
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b), "v"(addr4), "v"(addr8) : "v100", "v101", "v102", "v103");) }
        // packed f32: two adds / multiplies per lane and instruction
        if (OP == 10) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                                          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));) }
        if (OP == 11) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                                          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));) }
        // mixed stream as in the tile loop: one packed add per two plain adds and one v_min_f64
        if (OP == 12) { REP8(asm volatile("v_pk_add_f32 %4, %4, %8\n v_add_f32 v101, %6, %7\n v_min_f64 %0, %0, v[100:101]\n v_add_f32 v103, %6, %7\n v_min_f64 %1, %1, v[102:103]\n"
                                          "v_pk_add_f32 %5, %5, %8\n v_add_f32 v101, %6, %7\n v_min_f64 %2, %2, v[100:101]\n v_add_f32 v103, %6, %7\n v_min_f64 %3, %3, v[102:103]"
                                          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5) : "v"(a0), "v"(b), "v"(db) : "v100", "v101", "v102", "v103");) }
        // f32 variant of the final pass: add + v_min_f32
        if (OP == 9) { REP8(asm volatile("v_add_f32 v100, %4, %5\n v_min_f32 %0, %0, v100\n v_add_f32 v101, %4, %5\n v_min_f32 %1, %1, v101\n v_add_f32 v100, %4, %5\n v_min_f32 %2, %2, v100\n v_add_f32 v101, %4, %5\n v_min_f32 %3, %3, v101"
                                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(seed) : "v100", "v101");) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + q0.x + q1.y + q2.z + q3.w;
}

template <int OP>
static void run(const char* name, float* out, int wgPerCu, double perIter)
{
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(cus * wgPerCu), dim3(256), 0, 0, out, 10, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(cus * wgPerCu), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double units = (double)iters * 8 * perIter * wgPerCu;   // units per SIMD
    printf("%-44s waves/SIMD=%d  %8.3f ms  %.2f clk@2.4GHz per unit per SIMD\n", name, wgPerCu, ms, ms * 1e6 / units * 2.4);
}

// ---------------------------------------------------------------- 2. exactness of v_min_f64 as u64 min
__global__ void min64(const uint64_t* a, const uint64_t* b, uint64_t* o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = __builtin_bit_cast(double, a[i]), y = __builtin_bit_cast(double, b[i]), r;
    asm volatile("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    o[i] = __builtin_bit_cast(uint64_t, r);
}

// ---------------------------------------------------------------- 3. buffer addressing
__global__ void bufprobe(const uint32_t* base, uint32_t* out)
{
    const int lane = threadIdx.x;
    // raw buffer over the first 256 bytes; soffset moves the window by 512 bytes
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, 256, 0x00020000);
    out[lane] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 8, 0, 0);            // lanes >= 32 out of range
    out[64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 8, 512, 0);     // soffset 512: in range iff soffset is not checked
    out[128 + lane] = __builtin_amdgcn_raw_buffer_load_b32(r, (lane - 8) * 4, 0, 0); // negative offsets (lanes < 8)
    // structured buffer: stride 4, num_records = 100 elements; index = lane * 2 - 4
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t ba = reinterpret_cast<uint64_t>(base);
    u32x4 s;
    s[0] = __builtin_amdgcn_readfirstlane((uint32_t)ba);
    s[1] = __builtin_amdgcn_readfirstlane((uint32_t)(ba >> 32) | (4u << 16));      // stride 4
    s[2] = __builtin_amdgcn_readfirstlane(100u);                                   // num_records (elements)
    s[3] = __builtin_amdgcn_readfirstlane(0x00020000u);
    uint32_t v0, v1; int idx0 = lane * 2 - 4, idx1 = lane; uint32_t so = __builtin_amdgcn_readfirstlane(1024u);
    asm volatile("buffer_load_dword %0, %1, %2, 0 idxen\n s_waitcnt vmcnt(0)" : "=v"(v0) : "v"(idx0), "s"(s) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 idxen\n s_waitcnt vmcnt(0)" : "=v"(v1) : "v"(idx1), "s"(s), "s"(so) : "memory");
    out[192 + lane] = v0;
    out[256 + lane] = v1;
}

int main()
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {2, 4, 8}) {
        run<7>("v_add_f32 (unit = 1 instr)", out, w, 8);
        run<0>("v_min_f64 (unit = 1 instr)", out, w, 8);
        run<3>("v_mov_b64 (unit = 1 instr)", out, w, 8);
        run<1>("step: add_f32 + min_f64 (unit = step)", out, w, 4);
        run<8>("step: add_f32 + add_u32 + min_f64 (unit = step)", out, w, 4);
        run<2>("step r1: add + cmpx + 2 mov + s_mov (unit = step)", out, w, 4);
        run<9>("step f32: add + min_f32 (unit = step)", out, w, 4);
        run<10>("v_pk_add_f32 (unit = 1 instr)", out, w, 8);
        run<11>("v_pk_mul_f32 (unit = 1 instr)", out, w, 8);
        run<12>("pk_add + 2 x (add_f32 + min_f64) (unit = 5 instr)", out, w, 2);
        run<4>("ds_read_b32 x4 (unit = 1 instr)", out, w, 4);
        run<5>("ds_read_b64 x4 (unit = 1 instr)", out, w, 4);
        run<6>("ds_read_b128 x4 (unit = 1 instr)", out, w, 4);
    }
    {   // exactness
        const int n = 1 << 22;
        std::vector<uint64_t> a(n), b(n), o(n);
        srand(1);
        auto rnd32 = []() { return ((uint32_t)rand() << 16) ^ (uint32_t)rand(); };
        for (int i = 0; i < n; ++i) {
            auto hi = [&](int m) -> uint32_t {
                switch (m % 8) {
                case 0: return 0u;                               // d = 0
                case 1: return rnd32() % 0x00100000u;            // double-denormal range
                case 2: return 0x7F800000u;                      // +inf as float
                case 3: return 0x7F7FFFFFu;
                default: return rnd32() % 0x7F800001u;
                }
            };
            uint32_t ha = hi(rand()), hb = (rand() % 3 == 0) ? ha : hi(rand());
            a[i] = ((uint64_t)ha << 32) | rnd32();
            b[i] = ((uint64_t)hb << 32) | ((rand() % 5 == 0) ? (uint32_t)a[i] : rnd32());
            if (i % 97 == 0) a[i] &= 0xFFFFFFFF00000000ull;       // lo = 0 (the own candidate)
        }
        uint64_t *da, *db2, *dout;
        hipMalloc(&da, n * 8); hipMalloc(&db2, n * 8); hipMalloc(&dout, n * 8);
        hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db2, b.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(min64, dim3(n / 256), dim3(256), 0, 0, da, db2, dout, n);
        hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) { uint64_t e = a[i] < b[i] ? a[i] : b[i]; if (o[i] != e) { if (bad < 5) printf("  mismatch a=%016llx b=%016llx got=%016llx\n", (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)o[i]); ++bad; } }
        printf("v_min_f64 as u64 min: %ld mismatches of %d\n", bad, n);
    }
    {   // buffer semantics
        std::vector<uint32_t> h(4096);
        for (int i = 0; i < 4096; ++i) h[i] = 0x1000 + i;       // never 0
        uint32_t *d, *dout; hipMalloc(&d, 4096 * 4); hipMalloc(&dout, 320 * 4);
        hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(bufprobe, dim3(1), dim3(64), 0, 0, d, dout);
        std::vector<uint32_t> o(320); hipMemcpy(o.data(), dout, 320 * 4, hipMemcpyDeviceToHost);
        auto show = [&](const char* name, int off) { printf("%s:", name); for (int l = 0; l < 64; l += 1) printf(" %x", o[off + l]); printf("\n"); };
        show("raw 256B, voffset=lane*8        ", 0);
        show("raw 256B, voffset=lane*8, soff 512", 64);
        show("raw 256B, voffset=(lane-8)*4    ", 128);
        show("struct stride4 n=100 idx=2*lane-4", 192);
        show("struct idx=lane soffset=1024    ", 256);
    }
    return 0;
}
