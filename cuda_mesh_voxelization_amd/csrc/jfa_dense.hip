// jfa_dense.hip -- the tile kernel of the JFA passes (jfa_pass_dense) and its launcher.  Every pass of VP_ALGO_TILED at n >= 96 except
// the fused first two runs here: whole grids, the regions of the ghost-plane pipelines, the slabs of the halo pipelines -- all of them
// on id WINDOWS (jfa_common.h), the last pass fused with the id -> sdf conversion.
#include "jfa_common.h"

namespace vp {
namespace {

// A workgroup owns a TILE of RY output rows x CH output planes, both k apart: rows y_a = y0 + a*k, planes z_j = z0 + j*k.  Output
// (y_a, z_j) takes its candidates from rows y_a - k, y_a, y_a + k of planes z_j - k, z_j, z_j + k, i.e. from the tile's own rows / planes
// and one halo row / plane on each side.  Every source plane of the tile is therefore read ONCE -- (RY+2) rows x columns {x-k, x, x+k}
// per thread -- and serves up to 3 output rows x 3 output planes: 3(RY+2)(CH+2)/(RY*CH) = 5.6 loads per voxel (4 x 8) instead of 27.
// The kernel is input-stationary: every id is decoded once and scattered into the running (distance, rank) pairs of the outputs it is a
// candidate for.  LDS tables at fixed addresses turn id fields into seed x, dy^2 per output row and dz^2 per output plane (compact ids:
// the seed's z position); dx^2 is computed once per id, fl(dx^2 + dy^2) once per (id, output row), which keeps the reference's association
// ((dx^2 + dy^2) + dz^2) (jfa/jfa.h:19-20).
//
//  * Candidate update = ONE v_min_f64.  The running best of an output is the 64-bit pair (hi = bits of the distance, lo = rank of the
//    candidate).  A non-negative float's bit pattern orders like an unsigned integer, and a bit pattern with hi <= 0x7F800000 is a finite,
//    non-negative double whose order is that of the 64-bit pattern, so v_min_f64 on such pairs IS the lexicographic minimum of (distance,
//    rank) -- verified bit for bit on the part, denormal range included (tools/ubench/probe.hip) -- at 4.3 clocks per wave against 9 for
//    v_cmpx + 2 v_mov + the EXEC restore.  The rank makes the minimum the reference's "first minimum in scan order, own state first"
//    (jfa/sequential.cpp:84-112): the own voxel has rank 0, every other candidate the byte offset of its SOURCE voxel from the tile's
//    first source plane + 1, which increases along the scan order z, y, x.  v_add_f32 writes the distance straight into the high half of
//    the candidate pair (the low half is set once per loaded id): a candidate-step is 2 VALU, 6.3 clocks.
//    When an output is complete the seed id of its winner is fetched from where the winner was read: one gather load per voxel, issued a
//    plane ahead of its store.
//  * Rolling prefetch: the ids of a source row are spent once the row is scattered, so the same row of the NEXT source plane is requested
//    into their registers right away -- a whole plane of evaluation ahead of its use, without a second id buffer.
//  * FINAL (the last pass, k = 1, fused with the id -> sdf conversion) keeps distances only (v_min3_f32 / v_min_f32) and shares the tables.
//    The bitmask words of its output rows (the sign of the sdf) come from LDS (staged with the tables: 2 KB more, one workgroup per CU
//    less) with the 2-KB tables and straight from global memory above (profiles/r02/ab22.txt: global is +3.4 % at n = 512 and -4.7 % at
//    n = 1024, where it frees a third 512-thread workgroup per CU).
//  * Windows and strides.  `in` / `out` point at plane z0 of the frame inside an id WINDOW (jfa_common.h); the chain member P of a tile
//    (global plane zbase + P k) sits `P * ka` planes from its first member in the window: ka = k for a window of consecutive planes, ka =
//    the slab height for the [slab of z - k | own slab | slab of z + k] windows of the halo pipelines, whose steps k >= nz span whole slabs.
//  * Compact ids (IdC, n > 1024): the id state is a word plane and a byte plane (10 instead of 16 bytes per voxel and pass); 8-KB tables:
//    PX, the squared y differences per output row (TY) and ONE table of seed z POSITIONS -- (sz - pz)^2 is formed per id and output plane
//    (2 VALU each) -- so that the footprint is 52 KB and three 512-thread workgroups fit a CU.  "none" goes through the slots beyond 2048
//    of the z table (see IdC).  The byte offset of a source voxel no longer fits the rank: it is (index of the source row among the tile's
//    (CH + 2) x (RY + 2) source rows + 1) << 14 | byte offset of the column + 1 | top z bit << 1 (the own voxel: the top z bit alone, below
//    every other rank); an LDS table turns the row index of the winner back into its row number, the gather fetches one dword and the
//    byte of the output is rebuilt from the winning pair (distance = +inf: "none").
//  * Output stores leave with the nt policy (kStoreAux): the output is not read again before the next pass; reads -15 % / -17 % (n = 512 /
//    1024: fewer source lines evicted), dense passes -1.1 % / -2.2 % (profiles/r04/ab_store_*.txt).  The winner gather keeps the default
//    policy (nt loads bypass the L1, where the gathers of neighbouring lanes hit: +15 % / +8 %, ab_gnt_ry8_*.txt).
template <class ID> constexpr bool final_mask_global() { return ID::kTab > 512; }
constexpr int kStoreAux = 2;
// Pair mode (PM = 1, 2, 4, 8; round 3): the lanes of a wave are paired so that the voxels x and x + k sit in two lanes one DPP
// permutation apart (quad_perm for k = 1, 2; row_half_mirror for k = 4; row_ror:8 for k >= 8 -- the map from lane to x below keeps
// the 64 voxels of a wave inside at most two 128-byte runs).  Each lane then loads and decodes TWO ids per source row instead of
// three -- its own column and the column beyond it (x - k for the lower lane of a pair, x + k for the upper) -- and evaluates the
// third column, which IS its partner's own column, from the partner's decoded values: seed x, squared y / z differences arrive as
// the DPP operand of the very v_sub_f32 / v_add_f32 that consumes them.  3.75 instead of 5.6 loads, decodes and table lookups per
// voxel; the candidates, their order-defining ranks and every float operation are unchanged.  A DPP operand costs the instruction
// 1.7 clocks more (tools/ubench/probe3.hip), which eats most of the VALU saved by the decodes; dense passes -3 % / -6 % (n = 512 / 1024),
// fused last pass -6 % (profiles/r03/ab_pairs_*.txt).  Needs n a power of two and n % NT == 0 (every lane of every wave has its partner).
template <int PM>
__device__ __forceinline__ float from_partner(float v)
{
    constexpr int ctrl = PM == 1 ? 0xB1 : PM == 2 ? 0x4E : PM == 4 ? 0x141 : 0x128;   // quad_perm:[1,0,3,2] / [2,3,0,1] / row_half_mirror / row_ror:8
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}
// Closed tiles (CLOSED; round 3): at k = n/8 a chain of rows or planes has exactly eight members, so a tile of 8 rows x 8 planes IS a
// pair of whole chains: it has no halo rows and no halo planes at all -- where the 4 x 8 tile read 6 x 10 row-planes (two planes and
// one row of them outside the grid: a third of that pass's candidate steps were spent on "none") it reads 8 x 8, every one of them
// its own.  2 instead of 3.75 decoded ids per voxel (pair mode), 22.7 instead of 27 candidate steps (edge outputs have fewer
// neighbours), at the price of 48 running-pair registers (four waves per SIMD).  Compile-time: the loops simply lose their halo
// iterations.  Whole grids with n = 8 k only.  CLOSED is a mask: 1 = the rows of a tile are a whole chain (RY = n / k), 2 = the planes
// are (CH = n / k).  With the 4-KB tables (n = 1024) the 8 x 8 tile takes 68 KB of LDS (two 512-thread workgroups per CU): k = 128 at
// n = 1024 2.83 -> 2.66 ms against the planes-only form (profiles/r03/ab_clbig_1024.txt).
// FULL (round 4): every tile of the launch has all its RY output rows and CH output planes (n and the slab are multiples of RY k and
// CH k -- any whole power-of-two grid): the row / plane counts of a tile become compile-time constants and the ~220 workgroup-uniform
// branches around the stores and gathers of an x iteration (one per output row and plane, `if (a >= yout)`) disappear.
// Occupancy the register allocation aims at: six waves per SIMD where the LDS footprint allows six workgroups (2-KB tables) or three
// 512-thread ones (4-KB tables); four for 8-row tiles, the 8-KB tables of the compact ids (six would leave 80 VGPRs: the id passes spill
// 50 registers, 34.0 -> 46.2 ms) and the 256-thread variant with 4-KB tables (LDS-limited).
template <class ID, int RY, int NT, bool FINAL>
constexpr int dense_waves()
{
    return std::is_same<ID, IdC>::value ? 4 : RY > 4 ? 4 : (FINAL && !final_mask_global<ID>()) ? (ID::kTab == 512 ? 5 : 4) : (ID::kTab == 512 || NT == 512) ? 6 : 4;
}
// CYC (round 6): the planes of the frame are those of ONE rank of a CYCLIC distribution of the grid over 2^zsh ranks (the first phase of
// the transposed multi-GPU pipeline): frame plane l is the global plane zoff + (l << zsh).  A pass whose step k is a
// multiple of the rank count finds the planes z - k, z, z + k of every plane it owns on the same rank, k >> zsh frame planes away: the
// tile's plane chain, its bounds and the window offsets are those of a grid of n >> zsh planes walked with the step k >> zsh, rows and
// columns keep the step k, and only the POSITIONS of the output planes (pz) are taken from the global plane number -- the ids carry global
// coordinates as everywhere else.  `cyc` = zsh | zoff << 8.
template <class ID, int RY, int CH, int NT, bool FINAL, int PM, int CLOSED = 0, bool FULL = false, bool CYC = false>
__global__ void __launch_bounds__(NT, (dense_waves<ID, RY, NT, FINAL>()))
jfa_pass_dense(Frame f, uint32_t kArg, uint32_t ka, const char* __restrict__ in, const char* __restrict__ inB, char* __restrict__ out, char* __restrict__ outB,
               const char* __restrict__ none_row, const uint32_t* __restrict__ words, float fill, float* __restrict__ sdf,
               uint32_t tilesY, uint32_t tiles, uint32_t splitTiles, uint32_t cyc)
{
    static_assert(!CYC || !FINAL, "the last pass (k = 1) is never a multiple of the rank count");
    using T = typename ID::T;
    constexpr bool CPT = std::is_same<ID, IdC>::value;             // word plane + byte plane
    constexpr uint32_t IDB = 4u;                                   // bytes per voxel in the (word) plane the offsets below refer to
    constexpr int TAB = ID::kTab;
    constexpr int PXT = CPT ? TAB : TAB + 1;                       // 32-bit ids: slot TAB = the x index of "none" = +inf
    constexpr int TABZ = CPT ? IdC::kTabZ : TAB;                   // compact ids: "none" indexes the z table beyond its 2048 real slots: z position +inf
    constexpr int HY = (CLOSED & 1) ? 0 : 1;                       // halo rows on each side of the tile's output rows
    constexpr bool CZ = (CLOSED & 2) != 0;                         // no halo planes
    constexpr int NR = RY + 2 * HY;
    static_assert(!CLOSED || !FINAL, "closed tiles: id passes");
    constexpr int NC = PM ? 2 : 3;                                 // id columns a lane loads per source row
    constexpr int NI = NR * NC;
    constexpr int CHT = CPT ? 1 : CH;                              // z tables: squared differences per output plane / one table of positions
                                                                   // (one table for the fused last pass at n = 1024 too: +4 .. +7 %, profiles/r05/ab_zpos_last_1024.txt)
    using B = typename std::conditional<FINAL, float, double>::type;
    __shared__ float PX[PXT];
    __shared__ float TY[RY][TAB];
    __shared__ float TZ[CHT][TABZ];
    __shared__ uint32_t RB[CPT ? (CH + 2) * NR : 1];               // compact ids: row number (relative to `in`) of every source row of the tile
    constexpr bool GM = final_mask_global<ID>();
    __shared__ uint32_t WM[(FINAL && !GM) ? RY * CH * (TAB / 32) : 1];

    const int N = (int)f.n;
    const uint32_t k = FINAL ? 1u : kArg;                          // the fused last pass IS the pass with k = 1 (launch_dense refuses anything else)
    const int K = (int)k;
    // step between chain members and number of planes of the grid, in FRAME planes (CYC: those of the rank's share)
    const int zsh = CYC ? (int)(cyc & 31u) : 0, zoff = CYC ? (int)(cyc >> 8) : 0;
    const int KZ = CYC ? K >> zsh : K, NZ = CYC ? N >> zsh : N;
    auto zglobal = [&](int l) { return CYC ? zoff + (l << zsh) : l; };
    const int nzl = (int)(f.z1 - f.z0);
    const uint32_t tid = threadIdx.x;
    const int nresY = min(K, N);
    const bool rev = ((31 - __builtin_clz(k)) & 1) != 0;          // alternate the traversal direction between passes
    // Units in dispatch order: whole tiles first, then the last `splitTiles` tiles as two half-row units each (x halves), so
    // that what the chip runs while it drains is made of short units (see launch_dense).
    uint32_t lin = blockIdx.x;
    const uint32_t total = tiles;
    uint32_t xpart = 0, xparts = 1;
    if (lin >= total - splitTiles) {
        const uint32_t u = lin - (total - splitTiles);
        lin = total - splitTiles + (u >> 1); xpart = u & 1u; xparts = 2;
    }
    // Workgroups are dealt to the 8 XCDs by dispatch index mod 8, each with its own L2.  Tiles that share halo ROWS are k apart in the tile sequence (y residue
    // fastest): for k >= 8 they land on one XCD (one L2) anyway, for k = 4, 2, 1 they do not -- counters (profiles/r04/
    // pmc_bytes_*.txt, n = 1024): reads 2.8 x the id volume at k = 4 / 2 against 1.6 x at k >= 8.  For k < 8 each XCD therefore walks one
    // contiguous eighth of the whole-tile part of the sequence: reads at k = 4 / 2 fall to 1.57 x, the fused last pass 1.73 -> 1.36 x;
    // time -0.6 % / -1.8 % (the passes are not traffic-bound).  For every k the map LOSES at k >= 8: 2.3 x instead of 1.6 x --
    // neighbours in z are then a whole plane of tiles apart in time (profiles/r04/pmc_bytes_xcd_map.txt).
    {
        const uint32_t whole8 = (total - splitTiles) & ~7u;        // dispatch indices below total - splitTiles are whole tiles, index = tile
        if (K < 8 && lin < whole8) lin = (lin & 7u) * (whole8 >> 3) + (lin >> 3);
    }
    if (rev) lin = total - 1u - lin;
    const int nres = min(KZ, nzl);
    // Tile order: y residue fastest -- consecutive tiles are adjacent rows of the volume and share nothing.
    uint32_t bx, by, bxq, bxr, byq, byr;
    udivmod(lin, tilesY, by, bx); udivmod(bx, (uint32_t)nresY, bxq, bxr);
    const int ybase = (int)bxr + (int)bxq * RY * K;
    udivmod(by, (uint32_t)nres, byq, byr);
    const int lbase = (int)byr + (int)byq * CH * KZ;
    if (ybase >= N || lbase >= nzl) return;
    const int zbase = lbase + (int)f.z0;
    const int KA = (int)ka;                                        // planes between two chain members inside the window (see the header)
    float py[RY], pz[CH];
#pragma unroll
    for (int j = 0; j < RY; ++j) py[j] = axis_pos(f.oy, ybase + j * K, f.vs);
#pragma unroll
    for (int j = 0; j < CH; ++j) pz[j] = axis_pos(f.oz, zglobal(zbase + j * KZ), f.vs);
    for (uint32_t i = tid; i < (uint32_t)PXT; i += NT) PX[i] = i < (uint32_t)N ? axis_pos(f.ox, i, f.vs) : INFINITY;
    for (uint32_t i = tid; i < (uint32_t)TAB; i += NT) {
        if (i < (uint32_t)N) {
            const uint32_t si = scr(i);
            const float sy = axis_pos(f.oy, i, f.vs), sz = axis_pos(f.oz, i, f.vs);
#pragma unroll
            for (int j = 0; j < RY; ++j) { const float d = sy - py[j]; TY[j][si] = d * d; }
            if (CPT) TZ[0][si] = sz;
            else {
#pragma unroll
                for (int j = 0; j < CHT; ++j) { const float d = sz - pz[j]; TZ[j][si] = d * d; }
            }
        } else {                                                   // slots no real id refers to ("none" does: TAB - 1); finite: inf + it = inf
#pragma unroll
            for (int j = 0; j < RY; ++j) TY[j][i] = 0.0f;
#pragma unroll
            for (int j = 0; j < CHT; ++j) TZ[j][i] = 0.0f;
        }
    }
    if constexpr (TABZ > TAB)
        for (uint32_t i = (uint32_t)TAB + tid; i < (uint32_t)TABZ; i += NT) TZ[0][i] = INFINITY;
    if (CPT) {
        // row numbers of the tile's source rows: entry (pj + 1) * NR + rr = chain member pj (global plane zbase + pj k, window plane
        // lbase + pj ka from `in`), row ybase + (rr - HY) k (0 where outside)
        for (uint32_t i = tid; i < (uint32_t)((CH + 2) * NR); i += NT) {
            const int pj = (int)(i / NR) - 1, zg = zbase + pj * KZ, yy = ybase + ((int)(i % NR) - HY) * K;
            const bool ok = zg >= (int)f.z0 - KZ && zg < (int)f.z1 + KZ && zg >= 0 && zg < NZ && yy >= 0 && yy < N;
            RB[i] = ok ? (uint32_t)((lbase + pj * KA) * N + yy) : 0u;  // negative for the planes below the frame: read back as int
        }
    }
    if (FINAL && !GM) {
        for (uint32_t i = tid; i < (uint32_t)(RY * CH) * f.w; i += NT) {
            uint32_t ou, c;
            udivmod(i, f.w, ou, c);                                // (the divisor is uniform; word indices stay below 2^28: 32-bit arithmetic)
            const int o = (int)ou, a = o / CH, j = o % CH;
            const int oy = ybase + a * K, oz = zbase + j * K;
            WM[o * (TAB / 32) + (int)c] = (oy < N && oz < (int)f.z1) ? words[(uint32_t)((oz - (int)f.z0) * N + oy) * f.w + c] : 0u;
        }
    }
    __syncthreads();

    const char* tx = reinterpret_cast<const char*>(PX);
    const char* ty = reinterpret_cast<const char*>(TY);
    const char* tz = reinterpret_cast<const char*>(TZ);
    const uint32_t rowBytes = (uint32_t)N * IDB;
    const size_t planeBytes = (size_t)N * rowBytes;
    // compact ids: `inB` / `outB` = byte plane z0 of the source / output window; the byte row of "none" follows its word row
    const char* noneB = none_row + (size_t)TAB * 4u;
    int yout = 1, nout = 1;
    if constexpr (FULL) { yout = RY; nout = CH; }
    else {
#pragma unroll
        for (int j = 1; j < RY; ++j) yout += (ybase + j * K < N) ? 1 : 0;
#pragma unroll
        for (int j = 1; j < CH; ++j) nout += (zbase + j * KZ < (int)f.z1) ? 1 : 0;
    }
    const uint32_t kb = k * IDB;
    // 32-bit ids: ranks and the gather are relative to the first source plane of the tile that lies in the grid (chain member plo
    // below): at most the whole volume, 4 GiB at n = 1024, so byte offset + 1 <= 2^32 - 3 fits the low word.  Compact ids: see the header.

    const int zbase0 = zbase, lbase0 = lbase, ybase0 = ybase;
    const uint32_t xiters = ((uint32_t)N + NT - 1) / NT, xper = (xiters + xparts - 1) / xparts * NT;
    const uint32_t xbeg = xpart * xper, xend = min((uint32_t)N, xbeg + xper);
    for (uint32_t xb = xbeg; xb < xend; xb += NT) {
        uint32_t x = xb + tid;
        bool upper = false;                                        // PM: this lane is the x + k end of its pair
        if constexpr (PM != 0) {
            // pair index -> x: the pairs of a chain block of 2k voxels are (x0, x0 + k), x0 = block * 2k + (pair % k)
            uint32_t pl;
            if (PM == 1) { upper = tid & 1u; pl = tid >> 1; }
            else if (PM == 2) { upper = (tid >> 1) & 1u; pl = ((tid >> 2) << 1) | (tid & 1u); }
            else if (PM == 4) { const uint32_t l8 = tid & 7u; upper = l8 >> 2; pl = ((tid >> 3) << 2) | (upper ? 7u - l8 : l8); }
            else { upper = (tid >> 3) & 1u; pl = ((tid >> 4) << 3) | (tid & 7u); }
            const uint32_t pr = (xb >> 1) + pl;
            x = (((pr & ~(k - 1u)) << 1) | (pr & (k - 1u))) + (upper ? k : 0u);
        } else if (x >= xend) break;
        // The uniform bases are re-read through an empty asm in every x iteration: otherwise every address of the ~10
        // planes of the tile is hoisted out of the x loop, does not fit the SGPR file and is spilled to VGPR lanes
        // (v_readlane / v_writelane were 6 % of the VALU instructions of the loop).
        const int zbase = (int)opaque_uniform((size_t)(uint32_t)zbase0), lbase = (int)opaque_uniform((size_t)(uint32_t)lbase0);
        const int ybase = (int)opaque_uniform((size_t)(uint32_t)ybase0);
        const int plo = zbase - KZ >= 0 ? -1 : 0;
        const char* gbase = in + (ptrdiff_t)(lbase + plo * KA) * (ptrdiff_t)planeBytes;
        uint32_t ro[NR];
        bool yv[NR];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int ny = ybase + (rr - HY) * K;
            yv[rr] = ny >= 0 && ny < N && max(rr - HY - 1, 0) < yout;
            ro[rr] = (uint32_t)(yv[rr] ? ny : 0) * rowBytes;
        }
        const float px = PX[x];
        const bool hasM = x >= k, hasP = x + k < (uint32_t)N;
        const uint32_t xo = x * IDB, xmo = hasM ? xo - kb : xo, xpo = hasP ? xo + kb : xo;   // a column outside the grid reads the centre column
        // PM: the partner's column (always inside the grid) and the column beyond the own one
        const uint32_t xpart = upper ? xo - kb : xo + kb, xout = upper ? xpo : xmo;
        const uint32_t xoB = xo >> 2, xmoB = xmo >> 2, xpoB = xpo >> 2, xoutB = xout >> 2;      // compact ids: the same columns in the byte plane

        // Chain member P of the tile as a source plane: its base address (wave-uniform, computed once per plane, right where the plane
        // is first used) and whether it exists at all (inside the grid, and needed by an output plane of this tile that exists).  Rows
        // and planes that do not exist read a row of "none": the voxel loop is branch-free.
        struct Plane { const char* base; const char* baseB; bool ok; };
        auto plane_of = [&](int P, bool needed) {
            Plane pl;
            const int zg = zbase + P * KZ;
            pl.ok = needed && zg >= 0 && zg < NZ;
            const ptrdiff_t wp = pl.ok ? (ptrdiff_t)(lbase + P * KA) : 0;       // window plane relative to `in`
            pl.base = opaque_uniform(in + wp * (ptrdiff_t)planeBytes);
            pl.baseB = CPT ? opaque_uniform(inB + wp * (ptrdiff_t)((size_t)N * N)) : nullptr;
            return pl;
        };
        // ids of row rr of a source plane -> w[rr*3 ..]: columns {x-k, x, x+k}; "none" where outside the grid or not needed
        auto load_row = [&](const Plane& pl, int rr, T (&w)[NI]) {
            const bool ok = pl.ok && yv[rr];
            const __amdgpu_buffer_rsrc_t b = row_resource(ok ? pl.base + ro[rr] : none_row, rowBytes);
            if constexpr (CPT) {
                // word and byte of the three columns (a row of the byte plane is a quarter of the word row)
                const __amdgpu_buffer_rsrc_t bb = row_resource(ok ? pl.baseB + (ro[rr] >> 2) : noneB, (uint32_t)N);
                if constexpr (PM != 0) {
                    uint32_t w0, w1;
                    row_load(w0, b, xo); row_load(w1, b, xout);
                    w[rr * 2 + 0] = make_uint2(w0, row_load_u8(bb, xoB));
                    w[rr * 2 + 1] = make_uint2(w1, row_load_u8(bb, xoutB));
                } else {
                    uint32_t w0, w1, w2;
                    row_load(w0, b, xmo); row_load(w1, b, xo); row_load(w2, b, xpo);
                    w[rr * 3 + 0] = make_uint2(w0, row_load_u8(bb, xmoB));
                    w[rr * 3 + 1] = make_uint2(w1, row_load_u8(bb, xoB));
                    w[rr * 3 + 2] = make_uint2(w2, row_load_u8(bb, xpoB));
                }
            } else if constexpr (PM != 0) {
                row_load(w[rr * 2 + 0], b, xo);
                row_load(w[rr * 2 + 1], b, xout);
            } else {
                row_load(w[rr * 3 + 0], b, xmo);
                row_load(w[rr * 3 + 1], b, xo);
                row_load(w[rr * 3 + 2], b, xpo);
            }
        };

        B best[RY][CH];

        // What an id turns into before its candidate steps: seed x, squared y differences to the output rows it serves, squared
        // z differences to the output planes -- one LDS lookup each.
        struct Dec { float sx; float dy2[RY]; float dz2[CH]; uint32_t zt; };     // zt (compact ids): top z bit << 1, part of the rank
        auto lookup = [&](int P, int rr, T id, Dec& d) {
            const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1), olo = max(P - 1, 0), ohi = min(P + 1, CH - 1);
            d.sx = lds_f32(tx + ID::xoff(id));                              // 32-bit ids: "none" reads slot TAB = +inf
            if constexpr (CPT) d.zt = ID::zt2(id); else d.zt = 0u;
            const uint32_t yo = ID::yoff(id), zo = ID::zoff(id);
            // (table entries of 8 or 16 bytes -- one ds_read_b64 / b128 per id for several rows -- were measured: 40 % of the LDS cycles became
            // bank conflicts, +6 .. +16 % time, profiles/r02/ab*.txt)
#pragma unroll
            for (int a = alo; a <= ahi; ++a) d.dy2[a] = lds_f32(ty + a * (TAB * 4) + yo);
            if constexpr (CPT) {
                const float sz = lds_f32(tz + zo);                          // seed z position; the squares per output plane are formed here
#pragma unroll
                for (int o = olo; o <= ohi; ++o) { const float dzv = sz - pz[o]; d.dz2[o] = dzv * dzv; }
            } else {
#pragma unroll
                for (int o = olo; o <= ohi; ++o) d.dz2[o] = lds_f32(tz + o * (TAB * 4) + zo);
            }
        };
        float hold[3][3];                                                   // FINAL: distances of column x - k of the current source row
        // c = position of the candidate column in the sequence of a source row (0, 1, 2: the FINAL form pairs the first two in one
        // v_min3_f32); ownCol = it is the lane's own column (rank 0 for the own voxel); coloff = byte offset of the column in its row
        // (the rank); DPPV = the decoded values are the PARTNER's (pair mode): they are read through the DPP permutation
        auto steps_col = [&](int P, int rr, int c, bool ownCol, uint32_t coloff, const Dec& d, uint32_t prank, auto dppv) {
            constexpr bool DPPV = decltype(dppv)::value;
            auto val = [&](float v) { if constexpr (DPPV) return from_partner<PM ? PM : 1>(v); else return v; };
            auto valu = [&](uint32_t v) { if constexpr (DPPV) return __float_as_uint(from_partner<PM ? PM : 1>(__uint_as_float(v))); else return v; };
            const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1), olo = max(P - 1, 0), ohi = min(P + 1, CH - 1);
            const float dxv = val(d.sx) - px;
            const float dx2 = dxv * dxv;
            u32x2 cand;
            if (!FINAL) {
                if constexpr (CPT) cand.x = (uint32_t)(((P + 1) * NR + rr + 1) << 14) + 1u + coloff + valu(d.zt);
                else cand.x = prank + ro[rr] + coloff;
            }
#pragma unroll
            for (int a = alo; a <= ahi; ++a) {
                const float pre = val(d.dy2[a]) + dx2;
                const bool ownRow = (rr == a + HY) && ownCol;
#pragma unroll
                for (int o = olo; o <= ohi; ++o) {
                    const float dd = val(d.dz2[o]) + pre;
                    if constexpr (FINAL) {
                        // distances only, so the order of the candidates no longer matters: the left column's distance waits for
                        // the centre column's and both go through one v_min3_f32 (27 -> 18 minimum instructions per voxel)
                        if (c == 0) hold[a - alo][o - olo] = dd;
                        else if (c == 1) best[a][o] = min3_f32(best[a][o], hold[a - alo][o - olo], dd);
                        else best[a][o] = min_f32(best[a][o], dd);
                    } else {
                        u32x2 cd = cand;
                        if (ownRow && o == P) cd.x = CPT ? d.zt : 0u;      // the voxel's own state wins every tie (sequential.cpp:84,106)
                        cd.y = __float_as_uint(dd);
                        best[a][o] = min_f64(best[a][o], __builtin_bit_cast(double, cd));
                    }
                }
            }
        };
        // the candidate steps that follow the decode of id j of a plane (j = rr * NC + column slot)
        auto steps = [&](int P, int rr, int c, const Dec& d, uint32_t prank) {
            if constexpr (PM != 0) {
                if (c == 0) {                                                   // own column, then the partner's evaluation of ITS own column
                    steps_col(P, rr, 0, true, xo, d, prank, std::false_type{});
                    steps_col(P, rr, 1, false, xpart, d, prank, std::true_type{});
                } else {
                    steps_col(P, rr, 2, false, xout, d, prank, std::false_type{});
                }
            } else {
                steps_col(P, rr, c, c == 1, c == 0 ? xmo : c == 1 ? xo : xpo, d, prank, std::false_type{});
            }
        };
        // Software pipeline over the ids of a plane (2-KB tables, and the fused last pass with the 4-KB ones; tools/ab_step.py: -1.3 % at
        // n = 512, +0.8 % for the id passes at n = 1024, which keep the plain loop, as the compact ids do).
        constexpr bool PIPE = ID::kTab == 512 || (FINAL && !CPT);
        auto scatter = [&](int P, T (&w)[NI]) {
            // rank of the ids of this plane: byte offset of the row inside the gather window + 1 (wave-uniform) + the column offset
            // (compact ids: the row part is the unrolled constant of steps_col)
            const uint32_t prank = CPT ? 0u : (uint32_t)((P - plo) * KA) * (uint32_t)planeBytes + 1u;
            const int olo = max(P - 1, 0), ohi = min(P + 1, CH - 1);
            Plane next{nullptr, nullptr, false};
            const bool roll = P + 1 <= CH - (CZ ? 1 : 0);                  // there is a next source plane to prefetch
            if (roll) next = plane_of(P + 1, P + 1 <= nout);
            if constexpr (PIPE) {
                // the table lookups of id j + 1 are issued BEFORE the candidate steps of id j (the scheduling barriers keep the compiler
                // from sinking them back to their first use), so a wave waits for LDS data a whole id of VALU work after asking for it.
                // DEPTH = ids looked up ahead of the one being evaluated.  Two ahead (profiles/r03/ab_pipe_*.txt): fused last pass at
                // n = 1024 3.23 -> 3.05 ms (-6 %), at n = 512 +-0; dense passes +1 % (n = 512) and +24 % (n = 1024: 7 more VGPRs cost a
                // workgroup per CU)
                constexpr int DEPTH = (FINAL && ID::kTab == 1024) ? 2 : 1;
                Dec d[DEPTH + 1];
#pragma unroll
                for (int j = 0; j < DEPTH && j < NI; ++j) lookup(P, j / NC, w[j], d[j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int rr = j / NC, c = j % NC;
                    if (j + DEPTH < NI) lookup(P, (j + DEPTH) / NC, w[j + DEPTH], d[(j + DEPTH) % (DEPTH + 1)]);
                    __builtin_amdgcn_sched_barrier(0);
                    steps(P, rr, c, d[j % (DEPTH + 1)], prank);
                    if (c == NC - 1) {
                        const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1);
#pragma unroll
                        for (int a = alo; a <= ahi; ++a)
#pragma unroll
                            for (int o = olo; o <= ohi; ++o) pin(best[a][o]);
                        if (roll) load_row(next, rr, w);                   // rolling prefetch (see the header)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int alo = max(rr - HY - 1, 0), ahi = min(rr - HY + 1, RY - 1);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    Dec d;
                    lookup(P, rr, w[rr * NC + c], d);
                    steps(P, rr, c, d, prank);
                }
#pragma unroll
                for (int a = alo; a <= ahi; ++a)
#pragma unroll
                    for (int o = olo; o <= ohi; ++o) pin(best[a][o]);
                // the ids of this row are spent: the same row of the NEXT source plane is requested into their registers right away.  A
                // plane that is not needed reads "none" (never memory outside the window).
                if (roll) load_row(next, rr, w);
            }
        };

        T w[NI];
        T pend[RY];                                                // gathered winners of the previous output plane, stored a plane later
        {
            const Plane first = plane_of(CZ ? 0 : -1, true);         // closed tiles start at their own first plane
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) load_row(first, rr, w);
        }
#pragma clang loop unroll(full)
        for (int P = -1; P <= CH; ++P) {
            if (P + 1 < CH) {
#pragma unroll
                for (int a = 0; a < RY; ++a) {
                    if constexpr (FINAL) best[a][P + 1] = INFINITY;
                    else best[a][P + 1] = __builtin_bit_cast(double, (u32x2){0xFFFFFFFFu, 0x7F800000u});   // (+inf, last rank)
                }
            }
            uint32_t mw[RY] = {};
            if constexpr (FINAL && GM) {         // bitmask words of the rows stored after this plane: requested before its evaluation
                if (P >= 1 && P - 1 < nout) {
                    // one resource per output plane (its rows from ybase on), the row as the scalar offset of the load: no address arithmetic per row
                    const char* wpl = opaque_uniform(reinterpret_cast<const char*>(words) + ((size_t)(lbase + (P - 1) * K) * N + ybase) * (size_t)(f.w * 4u));
                    const __amdgpu_buffer_rsrc_t wr = row_resource(wpl, (uint32_t)(RY * K) * f.w * 4u);
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        mw[a] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wr, (int)((x >> 5) << 2), (int)((uint32_t)(a * K) * f.w * 4u), 0);
                    }
                }
            }
            // The planes before the first and after the last output plane may lie outside the grid (one of them does for every tile at
            // k = n/16, for half the tiles at n/32, ...): then they hold nothing but "none" and could be skipped with a workgroup-uniform
            // branch.  Measured: a branch around EVERY plane and around the first / last row +1.7 % (and spills with the rows), a branch at
            // the two ends of the chain only -0.3 % / +0.3 % (profiles/r03/ab_edge_*.txt, ab_endskip_*.txt) -- removed: what pays is
            // dropping the halo at compile time, which only the closed tiles of k = n/8 can.
            if (!(CZ && (P == -1 || P == CH))) scatter(P, w);        // closed tiles have no plane before the first or after the last
            if constexpr (FINAL) {
                if (P >= 1 && P - 1 < nout) {
                    // one resource per output plane, the row as the scalar offset of the store (with the mask loads above: 2,556 -> 1,560 scalar instructions
                    // per tile, no 64-bit vector address; fused last pass -1.8 % / -5.6 % / -1.9 % at n = 512 / 1024 / 2048, profiles/r05/ab_last_salu.txt)
                    const char* spl = opaque_uniform(reinterpret_cast<const char*>(sdf) + ((size_t)(lbase + (P - 1) * K) * N + ybase) * (size_t)rowBytes);
                    const __amdgpu_buffer_rsrc_t sr = row_resource(spl, (uint32_t)((RY - 1) * K + 1) * rowBytes);
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        const bool set = ((GM ? mw[a] : WM[(a * CH + (P - 1)) * (TAB / 32) + (x >> 5)]) >> (x & 31)) & 1u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(set ? best[a][P - 1] : copysignf(best[a][P - 1], fill)), sr, (int)xo,
                                                              (int)((uint32_t)(a * K) * rowBytes), kStoreAux);
                    }
                }
            } else {
                if (P >= 2 && P - 2 < nout) {                      // ids gathered during the previous plane
                    const char* orow = opaque_uniform(out + ((size_t)(lbase + (P - 2) * KA) * N + ybase) * rowBytes);
                    const char* orowB = CPT ? opaque_uniform(outB + ((size_t)(lbase + (P - 2) * KA) * N + ybase) * (size_t)N) : nullptr;
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        if constexpr (CPT) {
                            row_store<kStoreAux>(pend[a].x, row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                            row_store_u8<kStoreAux>(pend[a].y, row_resource(orowB + (size_t)(a * K) * N, (uint32_t)N), xoB);
                        } else {
                            row_store<kStoreAux>(pend[a], row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                        }
                    }
                }
                if (P >= 1 && P - 1 < nout) {                      // output plane P - 1 is complete: fetch the ids of its winners
                    const uint32_t orank = (uint32_t)((P - 1 - plo) * KA) * (uint32_t)planeBytes;
#pragma unroll
                    for (int a = 0; a < RY; ++a) {
                        if (a >= yout) continue;
                        const uint32_t lo = __builtin_bit_cast(u32x2, best[a][P - 1]).x;
                        if constexpr (CPT) {
                            // rank = (source row index + 1) << 14 | byte offset in the word row, + 1, + top z bit << 1; the own voxel (rank < 4) sits
                            // in row (P, a + HY) of the tile's source rows.  The byte of the output: top z bit from the rank, "none" iff nothing won.
                            const uint32_t zt = lo & 2u;
                            const uint32_t r = lo < 4u ? (uint32_t)((P * NR + a + HY + 1) << 14) + xo : lo - 1u - zt;
                            const ptrdiff_t row = (ptrdiff_t)(int)RB[(r >> 14) - 1u];
                            const uint32_t none1 = __builtin_bit_cast(u32x2, best[a][P - 1]).y == 0x7F800000u ? IdC::kNoneBit : 0u;
                            pend[a] = make_uint2(*reinterpret_cast<const uint32_t*>(in + row * (ptrdiff_t)rowBytes + (ptrdiff_t)(r & 16383u)),
                                                 zt | none1);
                        } else {
                            const uint32_t ownOff = orank + ro[a + HY] + xo;
                            const uint32_t off = lo ? lo - 1u : ownOff;
                            pend[a] = *reinterpret_cast<const T*>(gbase + off);
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < RY; ++a)
#pragma unroll
                for (int o = P; o <= P + 1; ++o)
                    if (o >= 0 && o < CH) pin(best[a][o]);
        }
        if constexpr (!FINAL) {
            if (CH - 1 < nout) {
                const char* orow = opaque_uniform(out + ((size_t)(lbase + (CH - 1) * KA) * N + ybase) * rowBytes);
                const char* orowB = CPT ? opaque_uniform(outB + ((size_t)(lbase + (CH - 1) * KA) * N + ybase) * (size_t)N) : nullptr;
#pragma unroll
                for (int a = 0; a < RY; ++a) {
                    if (a >= yout) continue;
                    if constexpr (CPT) {
                        row_store<kStoreAux>(pend[a].x, row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                        row_store_u8<kStoreAux>(pend[a].y, row_resource(orowB + (size_t)(a * K) * N, (uint32_t)N), xoB);
                    } else {
                        row_store<kStoreAux>(pend[a], row_resource(orow + (size_t)(a * K) * rowBytes, rowBytes), xo);
                    }
                }
            }
        }
    }
}

}  // namespace

// Tail of a dense launch.  A launch of T tiles on S = CUs x workgroups-per-CU slots runs ~T/S rounds; at n = 512 that is 5.3:
// while the chip drains, slots stand empty for about half a tile time (77 us of a 410-us pass).  The last ~4/3 S tiles are
// therefore dispatched as two half-row units each (x halves, one table prologue more per split tile): -1.9 % on the dense passes
// at n = 512 (profiles/r02/ab17.txt, ab18.txt).  Launches of 16 rounds and more (n = 1024: 43) are left whole (measured +-0).
static uint32_t tail_split(const vp_ctx* ctx, uint32_t tiles, uint32_t wgPerCu)
{
    const uint32_t slots = (uint32_t)ctx->cus * wgPerCu;
    if (tiles >= 16u * slots) return 0;
    return std::min(tiles / 2u, slots * 4u / 3u);
}

struct DenseArgs {
    vp_ctx* ctx; Frame f; uint32_t k, ka;
    const char *in, *inB; char *out, *outB; const char* none_row;
    const uint32_t* words; float fill; float* sdf;
    uint32_t nresY, ylen, nres, zlen;                              // residue classes and chain lengths of the rows / of the frame's planes
    uint32_t cyc = 0;                                              // cyclic plane distribution: log2(ranks) | rank << 8 (see jfa_pass_dense, CYC)
};
// the 8 x 16 tiles of the 2-KB-table format at n = 512: a build part of their own (the largest kernels of the file)
int launch_dense_id9_tile16(const DenseArgs& a, bool pairsOk);

namespace {

template <class ID, int RY, int CH, int NT, bool F, int PM, int CL = 0, bool FULL = false, bool CYC = false>
void launch_tile(const DenseArgs& a)
{
    const uint32_t ty = a.nresY * ((a.ylen + RY - 1) / RY), t = ty * a.nres * ((a.zlen + CH - 1) / CH);
    // workgroups a CU holds (what dense_waves and the LDS footprint allow): the unit of the tail split
    constexpr uint32_t perCu = std::is_same<ID, IdC>::value ? 2u
                             : ID::kTab == 512 ? (RY == 8 ? 4u : (F && !final_mask_global<ID>()) ? 5u : 6u)
                             : NT == 512 ? (RY == 8 ? 2u : 3u) : 4u;
    const uint32_t sp = a.f.n > (uint32_t)NT ? tail_split(a.ctx, t, perCu) : 0u;     // a row of <= NT voxels has no halves
    hipLaunchKernelGGL((jfa_pass_dense<ID, RY, CH, NT, F, PM, CL, FULL, CYC>), dim3(t + sp), dim3(NT), 0, a.ctx->stream, a.f, a.k, a.ka,
                       a.in, a.inB, a.out, a.outB, a.none_row, a.words, a.fill, a.sdf, ty, t, sp, a.cyc);
}

// Tile shape and lane pairing for one launch.  RY: 8 rows for the id passes with the 2-KB tables (109 VGPRs, four waves per SIMD, but half
// the table builds and 3.1 instead of 3.75 decoded ids per voxel: -1.1 %; with the 4-KB tables +-0 .. +1.1 %, the fused last pass +1 .. +5 %:
// those keep 4 rows, profiles/r03/ab_ry8_*.txt, r04/ab_gnt_ry8_1024.txt).  FULL (compile-time row / plane counts where every tile is whole):
// dense -3.2 % at n = 512, -5 % at n = 2048, fused last pass -1 / -4 / -3 %; the id passes with the 4-KB tables lose 4 % (80 VGPRs + 8
// spilled instead of 71 under the six-wave bound) and keep the run-time counts (profiles/r04/ab_full_*.txt).  Pair mode: 8-plane tiles of
// power-of-two grids whose rows are whole NT-thread iterations; the lane permutation follows k; k = 2 with the 2-KB tables keeps the plain
// form (7 % faster there: 0.370 vs 0.396 ms; with the 4-KB tables pairs win by 3 %).
template <int V> using int_c = std::integral_constant<int, V>;

// ALWAYS_FULL: the caller only comes with whole chains (no instantiation of the run-time row / plane counts).
// CYC: the passes of a cyclic plane distribution -- their chains are always whole (n / k is a power of two >= 8, see launch_dense), so each
// format instantiates ONE of the two forms: compile-time counts where the format uses them at all, run-time counts otherwise.
template <class ID, int CH, int NT, bool F, bool ALWAYS_FULL = false, bool CYC = false>
void launch_shape(const DenseArgs& a, bool pairsOk, bool wholeChains)
{
    constexpr int RY = (ID::kTab == 512 && !F && CH >= 8) ? 8 : 4;
    constexpr bool canFull = CH >= 8 && (ID::kTab != 1024 || F);
    auto go = [&](auto pm) {
        constexpr int PM = decltype(pm)::value;
        if constexpr (ALWAYS_FULL || (CYC && canFull)) launch_tile<ID, RY, CH, NT, F, PM, 0, true, CYC>(a);
        else {                                                     // (an else branch: the form not taken is not instantiated at all)
            if constexpr (canFull) {
                if (wholeChains && a.ylen % RY == 0) { launch_tile<ID, RY, CH, NT, F, PM, 0, true>(a); return; }
            }
            launch_tile<ID, RY, CH, NT, F, PM, 0, false, CYC>(a);
        }
    };
    if constexpr (CH >= 8) {
        if (pairsOk && a.f.n % NT == 0) {
            if constexpr (F) { go(int_c<1>{}); return; }
            else {
                if constexpr (ID::kTab != 512) { if (a.k == 2) { go(int_c<2>{}); return; } }
                if (a.k == 4) { go(int_c<4>{}); return; }
                if (a.k >= 8) { go(int_c<8>{}); return; }
            }
        }
    }
    go(int_c<0>{});
}

// FIN: the last pass (fused with the id -> sdf conversion); a template parameter so that the id passes and the last pass of a format are
// separate build parts.
template <class ID, bool FIN>
int launch_dense(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t stride,
                 const uint32_t* d_words, float fill, float* d_sdf)
{
    const uint32_t nz = f.z1 - f.z0, n = f.n;
    constexpr bool fin = FIN;
    const uint32_t nres = std::min(k, nz), zlen = (nz + k - 1) / k;
    const uint32_t nresY = std::min(k, n), ylen = (n + k - 1) / k;
    // Plane chains that are not a multiple of eight (the regions of the multi-GPU pipelines: 288 planes at k = 32 are chains of nine).  A
    // tile of CH planes walks CH + 2 plane iterations whatever it outputs, so 4-plane tiles cost 6 per 4 planes.  The first 8 q members of
    // every chain are the CONSECUTIVE planes [z0, z0 + 8 q k): they go to the 8-plane form as a frame of their own (each part sees the
    // other as its halo inside the window), the remaining r < 8 members as a second launch with whichever tile is cheaper for r (10
    // iterations for one 8-tile, 6 per 4-tile, 4 per 2-tile).  Measured per rank on one GPU: DESIGN.md, multi-GPU section.
    if (stride == k && nz % k == 0 && k < nz && zlen > 8 && zlen % 8 != 0) {
        const uint32_t planesA = (zlen / 8u) * 8u * k;
        Frame fa = f, fb = f;
        fa.z1 = f.z0 + planesA; fb.z0 = fa.z1;
        IdWin inB = in, outB = out;
        inB.at += planesA; outB.at += planesA;
        const size_t wordPlane = (size_t)n * f.w, sdfPlane = (size_t)n * n;
        VP_TRY((launch_dense<ID, FIN>(ctx, fa, k, in, out, stride, d_words, fill, d_sdf)));
        return launch_dense<ID, FIN>(ctx, fb, k, inB, outB, stride, d_words ? d_words + planesA * wordPlane : nullptr, fill, d_sdf ? d_sdf + planesA * sdfPlane : nullptr);
    }
    if (FIN && k != 1) return set_error(VP_ERR_INVALID, "jfa_pass_dense: the fused last pass is the pass with k = 1");
    VP_TRY(ensure_none_rows(ctx));
    DenseArgs a{ctx, f, k, stride, win_words(in, n, in.at), win_compact(n) ? win_bytes_plane(in, n, in.at) : nullptr,
                win_words(out, n, out.at), win_compact(n) ? win_bytes_plane(out, n, out.at) : nullptr,
                (const char*)(std::is_same<ID, Id9>::value ? none_row_id9(ctx) : std::is_same<ID, Id10>::value ? none_row_id10(ctx) : none_row_idc(ctx)),
                d_words, fill, d_sdf, nresY, ylen, nres, zlen};
    const bool pow2 = (n & (n - 1)) == 0;
    const bool wholeChains = zlen % 8 == 0 && nz % k == 0 && n % k == 0;
    // threads per workgroup: 256 with the 2-KB tables (26 KB of LDS, six workgroups per CU); with the 4-KB tables the 4 x 8 tile takes 52 KB,
    // shared by the 8 waves of a 512-thread workgroup (three per CU), as do the 8-KB tables of the compact ids (two per CU)
    constexpr int NTD = ID::kTab == 512 ? 256 : 512;               // 8-plane tiles
    constexpr int NTS = ID::kTab == 2048 ? 512 : 256;              // 4- and 2-plane tiles
    // closed 8 x 8 tiles at k = n/8 (see jfa_pass_dense): whole power-of-two grids on 32-bit ids.  (Compact ids, n = 2048: the closed tile's
    // 86 KB of tables need 1024 threads -- 128 VGPRs + 25 spilled -- and buy 5 % of that one pass; 8-row tiles for every pass cost 20 .. 38 %:
    // profiles/r05/ab_idc_tiles_2048.txt.)
    if constexpr (!std::is_same<ID, IdC>::value) {
        if constexpr (!fin) {
            if (pow2 && stride == k && f.z0 == 0 && f.z1 == n && n == 8u * k && n % NTD == 0) {
                launch_tile<ID, 8, 8, NTD, false, 8, 3, ID::kTab == 512>(a);
                VP_HIP(hipGetLastError());
                return 0;
            }
        }
    }
    // one 8-plane tile per chain also where the chain has 5 .. 7 members (10 plane iterations against 2 x 6); chains of one or two planes
    // (the remainders of the split above, and the slabs of a pass whose step spans whole slabs): 2-plane tiles
    const bool deep = zlen % 8 == 0 || (zlen > 4 && zlen < 8);
    const bool tiny = zlen <= 2;
    // Compact ids: 16-plane tiles where the chains allow it.  The kernel streams over its planes (three output planes are live whatever CH is) and
    // this format keeps ONE table of z positions, so a longer tile costs neither registers nor LDS, only code: (CH + 2) / CH plane reads per
    // output plane fall from 1.25 to 1.125 -- n = 2048: tile passes -2.7 %, fused last pass -6.2 % (profiles/r05/ab_ch16_2048.txt).
    if constexpr (std::is_same<ID, IdC>::value) {
        if (zlen % 16 == 0 && stride == k) {
            launch_shape<ID, 16, NTD, fin>(a, pow2 && (fin || k >= 2), wholeChains);
            VP_HIP(hipGetLastError());
            return 0;
        }
    }
    // ... and the 2-KB-table format at n = 512, on 512-thread workgroups (one x iteration per row).  The fused last pass on 4 x 16 tiles (46 KB
    // of tables + mask words: three workgroups per CU, the same six waves per SIMD): 0.319 -> 0.291 ms (-8.9 %, profiles/r05/ab_id9_ch16_512.txt).
    // The id passes on 8 x 16 tiles (50 KB: two workgroups per CU = the four waves per SIMD of the 8 x 8 tiles; halo factor 1.41 instead of
    // 1.56): -1.6 % (ab_id9_8x16_512.txt; on 4 x 16 tiles +-0).  With the 4-KB tables 16 planes cost two of three workgroups per CU: +12 %
    // (ab_id10_ch16_1024.txt).
    if constexpr (ID::kTab == 512) {
        if (n == 512 && zlen % 16 == 0 && stride == k && wholeChains) {      // (n = 512 = 2^9: every chain is whole)
            if constexpr (fin) launch_shape<ID, 16, 512, true, true>(a, pow2, true);
            else VP_TRY(launch_dense_id9_tile16(a, pow2 && k >= 2));
            VP_HIP(hipGetLastError());
            return 0;
        }
    }
    if (deep) launch_shape<ID, 8, NTD, fin>(a, pow2 && (fin || k >= 2), wholeChains);
    else if constexpr (!fin) { if (tiny) launch_shape<ID, 2, NTS, false>(a, false, false); else launch_shape<ID, 4, NTS, false>(a, false, false); }
    else launch_shape<ID, 4, NTS, true>(a, false, false);
    VP_HIP(hipGetLastError());
    return 0;
}

// One pass of a CYCLIC plane distribution (jfa_pass_dense, CYC): f is the whole-grid frame, the windows hold the n / ranks planes of rank
// `rank` (frame plane l = global plane rank + l * ranks), k is a multiple of `ranks`.  The chains of such a pass are always whole: every
// step of the halving sequence down to a multiple of `ranks` divides n exactly, so a chain of rows or planes has n / k = 2^j members,
// j >= 3 (the passes n/2 and n/4 are the fused start, jfa_first_two) -- 8-plane tiles (16 for the compact ids where n / k allows), closed
// 8 x 8 tiles at k = n/8 on power-of-two grids, pair mode as on one GPU.
template <class ID>
int launch_dense_cyclic(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t ranks, uint32_t rank)
{
    const uint32_t n = f.n, zsh = (uint32_t)__builtin_ctz(ranks);
    const uint32_t nzl = n >> zsh, kz = k >> zsh;
    if ((ranks & (ranks - 1)) != 0 || rank >= ranks || n % ranks != 0 || k % ranks != 0 || n % k != 0 || n / k < 8 || ((n / k) & (n / k - 1)) != 0)
        return set_error(VP_ERR_INVALID, "jfa_pass_dense (cyclic): n = %u, k = %u, %u ranks", n, k, ranks);
    VP_TRY(ensure_none_rows(ctx));
    Frame fl = f;                                                  // the rank's share as a frame of its own: planes [0, n / ranks)
    fl.z0 = 0; fl.z1 = nzl;
    const uint32_t zlen = nzl / kz, ylen = n / k;                  // = n / k both
    DenseArgs a{ctx, fl, k, kz, win_words(in, n, in.at), win_compact(n) ? win_bytes_plane(in, n, in.at) : nullptr,
                win_words(out, n, out.at), win_compact(n) ? win_bytes_plane(out, n, out.at) : nullptr,
                (const char*)(std::is_same<ID, Id9>::value ? none_row_id9(ctx) : std::is_same<ID, Id10>::value ? none_row_id10(ctx) : none_row_idc(ctx)),
                nullptr, 0.0f, nullptr, std::min(k, n), ylen, std::min(kz, nzl), zlen, zsh | (rank << 8)};
    const bool pow2 = (n & (n - 1)) == 0;
    constexpr int NTD = ID::kTab == 512 ? 256 : 512;
    if constexpr (!std::is_same<ID, IdC>::value) {
        if (pow2 && n == 8u * k && n % NTD == 0) {
            launch_tile<ID, 8, 8, NTD, false, 8, 3, ID::kTab == 512, true>(a);
            VP_HIP(hipGetLastError());
            return 0;
        }
    }
    if constexpr (std::is_same<ID, IdC>::value) {
        if (zlen % 16 == 0) {
            launch_shape<ID, 16, NTD, false, false, true>(a, pow2 && k >= 2, true);
            VP_HIP(hipGetLastError());
            return 0;
        }
    }
    launch_shape<ID, 8, NTD, false, false, true>(a, pow2 && k >= 2, true);
    VP_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// launch_dense<ID, FIN> per id format and pass kind, one build part each (-DVP_DENSE_PART=1..10; undefined: all of them in one unit)
#ifndef VP_DENSE_PART
#define VP_DENSE_PART 0
#endif
#define VP_DENSE_ENTRY(PART, NAME, ID, FIN)                                                                                                \
    int NAME(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t stride, const uint32_t* d_words, float fill, float* d_sdf) \
    { return launch_dense<ID, FIN>(ctx, f, k, in, out, stride, d_words, fill, d_sdf); }
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 1
VP_DENSE_ENTRY(1, launch_dense_id9_pass, Id9, false)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 2
VP_DENSE_ENTRY(2, launch_dense_id9_last, Id9, true)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 3
VP_DENSE_ENTRY(3, launch_dense_id10_pass, Id10, false)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 4
VP_DENSE_ENTRY(4, launch_dense_id10_last, Id10, true)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 5
VP_DENSE_ENTRY(5, launch_dense_idc_pass, IdC, false)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 6
VP_DENSE_ENTRY(6, launch_dense_idc_last, IdC, true)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 7
int launch_dense_id9_tile16(const DenseArgs& a, bool pairsOk) { launch_shape<Id9, 16, 512, false, true>(a, pairsOk, true); return 0; }
#endif
#undef VP_DENSE_ENTRY
#define VP_CYCLIC_ENTRY(NAME, ID)                                                                                                        \
    int NAME(vp_ctx* ctx, const Frame& f, uint32_t k, const IdWin& in, const IdWin& out, uint32_t ranks, uint32_t rank)                 \
    { return launch_dense_cyclic<ID>(ctx, f, k, in, out, ranks, rank); }
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 8
VP_CYCLIC_ENTRY(launch_cyclic_id9_pass, Id9)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 9
VP_CYCLIC_ENTRY(launch_cyclic_id10_pass, Id10)
#endif
#if VP_DENSE_PART == 0 || VP_DENSE_PART == 10
VP_CYCLIC_ENTRY(launch_cyclic_idc_pass, IdC)
#endif
#undef VP_CYCLIC_ENTRY

}  // namespace vp
