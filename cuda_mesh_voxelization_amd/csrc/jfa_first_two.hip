// jfa_first_two.hip -- the passes k = n/2 and k = n/4 of a whole grid in ONE launch, straight from the border bitmask (see the kernel).
#include "jfa_common.h"

namespace vp {
namespace {

// ------------------------------------------------------------------------------------------ seed scatter: one proposal round
// The seed at (sx, sy, sz), held by the voxel of slot s of a closed 4-chain tile (slot = ((plane * 4 + row) * 4 + segment) * XR +
// residue), proposes itself to the voxels STEP chain positions away along each axis, s itself included (rank 0).
//     key = distance bits << 32 | rank << 27 | tag,   rank = 1 + scan index of s as seen from the target (sequential.cpp:84-112)
// (tag: what the winner is to be known by -- the slot s, or the slot of the seed it carries; it never decides a comparison, since two
// proposals to one target with equal rank come from the same slot)
// The minimum is commutative, so the loops run (row, column, plane): dx^2 + dy^2 is formed once per row and column.  (Spreading
// the 27 proposals of a seed over nine threads was measured: -3 % at n = 1024, +5 .. 14 % at n = 512, profiles/r02/ab34.txt.)
// An empty slot holds (+inf, rank 0, slot 0): every key with a finite distance is smaller, and a proposal whose distance is not
// finite (sequential.cpp:106 never takes such a candidate) is >= it and leaves the slot as it is -- no test needed.
constexpr unsigned long long kEmptyKey = 0x7F80000000000000ull;

// STEP = 2 (the pass with k = n/2 inside a closed 4-chain tile): along each axis a voxel has exactly two in-grid chain positions two
// steps apart -- its own and position ^ 2 -- so a seed has 8 targets, all valid: no tests, no branches (the general form below
// compiles to 27 predicated blocks of which a wave of border voxels executes every one).  Scan index of s as seen from the partner
// along an axis: partner = position - t * 2 with t = +1 if position >= 2 else -1, i.e. t + 1 = position & 2.
// Measured (profiles/r03/ab_propose_*.txt): jfa_first_two 0.339 -> 0.311 ms at n = 512, 2.56 -> 2.37 ms at n = 1024.  The same idea for
// STEP = 1 -- 27 straight-line minima, targets outside the chain neutralised with the key ~0 -- trades 4 SALU for 2 VALU per candidate
// and gave the gain back (0.338 / 2.53): the kernel is bound by vector issue of the one wave that proposes.
// The positions of a tile's chain members are read from three small LDS tables the tile fills once (computing them -- cvt, mul, add and the
// index arithmetic before them -- is ~5 instructions per position, nine positions per proposal round, in the proposing wave's instruction
// stream, which is the tile's critical path).  Chain positions outside 0..3 are only ever asked for targets that are then skipped; the tables wrap them.
template <int XR>
struct ChainPosLds {
    const float* px; const float* py; const float* pz;             // [4 * XR], [4], [4]
    __device__ __forceinline__ float x(uint32_t seg, uint32_t xr) const { return px[(seg & 3u) * XR + xr]; }
    __device__ __forceinline__ float y(uint32_t j) const { return py[j & 3u]; }
    __device__ __forceinline__ float z(uint32_t j) const { return pz[j & 3u]; }
};

template <int XR, class POS>
__device__ __forceinline__ void propose_half(unsigned long long* keys, uint32_t s, float sx, float sy, float sz, const POS& pos)
{
    const uint32_t xr = s % XR, xs = (s / XR) & 3u, jr = (s / (4u * XR)) & 3u, jp = s / (16u * XR);
    float dx2[2], dy2[2], dz2[2];
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {                             // 0: the seed's own chain position, 1: the partner's
        const float dxv = sx - pos.x(xs ^ (2u * j), xr);
        const float dyv = sy - pos.y(jr ^ (2u * j));
        const float dzv = sz - pos.z(jp ^ (2u * j));
        dx2[j] = dxv * dxv; dy2[j] = dyv * dyv; dz2[j] = dzv * dzv;
    }
    const uint32_t ra[2] = {1u, xs & 2u}, rb[2] = {3u, (jr & 2u) * 3u}, rc[2] = {9u, (jp & 2u) * 9u};      // (t + 1) * {1, 3, 9}
#pragma unroll
    for (uint32_t ib = 0; ib < 2; ++ib)
#pragma unroll
        for (uint32_t ia = 0; ia < 2; ++ia) {
            const float pre = dx2[ia] + dy2[ib];
#pragma unroll
            for (uint32_t ic = 0; ic < 2; ++ic) {
                const float d = pre + dz2[ic];
                const uint32_t rank = (ia | ib | ic) ? rc[ic] + rb[ib] + ra[ia] + 1u : 0u;
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | ((rank << 27) | s);
                __hip_atomic_fetch_min(&keys[s ^ (ia * 2u * XR) ^ (ib * 8u * XR) ^ (ic * 32u * XR)], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
}

template <int XR, int STEP, class POS>
__device__ __forceinline__ void propose(unsigned long long* keys, uint32_t s, float sx, float sy, float sz, const POS& pos, uint32_t tag)
{
    const uint32_t xr = s % XR, xs = (s / XR) & 3u, jr = (s / (4u * XR)) & 3u, jp = s / (16u * XR);
    float dx2[3], dy2[3], dz2[3];
    bool va[3], vb[3], vc[3];
#pragma unroll
    for (int t = -1; t <= 1; ++t) {                                // target = s - t STEP positions: s is its neighbour at +t
        va[t + 1] = xs - t * STEP <= 3u; vb[t + 1] = jr - t * STEP <= 3u; vc[t + 1] = jp - t * STEP <= 3u;   // unsigned: also rejects < 0
        const float dxv = sx - pos.x(xs - t * STEP, xr);
        const float dyv = sy - pos.y(jr - t * STEP);
        const float dzv = sz - pos.z(jp - t * STEP);
        dx2[t + 1] = dxv * dxv; dy2[t + 1] = dyv * dyv; dz2[t + 1] = dzv * dzv;
    }
#pragma unroll
    for (int b = -1; b <= 1; ++b) {
        if (!vb[b + 1]) continue;
#pragma unroll
        for (int a = -1; a <= 1; ++a) {
            if (!va[a + 1]) continue;
            const float pre = dx2[a + 1] + dy2[b + 1];
            // lowest plane first: the three planes are then constant non-negative offsets from one index
            const uint32_t t0 = s - (uint32_t)(b * STEP) * (4u * XR) - (uint32_t)(a * STEP) * XR - (uint32_t)STEP * (16u * XR);
#pragma unroll
            for (int c = -1; c <= 1; ++c) {
                if (!vc[c + 1]) continue;
                const float d = pre + dz2[c + 1];
                const bool own = a == 0 && b == 0 && c == 0;       // (a distance that is not finite never wins: kEmptyKey)
                const uint32_t rank = own ? 0u : (uint32_t)((c + 1) * 9 + (b + 1) * 3 + (a + 1) + 1);
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | ((rank << 27) | tag);
                __hip_atomic_fetch_min(&keys[t0 + (uint32_t)((1 - c) * STEP) * (16u * XR)], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ first two passes from the mask
// Passes k = n/2 and k = n/4 in one kernel, straight from the border bitmask.  Both steps are steps along the closed 4-chains of
// the seed scatter (+-n/2 = two chain positions, +-n/4 = one), so the state after the first pass of a tile's 64 x XR voxels depends
// on the border bits of those same voxels only: nothing but 64 x XR bits is read, the first pass never touches HBM at all, and the
// id volume is written once.  Stage A scatters the border voxels (0.7 % on the headline mesh) two positions along each axis, which
// leaves every voxel with the slot of its pass-1 seed (or none); stage B scatters the voxels that have one (5 %) one position
// along each axis with that seed's coordinates.  Keys and ranks as above; a seed's coordinates are those of its slot.
// Measured and dropped (round 3, profiles/r03/ab_step_512.txt): a PERSISTENT form of this kernel -- 8 workgroups per CU walking the
// tile sequence, the next tile's mask words requested a tile ahead -- ran 0.493 ms against 0.404 (n = 512) and 3.44 against 2.92
// (n = 1024).  The launch already keeps 7.5 of 8 wave slots per SIMD occupied (SQ_WAVE_CYCLES is in quad-cycles), so there was no
// dispatch gap to close, and workgroups that start together walk their five stages in step and meet at the LDS.
// The id volume leaves with the nt policy (it is read again only by the next pass, after all of it has been written): 0.238 -> 0.222 ms
// at n = 512, 1.61 -> 1.50 ms at n = 1024 (profiles/r04/ab_ftnt_*.txt).
template <class T> __device__ __forceinline__ void ft_store(T* p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void ft_store(uint2* p, uint2 v)
{
    __builtin_nontemporal_store(__builtin_bit_cast(unsigned long long, v), reinterpret_cast<unsigned long long*>(p));
}
// (Round 6, measured and dropped: the id stores through a buffer resource per row -- the uniform row base in SGPRs, the lane's column as the
// 32-bit offset, no 64-bit vector address per store -- ran +2.5 % at n = 512 and +4.7 % at n = 1024, profiles/r06/ab_ft_bufstore.txt.  The kernel
// issues ~930 vector instructions per tile, four clocks each on a SIMD: 512 tiles per SIMD x 233 per wave x 4 = 199 of its 217 us at n = 512;
// its ~1,160 scalar instructions per tile run beside them on the scalar unit and are not what bounds it.)
// CPT (n > 1024): the result leaves in the compact layout of IdC -- `out` = the word planes, `outB` = the byte planes of the window --
// instead of ID's own; inside the kernel the ids stay ID's (Id64).
constexpr int kTilesPerWg = 2;    // tiles per workgroup.  Round 3, without the census fast path (profiles/r03/ab_tpw_*.txt): 1 / 2 / 4 / 8 tiles = 0.371 /
                                  // 0.373 / 0.370 / 0.406 ms at n = 512; with the census (round 4, ab_fttpw_*.txt) two tiles: -11 %, four: +-0 / +9 %
// CYC (round 6): the launch produces the planes of ONE rank of a cyclic distribution of the grid over 2^zsh ranks (cyc = zsh | rank << 8; the
// first phase of the transposed multi-GPU pipeline).  n/4 is a multiple of the rank count, so the four planes of a
// closed chain belong to one rank: the rank takes the tiles whose z residue is its own modulo the rank count -- rz below is then the
// residue's index among them, rzg the residue itself -- reads the border bits of those planes from the whole-grid mask and writes
// plane z at index z >> zsh of `out`.  Ids, positions and ranks are global as everywhere else.
template <class ID, int XR, int NT, int TPW, bool CPT = false, bool CYC = false>
__global__ void __launch_bounds__(NT)
jfa_first_two(Frame f, const uint32_t* __restrict__ border, typename ID::T* __restrict__ out, unsigned char* __restrict__ outB,
              uint32_t tilesX, uint32_t tiles, uint32_t shifts, FastDiv divTilesX, FastDiv divK, uint32_t cyc)
{
    static_assert(!CPT || std::is_same<ID, Id64>::value, "compact output: from 8-byte ids");
    using T = typename ID::T;
    constexpr uint32_t SLOTS = 64u * XR;                           // slot = ((plane * 4 + row) * 4 + segment) * XR + residue
    constexpr int PER = (int)(SLOTS / NT);
    constexpr unsigned long long kEmpty = kEmptyKey;
    static_assert(SLOTS <= 0x10000u && SLOTS % NT == 0, "tile");
    __shared__ unsigned long long keys[SLOTS];
    __shared__ T idOf[SLOTS];                                      // packed id of the voxel of a slot (what a seed at that slot is called)
    __shared__ uint32_t list[SLOTS];                               // entries: slot | slot of the seed it holds << 16
    __shared__ float posX[4 * XR], posY[4], posZ[4];
    __shared__ uint32_t cnt[2];
    __shared__ uint32_t latCnt[TPW][XR], latSeed[TPW][XR];         // border voxels per lattice of each tile, the slot of one of them
    const uint32_t tid = threadIdx.x, lane = tid & 63u, N = f.n, k = N / 4u;
    const uint32_t zsh = CYC ? cyc & 31u : 0u, zoff = CYC ? cyc >> 8 : 0u;
    const uint32_t kzl = k >> zsh;                                 // planes between chain members in `out`
    auto zres = [&](uint32_t rz) { return CYC ? zoff + (rz << zsh) : rz; };      // index of a z residue among the rank's -> the residue
    // slot tid + i NT = row-plane (rpb + i G) x column `col` (see jfa_pass_seeds): row addresses are scalar work
    constexpr uint32_t RPW = 4u * XR;
    constexpr uint32_t G = NT / RPW;
    static_assert(NT % RPW == 0 && RPW % 64u == 0, "a wave must stay inside one row-plane");
    const uint32_t col = tid % RPW;
    const uint32_t rpb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / RPW));
    // wave-cooperative append of the slots whose flag is set (one reservation per wave)
    auto append = [&](const bool (&flag)[PER], const uint32_t (&seed)[PER], uint32_t& counter) {
        unsigned long long ms[PER];
        uint32_t n = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { ms[i] = __builtin_amdgcn_ballot_w64(flag[i]); n += (uint32_t)__popcll(ms[i]); }
        if (n == 0) return;                                        // uniform; every lane of the wave is here (no divergence above)
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counter, n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(ms[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms[i], 0u));
            if (flag[i]) list[base + below] = (tid + (uint32_t)i * NT) | (seed[i] << 16);
            base += (uint32_t)__popcll(ms[i]);
        }
    };
    // shifts = log2(tilesX) | log2(k) << 8 | 1 << 16 when both are powers of two: two run-time divisions per tile are ~50 instructions
    // of a wave that executes ~450 in all (the kernel is issue-bound, see profiles/r03/first_two_stages_tpw2_n512.txt)
    auto tile_origin = [&](uint32_t t, uint32_t& rx0, uint32_t& ry, uint32_t& rz) {
        if (shifts >> 16) {
            rx0 = (t & (tilesX - 1u)) * XR; const uint32_t q = t >> (shifts & 31u); ry = q & (k - 1u); rz = q >> ((shifts >> 8) & 31u);
        } else {                                                   // sides that are not powers of two: multiply-shift division (see FastDiv)
            uint32_t q, r;
            divTilesX.divmod(t, q, r); rx0 = r * XR;
            divK.divmod(q, rz, ry);
        }
    };
    // A workgroup works through TPW consecutive tiles (default 1).  Their border words -- the only thing read from memory, and 37 % of
    // a one-tile workgroup's life spent waiting for them (profiles/r03/first_two_stages_n512.txt) -- are ALL requested before the
    // first tile is touched.  (Nothing is loaded inside the tile loop, so no wait in it ever covers the stores of the tile before:
    // that is what made the persistent form of round 3 slower.)  More tiles per workgroup bought nothing, see kTilesPerWg.
    const uint32_t tile0 = (blockIdx.x) * TPW;
    uint32_t mw[TPW][PER];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        uint32_t rx0, ry, rz;
        tile_origin(min(tile0 + u, tiles - 1u), rx0, ry, rz);
        const uint32_t rzg = zres(rz);
        const uint32_t myx = rx0 + col % XR + __umul24(col / XR, k);
        const bool xin = rx0 + col % XR < k;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t rp = rpb + (uint32_t)i * G, y = ry + (rp & 3u) * k, z = rzg + (rp >> 2) * k;
            mw[u][i] = xin ? border[(z * N + y) * f.w + (myx >> 5)] : 0u;          // < 2^28 words at n = 2048: 32-bit index arithmetic
        }
    }
    {                                                              // lattice census of every tile of this workgroup (see below); the border words are in flight
        for (uint32_t i = tid; i < (uint32_t)(TPW * XR); i += NT) (&latCnt[0][0])[i] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const uint32_t tile = tile0 + u;
        if (tile >= tiles) break;                                  // uniform
        uint32_t rx0, ry, rz;
        tile_origin(tile, rx0, ry, rz);
        const uint32_t rzg = zres(rz);
        const uint32_t myx = rx0 + col % XR + __umul24(col / XR, k);
        const bool xin = rx0 + col % XR < k;
        // The 64 voxels of one residue class (x, y, z mod k) -- a LATTICE, 4 x 4 x 4 chain positions -- only ever see each other in these
        // two passes.  A lattice without a border voxel stays "none"; a lattice with exactly ONE ends with that seed in all 64 voxels
        // whatever the distances are (after the step of two chain positions the seed sits at {s, s ^ 2} along every axis, and every
        // position 0 .. 3 has one of those within one step).  Only lattices with two or more border voxels need the proposals -- 7 % of
        // them at n = 512 on the benchmark mesh, and 61 % of the tiles have none (profiles/r04/first_two_lattices.txt): those tiles write
        // their result straight from the census below: one barrier, no lists, keys or proposals.  With that much less to issue the wait
        // for the border words shows again, and two tiles per workgroup (all loads up front) pay: 0.305 -> 0.269 -> 0.239 ms at n = 512,
        // 2.17 -> 1.77 -> 1.59 ms at n = 1024, 22.3 -> 17.2 ms at n = 2048 (profiles/r04/ab_ftfast_*.txt, ab_fttpw_*.txt).
        {
            const uint32_t res = col % XR;
            uint32_t mine = 0, mySlot = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (xin && ((mw[u][i] >> (myx & 31u)) & 1u)) { ++mine; mySlot = tid + (uint32_t)i * NT; }
            bool multi = false;
            if (mine) {
                multi = atomicAdd(&latCnt[u][res], mine) + mine >= 2u;   // whoever adds last to a lattice of two or more sees it
                latSeed[u][res] = mySlot;
            }
            if (!__syncthreads_or(multi ? 1 : 0)) {
                const uint32_t sSlot = latSeed[u][res];               // valid where latCnt[u][res] == 1
                const T one = ID::pack(rx0 + res + __umul24((sSlot / XR) & 3u, k), ry + ((sSlot / (4u * XR)) & 3u) * k, rzg + (sSlot / (16u * XR)) * k);
                const T id = latCnt[u][res] ? one : ID::none();
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    if (!xin) continue;
                    const uint32_t rp = rpb + (uint32_t)i * G;
                    const size_t vox = (size_t)((rz + (rp >> 2) * kzl) * N + (ry + (rp & 3u) * k)) * N + myx;
                    if constexpr (CPT) {
                        const uint2 c = IdC::from64(id);
                        ft_store(reinterpret_cast<uint32_t*>(out) + vox, c.x);
                        ft_store(outB + vox, (unsigned char)c.y);
                    } else {
                        ft_store(out + vox, id);
                    }
                }
                continue;                                              // next tile of the workgroup (uniform)
            }
        }
        if (tid < 2) cnt[tid] = 0;
        {                                                          // positions of the tile's 4 XR columns, 4 rows, 4 planes (see ChainPosLds)
            if (tid < RPW) posX[tid] = axis_pos(f.ox, myx, f.vs);  // tid < RPW: col == tid
            if (tid < 4) { posY[tid] = axis_pos(f.oy, ry + tid * k, f.vs); posZ[tid] = axis_pos(f.oz, rzg + tid * k, f.vs); }
        }
        __syncthreads();                                           // also: the previous tile's output stage has read keys / idOf
        // every entry of the list proposes the seed that sits at slot q (its coordinates are those of q) from slot s
        auto scatter = [&](uint32_t nlist, auto step) {
            constexpr int STEP = decltype(step)::value;
            auto run = [&](const auto& pos) {
                for (uint32_t e = tid; e < nlist; e += NT) {
                    const uint32_t entry = list[e], s = entry & 0xFFFFu, q = entry >> 16;
                    const float sx = pos.x((q / XR) & 3u, q % XR), sy = pos.y((q / (4u * XR)) & 3u), sz = pos.z(q / (16u * XR));
                    if constexpr (STEP == 2) propose_half<XR>(keys, s, sx, sy, sz, pos);
                    else propose<XR, STEP>(keys, s, sx, sy, sz, pos, q);
                }
            };
            run(ChainPosLds<XR>{posX, posY, posZ});
        };
        // ---- stage A: border voxels -> pass with k = n/2
        // (Measured and dropped, profiles/r03/ab_gather_*.txt: stage A as a GATHER -- the wave ballots of the border flags in LDS, "some
        // candidate of this voxel is a border voxel" as an OR of four ballot words, the <= 8 candidates evaluated by the thread that
        // proposes the voxel in stage B; no list, scatter, collect or key reset and two barriers fewer.  Bit-identical, 0.302 -> 0.300 ms
        // at n = 512, 2.12 -> 2.32 ms at n = 1024: what it adds to the one proposing wave outweighs what it takes from the others.
        // The opposite trade -- three waves per 64 entries of stage B, one target plane each, a third of the instruction stream per wave
        // but the set-up three times -- ran 0.305 -> 0.344 ms / 2.13 -> 2.27 ms (ab_split_*.txt): total issue decides, not the critical path.)
        bool flag[PER];
        uint32_t seed[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t s = tid + (uint32_t)i * NT;
            const uint32_t rp = rpb + (uint32_t)i * G, y = ry + (rp & 3u) * k, z = rzg + (rp >> 2) * k;
            flag[i] = xin && ((mw[u][i] >> (myx & 31u)) & 1u);
            keys[s] = kEmpty;
            idOf[s] = ID::pack(myx, y, z);
            seed[i] = s;                                           // a border voxel is its own seed
        }
        append(flag, seed, cnt[0]);
        __syncthreads();
        scatter(cnt[0], std::integral_constant<int, 2>{});
        __syncthreads();
        // ---- stage B: voxels that have a seed now -> pass with k = n/4
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t s = tid + (uint32_t)i * NT;
            const unsigned long long key = keys[s];
            flag[i] = key != kEmpty;
            seed[i] = (uint32_t)key & 0xFFFFu;                     // slot of the voxel's pass-1 seed (garbage where flag is false: not appended)
            keys[s] = kEmpty;                                      // own slots only: nobody else touches them before the barrier
        }
        append(flag, seed, cnt[1]);
        __syncthreads();
        scatter(cnt[1], std::integral_constant<int, 1>{});
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (!xin) continue;
            const unsigned long long key = keys[tid + (uint32_t)i * NT];
            const T id = key == kEmpty ? ID::none() : idOf[(uint32_t)key & 0x07FFFFFFu];       // stage B tags its proposals with the seed's slot
            // (Answering the lattices of at most one border voxel from the census in THESE tiles as well -- two thirds of their border voxels --
            // was measured: -1 % at n = 512, +4 % at n = 1024, profiles/r04/ab_ft3_*.txt: such a tile is bound by its fixed stages, not by
            // the number of proposals.)
            const uint32_t rp = rpb + (uint32_t)i * G;
            const size_t vox = (size_t)((rz + (rp >> 2) * kzl) * N + (ry + (rp & 3u) * k)) * N + myx;
            if constexpr (CPT) {
                const uint2 c = IdC::from64(id);
                ft_store(reinterpret_cast<uint32_t*>(out) + vox, c.x);
                ft_store(outB + vox, (unsigned char)c.y);
            } else {
                ft_store(out + vox, id);
            }
        }
    }
}

}  // namespace

// Passes k = n/2 and k = n/4 of a whole grid from its border mask in one launch; timed as the first pass.
// The chains {r, r + n/4, r + n/2, r + 3n/4} are closed for any n % 4 == 0 (every legal n), and a tile whose 16 / 32 residues reach past n/4
// masks the excess lanes: the fused start serves EVERY whole grid the tile kernels serve (profiles/r04/size_sweep.txt: whole step
// 2.89 -> 2.47 ms at n = 480, 29.3 -> 23.6 at 960 against init ids + two region passes).
// ranks > 1: the planes of rank `rank` of a cyclic distribution over `ranks` ranks (a power of two dividing n/4) into a window of
// n / ranks planes (see the kernel, CYC).
int launch_win_first_two(vp_ctx* ctx, const Frame& f, const uint32_t* d_border, const IdWin& out, uint32_t ranks, uint32_t rank)
{
    ProfScope p(ctx, VP_K_JFA_FIRST);
    const uint32_t k = f.n / 4;
    if (ranks > 1) {
        if ((ranks & (ranks - 1)) != 0 || k % ranks != 0 || rank >= ranks) return set_error(VP_ERR_INVALID, "jfa_first_two (cyclic): n = %u, %u ranks", f.n, ranks);
        const uint32_t zsh = (uint32_t)__builtin_ctz(ranks), cyc = zsh | (rank << 8);
        const bool small = f.n <= 512;
        const uint32_t xr = small ? 16u : 32u;
        const uint32_t tilesX = (k + xr - 1) / xr, tiles = tilesX * k * (k >> zsh);
        const dim3 grid((tiles + kTilesPerWg - 1) / kTilesPerWg);
        auto pow2 = [](uint32_t v) { return v != 0 && (v & (v - 1)) == 0; };
        const uint32_t shifts = (pow2(tilesX) && pow2(k)) ? ((uint32_t)__builtin_ctz(tilesX) | ((uint32_t)__builtin_ctz(k) << 8) | (1u << 16)) : 0u;
        char* w = win_words(out, f.n, 0);
        unsigned char* b = (unsigned char*)win_bytes_plane(out, f.n, 0);
        if (win_compact(f.n)) hipLaunchKernelGGL((jfa_first_two<Id64, 32, 512, kTilesPerWg, true, true>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint2*)w, b, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k), cyc);
        else if (small) hipLaunchKernelGGL((jfa_first_two<Id9, 16, 256, kTilesPerWg, false, true>), grid, dim3(256), 0, ctx->stream, f, d_border, (uint32_t*)w, nullptr, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k), cyc);
        else            hipLaunchKernelGGL((jfa_first_two<Id10, 32, 512, kTilesPerWg, false, true>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint32_t*)w, nullptr, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k), cyc);
        VP_HIP(hipGetLastError());
        return 0;
    }
    // tile = 4 x 4 x 4 chain positions x XR residues.  Measured (profiles/r02/ab33.txt, r04/ab_ftxr_*.txt): 16 residues x 256 threads is the
    // best shape at n <= 512, 32 x 512 above.  A workgroup takes kTilesPerWg consecutive tiles.
    const bool small = f.n <= 512;
    const uint32_t xr = small ? 16u : 32u;
    const uint32_t tilesX = (k + xr - 1) / xr, tiles = tilesX * k * k;
    const dim3 grid((tiles + kTilesPerWg - 1) / kTilesPerWg);
    auto pow2 = [](uint32_t v) { return v != 0 && (v & (v - 1)) == 0; };
    const uint32_t shifts = (pow2(tilesX) && pow2(k)) ? ((uint32_t)__builtin_ctz(tilesX) | ((uint32_t)__builtin_ctz(k) << 8) | (1u << 16)) : 0u;
    char* w = win_words(out, f.n, 0);
    unsigned char* b = (unsigned char*)win_bytes_plane(out, f.n, 0);
    if (win_compact(f.n)) hipLaunchKernelGGL((jfa_first_two<Id64, 32, 512, kTilesPerWg, true>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint2*)w, b, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k), 0u);
    else if (small) hipLaunchKernelGGL((jfa_first_two<Id9, 16, 256, kTilesPerWg>), grid, dim3(256), 0, ctx->stream, f, d_border, (uint32_t*)w, nullptr, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k), 0u);
    else            hipLaunchKernelGGL((jfa_first_two<Id10, 32, 512, kTilesPerWg>), grid, dim3(512), 0, ctx->stream, f, d_border, (uint32_t*)w, nullptr, tilesX, tiles, shifts, make_fastdiv(tilesX), make_fastdiv(k), 0u);
    VP_HIP(hipGetLastError());
    return 0;
}

}  // namespace vp
