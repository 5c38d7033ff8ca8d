// K4, bound-and-refine arithmetic, second form (round 6) -- replaces evaluate.py:78-81 (np.dot, + bias, np.argsort) and the
// filtering half of the rank walk (evaluate.py:96-105) for catalogues of up to 65,535 columns and factor widths up to 128.
//
// What the counters of the first form said (profiles/r06_start_pmc_k4.json, profiles/r06_k4_ablation_start.txt): a tile-wave
// (32 users x 32 items, 8 v_mfma_f32_32x32x16_f16 = 256 cycles of the matrix pipe) kept its wave resident for 4,700-5,600 cycles,
// 51-56 % of them parked in s_waitcnt / s_barrier, 11-14 % in issue stalls, 33-35 % issuing (200-270 VALU, 100-116 scalar, 31-42 LDS
// instructions); two waves per SIMD (256 registers, 65 KB of LDS per workgroup) cannot cover that; one workgroup per CU instead of
// two ran 1.5-1.8 x slower.  The final stage (exact rescoring of ~35 candidates per user and piece: 64 lanes x 16 B on 64 cache
// lines per instruction, texture-addresser-bound) was 18-27 % of a pass and ran once per PIECE of a user block.
//
// This form:
//  * candidate lists are PACKED: one 32-bit entry {upper 16 bits of the approximate score, 16-bit column}.  The approximate
//    scores only ever decide what is kept (every survivor is rescored exactly), and the scheduled trims already searched on the
//    upper 16 bits.  32 KB of lists per workgroup instead of 48: THREE workgroups per CU (48.3 KB, <= 168 registers).
//  * the two lanes of a user (k-group h = 0 / 1 of the MFMA layout) own a private SEGMENT of 32 slots each and count their
//    entries in a register: no LDS atomic, no length word in LDS, an append is one ds_write_b32.
//  * the kernel ends by DUMPING its lists; topk_finish2_kernel ranks every row ONCE over the union of its pieces: item rows are
//    staged through LDS by direct-to-LDS loads, eight lanes per 128-byte line (8 cache lines per instruction instead of 64), the
//    user's row comes through the scalar cache, the fp32 fma chain is the fp32-MFMA kernel's own (exact_score, topk_parts.h):
//    ids and score bits are those of tkr_topk_set_math(fp32).
// User blocks whose lists cannot hold their margin are flagged and redone by the fp32 kernel, as before (csrc/topk.hip).
#include "topk_refine.h"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "topk_parts.h"

namespace tkr {

#ifdef TKR_R2_PROF
// cycle sums per phase of the tile loop, all waves (scripts/probe_topk_phases.py with a -DTKR_R2_PROF build): [0] fragment reads + MFMA
// chain + bias, [1] wait for the next tile's direct loads, [2] barrier, [3] scheduled trims, [4] filter, [5] prologue, [6] final trim +
// dump, [7] tile-waves
__device__ unsigned long long g_r2_prof[8];
#define R2_MARK(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); r2p[i] += n_ - r2t; r2t = n_; }
#else
#define R2_MARK(i)
#endif

constexpr int kSeg = 32;                 // slots of one half-lane's segment
constexpr int kF2Batch = 40;             // candidates rescored per pass of the finish kernel (5 direct-to-LDS loads per 16 factors)

__device__ __forceinline__ uint32_t ord16(uint32_t raw) { return (raw & 0x8000u) ? (~raw & 0xffffu) : (raw | 0x8000u); }   // = ordered_bits(score) >> 16
__device__ __forceinline__ uint32_t unord16(uint32_t o) { return (o & 0x8000u) ? (o & 0x7fffu) : (~o & 0xffffu); }

// value of lane ^ 32, in the vector ALU (v_permlane32_swap: no trip through the LDS crossbar as ds_bpermute takes)
__device__ __forceinline__ uint32_t xor32(uint32_t v, int lane) {
    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ r = __builtin_amdgcn_permlane32_swap(v, v, false, false);       // x = [lower half, lower half], y = [upper, upper]
    return (lane & 32) ? r.x : r.y;
}

template <int SIZE, int STRIDE>
__device__ __forceinline__ void cmpx32(uint32_t& k, int lane) {
    const uint32_t o = lane_xor<STRIDE>(k);
    const bool upper = (lane & STRIDE) != 0, desc = (lane & SIZE) == 0;
    k = (upper != desc) ? max(k, o) : min(k, o);
}
// descending bitonic sort of one 32-bit key per lane across the wave
__device__ __forceinline__ uint32_t wave_sort_desc32(uint32_t k, int lane) {
    cmpx32<2, 1>(k, lane);
    cmpx32<4, 2>(k, lane); cmpx32<4, 1>(k, lane);
    cmpx32<8, 4>(k, lane); cmpx32<8, 2>(k, lane); cmpx32<8, 1>(k, lane);
    cmpx32<16, 8>(k, lane); cmpx32<16, 4>(k, lane); cmpx32<16, 2>(k, lane); cmpx32<16, 1>(k, lane);
    cmpx32<32, 16>(k, lane); cmpx32<32, 8>(k, lane); cmpx32<32, 4>(k, lane); cmpx32<32, 2>(k, lane); cmpx32<32, 1>(k, lane);
    cmpx32<64, 32>(k, lane); cmpx32<64, 16>(k, lane); cmpx32<64, 8>(k, lane); cmpx32<64, 4>(k, lane); cmpx32<64, 2>(k, lane);
    cmpx32<64, 1>(k, lane);
    return k;
}

// ---- the lists of a workgroup: ent[slot][user], slot = 32 * h + position in the segment of lane (user, h) -----------------------
// Exact trim of ONE user (wave-uniform u), first half: the up to 64 entries of its two segments, one per lane, sorted by their
// 16-bit keys; `keep` = how many may still reach (K-th best key's lower edge - m2), returns the user's new threshold.  An entry's
// score lies in [edge(key), edge(key + 1)): edge(K-th key) is a lower bound of the K-th best score, and an entry can only be
// dropped when edge(key + 1) <= the new threshold, i.e. key < key_of(threshold).
__device__ __forceinline__ float trim_user2_sort(const uint32_t* ent, int uq, int n0, int n1, int K, int lane, float thr_in, float m2,
                                                 uint32_t& key, int& keep) {
    const int n = n0 + n1;
    const int slot = lane < n0 ? lane : kSeg + (lane - n0);
    key = 0;
    if (lane < n) {
        const uint32_t e = ent[slot * kR2Users + uq];
        key = (ord16(e >> 16) << 16) | (e & 0xffffu);
    }
    int lane_here = lane;                                        // opaque: the 21 lane-pattern masks of the network are rebuilt HERE, on the rare
    asm volatile("" : "+v"(lane_here));                          // path, instead of living in 42 scalar registers across the tile loop (47 spills)
    key = wave_sort_desc32(key, lane_here);
    keep = n;
    float refined = thr_in;
    if (n >= K) {
        const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)key, K - 1) >> 16;
        refined = fmaxf(thr_in, unordered_bits(kb << 16) - m2);
        const uint32_t ck = ordered_bits(refined) >> 16;
        keep = __popcll(__ballot(lane < n && (key >> 16) >= ck));          // sorted: a prefix of the lanes
    }
    return refined;
}
// ... second half: the best c0 of the kept entries go back to segment 0, the others to segment 1
__device__ __forceinline__ void trim_user2_deal(uint32_t* ent, int uq, uint32_t key, int keep, int c0, int lane) {
    if (lane < keep) ent[(lane < c0 ? lane : kSeg + (lane - c0)) * kR2Users + uq] = (unord16(key >> 16) << 16) | (key & 0xffffu);
}

// Scheduled trim of all 32 users of a wave at once, one user per lane pair (csrc/topk.hip trim_all_users, REFINE form, on packed
// entries): a lower bound of the K-th best key by a bitwise search on 15-bit keys held two per register, then every lane compacts
// its OWN segment in place -- no cross-half phase, no length word.
__device__ __forceinline__ float trim_all2(uint32_t* ent, int uw, int h, int K, float thr, float m2, int& cnt) {
    const int n_mine = cnt;
    const int lane = 32 * h;                                     // (only its bit 5 matters below)
    const int n = n_mine + (int)xor32((uint32_t)n_mine, lane);
    uint32_t xo[16];
    const uint32_t tkey = ordered_bits(thr) >> 16;
    uint32_t unlike = 0;
    uint32_t* seg = ent + (h * kSeg) * kR2Users + uw;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        uint32_t e0 = seg[(2 * e) * kR2Users], e1 = seg[(2 * e + 1) * kR2Users];     // slots exist; masked below
        asm volatile("" : "+v"(e0), "+v"(e1));                   // (or every load sits in an exec-masked branch of its own)
        const uint32_t k0 = (2 * e < n_mine) ? ord16(e0 >> 16) : 0u;
        const uint32_t k1 = (2 * e + 1 < n_mine) ? ord16(e1 >> 16) : 0u;
        unlike |= ((2 * e < n_mine) ? (k0 ^ tkey) : 0u) | ((2 * e + 1 < n_mine) ? (k1 ^ tkey) : 0u);
        xo[e] = k0 | (k1 << 16);
    }
    uint32_t mixed = unlike & 0x8000u;                       // keys on both sides of zero: keep the top bit, drop the lowest
    mixed |= xor32(mixed, lane);
    const int s1 = mixed ? 1 : 0;
    const uint32_t top_bit = mixed ? 0u : (tkey & 0x8000u);
    u16x2 mx = {0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        xo[e] = ((xo[e] >> s1) & 0x7fff7fffu) | 0x80008000u;
        const u16x2 o = __builtin_bit_cast(u16x2, xo[e]);
        mx = u16x2{(unsigned short)max(mx.x, o.x), (unsigned short)max(mx.y, o.y)};
    }
    uint32_t hi = (uint32_t)max(mx.x, mx.y) & 0x7fffu;
    hi = max(hi, xor32(hi, lane));
    const uint32_t lo = (tkey >> s1) & 0x7fffu;
    const uint32_t differ = (n >= K) ? (lo ^ hi) : 0u;
    const uint32_t every = wave_or(differ);
    const int top = 31 - __clz((int)(every | 1u));
    uint32_t prefix = hi & ~((2u << top) - 1u);
#pragma unroll 1
    for (int b = top; b >= 0; --b) {
        const uint32_t cand = prefix | (1u << b);
        const uint32_t cand2 = cand | (cand << 16);
        u16x2 acc = {0, 0};
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = count_ge2(xo[e], cand2, acc);
        int c = (int)acc.x + (int)acc.y;
        c += (int)xor32((uint32_t)c, lane);
        if (c >= K) prefix = cand;
    }
    const bool active = n >= K && prefix != 0;
    const uint32_t ob = mixed ? (prefix << 17) : ((top_bit | prefix) << 16);        // the smallest ordered score with this key
    const float refined = fmaxf(thr, unordered_bits(ob) - m2);
    prefix = min(prefix, ((ordered_bits(refined) >> 16) >> s1) & 0x7fffu);
    if (!active) return thr;
    uint32_t* wq = seg;
    int kept = 0;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const uint32_t k15 = ((e & 1) ? (xo[e >> 1] >> 16) : xo[e >> 1]) & 0x7fffu;
        if (k15 >= prefix) {                                 // (absent entries carry key 0 < prefix)
            const uint32_t v = seg[e * kR2Users];
            *wq = v;                                         // at or below the slot it was read from
            wq += kR2Users;
            ++kept;
        }
    }
    cnt = kept;
    return refined;
}

// One visit of the filter: the lanes of mask m append register R of the score block to their segments.
template <bool MASKED, int R>
struct Visit2 {
    static __device__ __forceinline__ void run(const uint64_t (&hr)[16], uint32_t hits, const f32x16& sc, uint32_t& wp, uint32_t himask, int col0) {
        if (hr[R]) {
            asm volatile("" ::: "memory");                       // keeps the scalar branch
            uint64_t m = hr[R];
            if constexpr (MASKED) m = __ballot((hits & (1u << R)) != 0u);
            uint64_t saved;
            uint32_t e;
            asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                         "v_add_u32 %[e], %[c], %[col0]\n\t"
                         "v_and_or_b32 %[e], %[val], %[hm], %[e]\n\t"
                         "ds_write_b32 %[wp], %[e]\n\t"
                         "v_add_u32 %[wp], 0x200, %[wp]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [sv] "=&s"(saved), [e] "=&v"(e), [wp] "+v"(wp)
                         : [m] "s"(m), [c] "n"((R & 3) + 8 * (R >> 2)), [col0] "v"(col0), [val] "v"(sc[R]), [hm] "s"(himask)
                         : "memory", "scc");
        }
        if constexpr (R + 1 < 16) Visit2<MASKED, R + 1>::run(hr, hits, sc, wp, himask, col0);
    }
};
static_assert(kR2Users * 4 == 0x200, "the append's address step is a literal");

// One step of the filter: register R of the score block.  The compare's lane mask (hr[R], an SGPR pair) decides by a scalar branch
// whether anything happens; the visit itself is one asm statement: exec <- the candidate lanes that are not rated (bit C of the lane's mask word) --
// they count the hit --, of those the lanes whose segment has room append {upper half of the score, column}.
template <int R>
struct Visit3 {
    static constexpr int C = (R & 3) + 8 * (R >> 2);            // row of register R inside the lane's 4-row groups = its bit in mh
    static __device__ __forceinline__ void run(const uint64_t (&hr)[16], const f32x16& sc, uint32_t mh, uint32_t& wp, uint32_t wlim,
                                               uint32_t& nh, uint32_t himask, int col0) {
        if (hr[R]) {
            asm volatile("" ::: "memory");                       // keeps the scalar branch
            uint64_t saved;
            uint32_t e;
            asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                         "v_and_b32 %[e], %[bit], %[mh]\n\t"
                         "v_cmp_eq_u32 vcc, 0, %[e]\n\t"
                         "s_and_b64 exec, exec, vcc\n\t"
                         "v_add_u32 %[nh], 1, %[nh]\n\t"
                         "v_cmp_gt_u32 vcc, %[wlim], %[wp]\n\t"
                         "s_and_b64 exec, exec, vcc\n\t"
                         "v_add_u32 %[e], %[c], %[col0]\n\t"
                         "v_and_or_b32 %[e], %[val], %[hm], %[e]\n\t"
                         "ds_write_b32 %[wp], %[e]\n\t"
                         "v_add_u32 %[wp], 0x200, %[wp]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [sv] "=&s"(saved), [e] "=&v"(e), [wp] "+v"(wp), [nh] "+v"(nh)
                         : [m] "s"(hr[R]), [bit] "n"(1 << C), [c] "n"(C), [mh] "v"(mh), [wlim] "v"(wlim), [col0] "v"(col0), [val] "v"(sc[R]),
                           [hm] "s"(himask)
                         : "memory", "scc", "vcc");
        }
        if constexpr (R + 1 < 16) Visit3<R + 1>::run(hr, sc, mh, wp, wlim, nh, himask, col0);
    }
};

// Filter of one 32 x 32 block of approximate scores (register r of lane (ul, h) = item (r&3) + 8*(r>>2) + 4h of the tile, user ul
// of the wave; bias already folded in).  Per register: one compare, one scalar branch; nothing else unless a lane has a candidate.
// wp = LDS byte address of the next free slot of this lane's segment, seg0 / wlim = its first slot / its end.
__device__ __forceinline__ void filter3(uint32_t* ent, const f32x16& sc, uint32_t maskw, int t, int K, float& thr, uint32_t& wp, uint32_t seg0,
                                        float m2, bool& lost, int lane, int wave) {
    const int ul = lane & 31, h = lane >> 5;
    const uint32_t mh = maskw >> (4 * h);                        // bit (r&3)+8*(r>>2) <-> accumulator register r
    const uint32_t wlim = seg0 + kSeg * kR2Users * 4, wp_in = wp;
    int col0 = t * 32 + 4 * h;
    asm volatile("" : "+v"(col0));
    const uint32_t himask = 0xffff0000u;
    uint32_t nh = 0;                                             // unrated candidates of this lane in this tile, written or not
    {
        uint64_t hr[16], any = 0;                                // all sixteen compares first (back to back), then one scalar branch per register
#pragma unroll
        for (int r = 0; r < 16; ++r) { hr[r] = __ballot(sc[r] >= thr); any |= hr[r]; }
        if (!any) return;
#if defined(TKR_R2_ABL) && (TKR_R2_ABL & 2)                      // timing experiments only: compares, no visits
        if (any != 12345u) return;
#endif
        Visit3<0>::run(hr, sc, mh, wp, wlim, nh, himask, col0);
    }
    uint64_t ov = __ballot(wp_in + nh * (kR2Users * 4) > wlim);
    if (ov == 0) return;
    // rare: a segment had no room for all of its lane's candidates (a lane can bring 16 in one tile, and right after a trim its
    // segment holds about K / 2 + the margin's entries).  Exact trim of that user, both segments; its lanes look again at what
    // still qualifies under the new threshold, and the kept entries are dealt so that BOTH segments have room for what their lane
    // still has to write.  A list that has no room even then loses the candidates and says so (the fp32 kernel redoes the block).
    uint32_t hits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) hits |= (sc[r] >= thr) ? (1u << r) : 0u;
    const uint32_t nm = ~mh;
    hits &= (nm & 0xfu) | ((nm >> 4) & 0xf0u) | ((nm >> 8) & 0xf00u) | ((nm >> 12) & 0xf000u);
    const int written = (int)((wp - wp_in) >> 9);               // the visits ran in register order: the lowest set bits are done
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < written) hits &= hits - 1u;
    int cnt = (int)((wp - seg0) >> 9);
    int n_mine = __popc(hits);
    const uint64_t hr[16] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
    while (ov) {
        const int u = __builtin_amdgcn_readfirstlane((__ffsll((long long)ov) - 1) & 31);
        const int n0 = __builtin_amdgcn_readlane(cnt, u), n1 = __builtin_amdgcn_readlane(cnt, u + 32);
        uint32_t key;
        int keep;
        const float nt = trim_user2_sort(ent, wave * 32 + u, n0, n1, K, lane, __shfl(thr, u, 64), __shfl(m2, u, 64), key, keep);
        if (ul == u) {
            thr = fmaxf(thr, nt);
            uint32_t still = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) still |= (sc[r] >= thr) ? (1u << r) : 0u;
            hits &= still;
            n_mine = __popc(hits);
        }
        const int p0 = __builtin_amdgcn_readlane(n_mine, u), p1 = __builtin_amdgcn_readlane(n_mine, u + 32);
        const int lo = max(0, keep - (kSeg - p1)), hi = min(keep, kSeg - p0);      // entries segment 0 may take
        const bool fits = lo <= hi;
        const int c0 = fits ? min(max((keep + 1) >> 1, lo), hi) : min((keep + 1) >> 1, kSeg);
        trim_user2_deal(ent, wave * 32 + u, key, min(keep, c0 + kSeg), c0, lane);
        if (lane == u) cnt = c0;
        if (lane == u + 32) cnt = min(keep - c0, kSeg);
        if (ul == u && !fits) { lost = true; hits = 0; n_mine = 0; }
        ov &= ~((1ull << u) | (1ull << (u + 32)));
    }
    wp = seg0 + (uint32_t)cnt * (kR2Users * 4);
    Visit2<true, 0>::run(hr, hits, sc, wp, himask, col0);        // what is still to be written, lane by lane
}

template <int KS>
__global__ __launch_bounds__(256, 3) void score_topk_refine2_kernel(const Refine2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int ROWB = KS * 32;                                // bytes of an item row of the image (KS * 16 fp16)
    constexpr int TILEB = 32 * ROWB;
    constexpr int KPAD = KS * 16;
    constexpr int NT_ = 256;
    constexpr int NG = (TILEB / 16 + NT_ - 1) / NT_;             // 16-byte direct-to-LDS loads per thread and tile
    float* tbias = reinterpret_cast<float*>(smem_raw + 2 * TILEB);   // [2][32]
    uint32_t* ent = reinterpret_cast<uint32_t*>(tbias + 64);     // [64][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = lane & 31, h = lane >> 5;                     // h = k-group of the operands AND row group of the result
    const int uw = wave * 32 + ul;
#ifdef TKR_R2_PROF
    unsigned long long r2p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r2t = __builtin_amdgcn_s_memtime();
#endif
    int4 it = make_int4((int)blockIdx.x, (int)blockIdx.y * a.tiles_per_split, 0, (int)blockIdx.y | ((int)gridDim.y << 16));
    if (a.items) it = a.items[blockIdx.x];
    const int block = it.x, slot = it.w & 0xffff, stride = it.w >> 16;
    const int row = block * kR2Users + uw;
    const bool user_ok = row < a.n_rows;
    const int k = a.k;

    // ---- B operand: lane (user ul, k-group h) holds elements 16s + 8h .. +7 of its user's row, scaled to fp16
    f16x8 hreg[KS];
    float margin, bscale, sv;
    {
        const int urow = user_ok ? (a.uidx ? a.uidx[row] : row) : 0;
        const float* up = a.U + (size_t)urow * k;
        float uv[KS][8];
        if ((k & 3) == 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = 16 * s + 8 * h + 4 * q;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (user_ok && e < k) v = *reinterpret_cast<const float4*>(up + e);
                    uv[s][4 * q + 0] = v.x; uv[s][4 * q + 1] = v.y; uv[s][4 * q + 2] = v.z; uv[s][4 * q + 3] = v.w;
                }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 16 * s + 8 * h + i;
                    const float v = up[min(e, k - 1)];
                    uv[s][i] = (user_ok && e < k) ? v : 0.f;
                }
        }
        float nu = 0.f, amax = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                nu = fmaf(uv[s][i], uv[s][i], nu);
                amax = fmaxf(amax, fabsf(uv[s][i]));
            }
        nu += __shfl_xor(nu, 32, 64);                            // the other k-group of the same user
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const float su = pow2_scale(amax);
        sv = pow2_scale(__uint_as_float(a.extra[2]));
        bscale = su * sv;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) hreg[s][i] = (_Float16)(uv[s][i] * su);
        const float reach = sqrtf(nu) * 1.001f * __uint_as_float(a.extra[0]);        // >= |u| * max |v_i|
        margin = (fmaf(1.05f * 0.0009765625f, reach, 3.8146973e-6f * (reach + __uint_as_float(a.extra[1]))) + 7.7e-34f) * bscale +
                 2.01f * (float)KPAD;
    }
    const float m2 = 2.f * margin;
    bool lost = false;
    const uint32_t seg0 = (uint32_t)(uintptr_t)(ent + (h * kSeg) * kR2Users + uw);      // LDS byte address of this lane's segment
    uint32_t wp = seg0;                                          // ... and of its next free slot: (wp - seg0) >> 9 entries are in
    float thr = (a.thr_shared && user_ok) ? unordered_bits(a.thr_shared[row]) : -INFINITY;   // what other pieces of the block found so far
    thr = thr * bscale - margin;
    const int n_tiles_all = (a.n_cols + 31) >> 5;
    const int t_begin = it.y;
    const int n_tiles = a.items ? it.z : min(n_tiles_all, t_begin + a.tiles_per_split);

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // (provably uniform: scalar address arithmetic for the direct loads)
    auto stage_direct = [&](int t, int buf) {                     // tile t of the image -> LDS buffer buf, asynchronously (vmcnt)
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int c0 = q * NT_ + wave_u * 64;                // first chunk of this wave's 64 (LDS destination = wave base + 16 * lane)
            if ((TILEB / 16) % NT_ == 0 || c0 < TILEB / 16) {
                const int c = min(c0 + lane, TILEB / 16 - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.vimg + (size_t)t * TILEB + (size_t)c * 16),
                                                 (__attribute__((address_space(3))) void*)(smem_raw + buf * TILEB + c0 * 16), 16, 0, 0);
            }
        }
    };
    // the item biases of a tile: asked for a tile ahead by the first 32 threads, stored in front of the barrier behind which they are read
    const bool bias_lane = a.bias != nullptr && tid < 32;
    auto bias_of = [&](int t) {
        const int col = t * 32 + tid;
        return (bias_lane && col < a.n_cols) ? a.bias[col] : 0.f;
    };
    if (t_begin < n_tiles) stage_direct(t_begin, t_begin & 1);
    if (bias_lane) tbias[(t_begin & 1) * 32 + tid] = bias_of(t_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const uint32_t tail_mask = (a.n_cols & 31) ? (0xffffffffu << (a.n_cols & 31)) : 0u;
    int next_sched = t_begin + 2;
    // the rated-item word of (tile, user) comes from HBM (every workgroup reads its own 512 bytes per tile): asked for one tile ahead
    uint32_t mask_next = (a.mask && user_ok && t_begin < n_tiles) ? a.mask[(size_t)t_begin * a.mask_pitch + row] : 0u;
    R2_MARK(5)
    for (int t = t_begin; t < n_tiles; ++t) {
        const int buf = t & 1;
        uint32_t maskw = mask_next;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        float bias_next = 0.f;
        if (t + 1 < n_tiles) {
            stage_direct(t + 1, buf ^ 1);                        // nobody reads that buffer after the barrier of tile t-1
            bias_next = bias_of(t + 1);
        }
        f16x8 afrag[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s)
            afrag[s] = *reinterpret_cast<const f16x8*>(smem_raw + buf * TILEB + ul * ROWB + ((2 * s + h) ^ tile_swizzle<KS>(ul)) * 16);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[s], hreg[s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);       // all the fragment reads first, then the chain
        __builtin_amdgcn_sched_group_barrier(0x008, KS, 0);
        mfma_result_guard_8pass(acc);
        if (a.bias) {                                            // scores in the user's scaled units: fma(bias, scale, acc)
            const float* tb = tbias + buf * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4*>(tb + 8 * g + 4 * h);
                acc[4 * g + 0] = fmaf(bq.x, bscale, acc[4 * g + 0]); acc[4 * g + 1] = fmaf(bq.y, bscale, acc[4 * g + 1]);
                acc[4 * g + 2] = fmaf(bq.z, bscale, acc[4 * g + 2]); acc[4 * g + 3] = fmaf(bq.w, bscale, acc[4 * g + 3]);
            }
        }
#ifdef TKR_R2_PROF
        asm volatile("" : "+v"(acc));
        R2_MARK(0)
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tile t+1 has landed (issued a whole MFMA chain ago)
        R2_MARK(1)
        if (bias_lane) tbias[(buf ^ 1) * 32 + tid] = bias_next;
        __syncthreads();
        R2_MARK(2)
        if (a.mask && user_ok && t + 1 < n_tiles) mask_next = a.mask[(size_t)(t + 1) * a.mask_pitch + row];
        if (!user_ok) maskw = 0xffffffffu;
        if (t == n_tiles_all - 1) maskw |= tail_mask;
#if defined(TKR_R2_ABL) && (TKR_R2_ABL & 4)                      // timing experiments only: no scheduled trims
        if (t == -12345) {
#else
        if (t == next_sched) {
#endif
            int cnt = (int)((wp - seg0) >> 9);
            if (__ballot(cnt + (int)xor32((uint32_t)cnt, lane) > kSeg) != 0) {      // lists still short (bounds shared by earlier pieces): nothing to gain
                thr = trim_all2(ent, uw, h, a.K, thr, m2, cnt);
                wp = seg0 + (uint32_t)cnt * (kR2Users * 4);
                float bs = bscale;
                int row_here = row;
                asm volatile("" : "+v"(bs), "+v"(row_here));
                thr = share_bound(a.thr_shared, row_here, user_ok && h == 0, thr, margin, bscale,
                                  __uint_as_float((254u - ((__float_as_uint(bs) >> 23) & 0xffu)) << 23));
            }
            next_sched = t + ((t - t_begin + 1) >> 1);
        }
        R2_MARK(3)
#if defined(TKR_R2_ABL) && (TKR_R2_ABL & 1)                      // timing experiments only (results wrong): no filter
        if (acc[0] + acc[5] + acc[10] + acc[15] == 12345.678f) wp += 512;
#else
        filter3(ent, acc, maskw, t, a.K, thr, wp, seg0, m2, lost, lane, wave);
#endif
#ifdef TKR_R2_PROF
        asm volatile("" : "+v"(wp), "+v"(thr));
        R2_MARK(4)
        r2p[7] += 1;
#endif
    }
    int cnt = (int)((wp - seg0) >> 9);
    if (__ballot(lost) != 0 && lane == 0) a.extra[4 + block] = 1u;          // the exact kernel redoes this block
    if (__ballot(cnt + (int)xor32((uint32_t)cnt, lane) > kSeg) != 0) {
        thr = trim_all2(ent, uw, h, a.K, thr, m2, cnt);
        // what this piece knows at its end, for the pieces of the block that start later (the lead pieces: csrc/topk.hip item_table)
        thr = share_bound(a.thr_shared, row, user_ok && h == 0, thr, margin, bscale, __uint_as_float((254u - ((__float_as_uint(bscale) >> 23) & 0xffu)) << 23));
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dump: piece-major, a user's 64 slots as one 256-byte line; the header says how much of each segment is used
    const size_t piece = a.pbase ? (size_t)a.pbase[block] + slot : (size_t)block * stride + slot;
    const int n_other = __shfl_xor(cnt, 32, 64);
    if (h == 0 && user_ok) a.dhdr[piece * kR2Users + uw] = make_float2(__int_as_float(cnt | (n_other << 8)), thr);
    for (int u = 0; u < 32; ++u) {
        if (block * kR2Users + wave * 32 + u >= a.n_rows) break; // wave-uniform
        const int n0 = __builtin_amdgcn_readlane(cnt, u), n1 = __builtin_amdgcn_readlane(cnt, u + 32);
        if ((lane & 31) < (lane < 32 ? n0 : n1))
            a.dump[(piece * kR2Users + wave * 32 + u) * kR2Slots + lane] = ent[lane * kR2Users + wave * 32 + u];
    }
#ifdef TKR_R2_PROF
    R2_MARK(6)
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_r2_prof[i], r2p[i]);
#endif
}

// ---- the finish kernel: one wave per row ------------------------------------------------------------------------------------------
// The pieces of a row's user block dumped their lists (supersets of the row's best K among their tiles, by exact score) and
// their final thresholds (each a lower bound of the K-th best exact score minus the margin, in the user's scaled units: the largest
// one is valid for every piece).  The wave takes the union of what can still reach that bound, rescores it with the fp32 kernel's
// own fma chain and sorts.  KS > 0 (k == 16 * KS, KS in {2, 4, 8}): the candidates' item rows are STAGED -- 16 factors of each
// k-half at a time, 128 bytes per candidate -- by direct-to-LDS loads in which eight lanes fetch one candidate's two 64-byte runs
// (a wave instruction touches 16 half-lines instead of 64 lines: the lane-per-candidate gather of round 5 ran at the rate of the
// texture addresser); the 16-byte chunks of candidate j are rotated by (j >> 1) & 7 on the way in, so that the chain's ds_read_b128
// (lane j reads ITS candidate's chunks) is conflict-free.  The user's row is wave-uniform: scalar loads.  KS == 0: any k, every
// lane reads its candidate's row itself (exact_score).
struct Finish2Args {
    const uint32_t* dump;
    const float2* dhdr;
    int n_rows, S_uniform;
    const int32_t* nslots;
    const int32_t* pbase;
    const uint32_t* flagged;
    const float* U;
    const int32_t* uidx;
    const float* Vt;
    const float* bias;
    int k, K;
    int32_t* out_ids;
    float* out_scores;
};

constexpr int kF2Waves = 2;

template <int KS>
__global__ __launch_bounds__(kF2Waves * 64) void topk_finish2_kernel(const Finish2Args a) {
    constexpr int NSTEP = KS >= 2 ? KS / 2 : 1;
    constexpr int STAGEB = KS >= 2 ? kF2Batch * 128 : 16;
    constexpr int NI = kF2Batch / 8;                             // direct-to-LDS loads per step
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[kF2Waves][2][STAGEB];
    __shared__ uint32_t s_q[kF2Waves][128];
    __shared__ uint64_t s_key[kF2Waves][64];
    __shared__ __attribute__((aligned(16))) float s_u[kF2Waves][128];        // the user's row (KS > 0)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = gridDim.x * kF2Waves;
    const int k = a.k, K = a.K;
    // A wave walks rows r, r + waves, ...; what a row needs FIRST -- its pieces' headers (lane s: piece s), the entries of its first
    // four pieces (lane e: slot e of each), its user row -- is asked for while the row before it is ranked (an empty dump alone ran
    // 0.45 ms at the Netflix shape: one cold round trip per row in front of everything else).
    struct Row { int S; size_t p0; bool skip; float2 hme; uint32_t e0, e1, e2, e3; float4 u4; int ul; const float* up; };
    auto fetch = [&](int r) {
        Row w;
        w.skip = true; w.S = 0; w.p0 = 0; w.ul = 0; w.up = a.U; w.hme = make_float2(0.f, -INFINITY); w.u4 = make_float4(0.f, 0.f, 0.f, 0.f);
        w.e0 = w.e1 = w.e2 = w.e3 = 0u;
        if (r >= a.n_rows) return w;
        const int block = r / kR2Users;
        w.ul = r - block * kR2Users;
        if (a.flagged && a.flagged[block]) return w;             // a list of this block overflowed: the fp32 kernel redoes it
        w.skip = false;
        w.S = a.nslots ? a.nslots[block] : a.S_uniform;
        w.p0 = a.pbase ? (size_t)a.pbase[block] : (size_t)block * a.S_uniform;
        if (lane < w.S) w.hme = a.dhdr[(w.p0 + lane) * kR2Users + w.ul];
        w.e0 = a.dump[(w.p0 * kR2Users + w.ul) * kR2Slots + lane];
        w.e1 = a.dump[((w.p0 + min(1, w.S - 1)) * kR2Users + w.ul) * kR2Slots + lane];
        w.e2 = a.dump[((w.p0 + min(2, w.S - 1)) * kR2Users + w.ul) * kR2Slots + lane];
        w.e3 = a.dump[((w.p0 + min(3, w.S - 1)) * kR2Users + w.ul) * kR2Slots + lane];
        const int urow_i = __builtin_amdgcn_readfirstlane(a.uidx ? a.uidx[r] : r);
        w.up = a.U + (size_t)urow_i * k;
        if constexpr (KS >= 2) {
            if (lane < 4 * KS) w.u4 = reinterpret_cast<const float4*>(w.up)[lane];
        }
        return w;
    };
    int r = __builtin_amdgcn_readfirstlane(blockIdx.x * kF2Waves + wave);
    Row nxt = fetch(r);
    for (; r < a.n_rows; r += n_waves) {
    const Row cur = nxt;
    nxt = fetch(r + n_waves);
    if (cur.skip) continue;
    const int S = cur.S, ul = cur.ul;
    const size_t p0 = cur.p0;
    const float2 hme = cur.hme;
    const float* up = cur.up;
    __builtin_amdgcn_wave_barrier();
    if constexpr (KS >= 2) {                                     // (scalar loads of the row, 32 floats in front of every step, were a cold miss each)
        if (lane < 4 * KS) reinterpret_cast<float4*>(s_u[wave])[lane] = cur.u4;
    }
    float B = lane < S ? hme.y : -INFINITY;                      // the largest threshold any piece reached
    for (int s0 = 64; s0 < S; s0 += 64)
        if (s0 + lane < S) B = fmaxf(B, a.dhdr[(p0 + s0 + lane) * kR2Users + ul].y);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) B = fmaxf(B, __shfl_xor(B, o, 64));
    const uint32_t ckB = ordered_bits(B) >> 16;                  // an entry's score is below edge(key + 1): it can reach B only with key >= key_of(B)
    uint32_t* q = s_q[wave];
    uint64_t* keys = s_key[wave];
    unsigned char* stage = s_stage[wave][0];
    int queued = 0;
    uint64_t best = 0ull;                                        // lanes 0..31: the best so far, descending (0 = none)
    bool first = true, counted = false;
    int s = 0;
    auto flush = [&](int m) {                                    // rescore the first m <= kF2Batch queued columns, merge them into `best`
        const bool live = lane < m;
        const int col = live ? (int)q[lane] : 0;
        float sc;
        if constexpr (KS >= 2) {
            const int KH = 8 * KS;
            int colv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) colv[i] = (int)q[min(8 * i + (lane >> 3), m - 1)];
            auto issue = [&](int step, int b) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int j = 8 * i + (lane >> 3);
                    const int g = (lane & 7) ^ ((j >> 1) & 7);   // the chunk of candidate j that lands in position lane & 7 of its block
                    const float* src = a.Vt + (size_t)colv[i] * k + (g < 4 ? 16 * step + 4 * g : KH + 16 * step + 4 * (g - 4));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage + b * STAGEB + i * 1024), 16, 0, 0);
                }
            };
            asm volatile("" ::: "memory");
            issue(0, 0);
            if (NSTEP > 1) issue(1, 1);
            const int cl = min(lane, kF2Batch - 1);              // (lanes past the batch read a valid block; their result is dropped)
            const int rot = (cl >> 1) & 7;
            float acc = 0.f;
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                if (step + 1 < NSTEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned char* blk = stage + (step & 1) * STAGEB + cl * 128;
                float4 v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const float4*>(blk + ((c ^ rot) * 16));
                const float4* u0 = reinterpret_cast<const float4*>(s_u[wave] + 16 * step);       // every lane the same address: broadcasts
                const float4* u1 = reinterpret_cast<const float4*>(s_u[wave] + KH + 16 * step);
#pragma unroll
                for (int c = 0; c < 4; ++c) {                    // chunk c of half 0 and chunk c of half 1: k ascending, halves interleaved
                    const float4 b0 = u0[c], b1 = u1[c];
                    acc = fmaf(v[c].x, b0.x, acc); acc = fmaf(v[4 + c].x, b1.x, acc);
                    acc = fmaf(v[c].y, b0.y, acc); acc = fmaf(v[4 + c].y, b1.y, acc);
                    acc = fmaf(v[c].z, b0.z, acc); acc = fmaf(v[4 + c].z, b1.z, acc);
                    acc = fmaf(v[c].w, b0.w, acc); acc = fmaf(v[4 + c].w, b1.w, acc);
                }
                asm volatile("" : "+v"(acc) : : "memory");       // the reads of this buffer are done before it is filled again
                if (step + 2 < NSTEP) issue(step + 2, step & 1);
            }
            acc = acc + (a.bias ? a.bias[col] : 0.f);
            sc = acc + 0.0f;
        } else {
            sc = live ? exact_score(up, a.Vt + (size_t)col * k, k, a.bias, col) : 0.f;
        }
        const uint64_t mine = live ? (((uint64_t)ordered_bits(sc) << 32) | ((uint32_t)col + 1u)) : 0ull;
        keys[lane] = mine;
        __builtin_amdgcn_wave_barrier();
        if (first && m == queued && s >= S) {
            // the row's only batch (the rule): the order comes from COUNTING -- every lane counts the keys above its own, two per
            // broadcast read, and stores its result at that position: ~3 instructions per candidate instead of the 21-stage network
            // (keys are distinct: a column appears once in a row's union; absent lanes hold 0)
            int rank = 0;
            for (int j = 0; j < m; j += 2) {
                const uint64_t k0 = keys[j], k1 = keys[j + 1];
                rank += (k0 > mine) + (k1 > mine);
            }
            if (live && rank < K) {
                a.out_ids[(size_t)r * K + rank] = col;
                if (a.out_scores) a.out_scores[(size_t)r * K + rank] = sc;
            }
            if (lane >= m && lane < K) {
                a.out_ids[(size_t)r * K + lane] = -1;
                if (a.out_scores) a.out_scores[(size_t)r * K + lane] = -INFINITY;
            }
            queued = 0;
            counted = true;
            return;
        }
        const uint64_t key = keys[lane];
        if (first) {                                             // nothing to merge with: one sort of the 64
            best = wave_sort_desc(key, lane);
            first = false;
        } else {                                                 // best 32 so far in lanes 0..31 + 32 new ones, twice
#pragma unroll 1
            for (int hx = 0; hx < 2; ++hx) {
                const uint64_t in = (uint64_t)__shfl((unsigned long long)key, (lane & 31) + 32 * hx, 64);
                best = wave_sort_desc(lane < 32 ? best : in, lane);
            }
        }
        if (lane >= 32) best = 0ull;
        __builtin_amdgcn_wave_barrier();
        const uint32_t rest = (lane + m < queued) ? q[lane + m] : 0u;       // what stays queued moves to the front
        const uint32_t rest2 = (lane + 64 + m < queued) ? q[lane + 64 + m] : 0u;
        __builtin_amdgcn_wave_barrier();
        q[lane] = rest;
        q[lane + 64] = rest2;
        queued -= m;
        __builtin_amdgcn_wave_barrier();
    };
    __builtin_amdgcn_wave_barrier();
    while (s < S || queued > 0) {                                // (one call site of the rescoring)
        while (s < S && queued < kF2Batch) {
            const size_t at = p0 + s;
            const int nn = s < 64 ? __shfl(__float_as_int(hme.x), s, 64) : __float_as_int(a.dhdr[at * kR2Users + ul].x);
            uint32_t e = s == 0 ? cur.e0 : s == 1 ? cur.e1 : s == 2 ? cur.e2 : cur.e3;
            if (s >= 4) e = a.dump[(at * kR2Users + ul) * kR2Slots + lane];
            const int n_seg = lane < 32 ? (nn & 0xff) : ((nn >> 8) & 0xff);
            const bool keep = (lane & 31) < n_seg && ord16(e >> 16) >= ckB;
            const uint64_t m = __ballot(keep);
            if (keep) q[queued + __popcll(m & ((1ull << lane) - 1ull))] = e & 0xffffu;
            queued += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            ++s;
        }
        if (queued > 0) flush(min(queued, kF2Batch));
    }
    if (counted) continue;
    if (lane < K) {
        const bool have = best != 0ull;
        const uint32_t ob = (uint32_t)(best >> 32);
        const uint32_t f = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
        a.out_ids[(size_t)r * K + lane] = have ? (int32_t)((uint32_t)best - 1u) : -1;
        if (a.out_scores) a.out_scores[(size_t)r * K + lane] = have ? __uint_as_float(f) : -INFINITY;
    }
    }                                                            // rows of this wave
}

#ifdef TKR_R2_PROF
extern "C" int tkr_k4_prof_read(unsigned long long* out8) {      // timing builds only: read and reset the phase counters
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_r2_prof), sizeof(zero));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_r2_prof), zero, sizeof(zero));
    return (int)e;
}
#endif

size_t refine2_dump_bytes(size_t n_pieces) { return n_pieces * kR2Users * (kR2Slots * sizeof(uint32_t) + sizeof(float2)) + 256; }
bool refine2_supports(int n_cols, int k) { return n_cols <= 65535 && k <= 128; }

template <int KS>
static int launch_tile2(const Refine2Args& a, hipStream_t stream) {
    const size_t lds = (size_t)2 * 32 * KS * 32 + 64 * sizeof(float) + (size_t)kR2Slots * kR2Users * sizeof(uint32_t);
    auto kern = score_topk_refine2_kernel<KS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(a.grid_x, a.grid_y), dim3(256), lds, stream, a);
    return (int)hipGetLastError();
}

int launch_refine2(const Refine2Args& a, hipStream_t stream) {
    if (!refine2_supports(a.n_cols, a.k) || !a.extra || !a.vimg || !a.dump || !a.dhdr) return TKR_EUNSUPPORTED;
    const int KS = a.k <= 16 ? 1 : (a.k <= 32 ? 2 : (a.k <= 64 ? 4 : 8));
    int rc = KS == 1 ? launch_tile2<1>(a, stream) : KS == 2 ? launch_tile2<2>(a, stream) : KS == 4 ? launch_tile2<4>(a, stream) : launch_tile2<8>(a, stream);
    if (rc != TKR_OK) return rc;
    const Finish2Args f = {a.dump, a.dhdr, a.n_rows, a.grid_y, a.items ? a.nslots : nullptr, a.items ? a.pbase : nullptr, a.extra + 4, a.U, a.uidx,
                           a.Vt, a.bias, a.k, a.K, a.out_ids, a.out_scores};
    const bool staged = a.k == 16 * KS && KS >= 2;
    // one row per wave (TKR_TOPK_FINISH_ROWS=n: rows r, r + waves, ... on n workgroups per CU with the next row's headers asked for ahead
    // -- measured no faster: 0.261 vs 0.251 ms at the ML-10M shape, 1.19 vs 1.11 ms at the Netflix shape)
    static const int rows_per_cu = getenv("TKR_TOPK_FINISH_ROWS") ? atoi(getenv("TKR_TOPK_FINISH_ROWS")) : 0;
    auto launch_f = [&](auto kern) {
        int blocks = (a.n_rows + kF2Waves - 1) / kF2Waves;
        if (rows_per_cu > 0) blocks = std::min(blocks, 256 * rows_per_cu);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(kF2Waves * 64), 0, stream, f);
    };
    if (staged && KS == 8) launch_f(topk_finish2_kernel<8>);
    else if (staged && KS == 4) launch_f(topk_finish2_kernel<4>);
    else if (staged && KS == 2) launch_f(topk_finish2_kernel<2>);
    else launch_f(topk_finish2_kernel<0>);
    return (int)hipGetLastError();
}

}  // namespace tkr
