// K4 -- full-catalogue scoring fused with rated-item masking and per-user top-K selection.
//
// Replaces evaluate.py:78-81 (scores = np.dot(umat, temat.T) (+ bias); np.argsort over the whole
// row) and the filtering half of the rank walk (evaluate.py:96-105: skip train-rated items, keep
// the first `total`).  The [n_users, n_items] score matrix and its int64 argsort never exist:
// each workgroup keeps W x 32 users' factors in registers, streams 32-item tiles of V through
// LDS, multiplies with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and filters the 32x32 scores
// of every wave straight out of the accumulators into small per-user candidate lists in LDS.
//
// Orientation: MFMA rows = items, columns = users, so a lane owns ONE user (lane & 31) and 16
// items of the tile: the running K-th best score of its user is one VGPR, the rated-item
// bitmask of (user, tile) is one 32-bit word.  Candidates (score >= K-th best so far) are
// appended with an LDS atomic; when a user's list could overflow the wave sorts it
// (64-lane bitonic network on (score, column) keys), keeps K and raises the threshold.
//
// Canonical order: descending score, ties -> higher column first (= a stable
// ascending argsort read backwards).  -0.0 is canonicalised to +0.0 so it ties with 0.0 like
// numpy.  Accumulation order of the dot product: k-halves [0,KH) and [KH,2KH) interleaved by
// the MFMA, k ascending inside a half (one rounding per product; exact whenever all partial
// sums are representable, which is what the golden G5/G6 fixtures guarantee).
//
// Roofline: fp32 MFMA, 2*k*n_items flop per user; HBM traffic per user ~ 4k B factors + 8K B
// out + 4*n_items/32 B mask words -- three orders of magnitude below the flop ratio.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tkr_common.h"
#include "topk_parts.h"
#include "topk_refine.h"
#include "../../include/tkr.h"

#ifndef TKR_ABL
#define TKR_ABL 0      // timing experiments only (scripts/ablate_topk.sh): 1 no filter, 2 no staging, 4 no per-tile barrier, 8 no on-demand trims, 16 no scheduled trims, 32 no appends, 64 no final stage of the refine kernel, 256 per-phase cycle counters (tkr_k4_prof_read)
#endif

#if TKR_ABL & 256
// cycle sums per phase of the bf16 / refine tile loop, all waves: [0] MFMA chain + bias, [1] staging, [2] barrier, [3] scheduled
// trims, [4] filter, [5] prologue (operands, first tile), [6] final stage, [7] wave-tiles.  scripts/probe_topk_phases.py reads them.
__device__ unsigned long long g_k4_prof[8];
#define K4_MARK(i)                                                           \
    {                                                                        \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();          \
        k4p[i] += n_ - k4t;                                                  \
        k4t = n_;                                                            \
    }
#else
#define K4_MARK(i)
#endif

namespace tkr {

template <typename IdT>
struct TopkSmem {
    float* tile;      // [2][32][KP]
    float* tbias;     // [2][32]
    int* cnt;         // [users]
    float* cs;        // [kCap][users]   entry-major: the 32 users of a wave scan their lists conflict-free
    IdT* ci;          // [kCap][users]
    int users;
};

// Sort user `uw`'s candidate list, keep the best K, return the new threshold (K-th best, or -inf
// while fewer than K candidates exist).  Wave-uniform call.
// REFINE (bound-and-refine arithmetic): the scores are approximations within `m2`/2 of the exact ones, so everything within
// m2 of the K-th best stays: returns max(thr_in, K-th best - m2), keeps what reaches it.
template <typename IdT, bool REFINE = false>
__device__ __forceinline__ float trim_user(const TopkSmem<IdT>& sm, int uw, int K, int lane, uint64_t* sorted_out,
                                           float thr_in = -INFINITY, float m2 = 0.f) {
    const int n = min(sm.cnt[uw], kCap);                         // a reservation past the capacity wrote nothing
    uint64_t key = 0;                                            // below every real key (real keys have idx+1 > 0)
    if (lane < n) key = ((uint64_t)ordered_bits(sm.cs[lane * sm.users + uw]) << 32) | ((uint32_t)sm.ci[lane * sm.users + uw] + 1u);
    key = wave_sort_desc(key, lane);
    int keep = min(n, K);
    float refined = thr_in;
    if constexpr (REFINE) {
        if (n >= K) {
            const uint32_t kb = __builtin_amdgcn_readlane((uint32_t)(key >> 32), K - 1);
            refined = fmaxf(thr_in, unordered_bits(kb) - m2);
            keep = __popcll(__ballot(lane < n && unordered_bits((uint32_t)(key >> 32)) >= refined));     // sorted: a prefix of the lanes
        }
    }
    if (lane < keep) {
        const uint32_t ob = (uint32_t)(key >> 32);
        const uint32_t f = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
        sm.cs[lane * sm.users + uw] = __uint_as_float(f);
        sm.ci[lane * sm.users + uw] = (IdT)((uint32_t)key - 1u);
    }
    if (lane == 0) sm.cnt[uw] = keep;
    if (sorted_out) *sorted_out = key;
    if constexpr (REFINE) return refined;
    const uint32_t kb = __builtin_amdgcn_readlane((uint32_t)(key >> 32), K - 1);
    const uint32_t kf = (kb & 0x80000000u) ? (kb & 0x7fffffffu) : ~kb;
    return (n >= K) ? __uint_as_float(kf) : -INFINITY;
}

// Scheduled trim of ALL 32 users of a wave at once, one user per lane pair (half h scans entries
// [32h, 32h+32)).  The filter only needs a LOWER BOUND of the K-th best score (a superset of the top K may
// stay; order and exact cut come from the final sort), so the search runs on the upper 16 bits of the ordered
// score, held as 15-bit keys two per register: a list whose keys (and threshold) all carry the same top bit -- every list but
// one that straddles zero -- drops that bit, the others drop the lowest one.  One subtraction, one packed shift and one packed add
// then answer "key >= candidate" for a pair of keys (count_ge2; the compare / carry form of the same count took 4 instructions
// and 2 wait states per pair), and the bitwise search only walks the bits in which the threshold and the largest key of a list
// differ (the K-th best lies between them, or below the threshold, where the threshold wins anyway): 6-8 steps instead of 16.
// Then in-place compaction of the entries whose key reaches the bound.  A list that does not shrink enough is caught by the
// on-demand exact trim.
// REFINE: approximate scores, see trim_user -- the bound found is lowered by m2 before it becomes the threshold and the cut.
template <typename IdT, bool REFINE = false>
__device__ __forceinline__ float trim_all_users(const TopkSmem<IdT>& sm, int uw, int h, int K, float thr, float m2 = 0.f) {
    const int n = sm.cnt[uw];
    uint32_t xo[16];                                         // entries 2e (low half) and 2e+1 (high half)
    const uint32_t tkey = ordered_bits(thr) >> 16;
    uint32_t unlike = 0;                                     // bits in which some key present differs from the threshold's
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int p0 = 32 * h + 2 * e;
        const float v0 = sm.cs[p0 * sm.users + uw], v1 = sm.cs[(p0 + 1) * sm.users + uw];   // slots exist; masked below
        const uint32_t k0 = (p0 < n) ? (ordered_bits(v0) >> 16) : 0u;                    // absent: key 0
        const uint32_t k1 = (p0 + 1 < n) ? (ordered_bits(v1) >> 16) : 0u;
        unlike |= ((p0 < n) ? (k0 ^ tkey) : 0u) | ((p0 + 1 < n) ? (k1 ^ tkey) : 0u);
        xo[e] = k0 | (k1 << 16);
    }
    uint32_t mixed = unlike & 0x8000u;                       // keys on both sides of zero: keep the top bit, drop the lowest
    mixed |= (uint32_t)__shfl_xor((int)mixed, 32, 64);
    const int s1 = mixed ? 1 : 0;                            // per user (both halves agree)
    const uint32_t top_bit = mixed ? 0u : (tkey & 0x8000u);
    // the bits to search: below the common prefix of the threshold's key and the largest key of the list (over both halves)
    u16x2 mx = {0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        xo[e] = ((xo[e] >> s1) & 0x7fff7fffu) | 0x80008000u;
        const u16x2 o = __builtin_bit_cast(u16x2, xo[e]);
        mx = u16x2{(unsigned short)max(mx.x, o.x), (unsigned short)max(mx.y, o.y)};
    }
    uint32_t hi = (uint32_t)max(mx.x, mx.y) & 0x7fffu;
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, 32, 64));
    const uint32_t lo = (tkey >> s1) & 0x7fffu;
    const uint32_t differ = (n >= K) ? (lo ^ hi) : 0u;       // lists with fewer than K entries keep everything
    const uint32_t every = wave_or(differ);
    const int top = 31 - __clz((int)(every | 1u));            // wave-uniform: the highest bit any list has to decide
    uint32_t prefix = hi & ~((2u << top) - 1u);               // the bits above it: common to threshold and largest key
#pragma unroll 1
    for (int b = top; b >= 0; --b) {
        const uint32_t cand = prefix | (1u << b);
        const uint32_t cand2 = cand | (cand << 16);
        u16x2 acc = {0, 0};
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = count_ge2(xo[e], cand2, acc);
        int c = (int)acc.x + (int)acc.y;
        c += __shfl_xor(c, 32, 64);
        if (c >= K) prefix = cand;
    }
    const bool active = n >= K && prefix != 0;               // fewer than K candidates: keep all, threshold unchanged
    // the smallest ordered score with this key
    uint32_t ob = mixed ? (prefix << 17) : ((top_bit | prefix) << 16);
    float refined = thr;
    if constexpr (REFINE) {
        refined = fmaxf(thr, unordered_bits(ob) - m2);
        // every entry >= refined has a key >= this one (refined lies between the threshold and the largest entry: same top bit)
        prefix = min(prefix, ((ordered_bits(refined) >> 16) >> s1) & 0x7fffu);
    }
    const uint32_t cut2 = prefix | (prefix << 16);
    u16x2 macc = {0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) macc = count_ge2(xo[e], cut2, macc);
    const int mine = (int)macc.x + (int)macc.y;
    const int other = __shfl_xor(mine, 32, 64);
    // in-place compaction, half 0 first (writes at or below what it reads), then half 1 behind it
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
        if (active && h == phase) {
            int pos = h ? other : 0;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const uint32_t k15 = ((e & 1) ? (xo[e >> 1] >> 16) : xo[e >> 1]) & 0x7fffu;
                if (k15 >= prefix) {
                    const float v = sm.cs[(32 * h + e) * sm.users + uw];
                    const IdT id = sm.ci[(32 * h + e) * sm.users + uw];
                    sm.cs[pos * sm.users + uw] = v;
                    sm.ci[pos * sm.users + uw] = id;
                    ++pos;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (active && h == 0) sm.cnt[uw] = mine + other;
    if constexpr (REFINE) return active ? refined : thr;
    const uint32_t f = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
    return active ? fmaxf(thr, __uint_as_float(f)) : thr;
}

// One visit of the filter's fast path (filter_tile): the lanes of mask m append register R of the score block to their lists.
// as / ai: LDS byte addresses of the lane's next score / id slot, ss / si: their strides.  MASKED: m is taken from the lane's
// hit bits (a rated item of some lane passed its threshold in this tile), otherwise it is the compare's own SGPR mask.
template <typename IdT, bool REFINE, bool MASKED, int R>
struct FastVisit {
    static __device__ __forceinline__ void run(const uint64_t (&hr)[16], uint32_t hits, const float (&sc)[16], uint32_t& as, uint32_t& ai,
                                               uint32_t ss, uint32_t si, int col0) {
        static_assert(sizeof(IdT) == 2 || sizeof(IdT) == 4, "ids are 16 or 32 bits");
        if (hr[R]) {
            asm volatile("" ::: "memory");                       // keeps the scalar branch
            uint64_t m = hr[R];
            if constexpr (MASKED) m = __ballot((hits & (1u << R)) != 0u);
            const float val = REFINE ? sc[R] : sc[R] + 0.0f;     // -0.0 -> +0.0: ties with 0.0 like numpy (refine: rescored exactly later)
            uint64_t saved;
            uint32_t id;
            if constexpr (sizeof(IdT) == 2)
                asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                             "v_add_u32 %[id], %[c], %[col0]\n\t"
                             "ds_write_b32 %[as], %[val]\n\t"
                             "ds_write_b16 %[ai], %[id]\n\t"
                             "v_add_u32 %[as], %[ss], %[as]\n\t"
                             "v_add_u32 %[ai], %[si], %[ai]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved), [id] "=&v"(id), [as] "+v"(as), [ai] "+v"(ai)
                             : [m] "s"(m), [c] "n"((R & 3) + 8 * (R >> 2)), [col0] "v"(col0), [val] "v"(val), [ss] "s"(ss), [si] "s"(si)
                             : "memory", "scc");
            else
                asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                             "v_add_u32 %[id], %[c], %[col0]\n\t"
                             "ds_write_b32 %[as], %[val]\n\t"
                             "ds_write_b32 %[ai], %[id]\n\t"
                             "v_add_u32 %[as], %[ss], %[as]\n\t"
                             "v_add_u32 %[ai], %[si], %[ai]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(saved), [id] "=&v"(id), [as] "+v"(as), [ai] "+v"(ai)
                             : [m] "s"(m), [c] "n"((R & 3) + 8 * (R >> 2)), [col0] "v"(col0), [val] "v"(val), [ss] "s"(ss), [si] "s"(si)
                             : "memory", "scc");
        }
        if constexpr (R + 1 < 16) FastVisit<IdT, REFINE, MASKED, R + 1>::run(hr, hits, sc, as, ai, ss, si, col0);
    }
};

// Filter of one 32x32 score block (accumulator layout of the 32x32 MFMAs: register r of lane (ul, h) = item
// (r&3) + 8*(r>>2) + 4h of the tile, user ul of the wave) into the candidate lists.
// fp32 MFMA and VALU time add up on a SIMD (measured: nothing hides in the MFMA shadow, from the same wave or the
// other one), so the common path is kept to one compare per score: hr[r] is the wave's lane mask of "score r
// reaches my user's threshold" and lives in SGPRs; registers without a single candidate are skipped by a scalar
// branch, the rated/tail/no-user bits (maskw) are consulted only when a lane has one.
// BIASED: `acc` already holds fl(fl(dot) + bias) (add_bias_inplace), tbias is not read
// REFINE: approximate scores (see trim_user); a list whose exact trim leaves no room for this tile's candidates loses them and
// says so in `lost` (the exact arithmetic redoes the block).
template <typename IdT, bool BIASED = false, bool REFINE = false>
__device__ __forceinline__ void filter_tile(const TopkSmem<IdT>& sm, const f32x16& acc, const float* tbias, uint32_t maskw,
                                            int t, int K, float& thr, float m2 = 0.f, bool* lost = nullptr, float bscale = 1.f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ul = lane & 31, h = lane >> 5, users = sm.users;
    const int uw = wave * 32 + ul;
    const uint32_t mh = maskw >> (4 * h);                        // bit (r&3)+8*(r>>2) <-> accumulator register r
    float sc[16];
    if constexpr (BIASED) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = acc[r];
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {                            // rows 8g+4h .. 8g+4h+3 are registers 4g..4g+3
            const float4 bq = *reinterpret_cast<const float4*>(tbias + 8 * g + 4 * h);
            if constexpr (REFINE) {                              // scores in the user's scaled units
                sc[4 * g + 0] = fmaf(bq.x, bscale, acc[4 * g + 0]);
                sc[4 * g + 1] = fmaf(bq.y, bscale, acc[4 * g + 1]);
                sc[4 * g + 2] = fmaf(bq.z, bscale, acc[4 * g + 2]);
                sc[4 * g + 3] = fmaf(bq.w, bscale, acc[4 * g + 3]);
            } else {
                sc[4 * g + 0] = acc[4 * g + 0] + bq.x;           // fl(fl(dot)+b)
                sc[4 * g + 1] = acc[4 * g + 1] + bq.y;
                sc[4 * g + 2] = acc[4 * g + 2] + bq.z;
                sc[4 * g + 3] = acc[4 * g + 3] + bq.w;
            }
        }
    }
    uint64_t hr[16], any = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hr[r] = __ballot(sc[r] >= thr); any |= hr[r]; }
    if (!any) return;
#if TKR_ABL & 32
    if (any != 12345u) return;
#endif
    uint32_t hits = 0;                                           // bit r: register r of this lane is a candidate
#pragma unroll
    for (int r = 0; r < 16; ++r) hits |= (sc[r] >= thr) ? (1u << r) : 0u;
    // register r <-> mask bit (r&3) + 8*(r>>2): gather the four nibbles at bits 0, 8, 16, 24 of ~mh
    const uint32_t nm = ~mh, raw_hits = hits;
    hits &= (nm & 0xfu) | ((nm >> 4) & 0xf0u) | ((nm >> 8) & 0xf00u) | ((nm >> 12) & 0xf000u);
    uint32_t unplaced = 0;
    int pos = 0;
    const int n_mine = __popc(hits);
    if (hits) pos = atomicAdd(&sm.cnt[uw], n_mine);              // one LDS atomic per lane reserves all its slots
#if !(TKR_ABL & 512)
    if (__ballot(pos + n_mine > kCap) == 0) {
        // Every reservation of the wave fits (the rule; an overflow takes the general loop below).  The filter is bound by
        // instruction issue, and the general visit is ~25 issued instructions for the one or two lanes that hold a candidate in a
        // register; here a visit is exec <- the candidate lanes (the SGPR mask of the compare itself unless a lane of the wave had a
        // rated candidate in this tile), two stores, two address increments (FastVisit).
        uint32_t as = (uint32_t)(uintptr_t)(sm.cs + pos * users + uw), ai = (uint32_t)(uintptr_t)(sm.ci + pos * users + uw);
        const uint32_t ss = (uint32_t)users * 4u, si = (uint32_t)users * (uint32_t)sizeof(IdT);
        int col0 = t * 32 + 4 * h;
        asm volatile("" : "+v"(col0));
        if (__ballot(hits != raw_hits) == 0) FastVisit<IdT, REFINE, false, 0>::run(hr, hits, sc, as, ai, ss, si, col0);
        else FastVisit<IdT, REFINE, true, 0>::run(hr, hits, sc, as, ai, ss, si, col0);
        return;
    }
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (hr[r]) {                                             // scalar branch: most registers hold no candidate
            asm volatile("" ::: "memory");                       // (keeps the branch: the body must not be if-converted)
            if (hits & (1u << r)) {
                if (pos < kCap) {
                    sm.cs[pos * users + uw] = sc[r] + 0.0f;      // -0.0 -> +0.0: ties with 0.0 like numpy
                    sm.ci[pos * users + uw] = (IdT)(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
                } else {
                    unplaced |= 1u << r;                         // list full: trimmed below, then appended
                }
                ++pos;
            }
        }
    // rare: some user's list overflowed.  Exact trim of that user (keeps K, raises the threshold), then its
    // lanes append what still qualifies: at most 32 per user and tile, K + 32 <= kCap.
    uint64_t ov = __ballot(unplaced != 0);
#if TKR_ABL & 8
    ov = 0;
#endif
    while (ov) {
        const int u = (__ffsll((long long)ov) - 1) & 31;
        float nt;
        if constexpr (REFINE) nt = trim_user<IdT, true>(sm, wave * 32 + u, K, lane, nullptr, __shfl(thr, u, 64), __shfl(m2, u, 64));
        else nt = trim_user<IdT>(sm, wave * 32 + u, K, lane, nullptr);
        if (ul == u) {
            thr = fmaxf(thr, nt);                                // never below what another item range published
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((unplaced & (1u << r)) && sc[r] >= thr) {
                    const int p2 = atomicAdd(&sm.cnt[uw], 1);
                    if (!REFINE || p2 < kCap) {
                        sm.cs[p2 * users + uw] = sc[r] + 0.0f;
                        sm.ci[p2 * users + uw] = (IdT)(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
                    } else {
                        *lost = true;
                    }
                }
            unplaced = 0;
        }
        ov = __ballot(unplaced != 0);
    }
}

// Final exact sort of every user's list and output (or the partial list of this item range, merged later).
// One more lower-bound trim leaves most lists with K..32 entries; those are sorted two users at a time in the two
// 32-lane halves of the wave (15 compare-exchange stages, none across the halves) instead of one 21-stage sort each.
// acc <- fl(acc + bias) in the accumulator's own registers (no second 16-register tuple)
__device__ __forceinline__ void add_bias_inplace(f32x16& acc, const float* tbias, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bq = *reinterpret_cast<const float4*>(tbias + 8 * g + 4 * h);
        acc[4 * g + 0] += bq.x; acc[4 * g + 1] += bq.y; acc[4 * g + 2] += bq.z; acc[4 * g + 3] += bq.w;
    }
}

__device__ __forceinline__ void add_scaled_bias_inplace(f32x16& acc, const float* tbias, int h, float bscale) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bq = *reinterpret_cast<const float4*>(tbias + 8 * g + 4 * h);
        acc[4 * g + 0] = fmaf(bq.x, bscale, acc[4 * g + 0]); acc[4 * g + 1] = fmaf(bq.y, bscale, acc[4 * g + 1]);
        acc[4 * g + 2] = fmaf(bq.z, bscale, acc[4 * g + 2]); acc[4 * g + 3] = fmaf(bq.w, bscale, acc[4 * g + 3]);
    }
}

// where a workgroup's result goes: straight to the output, or as one of the sorted partial lists of its rows
struct TopkSlot {
    int block;        // user block (rows block*users .. +users-1)
    int slot;         // index of this partial list among the row's `stride` slots
    int stride;       // slots per row in the partial-list buffer; 1: final output, no merge
};

template <typename IdT>
__device__ __forceinline__ void emit_row(const TopkSlot& ws, int r, int p, bool have, uint64_t key, int K,
                                         int32_t* __restrict__ out_ids, float* __restrict__ out_scores, uint64_t* __restrict__ part) {
    if (ws.stride > 1) {
        part[((size_t)r * ws.stride + ws.slot) * K + p] = have ? key : 0ull;
        return;
    }
    const uint32_t ob = (uint32_t)(key >> 32);
    const uint32_t f = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
    out_ids[(size_t)r * K + p] = have ? (int32_t)((uint32_t)key - 1u) : -1;
    if (out_scores) out_scores[(size_t)r * K + p] = have ? __uint_as_float(f) : -INFINITY;
}

template <typename IdT>
__device__ __forceinline__ void write_rows(const TopkSmem<IdT>& sm, const TopkSlot& ws, int n_rows, int K, float thr,
                                           int32_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                           uint64_t* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int me = lane & 31, half = lane >> 5;
    if (__ballot(sm.cnt[wave * 32 + me] > 32) != 0) {
        (void)trim_all_users<IdT>(sm, wave * 32 + me, half, K, thr);
        __builtin_amdgcn_wave_barrier();
    }
    for (int j = 0; j < 16; ++j) {
        const int u0 = wave * 32 + 2 * j, r0 = ws.block * sm.users + u0;
        if (r0 >= n_rows) break;                                 // wave-uniform
        const int n0 = min(sm.cnt[u0], kCap);
        const int n1 = (r0 + 1 < n_rows) ? min(sm.cnt[u0 + 1], kCap) : 0;
        if (n0 <= 32 && n1 <= 32) {
            const int uu = u0 + half, nn = half ? n1 : n0;
            uint64_t key = 0;                                    // below every real key (real keys have idx+1 > 0)
            if (me < nn) key = ((uint64_t)ordered_bits(sm.cs[me * sm.users + uu]) << 32) | ((uint32_t)sm.ci[me * sm.users + uu] + 1u);
            key = wave_sort_halves(key, lane);
            const int p = half ? 31 - me : me;                   // the upper half comes out ascending
            if (r0 + half < n_rows && p < K) emit_row<IdT>(ws, r0 + half, p, p < nn, key, K, out_ids, out_scores, part);
            continue;
        }
        for (int q = 0; q < 2; ++q) {
            const int r = r0 + q;
            if (r >= n_rows) break;
            uint64_t key;
            const int n = q ? n1 : n0;
            trim_user<IdT>(sm, u0 + q, K, lane, &key);
            if (lane < K) emit_row<IdT>(ws, r, lane, lane < n, key, K, out_ids, out_scores, part);
        }
    }
}

// Final stage of the bound-and-refine kernel: every list holds a superset of its user's best K (by exact score) among the
// tiles of this workgroup.  Phase 1 scores the candidates exactly, one per lane, over the FLAT sequence of the wave's 32
// lists (a rescoring is four dependent memory round trips of 16 loads -- ~5 us whether 34 lanes work or 64; per list that
// was 32 passes per wave, flat it is ~17) and writes the exact score over the approximate one; phase 2 is the final
// stage of the exact kernels (lower-bound trim to K..32 entries, two users per sort).  `offs`: 33 ints of LDS of this wave.
constexpr int kStageUsers = 4;       // user rows staged per rescoring pass (write_rows_refine)
#ifndef TKR_RESCORE_IN_FLIGHT
#define TKR_RESCORE_IN_FLIGHT 8
#endif
constexpr int kRescoreInFlight = TKR_RESCORE_IN_FLIGHT;   // 16-byte loads of each half of an item row in flight per lane

template <typename IdT, int KS, int IN_FLIGHT = kRescoreInFlight>
__device__ __forceinline__ void write_rows_refine(const TopkSmem<IdT>& sm, const TopkSlot& ws, int n_rows, int K, float thr, float m2,
                                                  const float* __restrict__ U, const int32_t* __restrict__ uidx,
                                                  const float* __restrict__ Vt, const float* __restrict__ bias, int k,
                                                  int32_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                  uint64_t* __restrict__ part, unsigned char* scratch, int scratch_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int me = lane & 31, half = lane >> 5;
    const int n_waves = blockDim.x >> 6;
    int* offs = reinterpret_cast<int*>(scratch) + wave * 64;     // 33 ints of this wave
    // rows of the (up to kStageUsers) users of a pass, staged for the fast rescoring below: behind every wave's `offs`
    float* ust = reinterpret_cast<float*>(scratch + n_waves * 256) + (size_t)wave * kStageUsers * k;
    const bool staged = k == 16 * KS && KS <= 8 && n_waves * (256 + kStageUsers * k * 4) <= scratch_bytes;      // full-width rows: every trip count below is a constant
    if (__ballot(sm.cnt[wave * 32 + me] > 32) != 0) {
        (void)trim_all_users<IdT, true>(sm, wave * 32 + me, half, K, thr, m2);
        __builtin_amdgcn_wave_barrier();
    }
    const int n = lane < 32 ? min(sm.cnt[wave * 32 + lane], kCap) : 0;
    if (lane < 32) sm.cnt[wave * 32 + lane] = n;                 // a reservation past the capacity wrote nothing
    int incl = n;                                                // inclusive scan over the 32 lists
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 31, 64);
    if (lane < 32) offs[lane] = incl - n;
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < total; base += 64) {               // wave-uniform
        const int f = base + lane;
        int u = 0;                                               // the last list that starts at or before f (empty lists share a start)
#pragma unroll
        for (int step = 16; step > 0; step >>= 1)
            if (offs[u + step] <= f) u += step;
        const int u_first = __builtin_amdgcn_readfirstlane(u);
        const int u_last = __builtin_amdgcn_readlane(u, min(63, total - base - 1));
        if (staged && u_last - u_first < kStageUsers) {
            // The fast pass (k = 16, 32, 64 or 128; at most kStageUsers lists under the 64 candidates: the rule from K = 16 on).
            // exact_score runs four dependent memory round trips per candidate (32 of its 64 row loads fit the registers at a time)
            // and the final stage is bound by exactly that chain (68 round trips per wave: 16 % of the Netflix pass, 30 % of the
            // ML-10M pass).  Here ALL the loads of the candidate's item row go out first (128 registers), the few user rows of the
            // pass follow through LDS (every lane of a user reads the same addresses: broadcasts), and the fma chain -- the same
            // chain, element for element -- starts after ONE round trip.
            const bool live = f < total;
            const int e = f - offs[u], uq = wave * 32 + u;
            const int col = live ? (int)sm.ci[e * sm.users + uq] : 0;
            constexpr int q4 = 2 * KS;                           // float4 per k-half
            constexpr int NF = q4 < IN_FLIGHT ? q4 : IN_FLIGHT;       // float4 of each half in flight at a time
            const float4* vrow = reinterpret_cast<const float4*>(Vt + (size_t)col * (16 * KS));
            float4 va[NF], vb[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) { va[j] = vrow[j]; vb[j] = vrow[q4 + j]; }
            constexpr int per_row = 4 * KS;
            const int n_stage = (u_last - u_first + 1) * per_row;      // float4 of the staged rows: <= 128
            // (two named registers, not an array: a conditionally written array stays a stack object -- scratch)
            auto stage_ptr = [&](int idx) {
                const int sr = idx / per_row, sc4 = idx - sr * per_row;
                const int rr = min(ws.block * sm.users + wave * 32 + u_first + sr, n_rows - 1);
                return reinterpret_cast<const float4*>(U + (size_t)(uidx ? uidx[rr] : rr) * (16 * KS)) + sc4;
            };
            float4 su0 = make_float4(0.f, 0.f, 0.f, 0.f), su1 = su0;
            if (lane < n_stage) su0 = *stage_ptr(lane);
            if (lane + 64 < n_stage) su1 = *stage_ptr(lane + 64);
            if (lane < n_stage) reinterpret_cast<float4*>(ust)[lane] = su0;
            if (lane + 64 < n_stage) reinterpret_cast<float4*>(ust)[lane + 64] = su1;
            __builtin_amdgcn_wave_barrier();
            const float4* urow = reinterpret_cast<const float4*>(ust + (size_t)(u - u_first) * k);
            float acc = 0.f;
#pragma unroll 1
            for (int j0 = 0; j0 < q4; j0 += NF) {
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const float4 b0 = urow[j0 + j], b1 = urow[q4 + j0 + j];
                    acc = fmaf(va[j].x, b0.x, acc); acc = fmaf(vb[j].x, b1.x, acc);
                    acc = fmaf(va[j].y, b0.y, acc); acc = fmaf(vb[j].y, b1.y, acc);
                    acc = fmaf(va[j].z, b0.z, acc); acc = fmaf(vb[j].z, b1.z, acc);
                    acc = fmaf(va[j].w, b0.w, acc); acc = fmaf(vb[j].w, b1.w, acc);
                }
                if (j0 + NF < q4) {
#pragma unroll
                    for (int j = 0; j < NF; ++j) { va[j] = vrow[j0 + NF + j]; vb[j] = vrow[q4 + j0 + NF + j]; }
                }
            }
            acc = acc + (bias ? bias[col] : 0.f);
            if (live) sm.cs[e * sm.users + uq] = acc + 0.0f;
            __builtin_amdgcn_wave_barrier();                     // the next pass overwrites the staged rows
        } else if (f < total) {
            const int e = f - offs[u], uq = wave * 32 + u, r = ws.block * sm.users + uq;
            const int col = (int)sm.ci[e * sm.users + uq];
            sm.cs[e * sm.users + uq] = exact_score(U + (size_t)(uidx ? uidx[r] : r) * k, Vt + (size_t)col * k, k, bias, col);
        }
    }
    __builtin_amdgcn_wave_barrier();
    write_rows<IdT>(sm, ws, n_rows, K, -INFINITY, out_ids, out_scores, part);
}

// waves per workgroup of an instantiation: 8 (two per SIMD: one wave's filter overlaps the other's MFMA
// chain); 6 when candidate ids need 32 bits (LDS); 4 for wide factor rows (100+ operand registers)
template <int KHP, typename IdT>
constexpr int topk_waves() { return KHP > 64 ? 4 : (sizeof(IdT) == 2 ? kTopkMaxWaves : 6); }

// WAVES: waves per workgroup; the default of the width, or the bound-and-refine kernel's (its flagged user blocks are redone
// here on the SAME work items: both kernels must cut the users into the same blocks)
template <int KHP, typename IdT, int WAVES = topk_waves<KHP, IdT>()>
__global__ __launch_bounds__((WAVES * TKR_WAVE)) void score_topk_kernel(
    const float* __restrict__ U, const int32_t* __restrict__ uidx, int n_rows, const float* __restrict__ Vt,
    const float* __restrict__ bias, int n_cols, int k, const uint32_t* __restrict__ mask, int mask_pitch, int K,
    int32_t* __restrict__ out_ids, float* __restrict__ out_scores, int tiles_per_split,
    uint64_t* __restrict__ part /*[n_rows][gridDim.y][K] sorted keys, when gridDim.y > 1*/,
    uint32_t* __restrict__ thr_shared /*[n_rows] ordered bits of a lower bound of the row's K-th best score, or null*/,
    const int4* __restrict__ items /*balanced item table (block, t_begin, t_end, slot | stride << 16), or null: the grid*/,
    const uint32_t* __restrict__ only_flagged /*per user block, or null: rank every block.  Set by the bound-and-refine kernel
                                                for blocks it could not finish*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KP = 2 * KHP + 4;                              // padded LDS row (floats): conflict-free b128 reads
    const int W = blockDim.x >> 6;
    const int users = W * 32;
    TopkSmem<IdT> sm;
    sm.tile = reinterpret_cast<float*>(smem_raw);
    sm.tbias = sm.tile + 2 * 32 * KP;
    sm.cnt = reinterpret_cast<int*>(sm.tbias + 64);
    sm.cs = reinterpret_cast<float*>(sm.cnt + users);
    sm.ci = reinterpret_cast<IdT*>(sm.cs + (size_t)users * kCap);
    sm.users = users;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = lane & 31, h = lane >> 5;
    const int uw = wave * 32 + ul;                               // user slot inside the workgroup
    int4 it = make_int4((int)blockIdx.x, (int)blockIdx.y * tiles_per_split, 0, (int)blockIdx.y | ((int)gridDim.y << 16));
    if (items) it = items[blockIdx.x];
    if (only_flagged && only_flagged[it.x] == 0u) return;
    const TopkSlot ws = {it.x, it.w & 0xffff, it.w >> 16};
    const int row = ws.block * users + uw;                       // row of the output / index into uidx
    const bool user_ok = row < n_rows;
    const int KH = (k + 1) >> 1;                                 // k-range of half h: [h*KH, min(k, (h+1)*KH))

    // ---- B operand: this lane's user, half h of its factor row, resident for the whole kernel
    float breg[KHP];
    {
        const int urow = user_ok ? (uidx ? uidx[row] : row) : 0;
        const float* up = U + (size_t)urow * k + h * KH;
#pragma unroll
        for (int kk = 0; kk < KHP; ++kk) {
            const int e = h * KH + kk;
            breg[kk] = (user_ok && kk < KH && e < k) ? up[kk] : 0.f;
        }
    }
    for (int s = tid; s < users; s += blockDim.x) sm.cnt[s] = 0;
    float thr = (thr_shared && user_ok) ? unordered_bits(thr_shared[row]) : -INFINITY;   // what other item ranges found so far
    const int n_tiles_all = (n_cols + 31) >> 5;
    const int t_begin = it.y;                                    // this workgroup ranks items of tiles [t_begin, n_tiles)
    const int n_tiles = items ? it.z : min(n_tiles_all, t_begin + tiles_per_split);

    // ---- tile staging: global -> registers (issued early) -> LDS (written after the MFMA chain) ------
    // float4 path when every k-half starts 16-B aligned (k % 8 == 0); scalar path otherwise.
    constexpr int NT_ = WAVES * 64;
    constexpr int NC = (32 * 2 * KHP / 4 + NT_ - 1) / NT_;         // float4 chunks per thread
    const bool vec = (k & 7) == 0;                                // then KH % 4 == 0: no chunk straddles the halves
    const int nthreads = blockDim.x;
    const int k4 = k >> 2;
    int src_off[NC], dst_off[NC], item_of[NC];                    // per-chunk offsets, fixed for the whole kernel
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = tid + q * nthreads;
        const int item = vec ? c / k4 : 0, e = vec ? (c % k4) * 4 : 0;
        const bool live = vec && c < 32 * k4;
        src_off[q] = live ? item * k + e : -1;
        dst_off[q] = item * KP + (e / KH) * KHP + (e % KH);
        item_of[q] = item;
    }
    float4 stg[NC];
    float stg_bias = 0.f;
    auto stage_load = [&](int t) {                                // tile t -> registers
        if (vec) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                stg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src_off[q] >= 0 && t * 32 + item_of[q] < n_cols)
                    stg[q] = *reinterpret_cast<const float4*>(Vt + (size_t)t * 32 * k + src_off[q]);
            }
        }
        if (tid < 32) {
            const int col = t * 32 + tid;
            stg_bias = (bias && col < n_cols) ? bias[col] : 0.f;
        }
    };
    auto stage_store = [&](int t, int buf) {                      // registers (or, scalar path, global) -> LDS
        float* dst = sm.tile + buf * 32 * KP;
        if (vec) {
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (src_off[q] >= 0) *reinterpret_cast<float4*>(dst + dst_off[q]) = stg[q];
        } else {
            for (int c = tid; c < 32 * 2 * KHP; c += nthreads) {
                const int item = c / (2 * KHP), p = c % (2 * KHP);
                const int hh = p / KHP, kk = p % KHP;
                const int e = hh * KH + kk, col = t * 32 + item;
                float v = 0.f;
                if (kk < KH && e < k && col < n_cols) v = Vt[(size_t)col * k + e];
                dst[item * KP + p] = v;
            }
        }
        if (tid < 32) sm.tbias[buf * 32 + tid] = stg_bias;
    };
    if (vec) {                                                    // zero the LDS padding the vector path never writes
        for (int c = tid; c < 2 * 32 * KP; c += nthreads) sm.tile[c] = 0.f;
        __syncthreads();
    }
    stage_load(t_begin);
    stage_store(t_begin, t_begin & 1);
    __syncthreads();
    if (t_begin + 1 < n_tiles) stage_load(t_begin + 1);

    const uint32_t tail_mask = (n_cols & 31) ? (0xffffffffu << (n_cols & 31)) : 0u;
    int next_sched = t_begin + 2;                                // scheduled trims after 2, 3, 5, 8, 12, ... tiles (x1.5)
    for (int t = t_begin; t < n_tiles; ++t) {
        const int buf = t & 1;
        // rated / non-existent columns of this (user, tile) as one word; issued before the MFMA chain
        uint32_t maskw = (mask && user_ok) ? mask[(size_t)t * mask_pitch + row] : 0u;
        // ---- 32 items x 32 users x k: exact fp32 MFMA ------------------------------------------
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const float* arow = sm.tile + buf * 32 * KP + ul * KP + h * KHP;
#pragma unroll
        for (int kk = 0; kk < KHP; kk += 4) {
            const float4 a = *reinterpret_cast<const float4*>(arow + kk);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, breg[kk + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, breg[kk + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, breg[kk + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, breg[kk + 3], acc, 0, 0, 0);
        }
        mfma_result_guard(acc);
        // tile t+1: registers -> the other LDS buffer (free since the barrier of tile t-1); tile t+2 -> registers
        if (t + 1 < n_tiles) stage_store(t + 1, buf ^ 1);
        if (t + 2 < n_tiles) stage_load(t + 2);
        // ---- epilogue: bias, mask, threshold filter ----------------------------------------------
        if (!user_ok) maskw = 0xffffffffu;
        if (t == n_tiles_all - 1) maskw |= tail_mask;
        if (t == next_sched) {                                   // workgroup-uniform: every wave trims all its users now
            if (__ballot(sm.cnt[uw] > kCap / 2) != 0) {            // lists still short (thresholds shared by earlier ranges): nothing to gain
                thr = trim_all_users<IdT>(sm, uw, h, K, thr);
                thr = share_threshold(thr_shared, row, user_ok && h == 0, thr);
            }
            next_sched = t + ((t - t_begin + 1) >> 1);
        }
        filter_tile<IdT>(sm, acc, sm.tbias + buf * 32, maskw, t, K, thr);
        __syncthreads();                                         // tile t+1 staged; buffer `buf` may be overwritten next
    }

    write_rows<IdT>(sm, ws, n_rows, K, thr, out_ids, out_scores, part);
}

// ---- factor widths above 256: the k dimension in slabs of 256 --------------------------------------------------------------------
// evaluate.py:78 takes any width and the trainer here goes to k = 512.  score_topk_kernel keeps a lane's half of its user's factor
// row in registers (128 at k = 256) and a 32-item tile of the full width in LDS (2 x 33 KB at k = 256): neither scales.  Here the
// user operand of EVERY slab stays in registers (SLABS x 128: the 512-register file of a one-wave-per-SIMD workgroup holds two or
// three slabs) and the item tiles go through the same two LDS buffers one slab at a time -- unit (tile, slab) is staged while unit
// (tile, slab - 1) is multiplied; the accumulator runs over a tile's slabs and the filter sees it after the last.  Same fp32 MFMA
// chain, same lists, trims and merge as score_topk_kernel.
template <int SLABS, typename IdT>
__global__ __launch_bounds__(256) void score_topk_slab_kernel(
    const float* __restrict__ U, const int32_t* __restrict__ uidx, int n_rows, const float* __restrict__ Vt,
    const float* __restrict__ bias, int n_cols, int k, const uint32_t* __restrict__ mask, int mask_pitch, int K,
    int32_t* __restrict__ out_ids, float* __restrict__ out_scores, int tiles_per_split, uint64_t* __restrict__ part,
    uint32_t* __restrict__ thr_shared, const int4* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int KHP = 128, SW = 2 * KHP;                       // a slab: 256 factors, 128 per half-wave
    constexpr int KP = SW + 4;
    constexpr int W = 4, users = W * 32, NT_ = W * 64;
    TopkSmem<IdT> sm;
    sm.tile = reinterpret_cast<float*>(smem_raw);
    sm.tbias = sm.tile + 2 * 32 * KP;
    sm.cnt = reinterpret_cast<int*>(sm.tbias + 64);
    sm.cs = reinterpret_cast<float*>(sm.cnt + users);
    sm.ci = reinterpret_cast<IdT*>(sm.cs + (size_t)users * kCap);
    sm.users = users;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = lane & 31, h = lane >> 5;
    const int uw = wave * 32 + ul;
    int4 it = make_int4((int)blockIdx.x, (int)blockIdx.y * tiles_per_split, 0, (int)blockIdx.y | ((int)gridDim.y << 16));
    if (items) it = items[blockIdx.x];
    const TopkSlot ws = {it.x, it.w & 0xffff, it.w >> 16};
    const int row = ws.block * users + uw;
    const bool user_ok = row < n_rows;

    float breg[SLABS * KHP];                                     // element (slab s, kk) = factor 256 s + 128 h + kk of this lane's user
    {
        const int urow = user_ok ? (uidx ? uidx[row] : row) : 0;
        const float* up = U + (size_t)urow * k;
#pragma unroll
        for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
            for (int kk = 0; kk < KHP; ++kk) {
                const int e = sl * SW + h * KHP + kk;
                breg[sl * KHP + kk] = (user_ok && e < k) ? up[e] : 0.f;
            }
    }
    for (int s = tid; s < users; s += NT_) sm.cnt[s] = 0;
    float thr = (thr_shared && user_ok) ? unordered_bits(thr_shared[row]) : -INFINITY;
    const int n_tiles_all = (n_cols + 31) >> 5;
    const int t_begin = it.y;
    const int n_tiles = items ? it.z : min(n_tiles_all, t_begin + tiles_per_split);

    // ---- staging of unit (tile t, slab sl): global -> registers -> LDS buffer (unit index & 1)
    constexpr int NC = 32 * SW / 4 / NT_;                        // 8 float4 per thread
    const bool vec = (k & 3) == 0;                               // every item row and slab starts 16-byte aligned
    float4 stg[NC];
    float stg_bias = 0.f;
    auto stage_load = [&](int t, int sl) {
        if (vec) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = tid + q * NT_, item = c >> 6, e = sl * SW + (c & 63) * 4;
                stg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < k && t * 32 + item < n_cols) stg[q] = *reinterpret_cast<const float4*>(Vt + (size_t)(t * 32 + item) * k + e);
            }
        }
        if (sl == 0 && tid < 32) {
            const int col = t * 32 + tid;
            stg_bias = (bias && col < n_cols) ? bias[col] : 0.f;
        }
    };
    auto stage_store = [&](int t, int sl, int buf) {
        float* dst = sm.tile + buf * 32 * KP;
        if (vec) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int c = tid + q * NT_;
                *reinterpret_cast<float4*>(dst + (c >> 6) * KP + (c & 63) * 4) = stg[q];
            }
        } else {
            for (int c = tid; c < 32 * SW; c += NT_) {
                const int item = c / SW, e = sl * SW + c % SW, col = t * 32 + item;
                dst[item * KP + c % SW] = (e < k && col < n_cols) ? Vt[(size_t)col * k + e] : 0.f;
            }
        }
        if (sl == 0 && tid < 32) sm.tbias[(t & 1) * 32 + tid] = stg_bias;
    };
    // unit n = (t - t_begin) * SLABS + sl; the one after (t, sl):
    auto next_unit = [&](int& t, int& sl) { if (++sl == SLABS) { sl = 0; ++t; } };

    stage_load(t_begin, 0);
    stage_store(t_begin, 0, 0);
    __syncthreads();
    {
        int t1 = t_begin, s1 = 0;
        next_unit(t1, s1);
        if (t1 < n_tiles) stage_load(t1, s1);
    }
    const uint32_t tail_mask = (n_cols & 31) ? (0xffffffffu << (n_cols & 31)) : 0u;
    int next_sched = t_begin + 2;
    int unit = 0;
    for (int t = t_begin; t < n_tiles; ++t) {
        uint32_t maskw = (mask && user_ok) ? mask[(size_t)t * mask_pitch + row] : 0u;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int sl = 0; sl < SLABS; ++sl, ++unit) {
            const int buf = unit & 1;
            const float* arow = sm.tile + buf * 32 * KP + ul * KP + h * KHP;
#pragma unroll
            for (int kk = 0; kk < KHP; kk += 4) {
                const float4 a = *reinterpret_cast<const float4*>(arow + kk);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, breg[sl * KHP + kk + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, breg[sl * KHP + kk + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, breg[sl * KHP + kk + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, breg[sl * KHP + kk + 3], acc, 0, 0, 0);
            }
            if (sl + 1 == SLABS) mfma_result_guard(acc);
            // the next unit: registers -> the other buffer (free since the barrier before this unit); the one after it -> registers
            int t1 = t, s1 = sl;
            next_unit(t1, s1);
            if (t1 < n_tiles) stage_store(t1, s1, buf ^ 1);
            int t2 = t1, s2 = s1;
            next_unit(t2, s2);
            if (t1 < n_tiles && t2 < n_tiles) stage_load(t2, s2);
            if (sl + 1 < SLABS) __syncthreads();
        }
        if (!user_ok) maskw = 0xffffffffu;
        if (t == n_tiles_all - 1) maskw |= tail_mask;
        if (t == next_sched) {
            if (__ballot(sm.cnt[uw] > kCap / 2) != 0) {
                thr = trim_all_users<IdT>(sm, uw, h, K, thr);
                thr = share_threshold(thr_shared, row, user_ok && h == 0, thr);
            }
            next_sched = t + ((t - t_begin + 1) >> 1);
        }
        filter_tile<IdT>(sm, acc, sm.tbias + (t & 1) * 32, maskw, t, K, thr);
        __syncthreads();
    }
    write_rows<IdT>(sm, ws, n_rows, K, thr, out_ids, out_scores, part);
}

// ---- K4 on the dense matrix pipe: 6-product bf16 split of the fp32 factors ------------------------------------
// Every fp32 factor is split EXACTLY into three bf16 parts a = a1 + a2 + a3 (8 significant bits each, round to
// nearest at each step; bf16 has fp32's exponent range).  The product a*b is taken as the six partial products
// a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (the dropped a2b3, a3b2, a3b3 are below 2^-23 |ab|), each EXACT in fp32
// (8 x 8 significant bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- the rounding behaviour of an fp32
// dot product with a different summation order (measured error against fp64: same order as the fp32-MFMA path and as
// numpy's sgemm, tests/test_gpu_topk.py).  Inputs whose products and partial sums are representable (the golden
// fixtures) give exactly the same scores as the fp32 path.  Why: 48 bf16 MFMAs of 8 passes replace 64 fp32 MFMAs of
// 16 passes per 32x32xk=128 block (1.8 us against 3.4 us per 256-user tile, scripts/ubench/bf16x3_ubench.hip), and on
// gfx950 the fp32 MFMA does not overlap other work of the SIMD at all.  Finite inputs only (inf - inf in the split).
__device__ __forceinline__ void split3(float a, __bf16& p1, __bf16& p2, __bf16& p3) {
    p1 = (__bf16)a;
    const float r1 = a - (float)p1;
    p2 = (__bf16)r1;
    p3 = (__bf16)(r1 - (float)p2);
}

// waves per workgroup.  bf16x3: 8 (one workgroup per CU).  Bound-and-refine: 4 -- TWO workgroups per CU (LDS: 17 KB of tiles +
// 48 KB of lists each; 256 VGPRs per wave either way).  The one barrier of a tile makes every wave wait for the slowest filter of
// its workgroup; with four waves per barrier instead of eight, and a second workgroup to run while one waits, the Netflix
// shape went 10.18 -> 9.59 ms (ML-10M 1.44 -> 1.35).  The three-part tiles of bf16x3 do not fit twice (measured: 14.4 -> 30.8 ms).
template <int KS, typename IdT, bool REFINE = false>
constexpr int topk_waves_bf16() { return sizeof(IdT) == 2 ? (REFINE ? 4 : kTopkMaxWaves) : 6; }

// REFINE = bound-and-refine arithmetic (tkr_topk_set_math(2)): ONE fp16 product per element instead of six bf16 ones
// (v_mfma_f32_32x32x16_f16, the same rate).  fp16 has 11 significant bits but a narrow exponent range, so both sides are
// scaled by powers of two first: the item factors by sv (max |V| lands in [2^13, 2^14)), each user's row by its own su;
// the filter, the lists and the thresholds of a user live in these scaled units (scale = su * sv; the bias enters as
// fma(bias, scale, acc)).  In them the approximate score is within
//   margin = scale * (1.05 * 2^-10 * |u| * max_i |v_i| + 2^-18 * (|u| * max|v| + max|bias|)) + 2.01 * k
// of the exact fp32 score of exact_score(): each factor is rounded by <= 2^-11 relative -- or, below fp16's normal range,
// by <= 2^-14 absolute even if the matrix pipe flushes subnormals (the 2.01 * k term: 2 * k * 2^14 * 2^-14) --, Cauchy-Schwarz
// bounds the sum, and the 2^-18 term covers the fp32 roundings of both accumulations and of the bias add.  A list that keeps
// everything within 2 * margin of the K-th best approximate score therefore holds the exact best K; they are rescored exactly
// at the end.  `extra`: bits of [0] max |v_i|, [1] max |bias|, [2] max |V element| (topk_bounds_kernel); [4 + block] = 1
// when a list of the block overflowed.
// IMG (bound-and-refine only): the scaled fp16 item factors come PRE-CONVERTED, as the LDS image of every 32-item tile
// (topk_image_kernel below; `vimg`), and a tile is staged by ONE direct-to-LDS load per thread (global_load_lds_dwordx4): no
// staging registers, no conversion, no ds_write.  Without it every workgroup converts the same V again -- 1,876 times at the
// Netflix shape, 2.0 of 10.85 ms (scripts/ablate_topk.sh).  The direct load writes LDS linearly (wave base + 16 * lane), so the
// image carries no row padding; bank conflicts of the fragment reads are avoided by an XOR swizzle of the 16-byte chunks of a
// row instead (chunk j of item row r sits at j ^ swz(r), identical in the image and in the read: the b128 lane groups of
// MI355X_MICROARCH.md hold 16 distinct values of r & 15).
template <int KS>
__global__ __launch_bounds__(256) void topk_image_kernel(const float* __restrict__ Vt, int n_cols, int k, const uint32_t* __restrict__ extra,
                                                        unsigned char* __restrict__ vimg) {
    constexpr int CR = 2 * KS;
    const int n_tiles = (n_cols + 31) >> 5;
    const float sv = pow2_scale(__uint_as_float(extra[2]));
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < (long long)n_tiles * 32 * CR; c += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(c % CR), r = (int)((c / CR) & 31);
        const long long t = c / (32 * CR);
        const long long col = t * 32 + r;
        f16x8 out;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 8 * j + i;
            out[i] = (_Float16)((col < n_cols && e < k) ? Vt[col * k + e] * sv : 0.f);
        }
        *reinterpret_cast<f16x8*>(vimg + ((t * 32 + r) * CR + (j ^ tile_swizzle<KS>(r))) * 16) = out;
    }
}

template <int KS, typename IdT, bool REFINE = false, bool IMG = false>
__global__ __launch_bounds__((topk_waves_bf16<KS, IdT, REFINE>() * TKR_WAVE), 2) void score_topk_bf16_kernel(
    const float* __restrict__ U, const int32_t* __restrict__ uidx, int n_rows, const float* __restrict__ Vt,
    const float* __restrict__ bias, int n_cols, int k, const uint32_t* __restrict__ mask, int mask_pitch, int K,
    int32_t* __restrict__ out_ids, float* __restrict__ out_scores, int tiles_per_split, uint64_t* __restrict__ part,
    uint32_t* __restrict__ thr_shared, const int4* __restrict__ items /*(block, t_begin, t_end, slot | stride << 16) or null*/,
    uint32_t* __restrict__ extra, const unsigned char* __restrict__ vimg) {
    static_assert(!IMG || REFINE, "the tile image is the scaled fp16 operand of the bound-and-refine arithmetic");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#if TKR_ABL & 256
    const unsigned long long k4_start = __builtin_amdgcn_s_memtime();
#endif
    constexpr int NPART = REFINE ? 1 : 3;
    constexpr int PARTB = KS * 32;                               // bytes of one bf16 part of an item row (KS*16 elements)
    constexpr int ROWB = IMG ? PARTB : NPART * PARTB + 16;       // padded row: conflict-free ds_read_b128 (ROWB/4 = 4 mod 8); IMG: swizzled instead
    constexpr int KPAD = KS * 16;
    const int W = blockDim.x >> 6;
    const int users = W * 32;
    TopkSmem<IdT> sm;
    unsigned char* tile = smem_raw;                              // [2][32][ROWB]
    sm.tile = nullptr;
    sm.tbias = reinterpret_cast<float*>(smem_raw + 2 * 32 * ROWB);
    sm.cnt = reinterpret_cast<int*>(sm.tbias + 64);
    sm.cs = reinterpret_cast<float*>(sm.cnt + users);
    sm.ci = reinterpret_cast<IdT*>(sm.cs + (size_t)users * kCap);
    sm.users = users;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ul = lane & 31, h = lane >> 5;                     // h = k-group of the operands AND row group of the result
    const int uw = wave * 32 + ul;
    // work item: a (user block, tile range) from the balanced item table, or from the grid
    int4 it = make_int4((int)blockIdx.x, (int)blockIdx.y * tiles_per_split, 0, (int)blockIdx.y | ((int)gridDim.y << 16));
    if (items) it = items[blockIdx.x];
    const TopkSlot ws = {it.x, it.w & 0xffff, it.w >> 16};
    const int row = ws.block * users + uw;
    const bool user_ok = row < n_rows;

    // ---- B operand: lane (user ul, k-group h) holds elements 16s + 8h .. +7 of its user's row, three parts each
    bf16x8 breg[REFINE ? 1 : KS][NPART];
    f16x8 hreg[REFINE ? KS : 1];
    float margin = 0.f, bscale = 1.f, inv_bscale = 1.f, sv = 1.f;
    {
        const int urow = user_ok ? (uidx ? uidx[row] : row) : 0;
        const float* up = U + (size_t)urow * k;
        float uv[KS][8];
        if ((k & 3) == 0) {                                      // two 16-byte loads per 16-wide k step, all issued before use
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
        if constexpr (REFINE) {
            float nu = 0.f, amax = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    nu = fmaf(uv[s][i], uv[s][i], nu);
                    amax = fmaxf(amax, fabsf(uv[s][i]));
                }
            nu += __shfl_xor(nu, 32, 64);                          // the other k-group of the same user
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            const float su = pow2_scale(amax);
            sv = pow2_scale(__uint_as_float(extra[2]));
            bscale = su * sv;
            inv_bscale = __uint_as_float((254u - ((__float_as_uint(bscale) >> 23) & 0xffu)) << 23);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) hreg[s][i] = (_Float16)(uv[s][i] * su);
            const float reach = sqrtf(nu) * 1.001f * __uint_as_float(extra[0]);      // >= |u| * max |v_i|
            margin = (fmaf(1.05f * 0.0009765625f, reach, 3.8146973e-6f * (reach + __uint_as_float(extra[1]))) + 7.7e-34f) * bscale +
                     2.01f * (float)KPAD;
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __bf16 p1, p2, p3;
                    split3(uv[s][i], p1, p2, p3);
                    breg[s][0][i] = p1; breg[s][NPART > 1 ? 1 : 0][i] = p2; breg[s][NPART > 2 ? 2 : 0][i] = p3;
                }
        }
    }
    const float m2 = 2.f * margin;
    bool lost = false;
    for (int s = tid; s < users; s += blockDim.x) sm.cnt[s] = 0;
    float thr = (thr_shared && user_ok) ? unordered_bits(thr_shared[row]) : -INFINITY;   // what other item ranges found so far
    if constexpr (REFINE) thr = thr * bscale - margin;
    const int n_tiles_all = (n_cols + 31) >> 5;
    const int t_begin = it.y;
    const int n_tiles = items ? it.z : min(n_tiles_all, t_begin + tiles_per_split);

    // ---- tile staging: float4 of Vt -> registers (early) -> three 4 x bf16 parts -> LDS (after the MFMA chain)
    constexpr int NT_ = topk_waves_bf16<KS, IdT, REFINE>() * 64;
    constexpr int TILEB = 32 * ROWB;                             // bytes of one staged tile
    constexpr int NG = IMG ? (TILEB / 16 + NT_ - 1) / NT_ : 1;   // IMG: 16-byte direct-to-LDS loads per thread and tile
    auto stage_direct = [&](int t, int buf) {                     // IMG: tile t of the image -> LDS buffer buf, asynchronously (vmcnt)
        if constexpr (IMG) {
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const int c0 = q * NT_ + wave * 64;              // first chunk of this wave's 64 (LDS destination = wave base + 16 * lane)
                if (c0 < TILEB / 16) {
                    const int c = min(c0 + lane, TILEB / 16 - 1);
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(vimg + (size_t)t * TILEB + (size_t)c * 16),
                        (__attribute__((address_space(3))) void*)(smem_raw + buf * TILEB + c0 * 16), 16, 0, 0);
                }
            }
            if (tid < 32) {
                const int col = t * 32 + tid;
                sm.tbias[buf * 32 + tid] = (bias && col < n_cols) ? bias[col] : 0.f;
            }
        }
    };
    constexpr int NC = IMG ? 1 : (32 * KPAD / 4 + NT_ - 1) / NT_;          // float4 chunks per thread
    const bool vec = (k & 3) == 0;
    const int nthreads = blockDim.x;
    const int k4 = k >> 2;
    int src_off[NC], dst_off[NC], item_of[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = tid + q * nthreads;
        const int item = vec ? c / k4 : 0, e = vec ? (c % k4) * 4 : 0;
        const bool live = vec && c < 32 * k4;
        src_off[q] = live ? item * k + e : -1;
        dst_off[q] = item * ROWB + e * 2;
        item_of[q] = item;
    }
    float4 stg[NC];
    float stg_bias = 0.f;
    auto stage_load = [&](int t) {
        if (vec) {
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                stg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src_off[q] >= 0 && t * 32 + item_of[q] < n_cols)
                    stg[q] = *reinterpret_cast<const float4*>(Vt + (size_t)t * 32 * k + src_off[q]);
            }
        }
        if (tid < 32) {
            const int col = t * 32 + tid;
            stg_bias = (bias && col < n_cols) ? bias[col] : 0.f;
        }
    };
    auto stage_store = [&](int t, int buf) {
        unsigned char* dst = tile + buf * 32 * ROWB;
        if (vec) {
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (src_off[q] >= 0) {
                    const float a[4] = {stg[q].x, stg[q].y, stg[q].z, stg[q].w};
                    if constexpr (REFINE) {
                        f16x4 p1;
#pragma unroll
                        for (int i = 0; i < 4; ++i) p1[i] = (_Float16)(a[i] * sv);
                        *reinterpret_cast<f16x4*>(dst + dst_off[q]) = p1;
                    } else {
                        bf16x4 p1, p2, p3;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __bf16 x1, x2, x3;
                            split3(a[i], x1, x2, x3);
                            p1[i] = x1; p2[i] = x2; p3[i] = x3;
                        }
                        *reinterpret_cast<bf16x4*>(dst + dst_off[q]) = p1;
                        *reinterpret_cast<bf16x4*>(dst + dst_off[q] + (NPART - 1) / 2 * PARTB) = p2;
                        *reinterpret_cast<bf16x4*>(dst + dst_off[q] + (NPART - 1) * PARTB) = p3;
                    }
                }
        } else {
            for (int c = tid; c < 32 * KPAD; c += nthreads) {
                const int item = c / KPAD, e = c % KPAD, col = t * 32 + item;
                float v = 0.f;
                if (e < k && col < n_cols) v = Vt[(size_t)col * k + e];
                __bf16* rowp = reinterpret_cast<__bf16*>(dst + item * ROWB);
                if constexpr (REFINE) {
                    reinterpret_cast<_Float16*>(rowp)[e] = (_Float16)(v * sv);
                } else {
                    __bf16 x1, x2, x3;
                    split3(v, x1, x2, x3);
                    rowp[e] = x1; rowp[(NPART - 1) / 2 * KPAD + e] = x2; rowp[(NPART - 1) * KPAD + e] = x3;
                }
            }
        }
        if (tid < 32) sm.tbias[buf * 32 + tid] = stg_bias;
    };
    if constexpr (IMG) {
        stage_direct(t_begin, t_begin & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        if (vec) {                                                // zero what the vector path never writes (k..KPAD, padding)
            for (int c = tid; c < 2 * 32 * ROWB / 4; c += nthreads) reinterpret_cast<uint32_t*>(tile)[c] = 0u;
            __syncthreads();
        }
        stage_load(t_begin);
        stage_store(t_begin, t_begin & 1);
        __syncthreads();
        if (t_begin + 1 < n_tiles) stage_load(t_begin + 1);
    }


    const uint32_t tail_mask = (n_cols & 31) ? (0xffffffffu << (n_cols & 31)) : 0u;
    int next_sched = t_begin + 2;
#if TKR_ABL & 256
    unsigned long long k4p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k4t = k4_start;
    K4_MARK(5)
#endif
    // the rated-item word of (tile, user) comes from HBM (every workgroup reads its own 512 bytes per tile: never a cache hit): it
    // is asked for one tile ahead, right behind the barrier, and has the filter of this tile and the chain of the next to arrive
    uint32_t mask_next = (mask && user_ok && t_begin < n_tiles) ? mask[(size_t)t_begin * mask_pitch + row] : 0u;
    for (int t = t_begin; t < n_tiles; ++t) {
        const int buf = t & 1;
        uint32_t maskw = mask_next;
        // ---- 32 items x 32 users x k: six bf16 partial products per 16-wide k step, fp32 accumulation.
        // A operand: lane (item ul, k-group h) reads elements 16s + 8h .. +7 of each part; small terms first.
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned char* arow = tile + buf * 32 * ROWB + ul * ROWB + h * 16;
        // IMG: tile t+1 streams into the other buffer while this one is multiplied (nobody reads that buffer after the barrier
        // of tile t-1; the bias of a tile is folded into the accumulator before its barrier, so tbias is free as well)
        if constexpr (IMG) { if (t + 1 < n_tiles) stage_direct(t + 1, buf ^ 1); }
        if constexpr (REFINE) {                                   // one MFMA per fragment: all the LDS reads go out first
            f16x8 afrag[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if constexpr (IMG) afrag[s] = *reinterpret_cast<const f16x8*>(tile + buf * TILEB + ul * ROWB + ((2 * s + h) ^ tile_swizzle<KS>(ul)) * 16);
                else afrag[s] = *reinterpret_cast<const f16x8*>(arow + s * 32);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[s], hreg[s], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);   // the scheduler otherwise reloads one fragment register before every MFMA
            __builtin_amdgcn_sched_group_barrier(0x008, KS, 0);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if constexpr (!REFINE) {
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(arow + s * 32);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(arow + (NPART - 1) / 2 * PARTB + s * 32);
                const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(arow + (NPART - 1) * PARTB + s * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, breg[s][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, breg[s][NPART > 1 ? 1 : 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[s][NPART > 2 ? 2 : 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, breg[s][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[s][NPART > 1 ? 1 : 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[s][0], acc, 0, 0, 0);
            }
        }
        mfma_result_guard_8pass(acc);
        // Where the one barrier of the tile sits.  k <= 64: EARLY, right after the staging -- tile t+1 is in LDS, every wave
        // is done reading tile t, and the filter touches only the wave's own lists, so a wave whose filter is short starts
        // the next MFMA chain while its neighbours still append or trim (Netflix shape, k = 64: 11.2 -> 10.2 ms).  The bias
        // is folded into the accumulator first: tile t+2's bias may land in this slot before a slow wave filters.
        // k = 128: measured slower that way (14.4 -> 17.7 ms) -- the MFMA chains dominate there and the two waves of a SIMD
        // interleave them best when they start together (a lone dependent chain issues at ~44 instead of 32 cycles per
        // MFMA); there the barrier stays behind the filter.
        constexpr bool kEarlyBarrier = REFINE || KS <= 4;    // refine: the chain is 8 MFMAs at every width, the filter dominates
        if constexpr (kEarlyBarrier) {
            if constexpr (REFINE) {
                if (bias) add_scaled_bias_inplace(acc, sm.tbias + buf * 32, h, bscale);     // VBPR models have no item bias: 16 fma per tile less
            }
            else add_bias_inplace(acc, sm.tbias + buf * 32, h);
        }
#if TKR_ABL & 256
        asm volatile("" : "+v"(acc));
        K4_MARK(0)
#endif
#if !(TKR_ABL & 2)
        if constexpr (!IMG) {
            if (t + 1 < n_tiles) stage_store(t + 1, buf ^ 1);
            if (t + 2 < n_tiles) stage_load(t + 2);
        }
#endif
        if constexpr (IMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t+1 has landed (issued a whole MFMA chain ago)
        K4_MARK(1)
        if constexpr (kEarlyBarrier) __syncthreads();
        K4_MARK(2)
        if (mask && user_ok && t + 1 < n_tiles) mask_next = mask[(size_t)(t + 1) * mask_pitch + row];
        if (!user_ok) maskw = 0xffffffffu;
        if (t == n_tiles_all - 1) maskw |= tail_mask;
        if (t == next_sched && !(TKR_ABL & 16)) {
            if (__ballot(sm.cnt[uw] > kCap / 2) != 0) {            // lists still short (thresholds shared by earlier ranges): nothing to gain
                if constexpr (REFINE) {
                    thr = trim_all_users<IdT, true>(sm, uw, h, K, thr, m2);
                    // 1 / scale (a power of two) is rebuilt here from an opaque copy: hoisted out of the tile loop it is one more
                    // register alive across it, and the allocator, at its 256-register cap, spilled exactly that one
                    float bs = bscale;
                    int row_here = row;                          // likewise the address of the row's shared bound
                    asm volatile("" : "+v"(bs), "+v"(row_here));
                    thr = share_bound(thr_shared, row_here, user_ok && h == 0, thr, margin, bscale,
                                      __uint_as_float((254u - ((__float_as_uint(bs) >> 23) & 0xffu)) << 23));
                } else {
                    thr = trim_all_users<IdT>(sm, uw, h, K, thr);
                    int row_here = row;
                    asm volatile("" : "+v"(row_here));
                    thr = share_threshold(thr_shared, row_here, user_ok && h == 0, thr);
                }
            }
            next_sched = t + ((t - t_begin + 1) >> 1);
        }
        K4_MARK(3)
#if !(TKR_ABL & 1)
        filter_tile<IdT, kEarlyBarrier, REFINE>(sm, acc, sm.tbias + buf * 32, maskw, t, K, thr, m2, &lost, bscale);
#else
        if (acc[0] + acc[5] + acc[10] + acc[15] == 12345.678f) sm.cnt[uw] = 1;     // keeps the chain alive
#endif
#if !(TKR_ABL & 4)
        if constexpr (!kEarlyBarrier) __syncthreads();
#endif
#if TKR_ABL & 256
        asm volatile("" : "+v"(thr));
        K4_MARK(4)
        k4p[7] += 1;
#endif
    }
    if constexpr (REFINE) {
        if (__ballot(lost) != 0 && lane == 0) extra[4 + ws.block] = 1u;       // the exact kernel redoes this block
#if !(TKR_ABL & 64)
        // (the kernel that stages its tiles through registers has fewer to spare: 4 instead of 8 loads per half in flight, no scratch)
        write_rows_refine<IdT, KS, IMG ? kRescoreInFlight : 4>(sm, ws, n_rows, K, thr, m2, U, uidx, Vt, bias, k, out_ids, out_scores, part,
                                                               tile, 2 * TILEB);                                // the tile buffers are free now
#endif
    } else {
        write_rows<IdT>(sm, ws, n_rows, K, thr, out_ids, out_scores, part);
    }
#if TKR_ABL & 256
    K4_MARK(6)
    if (lane == 0)
        for (int q = 0; q < 8; ++q) atomicAdd(&g_k4_prof[q], k4p[q]);
#endif
}

// max_i |v_i| (2-norm, rounded up) and max_i |bias_i| for the margin of the bound-and-refine kernel; bounds[] zeroed before
__global__ __launch_bounds__(256) void topk_bounds_kernel(const float* __restrict__ Vt, const float* __restrict__ bias, int n_cols,
                                                         int k, uint32_t* __restrict__ bounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float vmax = 0.f, bmax = 0.f, emax = 0.f;
    if ((k & 3) == 0 && k <= 128) {                              // a column per half-wave, one float4 per lane, four columns in flight
        const int half = lane >> 5, l = lane & 31;
        const int stride = gridDim.x * 8;
        for (int c0 = (blockIdx.x * 4 + wave) * 2 + half; c0 < n_cols; c0 += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = c0 + u * stride;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col < n_cols && 4 * l < k) v[u] = *reinterpret_cast<const float4*>(Vt + (size_t)col * k + 4 * l);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = c0 + u * stride;
                float ss = fmaf(v[u].x, v[u].x, fmaf(v[u].y, v[u].y, fmaf(v[u].z, v[u].z, v[u].w * v[u].w)));
                emax = fmaxf(emax, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                vmax = fmaxf(vmax, sqrtf(ss) * 1.001f);
                if (bias && col < n_cols && l == 0) bmax = fmaxf(bmax, fabsf(bias[col]));
            }
        }
        vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
        bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
    } else {
        for (int col = blockIdx.x * 4 + wave; col < n_cols; col += gridDim.x * 4) {
            float ss = 0.f;
            for (int e = lane; e < k; e += 64) { const float v = Vt[(size_t)col * k + e]; ss = fmaf(v, v, ss); emax = fmaxf(emax, fabsf(v)); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            vmax = fmaxf(vmax, sqrtf(ss) * 1.001f);
            if (bias) bmax = fmaxf(bmax, fabsf(bias[col]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) emax = fmaxf(emax, __shfl_xor(emax, o, 64));
    __shared__ float red[3][4];                                  // one atomic per workgroup and word: 3,072 atomics on one line took 18 us
    if (lane == 0) { red[0][wave] = vmax; red[1][wave] = bmax; red[2][wave] = emax; }
    __syncthreads();
    if (threadIdx.x < 3) {                                       // non-negative floats order like their bits
        const float m = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
        if (m > 0.f) atomicMax(&bounds[threadIdx.x], __float_as_uint(m));
    }
}

// ---- merge of the per-item-range partial lists: one wave per row ----------------------------------
__global__ __launch_bounds__(256) void merge_topk_kernel(const uint64_t* __restrict__ part, int n_rows, int S, int K,
                                                        int32_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                        const int32_t* __restrict__ nslots /*per user block, or null: S everywhere*/,
                                                        int users_per_block, const uint32_t* __restrict__ only_flagged /*or null: every block*/) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    if (only_flagged && !only_flagged[r / users_per_block]) return;      // topk_finish2_kernel wrote this block's rows
    const uint64_t* p = part + (size_t)r * S * K;                // S slots per row; the block's first nslots are used
    if (nslots) {
        S = nslots[r / users_per_block];
        if (S <= 1) return;                                      // a single workgroup ranked the block: output already final
    }
    uint64_t key = (lane < K) ? p[lane] : 0ull;                  // best K so far in lanes 0..K-1 (K <= 32)
    for (int s = 1; s < S; ++s) {
        // best-so-far descending in lanes 0-31, the next (descending) list REVERSED in lanes 32-63: a bitonic sequence,
        // which the last six stages of the network sort
        const uint64_t in = (lane >= 32 && 63 - lane < K) ? p[(size_t)s * K + 63 - lane] : 0ull;
        uint64_t v = lane < 32 ? (lane < K ? key : 0ull) : in;
        uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
        cmpx<64, 32>(hi, lo, lane); cmpx<64, 16>(hi, lo, lane); cmpx<64, 8>(hi, lo, lane); cmpx<64, 4>(hi, lo, lane);
        cmpx<64, 2>(hi, lo, lane); cmpx<64, 1>(hi, lo, lane);
        key = ((uint64_t)hi << 32) | lo;
    }
    if (lane < K) {
        const bool have = key != 0ull;
        const uint32_t ob = (uint32_t)(key >> 32);
        const uint32_t f = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
        out_ids[(size_t)r * K + lane] = have ? (int32_t)((uint32_t)key - 1u) : -1;
        if (out_scores) out_scores[(size_t)r * K + lane] = have ? __uint_as_float(f) : -INFINITY;
    }
}

// Item-range splits: the hardware hands out the (user block, item range) workgroups in order as CUs free up, so
// many small ranges balance the load; ranges of a block share their thresholds (share_threshold), which leaves a
// fixed cost per range of a few tile-times (operand load, first tiles at a loose threshold, final sorts, merge).
constexpr int kMaxSplits = 32;
constexpr double kSplitFixedTiles = 16.0;    // measured: ~37 us final sorts + start-up, plus the looser thresholds of a short range
static int pick_splits(int n_rows, int users_per_wg, int n_tiles, int max_splits) {
    if (const char* e = getenv("TKR_TOPK_SPLITS")) {          // tuning aid: force the number of item ranges
        const int f = atoi(e);
        if (f >= 1) return std::min(std::min(f, max_splits), n_tiles);
    }
    const int wgs = (n_rows + users_per_wg - 1) / users_per_wg;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= max_splits && s <= n_tiles; ++s) {
        const int tps = (n_tiles + s - 1) / s;
        const int used = (n_tiles + tps - 1) / tps;              // splits that actually get tiles
        const double rounds = (double)(((size_t)wgs * used + 255) / 256);
        const double cost = rounds * (tps + kSplitFixedTiles) + (used > 1 ? 0.02 * used * kSplitFixedTiles : 0.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = used; }
    }
    return best;
}

// ---- rated-item bitmask: mask[(col>>5)*pitch + row] bit (col&31) --------------------------------
__global__ void build_mask_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ cols, int n_rows,
                                  int n_cols, uint32_t* __restrict__ mask, int pitch) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    for (int64_t p = ptr[row] + lane; p < ptr[row + 1]; p += 64) {
        const int c = cols[p];
        if (c >= 0 && c < n_cols) atomicOr(&mask[(size_t)(c >> 5) * pitch + row], 1u << (c & 31));
    }
}

// ---- K5: hits per bucket (evaluate.py:99-103) ------------------------------------------------------
// first_bucket[p / step] += 1 for every kept position p whose column is liked by the row's user;
// the cumulative sum over buckets (host) gives hits[q] = #liked in positions < (q+1)*step.
__global__ void count_hits_kernel(const int32_t* __restrict__ ids, int n_rows, int K, const int64_t* __restrict__ like_ptr,
                                  const int32_t* __restrict__ like_cols, int step, int interval,
                                  unsigned long long* __restrict__ first_bucket) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_rows * K) return;
    const int row = g / K, p = g % K;
    const int c = ids[g];
    if (c < 0 || p / step >= interval) return;
    int64_t lo = like_ptr[row], hi = like_ptr[row + 1];
    const int64_t end = hi;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (like_cols[mid] < c) lo = mid + 1; else hi = mid;
    }
    if (lo < end && like_cols[lo] == c) atomicAdd(&first_bucket[p / step], 1ull);
}

// ---- balanced item table ---------------------------------------------------------------------------------------------
// A grid of (user block, item range) workgroups only fills 256 CUs evenly when blocks x ranges happens to be a multiple of
// 256 (ML-10M: 273 blocks).  Instead the (block, tile) space is cut into G equal spans of consecutive tiles; a span is
// the tail of one block, whole blocks and the head of another, so the pieces ("items") that make up any span add up to
// the same number of tiles.  The items are dispatched largest first (LPT): the hardware hands workgroups to CUs in grid
// order as they free up, which with decreasing sizes packs the CUs to within a few tiles of equal.  Pieces of one block
// are its partial lists (slot = order inside the block); blocks that stay whole write their final output directly.
// The table depends only on the shape, so it is built once per shape and kept on the device (8 shapes).
struct ItemTable {
    int dev;                // device that owns d_items / d_nslots
    int n_rows, users, n_tiles, G;
    int n_items, n_blocks, stride;
    int4* d_items;          // [n_items] (block, t_begin, t_end, slot | stride << 16)
    int32_t* d_nslots;      // [n_blocks]
    int32_t* d_pbase;       // [n_blocks] pieces in front of the block's first (piece-major dumps: csrc/topk_refine.hip)
    uint64_t used;
};
static ItemTable g_tables[8];
static int g_tables_n = 0;
static uint64_t g_tables_tick = 0;

static const ItemTable* item_table(int n_rows, int users, int n_tiles, int G) {
    ++g_tables_tick;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < g_tables_n; ++i) {
        ItemTable& t = g_tables[i];
        if (t.dev == dev && t.n_rows == n_rows && t.users == users && t.n_tiles == n_tiles && t.G == G) { t.used = g_tables_tick; return &t; }
    }
    const int n_blocks = (n_rows + users - 1) / users;
    const long long total = (long long)n_blocks * n_tiles;
    std::vector<int4> items;
    std::vector<int32_t> nslots(n_blocks, 0);
    long long pos = 0;
    int g = 0;
    while (pos < total) {                                        // cut at block ends and at span ends
        while ((long long)(g + 1) * total / G <= pos) ++g;
        const long long span_end = (long long)(g + 1) * total / G;
        const int block = (int)(pos / n_tiles), t0 = (int)(pos % n_tiles);
        const long long block_end = (long long)(block + 1) * n_tiles;
        const long long end = std::min(span_end, block_end);
        items.push_back(make_int4(block, t0, t0 + (int)(end - pos), nslots[block]++));
        pos = end;
    }
    int stride = 1;
    std::vector<int32_t> pbase(n_blocks, 0);
    for (int b = 1; b < n_blocks; ++b) pbase[b] = pbase[b - 1] + nslots[b - 1];
    for (int b = 0; b < n_blocks; ++b) stride = std::max(stride, (int)nslots[b]);
    for (auto& it : items) it.w |= (nslots[it.x] > 1 ? stride : 1) << 16;
    std::stable_sort(items.begin(), items.end(), [](const int4& a, const int4& b) { return (a.z - a.y) > (b.z - b.y); });
    int slot = g_tables_n;
    if (g_tables_n < 8) {
        ++g_tables_n;
    } else {
        slot = 0;
        for (int i = 1; i < 8; ++i)
            if (g_tables[i].used < g_tables[slot].used) slot = i;
        (void)hipFree(g_tables[slot].d_items);
        (void)hipFree(g_tables[slot].d_nslots);
        (void)hipFree(g_tables[slot].d_pbase);
    }
    ItemTable& t = g_tables[slot];
    t = ItemTable{dev, n_rows, users, n_tiles, G, (int)items.size(), n_blocks, stride, nullptr, nullptr, nullptr, g_tables_tick};
    if (hipMalloc(&t.d_items, items.size() * sizeof(int4)) != hipSuccess || hipMalloc(&t.d_nslots, nslots.size() * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&t.d_pbase, pbase.size() * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(t.d_pbase, pbase.data(), pbase.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t.d_items, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t.d_nslots, nslots.data(), nslots.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        t.n_rows = -1;                                           // never matches; fall back to the plain grid
        return nullptr;
    }
    return &t;
}

// Number of spans for a problem of `total` (block, tile) cells: 256 x (spans per CU) when every span still gets a few
// dozen tiles (an item costs ~37 us = ~10 tile-times of its own), one span per CU for smaller problems, and 0 (plain
// grid) when even that would leave fewer than 8 tiles per span.
static int topk_spans_per_cu();
static int topk_span_count(long long total, int spans_per_cu = 0) {
    const int m = (spans_per_cu > 0 && topk_spans_per_cu() > 0) ? spans_per_cu : topk_spans_per_cu();
    if (m == 0 || total < 8 * 256) return 0;
    return (total >= 64LL * 256 * m) ? 256 * m : 256;
}

// spans per CU of the item table (TKR_TOPK_SPANS=0: plain grid of (block, range) workgroups)
static int topk_spans_per_cu() {
    static int m = -1;
    if (m < 0) {
        const char* e = getenv("TKR_TOPK_SPANS");
        m = e ? atoi(e) : 2;                                     // measured: 2 spans per CU at both benchmark shapes
        if (m < 0 || m > 4) m = 2;
    }
    return m;
}

// How one K4 call covers the (user block, tile) space and where its scratch lives in the workspace:
//   [lists x n_rows x K sorted keys][n_rows shared threshold words][extra words]
struct TopkPlan {
    dim3 grid;
    int tps;                  // tiles per item range of the plain grid (0 with the item table)
    uint64_t* part;           // partial lists
    uint32_t* thr_shared;     // or null
    const int4* items;        // item table, or null: the plain (block, range) grid
    int merge_lists;          // > 1: merge_topk_kernel over this many slots per row
    const int32_t* nslots;
    const int32_t* pbase;     // item table: pieces in front of a block's first
    int n_pieces;             // workgroups of the grid
    uint32_t* extra;          // `extra_words` zeroed uint32 behind the thresholds, or null when they did not fit
    size_t used_bytes;        // of the workspace
};

static int plan_topk(int users, int n_rows, int n_cols, int K, void* workspace, size_t workspace_bytes, size_t extra_words,
                     hipStream_t stream, TopkPlan& p, int spans_per_cu = 0) {
    const int grid = (n_rows + users - 1) / users;
    const int n_tiles = (n_cols + 31) / 32;
    const size_t per_split = (size_t)n_rows * K * sizeof(uint64_t);
    const size_t thr_bytes = (size_t)n_rows * sizeof(uint32_t), extra_bytes = extra_words * sizeof(uint32_t);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    p = TopkPlan{};
    const int G = topk_span_count((long long)grid * n_tiles, spans_per_cu);
    if (workspace && G > 0) {
        const ItemTable* tab = item_table(n_rows, users, n_tiles, G);
        if (tab && workspace_bytes >= (size_t)tab->stride * per_split + thr_bytes) {
            const size_t lists_bytes = (size_t)tab->stride * per_split;
            const bool fits = workspace_bytes >= lists_bytes + thr_bytes + extra_bytes;
            p.grid = dim3(tab->n_items);
            p.part = reinterpret_cast<uint64_t*>(ws);
            p.thr_shared = reinterpret_cast<uint32_t*>(ws + lists_bytes);
            p.items = tab->d_items;
            p.merge_lists = tab->stride;
            p.nslots = tab->d_nslots;
            p.pbase = tab->d_pbase;
            p.n_pieces = tab->n_items;
            p.extra = (fits && extra_words) ? p.thr_shared + n_rows : nullptr;
            p.used_bytes = lists_bytes + thr_bytes + (p.extra ? extra_bytes : 0);
            TKR_CHECK(hipMemsetAsync(p.thr_shared, 0, thr_bytes + (p.extra ? extra_bytes : 0), stream));
            return TKR_OK;
        }
    }
    const size_t fixed = thr_bytes + extra_bytes;
    const int max_splits = (workspace && workspace_bytes > fixed)
                               ? (int)std::min<size_t>(kMaxSplits, (workspace_bytes - fixed) / (per_split ? per_split : 1)) : 1;
    int S = pick_splits(n_rows, users, n_tiles, max_splits < 1 ? 1 : max_splits);
    const int tps = (n_tiles + S - 1) / S;
    S = (n_tiles + tps - 1) / tps;
    p.grid = dim3(grid, S);
    p.tps = tps;
    p.part = reinterpret_cast<uint64_t*>(ws);
    p.merge_lists = S;
    p.n_pieces = grid * S;
    const size_t lists_bytes = S > 1 ? (size_t)S * per_split : 0;
    const bool fits = workspace && workspace_bytes >= lists_bytes + fixed;
    p.used_bytes = lists_bytes + (fits ? fixed : 0);
    if (S > 1 || (fits && extra_words)) {                        // thresholds live behind the S partial lists
        p.thr_shared = reinterpret_cast<uint32_t*>(ws + lists_bytes);
        p.extra = (fits && extra_words) ? p.thr_shared + n_rows : nullptr;
        TKR_CHECK(hipMemsetAsync(p.thr_shared, 0, thr_bytes + (p.extra ? extra_bytes : 0), stream));
    }
    return TKR_OK;
}

static int merge_planned(const TopkPlan& p, int users, int n_rows, int K, int32_t* out_ids, float* out_scores, hipStream_t stream,
                         const uint32_t* only_flagged = nullptr) {
    if (p.merge_lists > 1)
        hipLaunchKernelGGL(merge_topk_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, reinterpret_cast<const uint64_t*>(p.part),
                           n_rows, p.merge_lists, K, out_ids, out_scores, p.nslots, users, only_flagged);
    return (int)hipGetLastError();
}

template <int KHP, typename IdT, int WAVES = topk_waves<KHP, IdT>()>
static int launch_topk_planned(const TopkPlan& p, const float* U, const int32_t* uidx, int n_rows, const float* Vt, const float* bias,
                               int n_cols, int k, const uint32_t* mask, int pitch, int K, int32_t* out_ids, float* out_scores,
                               const uint32_t* only_flagged, hipStream_t stream) {
    constexpr int KP = 2 * KHP + 4;
    const int W = WAVES, users = W * 32;
    const size_t lds = (size_t)(2 * 32 * KP + 64) * 4 + (size_t)users * 4 + (size_t)users * kCap * (4 + sizeof(IdT));
    if (lds > 160 * 1024) return TKR_EUNSUPPORTED;
    auto kern = score_topk_kernel<KHP, IdT, WAVES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, p.grid, dim3(W * 64), lds, stream, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids,
                       out_scores, p.tps, p.part, p.thr_shared, p.items, only_flagged);
    return (int)hipGetLastError();
}

template <int KHP, typename IdT>
static int launch_topk(const float* U, const int32_t* uidx, int n_rows, const float* Vt, const float* bias, int n_cols, int k,
                       const uint32_t* mask, int pitch, int K, int32_t* out_ids, float* out_scores, void* workspace,
                       size_t workspace_bytes, hipStream_t stream) {
    const int users = topk_waves<KHP, IdT>() * 32;
    TopkPlan p;
    int rc = plan_topk(users, n_rows, n_cols, K, workspace, workspace_bytes, 0, stream, p);
    if (rc != TKR_OK) return rc;
    rc = launch_topk_planned<KHP, IdT>(p, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores, nullptr, stream);
    if (rc != TKR_OK) return rc;
    return merge_planned(p, users, n_rows, K, out_ids, out_scores, stream);
}

// the fp32-MFMA kernel for the factor width k (k <= 128 here: same workgroup shape as the bf16 kernels)
template <typename IdT, int WAVES>
static int launch_fp32_planned(const TopkPlan& p, const float* U, const int32_t* uidx, int n_rows, const float* Vt, const float* bias,
                               int n_cols, int k, const uint32_t* mask, int pitch, int K, int32_t* out_ids, float* out_scores,
                               const uint32_t* only_flagged, hipStream_t stream) {
    const int kh = (k + 1) / 2;
#define TKR_TOPK_CASE(KHP)                                                                                                    \
    if (kh <= KHP)                                                                                                            \
        return launch_topk_planned<KHP, IdT, WAVES>(p, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores, \
                                             only_flagged, stream);
    TKR_TOPK_CASE(16)
    TKR_TOPK_CASE(28)
    TKR_TOPK_CASE(32)
    TKR_TOPK_CASE(52)
    TKR_TOPK_CASE(64)
#undef TKR_TOPK_CASE
    return TKR_EUNSUPPORTED;
}

static size_t topk_image_bytes(int n_cols, int k) {               // the fp16 tile image of the bound-and-refine arithmetic (k <= 128)
    const int KS = k <= 16 ? 1 : (k <= 32 ? 2 : (k <= 64 ? 4 : 8));
    return (size_t)((n_cols + 31) / 32) * 32 * KS * 32;
}


template <int KS, typename IdT, bool REFINE>
static int launch_topk_bf16(const float* U, const int32_t* uidx, int n_rows, const float* Vt, const float* bias, int n_cols,
                            int k, const uint32_t* mask, int pitch, int K, int32_t* out_ids, float* out_scores,
                            void* workspace, size_t workspace_bytes, hipStream_t stream) {
    // REFINE: the tile image lives at the END of the workspace when the caller sized it with tkr_topk_workspace_bytes_for
    unsigned char* vimg = nullptr;
    if constexpr (REFINE) {
        static const bool no_img = getenv("TKR_TOPK_IMAGE") && getenv("TKR_TOPK_IMAGE")[0] == '0';
        const size_t img = (topk_image_bytes(n_cols, k) + 255) & ~(size_t)255;
        const size_t base = (size_t)tkr_topk_workspace_bytes(n_rows, K);
        if (!no_img && workspace && workspace_bytes >= base + img + 256) {
            const size_t off = (workspace_bytes - img) & ~(size_t)255;
            vimg = static_cast<unsigned char*>(workspace) + off;
            workspace_bytes = off;
        }
    }
    const bool use_img = vimg != nullptr;
    const int ROWB = use_img ? KS * 32 : (REFINE ? 1 : 3) * KS * 32 + 16;
    const int W = topk_waves_bf16<KS, IdT, REFINE>();
    const int users = W * 32;
    size_t lds = (size_t)2 * 32 * ROWB + 64 * 4 + (size_t)users * 8 + (size_t)users * kCap * (4 + sizeof(IdT));
    if (const char* e = getenv("TKR_TOPK_LDS_PAD")) lds += (size_t)atoi(e);      // occupancy experiments (scripts/)
    if (lds > 160 * 1024) return TKR_EUNSUPPORTED;
    auto kern = score_topk_bf16_kernel<KS, IdT, REFINE, false>;
    auto kern_img = score_topk_bf16_kernel<KS, IdT, REFINE, REFINE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(use_img ? kern_img : kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    const int n_blocks = (n_rows + users - 1) / users;
    // the second form of bound-and-refine (csrc/topk_refine.hip: packed lists, three workgroups per CU, every row ranked once by
    // topk_finish2_kernel): 16-bit column ids, the tile image, room for the pieces' dumps.  TKR_TOPK_V2=0: the first form.
    static const bool v2_off = getenv("TKR_TOPK_V2") && getenv("TKR_TOPK_V2")[0] == '0';
    const bool want_v2 = REFINE && sizeof(IdT) == 2 && use_img && !v2_off && refine2_supports(n_cols, k) && users == kR2Users;
    TopkPlan p;
    int rc = plan_topk(users, n_rows, n_cols, K, workspace, workspace_bytes, REFINE ? (size_t)(4 + n_blocks) : 0, stream, p,
                       want_v2 ? kR2SpansPerCU : 0);
    if (rc != TKR_OK) return rc;
    if constexpr (REFINE && sizeof(IdT) == 2) {
        const size_t used = (p.used_bytes + 255) & ~(size_t)255;
        if (want_v2 && p.extra && used + refine2_dump_bytes((size_t)p.n_pieces) <= workspace_bytes) {
            hipLaunchKernelGGL(topk_bounds_kernel, dim3(std::min(256, (n_cols + 3) / 4)), dim3(256), 0, stream, Vt, bias, n_cols, k, p.extra);
            hipLaunchKernelGGL(topk_image_kernel<KS>, dim3(std::min(2048, ((n_cols + 31) / 32 * 32 * 2 * KS + 255) / 256)), dim3(256), 0, stream, Vt,
                               n_cols, k, p.extra, vimg);
            Refine2Args a{};
            a.U = U; a.uidx = uidx; a.n_rows = n_rows; a.Vt = Vt; a.bias = bias; a.n_cols = n_cols; a.k = k;
            a.mask = mask; a.mask_pitch = pitch; a.K = K;
            a.grid_x = (int)p.grid.x; a.grid_y = (int)p.grid.y; a.tiles_per_split = p.tps;
            a.thr_shared = p.thr_shared; a.items = p.items; a.nslots = p.nslots; a.pbase = p.pbase;
            a.extra = p.extra; a.vimg = vimg;
            a.dump = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(workspace) + used);
            a.dhdr = reinterpret_cast<float2*>(a.dump + (size_t)p.n_pieces * kR2Users * kR2Slots);
            a.out_ids = out_ids; a.out_scores = out_scores;
            rc = launch_refine2(a, stream);
            if (rc != TKR_OK) return rc;
            // blocks with an overflowed list: the fp32 kernel on the same work items, merged where a block was cut
            rc = launch_fp32_planned<IdT, topk_waves_bf16<KS, IdT, true>()>(p, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores,
                                                                            p.extra + 4, stream);
            if (rc != TKR_OK) return rc;
            return merge_planned(p, users, n_rows, K, out_ids, out_scores, stream, p.extra + 4);
        }
    }
    if constexpr (REFINE) {
        if (!p.extra) return TKR_EAGAIN_EXACT;                    // no room for the block flags: the caller runs the fp32 kernel
        hipLaunchKernelGGL(topk_bounds_kernel, dim3(std::min(256, (n_cols + 3) / 4)), dim3(256), 0, stream, Vt, bias, n_cols, k, p.extra);
        if (use_img)
            hipLaunchKernelGGL(topk_image_kernel<KS>, dim3(std::min(2048, ((n_cols + 31) / 32 * 32 * 2 * KS + 255) / 256)), dim3(256), 0, stream, Vt,
                               n_cols, k, p.extra, vimg);
    }
    hipLaunchKernelGGL(use_img ? kern_img : kern, p.grid, dim3(W * 64), lds, stream, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K,
                       out_ids, out_scores, p.tps, p.part, p.thr_shared, p.items, p.extra, (const unsigned char*)vimg);
    rc = (int)hipGetLastError();
    if (rc != TKR_OK) return rc;
    if constexpr (REFINE) {
        // blocks with an overflowed list: the fp32 kernel, same work items
        rc = launch_fp32_planned<IdT, topk_waves_bf16<KS, IdT, true>()>(p, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores,
                                                                        p.extra + 4, stream);
        if (rc != TKR_OK) return rc;
    }
    return merge_planned(p, users, n_rows, K, out_ids, out_scores, stream);
}

template <int SLABS, typename IdT>
static int launch_topk_slabs(const float* U, const int32_t* uidx, int n_rows, const float* Vt, const float* bias, int n_cols, int k,
                             const uint32_t* mask, int pitch, int K, int32_t* out_ids, float* out_scores, void* workspace,
                             size_t workspace_bytes, hipStream_t stream) {
    constexpr int users = 128, KP = 260;
    TopkPlan p;
    int rc = plan_topk(users, n_rows, n_cols, K, workspace, workspace_bytes, 0, stream, p);
    if (rc != TKR_OK) return rc;
    const size_t lds = (size_t)(2 * 32 * KP + 64) * 4 + (size_t)users * 4 + (size_t)users * kCap * (4 + sizeof(IdT));
    auto kern = score_topk_slab_kernel<SLABS, IdT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, p.grid, dim3(256), lds, stream, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores,
                       p.tps, p.part, p.thr_shared, p.items);
    rc = (int)hipGetLastError();
    if (rc != TKR_OK) return rc;
    return merge_planned(p, users, n_rows, K, out_ids, out_scores, stream);
}

// ---- any width: one wave per row, a lane per item (k > 768) -------------------------------------------------------------------------
// The MFMA kernels keep a workgroup's users resident as operands: three 256-wide slabs fill the register file.  Beyond that nothing
// is resident: a wave takes ONE row, its lanes take 64 items at a time and each runs the fp32 fma chain of exact_score over the whole
// width (k-halves interleaved, k ascending: the chain of the other kernels with KH = ceil(k / 2)); what reaches the K-th best so far
// is merged into the row's sorted list (lanes 0..31) by the bitonic network.  Slow -- every lane gathers its own item row -- but the
// same interface and the same canonical order for every k.
__global__ __launch_bounds__(256) void score_topk_wide_kernel(const float* __restrict__ U, const int32_t* __restrict__ uidx, int n_rows,
                                                             const float* __restrict__ Vt, const float* __restrict__ bias, int n_cols, int k,
                                                             const uint32_t* __restrict__ mask, int mask_pitch, int K,
                                                             int32_t* __restrict__ out_ids, float* __restrict__ out_scores) {
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n_rows) return;
    const float* up = U + (size_t)(uidx ? uidx[r] : r) * k;
    uint64_t best = 0ull;                                        // lanes 0..31: the best so far, descending (0 = none)
    uint32_t kth = 0u;                                           // ordered bits of the K-th best score so far (0: fewer than K yet)
    for (int c0 = 0; c0 < n_cols; c0 += 64) {
        const int col = c0 + lane;
        bool ok = col < n_cols;
        if (ok && mask) ok = ((mask[(size_t)(col >> 5) * mask_pitch + r] >> (col & 31)) & 1u) == 0u;
        uint64_t key = 0ull;
        if (ok) {
            const float sc = exact_score(up, Vt + (size_t)col * k, k, bias, col);
            const uint32_t ob = ordered_bits(sc);
            if (ob >= kth) key = ((uint64_t)ob << 32) | ((uint32_t)col + 1u);
        }
        if (__ballot(key != 0ull) == 0) continue;
#pragma unroll 1
        for (int hx = 0; hx < 2; ++hx) {                         // the best 32 so far + 32 of the new ones, twice
            const uint64_t in = (uint64_t)__shfl((unsigned long long)key, (lane & 31) + 32 * hx, 64);
            best = wave_sort_desc(lane < 32 ? best : in, lane);
            if (lane >= 32) best = 0ull;
        }
        const uint64_t kb = (uint64_t)__shfl((unsigned long long)best, K - 1, 64);
        kth = kb ? (uint32_t)(kb >> 32) : 0u;
    }
    if (lane < K) {
        const bool have = best != 0ull;
        const uint32_t ob = (uint32_t)(best >> 32);
        const uint32_t f = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
        out_ids[(size_t)r * K + lane] = have ? (int32_t)((uint32_t)best - 1u) : -1;
        if (out_scores) out_scores[(size_t)r * K + lane] = have ? __uint_as_float(f) : -INFINITY;
    }
}

// arithmetic of the score block: 2 = bound-and-refine (default; k <= 128 and a workspace, else it runs as 1), 0 = bf16-split
// products on the dense matrix pipe (k <= 128), 1 = fp32 MFMA for every k.
// Initial value from TKR_TOPK_MATH=refine|bf16x3|fp32; tkr_topk_set_math changes it for the process.
static int g_topk_math = -1;
static int topk_math() {
    if (g_topk_math < 0) {
        const char* e = getenv("TKR_TOPK_MATH");
        g_topk_math = (e && strcmp(e, "fp32") == 0) ? 1 : (e && strcmp(e, "bf16x3") == 0) ? 0 : 2;
    }
    return g_topk_math;
}

template <typename IdT>
static int dispatch_topk(const float* U, const int32_t* uidx, int n_rows, const float* Vt, const float* bias, int n_cols,
                         int k, const uint32_t* mask, int pitch, int K, int32_t* out_ids, float* out_scores,
                         void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (k <= 128 && topk_math() == 2) {                          // bound-and-refine; without room for its flags: the fp32 kernel (same results)
        int rc = TKR_EAGAIN_EXACT;
#define TKR_TOPK_REFINE_CASE(KS)                                                                                        \
    if (rc == TKR_EAGAIN_EXACT && k <= 16 * KS) {                                                                       \
        rc = launch_topk_bf16<KS, IdT, true>(U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores, \
                                             workspace, workspace_bytes, stream);                                      \
        if (rc != TKR_EAGAIN_EXACT) return rc;                                                                          \
        rc = TKR_OK;                                                                                                    \
    }
        TKR_TOPK_REFINE_CASE(1)
        TKR_TOPK_REFINE_CASE(2)
        TKR_TOPK_REFINE_CASE(4)
        TKR_TOPK_REFINE_CASE(8)
#undef TKR_TOPK_REFINE_CASE
    }
#ifdef TKR_LAB                                                   // bf16x3 (six exact partial products): measured slower than bound-and-refine; `make LAB=1`
    if (k <= 128 && topk_math() == 0) {
#define TKR_TOPK_BF16_CASE(KS)                                                                                          \
    if (k <= 16 * KS)                                                                                                   \
        return launch_topk_bf16<KS, IdT, false>(U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores, \
                                                workspace, workspace_bytes, stream);
        TKR_TOPK_BF16_CASE(1)
        TKR_TOPK_BF16_CASE(2)
        TKR_TOPK_BF16_CASE(4)
        TKR_TOPK_BF16_CASE(8)
#undef TKR_TOPK_BF16_CASE
    }
#endif
    const int kh = (k + 1) / 2;
#define TKR_TOPK_CASE(KHP)                                                                                   \
    if (kh <= KHP)                                                                                           \
        return launch_topk<KHP, IdT>(U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K,                   \
                                     out_ids, out_scores, workspace, workspace_bytes, stream);
    TKR_TOPK_CASE(16)
    TKR_TOPK_CASE(28)
    TKR_TOPK_CASE(32)
    TKR_TOPK_CASE(52)
    TKR_TOPK_CASE(64)
    TKR_TOPK_CASE(100)
    TKR_TOPK_CASE(128)
#undef TKR_TOPK_CASE
    if (k <= 512)
        return launch_topk_slabs<2, IdT>(U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores, workspace, workspace_bytes, stream);
    if (k <= 768)
        return launch_topk_slabs<3, IdT>(U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K, out_ids, out_scores, workspace, workspace_bytes, stream);
    hipLaunchKernelGGL(score_topk_wide_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, stream, U, uidx, n_rows, Vt, bias, n_cols, k, mask, pitch, K,
                       out_ids, out_scores);
    return (int)hipGetLastError();
}

}  // namespace tkr

extern "C" int tkr_build_rated_mask(const int64_t* rated_ptr, const int32_t* rated_cols, int32_t n_rows, int32_t n_cols,
                                    uint32_t* mask, int32_t pitch, void* stream) {
    if (!rated_ptr || !mask || n_rows <= 0 || n_cols <= 0 || pitch < n_rows) return TKR_EINVAL;
    hipLaunchKernelGGL(tkr::build_mask_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rated_ptr,
                       rated_cols, n_rows, n_cols, mask, pitch);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

#if TKR_ABL & 256
extern "C" int tkr_k4_prof_read(unsigned long long* out8) {      // timing builds only: read and reset the phase counters
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_k4_prof), sizeof(zero));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_k4_prof), zero, sizeof(zero));
    return (int)e;
}
#endif

extern "C" int tkr_topk_set_math(int32_t mode) {
    if (mode < 0 || mode > 2) return TKR_EINVAL;
#ifndef TKR_LAB
    if (mode == 0) return TKR_EUNSUPPORTED;                      // bf16x3 lives in the lab library (make LAB=1)
#endif
    tkr::g_topk_math = mode;
    return TKR_OK;
}

extern "C" int64_t tkr_topk_workspace_bytes(int32_t n_rows, int32_t K) {
    // room for the partial lists a launch can use + one shared threshold word per row.  Plain grid of (block, range)
    // workgroups: enough ranges to balance 256 CUs, at most kMaxSplits.  Item table: the pieces one block can be cut
    // into, at most ceil(G / blocks) + 1 with G <= 512 spans.
    const int64_t blocks = ((int64_t)n_rows + 255) / 256;        // (a lower bound of the blocks: 128-256 users each)
    int64_t splits = (16 * 256 + blocks - 1) / blocks;
    if (splits > tkr::kMaxSplits) splits = tkr::kMaxSplits;
    if (splits < 2) splits = 2;
    const int64_t pieces = (1024 + blocks - 1) / blocks + 1;     // 256 x (spans per CU <= 4)
    const int64_t lists = splits > pieces ? splits : pieces;
    const int64_t refine_words = 4 + ((int64_t)n_rows + 127) / 128;      // bounds + one flag per user block (>= 128 users each)
    return lists * n_rows * K * (int64_t)sizeof(uint64_t) + (int64_t)n_rows * (int64_t)sizeof(uint32_t) + refine_words * 4 + 512;
}

extern "C" int64_t tkr_topk_workspace_bytes_for(int32_t n_rows, int32_t n_cols, int32_t k, int32_t K) {
    // tkr_topk_workspace_bytes + room for the pre-converted item factors of the bound-and-refine arithmetic (k <= 128)
    int64_t n = tkr_topk_workspace_bytes(n_rows, K);
    if (k <= 128 && n_cols > 0) n = ((n + 255) & ~(int64_t)255) + (int64_t)((tkr::topk_image_bytes(n_cols, k) + 255) & ~(size_t)255) + 512;
    if (tkr::refine2_supports(n_cols, k)) {                      // + the pieces' packed lists for topk_finish2_kernel (csrc/topk_refine.hip)
        const int64_t blocks = ((int64_t)n_rows + tkr::kR2Users - 1) / tkr::kR2Users, n_tiles = ((int64_t)n_cols + 31) / 32;
        const int64_t pieces = blocks * n_tiles >= 8 * 256 ? blocks + 256 * 4 : blocks * (n_tiles < tkr::kMaxSplits ? n_tiles : tkr::kMaxSplits);
        n = ((n + 255) & ~(int64_t)255) + (int64_t)tkr::refine2_dump_bytes((size_t)pieces) + 256;
    }
    return n;
}

extern "C" int tkr_score_topk(const float* U, const int32_t* user_idx, int32_t n_rows, const float* Vt,
                              const float* bias, int32_t n_cols, int32_t k, const uint32_t* mask, int32_t mask_pitch,
                              int32_t K, int32_t* out_ids, float* out_scores, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    if (!U || !Vt || !out_ids || n_rows <= 0 || n_cols <= 0 || k <= 0 || K <= 0) return TKR_EINVAL;
    if (mask && mask_pitch < n_rows) return TKR_EINVAL;
    if (K > tkr::kMaxK) return TKR_EUNSUPPORTED;
    if (n_cols <= 65535)
        return tkr::dispatch_topk<uint16_t>(U, user_idx, n_rows, Vt, bias, n_cols, k, mask, mask_pitch, K, out_ids,
                                            out_scores, workspace, (size_t)(workspace_bytes > 0 ? workspace_bytes : 0),
                                            (hipStream_t)stream);
    return tkr::dispatch_topk<uint32_t>(U, user_idx, n_rows, Vt, bias, n_cols, k, mask, mask_pitch, K, out_ids, out_scores,
                                        workspace, (size_t)(workspace_bytes > 0 ? workspace_bytes : 0), (hipStream_t)stream);
}

extern "C" int tkr_count_hits(const int32_t* ids, int32_t n_rows, int32_t K, const int64_t* like_ptr,
                              const int32_t* like_cols, int32_t step, int32_t interval, uint64_t* first_bucket,
                              void* stream) {
    if (!ids || !like_ptr || !first_bucket || n_rows <= 0 || K <= 0 || step <= 0 || interval < 0) return TKR_EINVAL;
    const int n = n_rows * K;
    hipLaunchKernelGGL(tkr::count_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, n_rows, K,
                       like_ptr, like_cols, step, interval, reinterpret_cast<unsigned long long*>(first_bucket));
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}
