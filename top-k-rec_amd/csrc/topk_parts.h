// Device helpers shared by the K4 translation units (csrc/topk.hip, csrc/topk_refine.hip): ordered keys, the wave-wide bitonic
// sorts, the bounds the item ranges of a user block tell each other, the exact fp32 score of one (user, item) pair.
#pragma once
#include "tkr_common.h"

namespace tkr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTopkMaxWaves = 8;
constexpr int kCap = 64;                 // candidate slots per user (= wave width: one entry per lane in a trim)
constexpr int kMaxK = 32;                // K + 32 (largest per-tile inflow) <= kCap
constexpr int TKR_EAGAIN_EXACT = -100;   // internal: the bound-and-refine launch cannot run here (no workspace for its flags)

__device__ __forceinline__ uint32_t ordered_bits(float s) {      // monotone float -> uint
    const uint32_t f = __float_as_uint(s);
    return (f & 0x80000000u) ? ~f : (f | 0x80000000u);
}

// value of lane (lane ^ STRIDE).  Strides 1..8 stay in the VALU (DPP), 16 uses the LDS crossbar without
// an address (ds_swizzle), only 32 needs a bpermute.
__device__ __forceinline__ float unordered_bits(uint32_t ob) {    // inverse of ordered_bits; 0 -> below every float
    if (ob == 0u) return -INFINITY;
    return __uint_as_float((ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob);
}

// Item-range splits of one user block cooperate through thr_shared[row]: the K-th best score inside ANY subset of
// the catalogue is a lower bound of the K-th best overall, so every range may filter with the largest bound any
// range has published.  Ranges are dispatched range-major (blockIdx.x fastest), so later ranges start with the
// thresholds of earlier ones instead of -inf and skip the expensive low-threshold phase.  Results do not depend on
// the timing: a column of the global top K passes every such bound, and the final order comes from the exact sorts.
__device__ __forceinline__ float share_threshold(uint32_t* thr_shared, int row, bool publish, float thr) {
    if (!thr_shared) return thr;
    uint32_t seen = 0u;
    if (publish && thr > -INFINITY) seen = atomicMax(&thr_shared[row], ordered_bits(thr));
    seen = max(seen, (uint32_t)__shfl_xor((int)seen, 32, 64));  // the h = 1 lane of the user gets it too
    return fmaxf(thr, unordered_bits(seen));
}
// Bound-and-refine arithmetic: the filter runs on approximate scores with `thr` = (lower bound of the K-th best EXACT score)
// - margin; what the item ranges of a block tell each other is the bound itself.
// thr and margin are in the user's scaled units (scale = a power of two), the shared word is not.
__device__ __forceinline__ float share_bound(uint32_t* thr_shared, int row, bool publish, float thr, float margin, float scale,
                                             float inv_scale) {
    if (!thr_shared) return thr;
    uint32_t seen = 0u;
    if (publish && thr > -INFINITY) seen = atomicMax(&thr_shared[row], ordered_bits((thr + margin) * inv_scale));
    seen = max(seen, (uint32_t)__shfl_xor((int)seen, 32, 64));
    return fmaxf(thr, unordered_bits(seen) * scale - margin);
}

// 2^e with amax * 2^e in [2^13, 2^14) (|e| <= 60; 1 for amax = 0): the power-of-two scaling of the fp16 pass
__device__ __forceinline__ float pow2_scale(float amax) {
    if (!(amax > 0.f)) return 1.f;
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
    const int sft = max(-60, min(60, 13 - e));
    return __uint_as_float((uint32_t)(127 + sft) << 23);
}

template <int STRIDE>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (STRIDE == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    else if constexpr (STRIDE == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // [2,3,0,1]
    else if constexpr (STRIDE == 4) {   // half_mirror (i ^ 7) then quad reverse (i ^ 3)
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);
        return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, true);
    } else if constexpr (STRIDE == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);   // row_ror:8
    else if constexpr (STRIDE == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);   // xor 16 inside each 32 lanes
    else return (uint32_t)__shfl_xor((int)v, 32, 64);
}

template <int SIZE, int STRIDE>
__device__ __forceinline__ void cmpx(uint32_t& hi, uint32_t& lo, int lane) {
    const uint32_t ohi = lane_xor<STRIDE>(hi), olo = lane_xor<STRIDE>(lo);
    const bool other_gt = (ohi > hi) || (ohi == hi && olo > lo);
    const bool upper = (lane & STRIDE) != 0;                 // I am the higher lane of the pair
    const bool desc = (lane & SIZE) == 0;                    // this block sorts descending
    const bool take_max = (upper != desc);                   // lower lane of a descending block keeps the max
    const bool take_other = (take_max == other_gt);
    hi = take_other ? ohi : hi;
    lo = take_other ? olo : lo;
}

// descending bitonic sort of one 64-bit key per lane across the wave (keys are distinct)
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key, int lane) {
    uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    cmpx<2, 1>(hi, lo, lane);
    cmpx<4, 2>(hi, lo, lane); cmpx<4, 1>(hi, lo, lane);
    cmpx<8, 4>(hi, lo, lane); cmpx<8, 2>(hi, lo, lane); cmpx<8, 1>(hi, lo, lane);
    cmpx<16, 8>(hi, lo, lane); cmpx<16, 4>(hi, lo, lane); cmpx<16, 2>(hi, lo, lane); cmpx<16, 1>(hi, lo, lane);
    cmpx<32, 16>(hi, lo, lane); cmpx<32, 8>(hi, lo, lane); cmpx<32, 4>(hi, lo, lane); cmpx<32, 2>(hi, lo, lane);
    cmpx<32, 1>(hi, lo, lane);
    cmpx<64, 32>(hi, lo, lane); cmpx<64, 16>(hi, lo, lane); cmpx<64, 8>(hi, lo, lane); cmpx<64, 4>(hi, lo, lane);
    cmpx<64, 2>(hi, lo, lane); cmpx<64, 1>(hi, lo, lane);
    return ((uint64_t)hi << 32) | lo;
}

// the first 15 stages of the network: lanes 0-31 end up sorted descending, lanes 32-63 ASCENDING (two independent
// 32-key sorts, no exchange across the halves)
__device__ __forceinline__ uint64_t wave_sort_halves(uint64_t key, int lane) {
    uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    cmpx<2, 1>(hi, lo, lane);
    cmpx<4, 2>(hi, lo, lane); cmpx<4, 1>(hi, lo, lane);
    cmpx<8, 4>(hi, lo, lane); cmpx<8, 2>(hi, lo, lane); cmpx<8, 1>(hi, lo, lane);
    cmpx<16, 8>(hi, lo, lane); cmpx<16, 4>(hi, lo, lane); cmpx<16, 2>(hi, lo, lane); cmpx<16, 1>(hi, lo, lane);
    cmpx<32, 16>(hi, lo, lane); cmpx<32, 8>(hi, lo, lane); cmpx<32, 4>(hi, lo, lane); cmpx<32, 2>(hi, lo, lane);
    cmpx<32, 1>(hi, lo, lane);
    return ((uint64_t)hi << 32) | lo;
}


typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// wave-wide OR, returned to every lane (uniform)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_or(uint32_t v) {
    return v | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    v = dpp_or<0xb1>(v); v = dpp_or<0x4e>(v); v = dpp_or<0x124>(v); v = dpp_or<0x128>(v);
    v = dpp_or<0x142, 0xa>(v); v = dpp_or<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// how many of the two 15-bit keys packed in xo (each with bit 15 set on top) reach cand: (key | 0x8000) - cand keeps bit 15
// exactly when key >= cand and never borrows from the neighbouring field; the answers are added up as two 16-bit counters
__device__ __forceinline__ u16x2 count_ge2(uint32_t xo, uint32_t cand2, u16x2 acc) {
    return acc + (__builtin_bit_cast(u16x2, xo - cand2) >> (unsigned short)15);
}


// ---- bound-and-refine: the exact score of one candidate ------------------------------------------------------------------
// The fp32 dot product of the fp32-MFMA kernel, bit for bit: v_mfma_f32_32x32x2_f32 adds the product of k-half 0, then the
// product of k-half 1, one fused multiply-add each (measured: scripts/probe_mfma_order.py, 100 % of 7,680 scores at k = 50,
// 64, 100, 128) -- so acc <- fma(v[kk], u[kk], acc); acc <- fma(v[KH+kk], u[KH+kk], acc) for kk = 0 .. KH-1, then fl(acc + bias)
// and -0.0 -> +0.0 as the filter of that kernel does.
__device__ __forceinline__ float exact_score(const float* __restrict__ up, const float* __restrict__ vp, int k, const float* bias, int col) {
    const int KH = (k + 1) >> 1;
    float acc = 0.f;
    if ((k & 7) == 0) {                                         // both halves 16-byte aligned
#pragma unroll 8                                                // 32 loads in flight: two memory round trips per 128 factors
        for (int kk = 0; kk < KH; kk += 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(vp + kk), a1 = *reinterpret_cast<const float4*>(vp + KH + kk);
            const float4 b0 = *reinterpret_cast<const float4*>(up + kk), b1 = *reinterpret_cast<const float4*>(up + KH + kk);
            acc = fmaf(a0.x, b0.x, acc); acc = fmaf(a1.x, b1.x, acc);
            acc = fmaf(a0.y, b0.y, acc); acc = fmaf(a1.y, b1.y, acc);
            acc = fmaf(a0.z, b0.z, acc); acc = fmaf(a1.z, b1.z, acc);
            acc = fmaf(a0.w, b0.w, acc); acc = fmaf(a1.w, b1.w, acc);
        }
    } else {
        for (int kk = 0; kk < KH; ++kk) {
            acc = fmaf(vp[kk], up[kk], acc);
            if (KH + kk < k) acc = fmaf(vp[KH + kk], up[KH + kk], acc);
        }
    }
    acc = acc + (bias ? bias[col] : 0.f);
    return acc + 0.0f;
}


// A compiler hole, ROCm 7.2 / gfx950 (found in round 5; tests/test_gpu_topk.py test_exact_arithmetic_lists caught it): the wait states
// between a 16-pass MFMA and the first VALU read of its result (19 on this chip; the hardware does NOT interlock them) are counted by
// hipcc along the LAYOUT of the code, not along the control flow.  On a workgroup's LAST tile the staging code between the chain and
// the filter is skipped by two scalar branches, the `s_nop 2` hipcc had placed in front of `v_accvgpr_read a15` is all that is left,
// and the lane reads register 15 -- the last the matrix pipe writes -- one MFMA early: scores of tile rows 27 and 31 came out short
// of their last two products, depending on how the rest of the file happened to be laid out.  The wait goes in by hand, right behind
// the chain: 20 states of ~6,400 per tile.
// (`acc` is an operand of the statement: it cannot move above the chain or below the first read; where the accumulators live in
// AGPRs hipcc copies them out in front of it -- in line with the chain, where its own count is right)
__device__ __forceinline__ void mfma_result_guard(f32x16& acc) {            // behind a chain of 16-pass MFMAs (v_mfma_f32_32x32x2_f32)
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));
}
__device__ __forceinline__ void mfma_result_guard_8pass(f32x16& acc) {      // behind 8-pass MFMAs (v_mfma_f32_32x32x16_f16 / _bf16): 11 states
    asm volatile("s_nop 10" : "+v"(acc));
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));


template <int KS>
__device__ __forceinline__ int tile_swizzle(int r) {             // KS in {1, 2, 4, 8}: CR = 2 * KS chunks per row, 16 / CR rows per 256 B
    constexpr int CR = 2 * KS;
    return (r / (16 / CR)) & (CR - 1);
}


}  // namespace tkr
