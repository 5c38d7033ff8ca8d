// Shared device helpers for the top-k-rec MI355X (gfx950) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TKR_WAVE 64

// Library error codes (negative); positive return values are hipError_t.
#define TKR_OK 0
#define TKR_EINVAL (-1)
#define TKR_EUNSUPPORTED (-2)

#define TKR_CHECK(expr)                       \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

#define TKR_CHECK_RC(expr) do { const int rc_ = (expr); if (rc_ != TKR_OK) return rc_; } while (0)
#define TKR_LAUNCH_CHECK()                    \
    do {                                      \
        hipError_t _e = hipGetLastError();    \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

namespace tkr {

// ---- Philox4x32-10 (Salmon et al.), the build's counter-based stream -----------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// floor((hi<<32 | lo) * n / 2^64), 0 < n < 2^32
__device__ __forceinline__ uint32_t mulhi64(uint32_t lo, uint32_t hi, uint32_t n) {
    return (uint32_t)__umul64hi(((uint64_t)hi << 32) | lo, (uint64_t)n);
}

// ---- wave64 reductions (DPP; fixed tree -> bitwise repeatable) --------------------------------
// v += dpp(v): lanes whose source is masked off add 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __int_as_float(moved);
}

// sum over the 64 lanes, returned to every lane (uniform)
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xb1>(v);              // quad_perm [1,0,3,2]
    v = dpp_add<0x4e>(v);              // quad_perm [2,3,0,1]
    v = dpp_add<0x124>(v);             // row_ror:4
    v = dpp_add<0x128>(v);             // row_ror:8   -> every lane of a 16-lane row holds the row sum
    v = dpp_add<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ void wave_sum2(float& a, float& b) {
    a = wave_sum(a);
    b = wave_sum(b);
}

// One batch's loss is summed by hundreds of waves on ONE word.  atomicAdd(float*) compiles to a compare-and-swap LOOP here (the
// compiler must assume fine-grained memory): 250 waves retrying against each other cost 0.85 us of a 2.2 us batch.  The tables of
// this library live in ordinary device memory, where the hardware's global_atomic_add_f32 is exact enough (an fp32 add in L2).
__device__ __forceinline__ void loss_add(float* p, float v) { unsafeAtomicAdd(p, v); }
// ... and where many workgroups of one launch add to one batch's loss (K3): 64 slots, one cache line apart, picked by the
// workgroup number -- a word takes ~10 ns per atomic, one after the other; 64 words take them side by side.  loss_slots_kernel
// adds a call's slots up afterwards (csrc/vbpr_step.hip).
constexpr int kLossSlots = 64, kLossSlotStride = 32;                 // floats
__device__ __forceinline__ void loss_add_spread(float* slots, float v) {
    unsafeAtomicAdd(slots + (blockIdx.x & (kLossSlots - 1)) * kLossSlotStride, v);
}
__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float bcast_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// sigma(-x) and log(1+exp(-x)) in the stable forms the oracle uses (oracle/ref_np.py).
__device__ __forceinline__ float sigmoid_neg(float x) {
    const float e = expf(-fabsf(x));
    return x >= 0.f ? e / (1.f + e) : 1.f / (1.f + e);
}
__device__ __forceinline__ float softplus_neg(float x) { return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))); }

// The [B, B] pair sums of the VBPR objective evaluate sigma(-(alpha_a + beta_b)) for every pair of a batch: B^2 exponentials
// (67 M at batch 8192: 110 of the step's 210 us).  sigma(-(a + b)) = 1 / (1 + e^a * e^b): the producers of alpha / beta also store
// e^alpha, e^beta (pair_exp: arguments clamped to +-80, beyond which the sigmoid is 0 or 1 in fp32 anyway, so the product never
// meets 0 * inf), and a pair term is one fma and one reciprocal (pair_sigmoid).  Relative error ~3e-7 per term against the
// stable form of the oracle; the tolerance of the step tests is 3e-4.
__device__ __forceinline__ float pair_exp(float x) { return __expf(fminf(fmaxf(x, -80.f), 80.f)); }
__device__ __forceinline__ float pair_sigmoid(float ea, float eb) { return __builtin_amdgcn_rcpf(fmaf(ea, eb, 1.f)); }
// ... and the pair's term of the loss, log(1 + e^-(a + b)) = log(1 + e^a e^b) - (a + b): one hardware log on the denominator the
// sigmoid needs anyway (the stable libm form -- an exp and a log1p per pair, B^2 pairs -- was 2.9 of the 25.7 us of a 256-batch).
// Beyond x = 30 the term is below 1e-13 (and the difference of two 30s has no bits left for it): 0.
__device__ __forceinline__ float pair_softplus_neg(float ea, float eb, float x) {
    return x > 30.f ? 0.f : __logf(fmaf(ea, eb, 1.f)) - x;
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

}  // namespace tkr
