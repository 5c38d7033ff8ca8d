// Pack / unpack of the replicated (item-side) tables around the per-epoch all-reduce (top-k-rec_amd/dist.py).
//
// The reference is single-process; users are sharded over the GPUs here and the item tables are reconciled once per
// epoch by   P <- P0 + sum_g (P_g - P0),   ms <- mean_g ms_g   (SURVEY.md §8e, H4).  A double-buffered table keeps
// the current value of row r in buffer (cnt[r] & 1) (see bpr_step.hip), so "current" is a gather by parity.  These three
// kernels replace ~25 framework ops per exchange (240 us of launch overhead against a 2.1 ms epoch at 8 GPUs) by
// one launch per table and direction:
//   snapshot  start[r]       = P[par(r)][r]
//   pack      flat_delta[r]  = P[par(r)][r] - start[r];   flat_ms[r] = ms[par(r)][r] * inv_world
//   unpack    P[0][r] = start[r] + flat_delta[r];  ms[0][r] = flat_ms[r]      (the caller zeroes cnt afterwards)
// cnt == NULL means a single-buffered dense table (VBPR cem / icb).  Pure bandwidth: 3-4 floats moved per element.
#include "tkr_common.h"
#include "../../include/tkr.h"

namespace tkr {

__global__ void sync_snapshot_kernel(const float* __restrict__ P, const int32_t* __restrict__ cnt, float* __restrict__ start,
                                     int64_t n, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * w) return;
    const int64_t r = i / w;
    const int64_t par = cnt ? (cnt[r] & 1) : 0;
    start[i] = P[par * n * w + i];
}

__global__ void sync_pack_kernel(const float* __restrict__ P, const float* __restrict__ ms, const int32_t* __restrict__ cnt,
                                 const float* __restrict__ start, float* __restrict__ flat_delta, float* __restrict__ flat_ms,
                                 int64_t n, int w, float inv_world) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * w) return;
    const int64_t r = i / w;
    const int64_t par = cnt ? (cnt[r] & 1) : 0;
    flat_delta[i] = P[par * n * w + i] - start[i];
    flat_ms[i] = ms[par * n * w + i] * inv_world;
}

__global__ void sync_unpack_kernel(float* __restrict__ P, float* __restrict__ ms, const float* __restrict__ start,
                                   const float* __restrict__ flat_delta, const float* __restrict__ flat_ms, int64_t n, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * w) return;
    P[i] = start[i] + flat_delta[i];                             // buffer 0 becomes current
    ms[i] = flat_ms[i];
}

// ---- the same three steps for the GRANULE tables of the dataflow step (csrc/bpr_flow.hip) -----------------------------
// V / msV: [nbuf][n][kp] granules {fp32 value, uint32 version tag}, nbuf = 2 or 4 (tkr_flow_state.item_bufs); the item bias and its
// slot are granules 0 and 1 of the row's tail [nbuf][n][tg], tg = 4 (two buffers) or 8 (four).  "Current" is buffer cnt[r] & (nbuf - 1).  The flat vectors hold the n*k elements of V followed by the n biases.
// unpack re-creates what a fresh assignment of the tables looks like: version 0 of every row in buffer 0 (tags 0, padding 0 / slot
// padding 1), no version in buffer 1 (tags 0xffffffff), expect = 0, rd = 0, update counters 0 -- one launch instead of ~20
// framework ops (fills, strided copies, tag resets) on 2 x 21 MB: measured 464 -> ~60 us per exchange at the ML-10M shape.
// One thread per PAIR of granules (16 bytes: the access width of csrc/bpr_flow.hip), 64 threads per 128-granule row, no
// integer division (round 2: one thread per element with i / (k + 1) and 8-byte accesses: 110 us for the three launches of an
// exchange at the ML-10M shape).  The row's tail (bias, its slot) rides with thread 0 of the row.
__global__ __launch_bounds__(256) void sync_flow_snapshot_kernel(const float4* __restrict__ P, const float2* __restrict__ tail,
                                                                const int32_t* __restrict__ cnt, float* __restrict__ start, int n,
                                                                int k, int kp, int nbuf) {
    const int hp = kp >> 1;                                      // pairs per row
    const int tg = nbuf == 4 ? 8 : 4;
    const int r = blockIdx.x * (256 / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t par = cnt[r] & (nbuf - 1);
    for (int q = threadIdx.x & 63; q < hp; q += 64) {
        const float4 g = P[(par * n + r) * hp + q];
        const int c = 2 * q;
        if (c < k) start[(int64_t)r * k + c] = g.x;
        if (c + 1 < k) start[(int64_t)r * k + c + 1] = g.z;
    }
    if ((threadIdx.x & 63) == 0) start[(int64_t)n * k + r] = tail[(par * n + r) * tg + 0].x;
}

__global__ __launch_bounds__(256) void sync_flow_pack_kernel(const float4* __restrict__ P, const float4* __restrict__ M,
                                                            const float2* __restrict__ tail, const int32_t* __restrict__ cnt,
                                                            const float* __restrict__ start, float* __restrict__ flat_delta,
                                                            float* __restrict__ flat_ms, int n, int k, int kp, float inv_world, int nbuf) {
    const int hp = kp >> 1;
    const int tg = nbuf == 4 ? 8 : 4;
    const int r = blockIdx.x * (256 / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t par = cnt[r] & (nbuf - 1);
    for (int q = threadIdx.x & 63; q < hp; q += 64) {
        const float4 g = P[(par * n + r) * hp + q], m = M[(par * n + r) * hp + q];
        const int c = 2 * q;
        const int64_t o = (int64_t)r * k + c;
        if (c < k) { flat_delta[o] = g.x - start[o]; flat_ms[o] = m.x * inv_world; }
        if (c + 1 < k) { flat_delta[o + 1] = g.z - start[o + 1]; flat_ms[o + 1] = m.z * inv_world; }
    }
    if ((threadIdx.x & 63) == 0) {
        const int64_t o = (int64_t)n * k + r, g = (par * n + r) * tg;
        flat_delta[o] = tail[g + 0].x - start[o];
        flat_ms[o] = tail[g + 1].x * inv_world;
    }
}

// unpack also leaves the NEW values in `start`: they are the next epoch's starting point, so the next exchange needs no
// snapshot launch (dist.ItemSync keeps track of whether anything else touched the tables in between)
__global__ __launch_bounds__(256) void sync_flow_unpack_kernel(float4* __restrict__ P, float4* __restrict__ M, float2* __restrict__ tail,
                                                              uint32_t* __restrict__ rd, int32_t* __restrict__ cnt,
                                                              float* __restrict__ start, const float* __restrict__ flat_delta,
                                                              const float* __restrict__ flat_ms, int n, int k, int kp, int nbuf) {
    const int hp = kp >> 1;
    const int tg = nbuf == 4 ? 8 : 4;
    const int r = blockIdx.x * (256 / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    const float none = __uint_as_float(0xffffffffu), zero_tag = __uint_as_float(0u);
    for (int q = threadIdx.x & 63; q < hp; q += 64) {
        const int c = 2 * q;
        const int64_t o = (int64_t)r * k + c;
        float v0 = 0.f, v1 = 0.f, m0 = 1.f, m1 = 1.f;            // padding: value 0, slot 1
        if (c < k) { v0 = start[o] + flat_delta[o]; m0 = flat_ms[o]; start[o] = v0; }
        if (c + 1 < k) { v1 = start[o + 1] + flat_delta[o + 1]; m1 = flat_ms[o + 1]; start[o + 1] = v1; }
        P[(int64_t)r * hp + q] = make_float4(v0, zero_tag, v1, zero_tag);
        M[(int64_t)r * hp + q] = make_float4(m0, zero_tag, m1, zero_tag);
        for (int64_t bf = 1; bf < nbuf; ++bf) {
            P[(bf * n + r) * hp + q] = make_float4(0.f, none, 0.f, none);
            M[(bf * n + r) * hp + q] = make_float4(0.f, none, 0.f, none);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        const int64_t ob = (int64_t)n * k + r;
        const float b = start[ob] + flat_delta[ob];
        start[ob] = b;
        tail[(int64_t)r * tg + 0] = make_float2(b, zero_tag);
        tail[(int64_t)r * tg + 1] = make_float2(flat_ms[ob], zero_tag);
        for (int c = 2; c < tg; ++c) tail[(int64_t)r * tg + c] = make_float2(0.f, zero_tag);
        for (int64_t bf = 1; bf < nbuf; ++bf)
            for (int c = 0; c < tg; ++c) tail[(bf * n + r) * tg + c] = make_float2(0.f, none);
        for (int c = 0; c < nbuf; ++c) rd[(int64_t)nbuf * r + c] = 0u;
        cnt[r] = 0;
    }
}

}  // namespace tkr

extern "C" int tkr_sync_flow_snapshot(const void* P, const void* tail, const int32_t* cnt, float* start, int32_t n, int32_t k,
                                      int32_t item_bufs, void* stream) {
    if (!P || !tail || !cnt || !start || n <= 0 || k <= 0 || (item_bufs != 2 && item_bufs != 4)) return TKR_EINVAL;
    const int kp = (k + 127) / 128 * 128;
    hipLaunchKernelGGL(tkr::sync_flow_snapshot_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const float4*>(P), static_cast<const float2*>(tail), cnt, start, n, k, kp, item_bufs);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_sync_flow_pack(const void* P, const void* M, const void* tail, const int32_t* cnt, const float* start,
                                  float* flat_delta, float* flat_ms, int32_t n, int32_t k, float inv_world, int32_t item_bufs, void* stream) {
    if (!P || !M || !tail || !cnt || !start || !flat_delta || !flat_ms || n <= 0 || k <= 0 || (item_bufs != 2 && item_bufs != 4)) return TKR_EINVAL;
    const int kp = (k + 127) / 128 * 128;
    hipLaunchKernelGGL(tkr::sync_flow_pack_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const float4*>(P), static_cast<const float4*>(M), static_cast<const float2*>(tail), cnt, start,
                       flat_delta, flat_ms, n, k, kp, inv_world, item_bufs);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_sync_flow_unpack(void* P, void* M, void* tail, uint32_t* rd, int32_t* cnt, float* start,
                                    const float* flat_delta, const float* flat_ms, int32_t n, int32_t k, int32_t item_bufs, void* stream) {
    if (!P || !M || !tail || !rd || !cnt || !start || !flat_delta || !flat_ms || n <= 0 || k <= 0 || (item_bufs != 2 && item_bufs != 4)) return TKR_EINVAL;
    const int kp = (k + 127) / 128 * 128;
    hipLaunchKernelGGL(tkr::sync_flow_unpack_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<float4*>(P), static_cast<float4*>(M), static_cast<float2*>(tail), rd, cnt, start, flat_delta,
                       flat_ms, n, k, kp, item_bufs);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_sync_snapshot(const float* P, const int32_t* cnt, float* start, int64_t n, int32_t w, void* stream) {
    if (!P || !start || n <= 0 || w <= 0) return TKR_EINVAL;
    hipLaunchKernelGGL(tkr::sync_snapshot_kernel, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, cnt,
                       start, n, w);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_sync_pack(const float* P, const float* ms, const int32_t* cnt, const float* start, float* flat_delta,
                             float* flat_ms, int64_t n, int32_t w, float inv_world, void* stream) {
    if (!P || !ms || !start || !flat_delta || !flat_ms || n <= 0 || w <= 0) return TKR_EINVAL;
    hipLaunchKernelGGL(tkr::sync_pack_kernel, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, ms, cnt,
                       start, flat_delta, flat_ms, n, w, inv_world);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_sync_unpack(float* P, float* ms, const float* start, const float* flat_delta, const float* flat_ms, int64_t n,
                               int32_t w, void* stream) {
    if (!P || !ms || !start || !flat_delta || !flat_ms || n <= 0 || w <= 0) return TKR_EINVAL;
    hipLaunchKernelGGL(tkr::sync_unpack_kernel, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, ms,
                       start, flat_delta, flat_ms, n, w);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}
