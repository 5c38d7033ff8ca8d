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

}  // namespace tkr

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
