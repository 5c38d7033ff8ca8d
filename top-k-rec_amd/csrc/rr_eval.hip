// K6 / K7 -- the rank walk of utils.evaluate (utils.py:101-127) on top of K4's filtered lists.
//
// utils.evaluate differs from the CLI's walk (evaluate.py:96-105) in two ways: a hit is bucketed by the item's
// RAW rank t (train-rated items included, utils.py:113-117: j = t // step) and reciprocal ranks 1/(t+1) are
// summed next to the hit counts.  K4 yields the first `total` unrated columns of a user in order; the raw rank of
// the p-th of them is p + (number of the user's train-rated columns ranked before it).  K6 computes exactly that
// count: one wave per row scores the kept columns and the rated columns of its user with ONE dot routine (so
// the comparisons are self-consistent) and counts, for every kept column, the rated ones in front of it under the
// canonical order (descending score, ties -> higher column first).  K7 turns (kept ids, raw ranks, like lists) into
// per-row first-bucket hit counts and reciprocal-rank sums; the host sums rows and accumulates buckets
// (utils.py:115-117: for k in range(j, interval)).
//
// Integer/byte work bound by the row gathers of Vt: (n_rated + total) rows of k floats per user.
#include "tkr_common.h"
#include "../../include/tkr.h"

namespace tkr {

constexpr int kRawMaxK = 256;      // kept columns per row a launch handles

template <int NE>
__device__ __forceinline__ float row_dot(const float (&u)[NE], const float* __restrict__ row, int k, int lane) {
    float p = 0.f;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = lane * NE + q;
        const float v = row[min(e, k - 1)];
        p = fmaf(u[q], (e < k) ? v : 0.f, p);
    }
    return wave_sum(p);
}

template <int NE>
__global__ __launch_bounds__(256) void raw_rank_kernel(const float* __restrict__ U, const int32_t* __restrict__ uidx,
                                                       int n_rows, const float* __restrict__ Vt,
                                                       const float* __restrict__ bias, int k,
                                                       const int64_t* __restrict__ rated_ptr,
                                                       const int32_t* __restrict__ rated_cols,
                                                       const int32_t* __restrict__ ids, int K,
                                                       int32_t* __restrict__ raw_rank) {
    __shared__ float s_score[4][kRawMaxK];
    __shared__ int s_before[4][kRawMaxK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    const int urow = uidx ? uidx[r] : r;
    float u[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = lane * NE + q;
        const float v = U[(size_t)urow * k + min(e, k - 1)];
        u[q] = (e < k) ? v : 0.f;
    }
    // kept columns: score with the same routine as the rated ones
    for (int p = 0; p < K; ++p) {
        const int c = ids[(size_t)r * K + p];                    // wave-uniform
        float s = -INFINITY;
        if (c >= 0) s = row_dot<NE>(u, Vt + (size_t)c * k, k, lane) + (bias ? bias[c] : 0.f);
        if (lane == 0) { s_score[wave][p] = s + 0.0f; s_before[wave][p] = 0; }
    }
    __builtin_amdgcn_wave_barrier();
    for (int64_t q = rated_ptr[r]; q < rated_ptr[r + 1]; ++q) {
        const int c = rated_cols[q];
        const float s = row_dot<NE>(u, Vt + (size_t)c * k, k, lane) + (bias ? bias[c] : 0.f) + 0.0f;
        for (int p = lane; p < K; p += 64) {
            const int cp = ids[(size_t)r * K + p];
            const float sp = s_score[wave][p];
            if (cp >= 0 && (s > sp || (s == sp && c > cp))) s_before[wave][p] += 1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < K; p += 64)
        raw_rank[(size_t)r * K + p] = ids[(size_t)r * K + p] >= 0 ? p + s_before[wave][p] : -1;
}

// one thread per row: sequential over the kept positions, so the reciprocal-rank sums are reproducible
__global__ void count_hits_rr_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ raw_rank, int n_rows, int K,
                                     const int64_t* __restrict__ like_ptr, const int32_t* __restrict__ like_cols, int step,
                                     int interval, int32_t* __restrict__ hit_first, double* __restrict__ rr_first) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t lo0 = like_ptr[r], hi0 = like_ptr[r + 1];
    for (int p = 0; p < K; ++p) {
        const int c = ids[(size_t)r * K + p];
        if (c < 0) break;
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (like_cols[mid] < c) lo = mid + 1; else hi = mid;
        }
        if (lo < hi0 && like_cols[lo] == c) {
            const int t = raw_rank[(size_t)r * K + p];
            const int j = t / step;
            if (j < interval) {
                hit_first[(size_t)r * interval + j] += 1;
                rr_first[(size_t)r * interval + j] += 1.0 / (double)(t + 1);
            }
        }
    }
}

}  // namespace tkr

extern "C" int tkr_raw_ranks(const float* U, const int32_t* user_idx, int32_t n_rows, const float* Vt, const float* bias,
                             int32_t k, const int64_t* rated_ptr, const int32_t* rated_cols, const int32_t* ids, int32_t K,
                             int32_t* raw_rank, void* stream) {
    if (!U || !Vt || !rated_ptr || !ids || !raw_rank || n_rows <= 0 || k <= 0 || K <= 0) return TKR_EINVAL;
    if (K > tkr::kRawMaxK || k > 256) return TKR_EUNSUPPORTED;
    const dim3 grid((n_rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch ((k + 63) / 64) {
        case 1: hipLaunchKernelGGL(tkr::raw_rank_kernel<1>, grid, block, 0, s, U, user_idx, n_rows, Vt, bias, k, rated_ptr, rated_cols, ids, K, raw_rank); break;
        case 2: hipLaunchKernelGGL(tkr::raw_rank_kernel<2>, grid, block, 0, s, U, user_idx, n_rows, Vt, bias, k, rated_ptr, rated_cols, ids, K, raw_rank); break;
        case 3: hipLaunchKernelGGL(tkr::raw_rank_kernel<3>, grid, block, 0, s, U, user_idx, n_rows, Vt, bias, k, rated_ptr, rated_cols, ids, K, raw_rank); break;
        default: hipLaunchKernelGGL(tkr::raw_rank_kernel<4>, grid, block, 0, s, U, user_idx, n_rows, Vt, bias, k, rated_ptr, rated_cols, ids, K, raw_rank); break;
    }
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_count_hits_rr(const int32_t* ids, const int32_t* raw_rank, int32_t n_rows, int32_t K,
                                 const int64_t* like_ptr, const int32_t* like_cols, int32_t step, int32_t interval,
                                 int32_t* hit_first, double* rr_first, void* stream) {
    if (!ids || !raw_rank || !like_ptr || !hit_first || !rr_first || n_rows <= 0 || K <= 0 || step <= 0 || interval <= 0)
        return TKR_EINVAL;
    hipLaunchKernelGGL(tkr::count_hits_rr_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, raw_rank,
                       n_rows, K, like_ptr, like_cols, step, interval, hit_first, rr_first);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}
