// Profiling aid (not on the product path): a row copy with EXACTLY the access pattern of the step
// kernels (one wave per row, 4 B per lane, 256 B per wave-instruction, rows picked by an index list),
// used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against a known byte count
// (MI355X_MICROARCH.md §HBM: counters must be calibrated in your own access pattern).
#include "tkr_common.h"
#include "../../include/tkr.h"

namespace tkr {
__global__ __launch_bounds__(1024) void calib_rowcopy_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            const int32_t* __restrict__ rows, int n, int k) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (w >= n) return;
    const int r = rows[w];
    for (int e = lane; e < k; e += 64) dst[(size_t)r * k + e] = src[(size_t)r * k + e] + 1.0f;
}
}  // namespace tkr

extern "C" int tkr_calib_rowcopy(const float* src, float* dst, const int32_t* rows, int32_t n, int32_t k, void* stream) {
    if (!src || !dst || !rows || n <= 0 || k <= 0) return TKR_EINVAL;
    hipLaunchKernelGGL(tkr::calib_rowcopy_kernel, dim3((n + 15) / 16), dim3(1024), 0, (hipStream_t)stream, src, dst, rows, n, k);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}
