// microbenchmark: fp32 MFMA 32x32x2 issue rate per SIMD vs waves per SIMD and number of accumulator chains,
// operands fed from LDS with ds_read_b128 exactly as K4 does.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float tile[32 * 132];
    for (int i = threadIdx.x; i < 32 * 132; i += blockDim.x) tile[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, ul = lane & 31, h = lane >> 5;
    float b[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) b[i] = 0.001f * (float)(i + lane);
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x16){0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    const float* arow = tile + ul * 132 + h * 64;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(arow + 4 * g);
            acc[g % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * g + 0], acc[g % CHAINS], 0, 0, 0);
            acc[g % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * g + 1], acc[g % CHAINS], 0, 0, 0);
            acc[g % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * g + 2], acc[g % CHAINS], 0, 0, 0);
            acc[g % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * g + 3], acc[g % CHAINS], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run(int waves_per_simd, float* d) {
    const int threads = 256 * waves_per_simd, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 64.0 * iters * waves_per_simd;         // per SIMD
    const double tf = 256.0 * 4 * mfma * 4096 / (ms * 1e-3) / 1e12;
    printf("chains=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", CHAINS, waves_per_simd, ms, tf, ms * 1e6 / mfma);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int w = 1; w <= 2; ++w) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
    return 0;
}
