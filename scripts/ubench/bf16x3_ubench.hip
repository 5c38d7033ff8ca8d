// K4-shaped inner loop with the 6-product bf16 split: 8 waves/WG, per "tile" 8 k-steps x (3 ds_read_b128 + 6 MFMA 32x32x16 bf16) + barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ROWB = 3 * 256 + 16;     // bytes per item row: 3 parts x 128 bf16 + pad
__global__ __launch_bounds__(512) void k(float* out, int tiles, int do_barrier) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 2 * 32 * ROWB / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.25f;
    __syncthreads();
    const int lane = threadIdx.x & 63, m = lane & 31, kg = lane >> 5;
    bf16x8 b[8][3];
#pragma unroll
    for (int s = 0; s < 8; ++s) for (int p = 0; p < 3; ++p) for (int e = 0; e < 8; ++e) b[s][p][e] = (__bf16)(0.001f * (float)(s + p + e + lane));
    float tot = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const unsigned char* arow = lds + (t & 1) * 32 * ROWB + m * ROWB + kg * 16;
        f32x16 acc = (f32x16){0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(arow + s * 32);
            const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(arow + 256 + s * 32);
            const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(arow + 512 + s * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b[s][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[s][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[s][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[s][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[s][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[s][0], acc, 0, 0, 0);
        }
        float mx = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
        tot += mx;
        if (do_barrier) __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int bar = 0; bar < 2; ++bar) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int tiles = 4448;        // 556 x 8 rounds
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 2 * 32 * ROWB, 0, d, 10, bar);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 2 * 32 * ROWB, 0, d, tiles, bar);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("barrier=%d: %d tiles %.3f ms = %.2f us/tile  (fp32-equivalent %.1f TFLOP/s)\n", bar, tiles, ms, ms * 1e3 / tiles,
               256.0 * 256 * 32 * 128 * 2 * tiles / (ms * 1e-3) / 1e12);
    }
    return 0;
}
