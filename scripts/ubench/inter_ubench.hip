// same-wave interleave: N independent VALU fmas between consecutive (dependent) MFMAs -- do they hide in the MFMA shadow?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
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
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.01f * (i + lane);
    const float* arow = tile + ul * 132 + h * 64;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(arow + 4 * g);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[(4 * g + q) % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b[4 * g + q], acc[(4 * g + q) % CHAINS], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) x[v % 16] = fmaf(x[v % 16], 1.0001f, 0.0001f);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
                if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);   // NV VALU
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int CHAINS> void run(float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, CHAINS>), dim3(256), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, CHAINS>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chains=%d, %2d VALU per MFMA: %.3f ms\n", CHAINS, NV, ms);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 1>(d); run<4, 1>(d); run<8, 1>(d); run<12, 1>(d); run<16, 1>(d); run<24, 1>(d);
    run<0, 2>(d); run<8, 2>(d); run<16, 2>(d);
    return 0;
}
