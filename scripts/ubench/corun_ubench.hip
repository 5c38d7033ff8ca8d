// does VALU/LDS work of one wave overlap the MFMA chain of the other wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// mode bit0: waves 0-3 run MFMA chains; bit1: waves 4-7 run a filter-like VALU+LDS loop
__global__ __launch_bounds__(512) void k(float* out, int iters, int viters, int mode, int kind) {
    __shared__ __attribute__((aligned(16))) float tile[32 * 132];
    __shared__ int cnt[256];
    __shared__ float cs[64 * 256];
    for (int i = threadIdx.x; i < 32 * 132; i += blockDim.x) tile[i] = (float)(i & 7) * 0.125f;
    if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ul = lane & 31, h = lane >> 5;
    float s = 0.f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        float b[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) b[i] = 0.001f * (float)(i + lane);
        f32x16 acc = (f32x16){0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
        const float* arow = tile + ul * 132 + h * 64;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 a = *reinterpret_cast<const float4*>(arow + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * g + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * g + 3], acc, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += acc[r];
    } else {
        if (!(mode & 2)) return;
        if (mode & 4) __builtin_amdgcn_s_setprio(3);
        const int uw = (wave - 4) * 32 + ul;
        float thr = 0.5f, x = 0.001f * lane;
        if (kind == 0) {                                    // pure VALU: dependent fma chain, 64 per iteration
            for (int it = 0; it < viters; ++it) {
#pragma unroll
                for (int q = 0; q < 64; ++q) x = fmaf(x, 1.0001f, 0.0001f);
            }
            s = x;
        } else if (kind == 1) {                             // filter-like: 16 compares, popc, shuffle, LDS count read, ballot
            for (int it = 0; it < viters; ++it) {
                unsigned hits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) hits |= (unsigned)((x + 0.01f * r) >= thr) << r;
                const int mine = __popc(hits);
                const int total = mine + __shfl_xor(mine, 32, 64);
                const bool need = total > 0 && (cnt[uw] + total > 64);
                unsigned long long pending = __ballot(need);
                if (pending) { cnt[uw] = 0; thr += 0.001f; }
                if (mine) { int pos = atomicAdd(&cnt[uw], 1) & 63; cs[pos * 128 + uw] = x; }
                x = x * 0.999f + 0.0001f * (float)it;
            }
            s = x + thr;
        } else {                                            // LDS-only: dependent ds_read chain
            int p = uw;
            for (int it = 0; it < viters; ++it) {
#pragma unroll
                for (int q = 0; q < 16; ++q) p = (cnt[p & 255] + p + 1) & 255;
            }
            s = (float)p;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
float run(float* d, int iters, int viters, int mode, int kind) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 10, 10, mode, kind);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, viters, mode, kind);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 2000;
    const float m = run(d, iters, 0, 1, 0);
    printf("MFMA alone (1 wave/SIMD, %d x 64 MFMA): %.3f ms\n", iters, m);
    const char* names[3] = {"VALU fma chain", "filter-like", "LDS chain"};
    for (int kind = 0; kind < 3; ++kind) {
        int viters = 2000;
        float v = run(d, 0, viters, 2, kind);
        viters = (int)(viters * m / v);                      // calibrate to the MFMA duration
        v = run(d, 0, viters, 2, kind);
        const float both = run(d, iters, viters, 3, kind);
        const float prio = run(d, iters, viters, 7, kind);
        printf("%-16s alone %.3f ms | both %.3f ms, with setprio %.3f ms (sum %.3f, max %.3f)\n", names[kind], v, both, prio, m + v, m > v ? m : v);
    }
    return 0;
}
