// Ceiling of K2's access pattern by row layout: a wave per task reads a random row (param + slot), adds, writes it back.
//   split:       two 512-byte rows in two tables ([n][k] param, [n][k] slot)   -- the DoubleTable layout of csrc/bpr_step.hip
//   interleaved: one 1 KB row [param | slot]
// hipcc --offload-arch=gfx950 -O3 scripts/micro/rowrmw.hip -o scripts/micro/rowrmw && scripts/micro/rowrmw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool INTER>
__global__ __launch_bounds__(256) void rmw(float* __restrict__ P, float* __restrict__ M, const int* __restrict__ idx, int n_tasks, int k) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = (gridDim.x * blockDim.x) >> 6;
    for (int t = wave; t < n_tasks; t += nw) {
        const int r = idx[t];
        float2* p = reinterpret_cast<float2*>(INTER ? P + (size_t)r * 2 * k : P + (size_t)r * k) + lane;
        float2* m = reinterpret_cast<float2*>(INTER ? P + (size_t)r * 2 * k + k : M + (size_t)r * k) + lane;
        float2 a = *p, b = *m;
        b.x = 0.9f * b.x + 0.1f * a.x * a.x; b.y = 0.9f * b.y + 0.1f * a.y * a.y;
        a.x -= 1e-4f * a.x * rsqrtf(b.x + 1e-10f); a.y -= 1e-4f * a.y * rsqrtf(b.y + 1e-10f);
        *p = a; *m = b;
    }
}

int main() {
    const int k = 128, n = 1 << 20;                     // 1 M rows x 1 KB = 1 GB: nothing stays in a cache
    float *P, *M; int* idx;
    hipMalloc(&P, (size_t)n * 2 * k * 4); hipMalloc(&M, (size_t)n * k * 4);
    hipMemset(P, 0x3c, (size_t)n * 2 * k * 4); hipMemset(M, 0x3c, (size_t)n * k * 4);
    for (int tasks : {24576, 196608, 1 << 20}) {
        std::vector<int> h(tasks);
        // distinct rows (a batch's tasks are distinct rows): a random permutation's prefix
        std::vector<int> perm(n); for (int i = 0; i < n; ++i) perm[i] = i;
        srand(7); for (int i = 0; i < tasks; ++i) { int j = i + rand() % (n - i); std::swap(perm[i], perm[j]); h[i] = perm[i]; }
        hipMalloc(&idx, tasks * 4); hipMemcpy(idx, h.data(), tasks * 4, hipMemcpyHostToDevice);
        for (int inter = 0; inter < 2; ++inter) {
            for (int grid : {1024, 2048, 4096}) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 3; ++rep) { if (inter) rmw<true><<<grid, 256>>>(P, M, idx, tasks, k); else rmw<false><<<grid, 256>>>(P, M, idx, tasks, k); }
                hipEventRecord(e0);
                for (int rep = 0; rep < 10; ++rep) { if (inter) rmw<true><<<grid, 256>>>(P, M, idx, tasks, k); else rmw<false><<<grid, 256>>>(P, M, idx, tasks, k); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
                printf("%-11s %8d tasks grid %4d: %7.1f us  %.2f TB/s (read + write, 2 KB per task)\n", inter ? "interleaved" : "split", tasks, grid, ms * 1e3,
                       (double)tasks * 2048 / ms / 1e9);
            }
        }
        hipFree(idx);
    }
    return 0;
}
