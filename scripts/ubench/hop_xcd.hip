// Hand-off latency of one 1 KiB tagged row between two waves, same XCD vs different XCDs, for the store flavours a
// forwarding scheme could use.  Ping-pong: A writes row R0 with tag i, B polls R0 for tag i and writes R1 with tag i, A polls R1.
// One hop = total / (2 N).   hipcc --offload-arch=gfx950 -O3 scripts/ubench/hop_xcd.hip -o /tmp/hop_xcd && /tmp/hop_xcd
//   store 0: sc1 (write-through, what K2f does)      load 16: sc1
//   store 1: sc1 followed by a plain store of the same bytes (keeps the line in the writer's L2)
//   store 2: plain only (valid on the same XCD only)
//   store 3: sc0 sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int STORE>
__device__ __forceinline__ void put(uint32_t* row, int lane, uint32_t tag) {
    v4u x; x.x = lane; x.y = tag; x.z = lane + 64; x.w = tag;
    const __amdgpu_buffer_rsrc_t r = rsrc(row, 1024);
    if (STORE == 0) __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, 0, 16);
    if (STORE == 1) { __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, 0, 16); __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, 0, 0); }
    if (STORE == 2) __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, 0, 0);
    if (STORE == 3) __builtin_amdgcn_raw_buffer_store_b128(x, r, lane * 16, 0, 17);
}
template <int LOAD>
__device__ __forceinline__ bool poll(const uint32_t* row, int lane, uint32_t tag, uint32_t& spins) {
    const __amdgpu_buffer_rsrc_t r = rsrc(row, 1024);
    for (uint32_t it = 0; it < 2000000u; ++it) {
        const v4u x = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, LOAD);
        asm volatile("" ::: "memory");
        if (__all(x.y == tag && x.w == tag)) { spins += it; return true; }
    }
    return false;
}
// the link as K2f builds it: two full rows and a 32-byte tail (two lanes), all three polled and validated per pass
template <int TAIL_LANES>
__device__ __forceinline__ void put3(uint32_t* base, int lane, uint32_t tag) {
    v4u x; x.x = lane; x.y = tag; x.z = lane + 64; x.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc(base, 1024), lane * 16, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc(base + 1024, 1024), lane * 16, 0, 16);
    if (lane < TAIL_LANES) __builtin_amdgcn_raw_buffer_store_b128(x, rsrc(base + 2048, 1024), lane * 16, 0, 16);
}
template <int TAIL_LANES>
__device__ __forceinline__ bool poll3(const uint32_t* base, int lane, uint32_t tag, uint32_t& spins) {
    for (uint32_t it = 0; it < 2000000u; ++it) {
        const v4u a = __builtin_amdgcn_raw_buffer_load_b128(rsrc(base, 1024), lane * 16, 0, 16);
        const v4u b = __builtin_amdgcn_raw_buffer_load_b128(rsrc(base + 1024, 1024), lane * 16, 0, 16);
        const v4u c = __builtin_amdgcn_raw_buffer_load_b128(rsrc(base + 2048, 1024), (lane % TAIL_LANES) * 16, 0, 16);
        asm volatile("" ::: "memory");
        if (__all(a.y == tag && a.w == tag && b.y == tag && b.w == tag && c.y == tag && c.w == tag)) { spins += it; return true; }
    }
    return false;
}
template <int TAIL_LANES>
__global__ void pingpong3(uint32_t* rows, int a_block, int b_block, int n, unsigned long long* out) {
    const int lane = threadIdx.x;
    if ((int)blockIdx.x != a_block && (int)blockIdx.x != b_block) return;
    uint32_t* r0 = rows; uint32_t* r1 = rows + 4096;
    uint32_t spins = 0;
    bool ok = true;
    if ((int)blockIdx.x == a_block) {
        for (int i = 1; i <= n && ok; ++i) { put3<TAIL_LANES>(r0, lane, i); ok = poll3<TAIL_LANES>(r1, lane, i, spins); }
    } else {
        for (int i = 1; i <= n && ok; ++i) { ok = poll3<TAIL_LANES>(r0, lane, i, spins); put3<TAIL_LANES>(r1, lane, i); }
    }
    if (lane == 0) { out[(blockIdx.x == a_block ? 0 : 2)] = ok; out[(blockIdx.x == a_block ? 1 : 3)] = spins; }
}
template <int TAIL_LANES>
static void run3(const char* what, uint32_t* rows, int a, int b, unsigned long long* out, int n) {
    hipMemset(rows, 0, 16384 * 4); hipMemset(out, 0, 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pingpong3<TAIL_LANES>), dim3(64), dim3(64), 0, 0, rows, a, b, n, out);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("  %-44s %7.3f us per hop (%s; %.1f polls per hop)\n", what, ms * 1e3 / (2.0 * n), h[0] && h[2] ? "ok" : "TIMED OUT",
           (double)(h[1] + h[3]) / (2.0 * n));
}

// blocks: every block reports its XCC; the host picked roles (block ids of A and B) from a census run
template <int STORE, int LOAD>
__global__ void pingpong(uint32_t* rows, int a_block, int b_block, int n, unsigned long long* out, uint32_t* xcc) {
    const int lane = threadIdx.x;
    if (lane == 0) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blockIdx.x] = id & 0xf; }
    if ((int)blockIdx.x != a_block && (int)blockIdx.x != b_block) return;
    uint32_t* r0 = rows; uint32_t* r1 = rows + 4096;
    uint32_t spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bool ok = true;
    if ((int)blockIdx.x == a_block) {
        for (int i = 1; i <= n && ok; ++i) { put<STORE>(r0, lane, i); ok = poll<LOAD>(r1, lane, i, spins); }
    } else {
        for (int i = 1; i <= n && ok; ++i) { ok = poll<LOAD>(r0, lane, i, spins); put<STORE>(r1, lane, i); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x == a_block ? 0 : 2)] = ok ? t1 - t0 : 0ull; out[(blockIdx.x == a_block ? 1 : 3)] = spins; }
}

template <int STORE, int LOAD>
static void run(const char* what, uint32_t* rows, int a, int b, unsigned long long* out, uint32_t* xcc, int n) {
    hipMemset(rows, 0, 16384 * 4); hipMemset(out, 0, 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pingpong<STORE, LOAD>), dim3(64), dim3(64), 0, 0, rows, a, b, n, out, xcc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("  %-44s %7.3f us per hop (kernel %.1f us; %s; %.1f polls per hop)\n", what, ms * 1e3 / (2.0 * n), ms * 1e3,
           h[0] && h[2] ? "ok" : "TIMED OUT", (double)(h[1] + h[3]) / (2.0 * n));
}

int main() {
    uint32_t *rows, *xcc; unsigned long long* out;
    hipMalloc(&rows, 16384 * 4); hipMalloc(&xcc, 64 * 4); hipMalloc(&out, 32);
    hipMemset(xcc, 0xff, 64 * 4);
    hipLaunchKernelGGL((pingpong<0, 16>), dim3(64), dim3(64), 0, 0, rows, -1, -1, 0, out, xcc);      // census
    hipDeviceSynchronize();
    std::vector<uint32_t> x(64); hipMemcpy(x.data(), xcc, 64 * 4, hipMemcpyDeviceToHost);
    int same = -1, other = -1;
    for (int b = 1; b < 64; ++b) { if (same < 0 && x[b] == x[0]) same = b; if (other < 0 && x[b] != x[0]) other = b; }
    printf("block 0 on XCC %u; block %d on the same XCC, block %d on XCC %u\n", x[0], same, other, x[other]);
    const int n = 2000;
    for (int pass = 0; pass < 2; ++pass) {
        const int b = pass ? other : same;
        printf("%s:\n", pass ? "different XCDs" : "same XCD");
        run<0, 16>("store sc1, load sc1 (K2f today)", rows, 0, b, out, xcc, n);
        run<1, 16>("store sc1 + plain, load sc1", rows, 0, b, out, xcc, n);
        run<3, 16>("store sc0 sc1, load sc1", rows, 0, b, out, xcc, n);
        run<3, 17>("store sc0 sc1, load sc0 sc1", rows, 0, b, out, xcc, n);
        if (!pass) run<2, 16>("store plain, load sc1", rows, 0, b, out, xcc, n);
        run3<64>("two rows + a full third row", rows, 0, b, out, n);
        run3<2>("two rows + a 32-byte tail (K2f's link)", rows, 0, b, out, n);
    }
    return 0;
}
