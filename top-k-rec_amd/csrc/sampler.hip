// K1 -- uniform-user (u,i,j) sampler + batch planner.
//
// Replaces the reference's Python generator BPR._uniform_user_sampling
// (single/bpr.py:155-165) and prepares, off the sequential critical path, the in-batch
// duplicate structure that TF's optimizer builds inside sess.run (unique +
// unsorted_segment_sum on the IndexedSlices gradients, single/bpr.py:100).
//
// One workgroup per batch.  Integer work only; every output word is defined by
// oracle/plan_np.py and must match it bit for bit:
//   draw   : Philox4x32-10 keyed by the seed, counter = global triplet index + round
//   sort   : LDS bitonic sort of 64-bit (row<<32 | occurrence) keys -> stable grouping
//   plan   : task[3B] = (row|kind<<31, occ_start, occ_count, 0), occ[3B] = per-occurrence
//            partner ids in group order (users first, then items; i-roles before j-roles)
//
// HBM traffic per triplet: 8 B row_ptr pair + 4 B positive + ~4*log2(deg) B membership
// probes + 12 B triplet + 24 B occ + <=48 B task  ~= 0.1 KB; latency-bound, not on the
// critical path (the step kernels of earlier batches run while later batches are planned).
#include "tkr_common.h"

namespace tkr {

constexpr int kPlanThreads = 256;
constexpr int kMaxRounds = 64;   // oracle/plan_np.py MAX_ROUNDS

__device__ __forceinline__ bool is_member(const int32_t* __restrict__ cols_sorted, int lo, int hi, int item) {
    int a = lo, b = hi;
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (cols_sorted[mid] < item) a = mid + 1; else b = mid;
    }
    return a < hi && cols_sorted[a] == item;
}

__device__ __forceinline__ void draw_triplet(const int32_t* __restrict__ tr_users, uint32_t n_tr,
                                             const int32_t* __restrict__ row_ptr,
                                             const int32_t* __restrict__ pos_cols,
                                             const int32_t* __restrict__ cols_sorted, uint32_t n_items,
                                             uint32_t k0, uint32_t k1, uint64_t g, int& u, int& i, int& j) {
    const uint32_t c0 = (uint32_t)g, c1 = (uint32_t)(g >> 32);
    u32x4 w = philox4x32_10(c0, c1, 0u, 0u, k0, k1);
    u = tr_users[mulhi64(w.x, w.y, n_tr)];
    const int lo = row_ptr[u], hi = row_ptr[u + 1];
    i = pos_cols[lo + (int)mulhi64(w.z, w.w, (uint32_t)(hi - lo))];
    int cand = 0;
    bool found = false;
    for (uint32_t r = 1; r <= (uint32_t)kMaxRounds && !found; ++r) {
        w = philox4x32_10(c0, c1, r, 0u, k0, k1);
        cand = (int)mulhi64(w.x, w.y, n_items);
        if (!is_member(cols_sorted, lo, hi, cand)) { found = true; break; }
        cand = (int)mulhi64(w.z, w.w, n_items);
        if (!is_member(cols_sorted, lo, hi, cand)) { found = true; break; }
    }
    if (!found) {   // cyclic scan fallback (user rated almost everything)
        for (uint32_t s = 0; s < n_items && is_member(cols_sorted, lo, hi, cand); ++s)
            cand = (cand + 1 == (int)n_items) ? 0 : cand + 1;
    }
    j = cand;
}

// In-LDS bitonic sort of n (power of two) 64-bit keys, ascending.
__device__ __forceinline__ void bitonic_sort(uint64_t* keys, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int p = threadIdx.x; p < (n >> 1); p += kPlanThreads) {
                const int lo = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                const int hi = lo | stride;
                const bool asc = ((lo & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

// Turn sorted keys[0..n) (row<<32 | occurrence) into task heads + counts.  Returns the
// number of groups (uniform across the block).  `slot0` = first task slot to fill,
// `occ0` = occ offset of sorted position 0, `kind` = 0 users / 1 items.
__device__ __forceinline__ int emit_tasks(const uint64_t* keys, int n, int4* task, int slot0, int occ0,
                                          int kind, int* scan /*LDS [kPlanThreads+1]*/) {
    const int per = (n + kPlanThreads - 1) / kPlanThreads;
    const int beg = min((int)threadIdx.x * per, n), end = min(beg + per, n);
    int cnt = 0;
    for (int p = beg; p < end; ++p)
        cnt += (p == 0) || ((uint32_t)(keys[p] >> 32) != (uint32_t)(keys[p - 1] >> 32));
    scan[threadIdx.x + 1] = cnt;
    if (threadIdx.x == 0) scan[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int t = 1; t <= kPlanThreads; ++t) scan[t] += scan[t - 1];
    __syncthreads();
    int s = scan[threadIdx.x];
    const int total = scan[kPlanThreads];
    for (int p = beg; p < end; ++p) {
        const uint32_t row = (uint32_t)(keys[p] >> 32);
        if ((p == 0) || (row != (uint32_t)(keys[p - 1] >> 32))) {
            // length of this group: scan forward to the next head (groups are short on
            // average; long ones cost O(len) once)
            int q = p + 1;
            while (q < n && (uint32_t)(keys[q] >> 32) == row) ++q;
            task[slot0 + s] = make_int4((int)(row | ((uint32_t)kind << 31)), occ0 + p, q - p, 0);
            ++s;
        }
    }
    return total;
}

__global__ __launch_bounds__(kPlanThreads) void sample_plan_kernel(
    const int32_t* __restrict__ tr_users, uint32_t n_tr, const int32_t* __restrict__ row_ptr,
    const int32_t* __restrict__ pos_cols, const int32_t* __restrict__ cols_sorted, uint32_t n_items,
    uint64_t seed, uint64_t first_triplet, const int64_t* __restrict__ ctl, int B, int npad_items,
    int32_t* __restrict__ out_u, int32_t* __restrict__ out_i, int32_t* __restrict__ out_j,
    int4* __restrict__ task_all, int2* __restrict__ occ_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                       // [npad_items]
    int* scan = reinterpret_cast<int*>(smem + (size_t)npad_items * 8);        // [kPlanThreads+1]

    const int b = blockIdx.x;
    const uint64_t batch0 = ctl ? (uint64_t)ctl[0] : 0ull;                    // device-side chunk base
    const uint64_t g0 = first_triplet + (batch0 + (uint64_t)b) * (uint64_t)B;
    int32_t* bu = out_u + (size_t)b * B;
    int32_t* bi = out_i + (size_t)b * B;
    int32_t* bj = out_j + (size_t)b * B;
    int4* task = task_all + (size_t)b * 3 * B;
    int2* occ = occ_all + (size_t)b * 3 * B;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);

    // ---- draw; item keys go to LDS, triplets to HBM ---------------------------------
    for (int t = threadIdx.x; t < B; t += kPlanThreads) {
        int u, i, j;
        draw_triplet(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_items, k0, k1, g0 + t, u, i, j);
        bu[t] = u; bi[t] = i; bj[t] = j;
    }
    __threadfence_block();
    __syncthreads();

    // ---- users: sort (u<<32 | t) ----------------------------------------------------------
    int npad_u = 1;
    while (npad_u < B) npad_u <<= 1;
    for (int t = threadIdx.x; t < npad_u; t += kPlanThreads)
        keys[t] = (t < B) ? (((uint64_t)(uint32_t)bu[t] << 32) | (uint32_t)t) : ~0ull;
    bitonic_sort(keys, npad_u);
    const int n_uq = emit_tasks(keys, B, task, 0, 0, 0, scan);
    for (int p = threadIdx.x; p < B; p += kPlanThreads) {
        const int t = (int)(uint32_t)keys[p];
        occ[p] = make_int2(bi[t], bj[t]);
    }
    __syncthreads();

    // ---- items: sort (item<<32 | o), o<B: i-role of triplet o, else j-role of o-B ---------
    for (int o = threadIdx.x; o < npad_items; o += kPlanThreads) {
        uint64_t key = ~0ull;
        if (o < B) key = ((uint64_t)(uint32_t)bi[o] << 32) | (uint32_t)o;
        else if (o < 2 * B) key = ((uint64_t)(uint32_t)bj[o - B] << 32) | (uint32_t)o;
        keys[o] = key;
    }
    bitonic_sort(keys, npad_items);
    const int n_iq = emit_tasks(keys, 2 * B, task, n_uq, B, 1, scan);
    for (int p = threadIdx.x; p < 2 * B; p += kPlanThreads) {
        const int o = (int)(uint32_t)keys[p];
        const bool role = o >= B;
        const int t = role ? o - B : o;
        const uint32_t other = (uint32_t)(role ? bi[t] : bj[t]);
        occ[B + p] = make_int2(bu[t], (int)(other | ((uint32_t)role << 31)));
    }
    for (int s = n_uq + n_iq + threadIdx.x; s < 3 * B; s += kPlanThreads) task[s] = make_int4(-1, 0, 0, 0);
}

}  // namespace tkr

extern "C" int tkr_sample_plan(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr,
                               const int32_t* pos_cols, const int32_t* cols_sorted, int32_t n_items,
                               uint64_t seed, uint64_t first_triplet, const int64_t* ctl, int32_t n_batches,
                               int32_t batch_size, int32_t* out_u, int32_t* out_i, int32_t* out_j,
                               int32_t* task, int32_t* occ, void* stream) {
    if (n_tr <= 0 || n_items <= 0 || batch_size <= 0 || n_batches < 0) return TKR_EINVAL;
    if (batch_size > 8192) return TKR_EUNSUPPORTED;   // 2B 64-bit keys must fit the 160 KiB LDS
    if (n_batches == 0) return TKR_OK;
    int npad = 1;
    while (npad < 2 * batch_size) npad <<= 1;
    const size_t lds = (size_t)npad * 8 + (tkr::kPlanThreads + 1) * sizeof(int);
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        TKR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tkr::sample_plan_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(tkr::sample_plan_kernel, dim3(n_batches), dim3(tkr::kPlanThreads), lds,
                       (hipStream_t)stream, tr_users, (uint32_t)n_tr, row_ptr, pos_cols, cols_sorted,
                       (uint32_t)n_items, seed, first_triplet, ctl, batch_size, npad, out_u, out_i, out_j,
                       reinterpret_cast<int4*>(task), reinterpret_cast<int2*>(occ));
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}
