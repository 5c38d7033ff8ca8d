// K1 for LARGE batches (batch_size > 8192): the same plan as csrc/sampler.hip, word for word (oracle/plan_np.py), built
// by grid-wide kernels instead of one workgroup per batch.
//
// single/bpr.py:103-113 takes any batch_size; sampler.hip sorts a batch inside one workgroup's LDS (2B 64-bit keys must fit
// 160 KiB: B <= 8192).  Here the duplicate structure TF's optimizer builds per batch (unique + unsorted_segment_sum on the
// IndexedSlices gradients, single/bpr.py:100) comes from a DEVICE-WIDE radix sort of (batch | row | occurrence) keys over all
// batches of the call at once, group heads from a flag + exclusive scan, and the parity / classification / record steps of
// resolve_kernel run one thread per task or occurrence with their per-batch ranks taken from two more scans.
//
//   draw      one thread per triplet: Philox stream of sampler.hip (draw_triplet), keys out
//   sort      hipcub::DeviceRadixSort over keys  batch | row | occurrence, packed to the bits the sizes need (users: n*B keys, items: n*2B)
//   heads     flag = first key of a (batch, row) group; exclusive scan -> group index; task = (row|kind, start, count, 0)
//   resolve   parities from the touch bitmaps (as sampler.hip), light / heavy ranks by scan, 64-byte wave records, headers
//
// Integer work, HBM-bound streaming: ~150 B per triplet over all passes; it runs on the planner's side stream under the steps
// of the previous chunk.  Every output word is defined by oracle/plan_np.py and must match it bit for bit (tests/test_gpu_bpr.py).
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "tkr_common.h"
#include "sampler_draw.h"

namespace tkr {

// key = batch | row | occurrence, packed as tightly as the problem allows (the radix sort costs a pass per 8 bits):
// `ob` bits of occurrence index, `rb` bits of row
struct KeyBits { int ob, rb; };
__host__ __device__ inline uint64_t big_key(KeyBits kb, uint32_t batch, uint32_t row, uint32_t o) {
    return ((uint64_t)batch << (kb.rb + kb.ob)) | ((uint64_t)row << kb.ob) | o;
}
__device__ __forceinline__ uint32_t key_row(KeyBits kb, uint64_t k) { return (uint32_t)((k >> kb.ob) & ((1ull << kb.rb) - 1)); }
__device__ __forceinline__ uint32_t key_occ(KeyBits kb, uint64_t k) { return (uint32_t)(k & ((1ull << kb.ob) - 1)); }
static int bits_for(uint64_t n) { int b = 1; while ((1ull << b) < n) ++b; return b; }        // values 0 .. n-1

__global__ void big_draw_kernel(const int32_t* __restrict__ tr_users, uint32_t n_tr, const int32_t* __restrict__ row_ptr,
                                const int32_t* __restrict__ pos_cols, const int32_t* __restrict__ cols_sorted, uint32_t n_items,
                                uint64_t seed, uint64_t first_triplet, const int64_t* __restrict__ ctl, int B, size_t total,
                                int32_t* __restrict__ out_u, int32_t* __restrict__ out_i, int32_t* __restrict__ out_j,
                                uint64_t* __restrict__ ukeys, uint64_t* __restrict__ ikeys, KeyBits ku, KeyBits ki) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const uint32_t b = (uint32_t)(g / B), t = (uint32_t)(g % B);
    const uint64_t batch0 = ctl ? (uint64_t)ctl[0] : 0ull;
    int u, i, j;
    draw_triplet(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_items, (uint32_t)seed, (uint32_t)(seed >> 32),
                 first_triplet + batch0 * (uint64_t)B + g, u, i, j);
    out_u[g] = u; out_i[g] = i; out_j[g] = j;
    ukeys[g] = big_key(ku, b, (uint32_t)u, t);
    ikeys[(size_t)b * 2 * B + t] = big_key(ki, b, (uint32_t)i, t);
    ikeys[(size_t)b * 2 * B + B + t] = big_key(ki, b, (uint32_t)j, (uint32_t)B + t);
}

// flag[p] = 1 at the first key of every (batch, row) group; flag[total] = 0 (so that the exclusive scan's last entry is the count)
__global__ void big_flag_kernel(const uint64_t* __restrict__ keys, size_t total, int per_batch, int32_t* __restrict__ flag, KeyBits kb) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > total) return;
    flag[p] = (p < total) && ((p % per_batch) == 0 || key_row(kb, keys[p]) != key_row(kb, keys[p - 1]));
}

// heads write (row | kind<<31, start, -, 0), their position and the row's touch bit; every key writes its occurrence
template <bool ITEM>
__global__ void big_emit_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ flag, const int32_t* __restrict__ rank,
                                const int32_t* __restrict__ urank /*items: group counts of the users*/, size_t total, int B,
                                const int32_t* __restrict__ bu, const int32_t* __restrict__ bi, const int32_t* __restrict__ bj,
                                int4* __restrict__ task_all, int2* __restrict__ occ_all, int32_t* __restrict__ occt_all,
                                int32_t* __restrict__ headpos, uint32_t* __restrict__ touch, KeyBits kb) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int per = ITEM ? 2 * B : B;
    const size_t b = p / per;
    const int lp = (int)(p % per);
    const uint64_t key = keys[p];
    const uint32_t row = key_row(kb, key), o = key_occ(kb, key);
    int2* occ = occ_all + b * 3 * B;
    int32_t* occt = occt_all + b * 3 * B;
    const int32_t* u_ = bu + b * B;
    const int32_t* i_ = bi + b * B;
    const int32_t* j_ = bj + b * B;
    if (ITEM) {
        const bool role = o >= (uint32_t)B;
        const int t = role ? (int)o - B : (int)o;
        const uint32_t other = (uint32_t)(role ? i_[t] : j_[t]);
        occ[B + lp] = make_int2(u_[t], (int)(other | ((uint32_t)role << 31)));
        occt[B + lp] = t;
    } else {
        occ[lp] = make_int2(i_[o], j_[o]);
        occt[lp] = (int)o;
    }
    if (flag[p]) {
        const int gid = rank[p];                                     // exclusive scan at a head = its group index
        const int slot = (ITEM ? (urank[(b + 1) * B] - urank[b * B]) : 0) + gid - rank[b * per];
        headpos[gid] = lp;
        task_all[b * 3 * B + slot] = make_int4((int)(row | ((uint32_t)ITEM << 31)), (ITEM ? B : 0) + lp, 0, 0);
        atomicOr(&touch[(size_t)row * 16 + (b >> 5)], 1u << (b & 31));
    }
}

// the last key of every group knows the group's length; slots past the last task of a batch are (-1, 0, 0, 0)
template <bool ITEM>
__global__ void big_count_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ rank, const int32_t* __restrict__ urank,
                                 size_t total, int B, const int32_t* __restrict__ headpos, int4* __restrict__ task_all) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int per = ITEM ? 2 * B : B;
    const size_t b = p / per;
    const int lp = (int)(p % per);
    if (lp + 1 != per && !flag[p + 1]) return;                       // not the last key of its group
    const int gid = rank[p] - (flag[p] ? 0 : 1);
    const int slot = (ITEM ? (urank[(b + 1) * B] - urank[b * B]) : 0) + gid - rank[b * per];
    task_all[b * 3 * B + slot].z = lp + 1 - headpos[gid];
}

__global__ void big_fill_kernel(const int32_t* __restrict__ urank, const int32_t* __restrict__ irank, int n_batches, int B,
                                int4* __restrict__ task_all) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)n_batches * 3 * B) return;
    const size_t b = g / (3 * (size_t)B);
    const int s = (int)(g % (3 * (size_t)B));
    const int n_tasks = (urank[(b + 1) * B] - urank[b * B]) + (irank[(b + 1) * 2 * B] - irank[b * 2 * B]);
    if (s >= n_tasks) task_all[g] = make_int4(-1, 0, 0, 0);
}

__device__ __forceinline__ int big_parity(const int32_t* __restrict__ cnt, const uint32_t* __restrict__ touch, int row, int batch) {
    const uint32_t* w = touch + (size_t)row * 16;
    int c = cnt[row];
    const int full = batch >> 5;
    for (int q = 0; q < full; ++q) c += __popc(w[q]);
    c += __popc(w[full] & ((1u << (batch & 31)) - 1u));
    return c & 1;
}

// parities of every task's own row and of every occurrence's partner rows (+ per-triplet parities); light / heavy flags
__global__ void big_parity_kernel(int n_batches, int B, int lmax, int4* __restrict__ task_all, int2* __restrict__ occ_all,
                                  const int32_t* __restrict__ occt_all, const int32_t* __restrict__ out_u,
                                  const int32_t* __restrict__ ucnt, const int32_t* __restrict__ icnt,
                                  const uint32_t* __restrict__ touch_u, const uint32_t* __restrict__ touch_i,
                                  int32_t* __restrict__ tpar_all, int32_t* __restrict__ is_light, int32_t* __restrict__ is_heavy) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_batches * 3 * B;
    if (g > total) return;
    if (g == total) { is_light[g] = 0; is_heavy[g] = 0; return; }
    const int b = (int)(g / (3 * (size_t)B));
    const int s = (int)(g % (3 * (size_t)B));
    int4 t = task_all[g];
    int l = 0, h = 0;
    if (t.x != -1) {
        const int row = t.x & 0x7fffffff;
        t.w = (t.x < 0) ? big_parity(icnt, touch_i, row, b) : big_parity(ucnt, touch_u, row, b);
        task_all[g] = t;
        if (t.z <= lmax) l = 1; else h = 1;
    }
    is_light[g] = l;
    is_heavy[g] = h;
    int2 o = occ_all[g];
    if (s < B) {                                                     // user occurrence: (i, j)
        const int pi = big_parity(icnt, touch_i, o.x, b), pj = big_parity(icnt, touch_i, o.y, b);
        if (tpar_all) {
            const int tt = occt_all[g];
            const int u = out_u[(size_t)b * B + tt];
            tpar_all[(size_t)b * B + tt] = big_parity(ucnt, touch_u, u, b) | (pi << 1) | (pj << 2);
        }
        o.x |= pi << 30;
        o.y |= pj << 30;
    } else {                                                         // item occurrence: (u, other | role<<31)
        const int pu = big_parity(ucnt, touch_u, o.x, b);
        const int po = big_parity(icnt, touch_i, o.y & 0x3fffffff, b);
        o.x |= pu << 30;
        o.y |= po << 30;
    }
    occ_all[g] = o;
}

__device__ __forceinline__ int pack_t16(int a, int b) { return (int)(((uint32_t)a & 0xffffu) | (((uint32_t)b & 0xffffu) << 16)); }

// one thread per task: its wave record(s), as resolve_kernel writes them
__global__ void big_record_kernel(int n_batches, int B, int team, int lmax, int rec_stride, const int4* __restrict__ task_all,
                                  const int2* __restrict__ occ_all, const int32_t* __restrict__ occt_all,
                                  const int32_t* __restrict__ lrank, const int32_t* __restrict__ hrank, int32_t* __restrict__ rec_all,
                                  int4* __restrict__ hdr_all) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = 3 * (size_t)B;
    if (g >= (size_t)n_batches * per) return;
    const size_t b = g / per;
    const int s = (int)(g % per);
    const int tot_l = lrank[(b + 1) * per] - lrank[b * per], tot_h = hrank[(b + 1) * per] - hrank[b * per];
    const int nlb = (tot_l + team - 1) / team;                       // light tasks per workgroup = team (sampler.hip light_per_block)
    int32_t* rec = rec_all + b * (size_t)rec_stride * 16;
    const int2* occ = occ_all + b * per;
    const int32_t* occt = occt_all + b * per;
    if (s == 0) hdr_all[b] = make_int4(nlb + tot_h, nlb, tot_h, tot_l + tot_h);
    if (s < nlb * team - tot_l) {                                    // idle wave slots of the last light workgroup
        int32_t* r = rec + (size_t)(tot_l + s) * 16;
        r[0] = -1;
        for (int q = 1; q < 16; ++q) r[q] = 0;
    }
    const int4 t = task_all[g];
    if (t.x == -1) return;
    if (t.z <= lmax) {
        const int li = lrank[g] - lrank[b * per];
        int32_t* r = rec + (size_t)li * 16;                          // (li / lpb) * team + li % lpb with lpb == team
        r[0] = t.x; r[1] = t.w | (1 << 8); r[2] = t.z; r[3] = t.y;
        int tt[4];
        for (int q = 0; q < 4; ++q) {
            const int2 o = (q < t.z) ? occ[t.y + q] : make_int2(0, 0);
            r[4 + 2 * q] = o.x; r[5 + 2 * q] = o.y;
            tt[q] = (q < t.z) ? occt[t.y + q] : 0;
        }
        r[12] = t.z; r[13] = pack_t16(tt[0], tt[1]); r[14] = pack_t16(tt[2], tt[3]); r[15] = 0;
    } else {
        const int hi = hrank[g] - hrank[b * per];
        for (int w = 0; w < team; ++w) {
            int32_t* r = rec + ((size_t)(nlb + hi) * team + w) * 16;
            const int mine = (t.z > w) ? (t.z - w + team - 1) / team : 0;
            r[0] = t.x; r[1] = t.w | (team << 8) | (w << 16); r[2] = mine; r[3] = t.y + w;
            int tt[4];
            for (int q = 0; q < 4; ++q) {
                const int2 o = (q < mine) ? occ[t.y + w + q * team] : make_int2(0, 0);
                r[4 + 2 * q] = o.x; r[5 + 2 * q] = o.y;
                tt[q] = (q < mine) ? occt[t.y + w + q * team] : 0;
            }
            r[12] = t.z; r[13] = pack_t16(tt[0], tt[1]); r[14] = pack_t16(tt[2], tt[3]); r[15] = 0;
        }
    }
}

struct BigLayout {             // carve of the caller's workspace
    uint64_t *ukeys, *ukeys2, *ikeys, *ikeys2;
    int32_t *uflag, *urank, *iflag, *irank, *uhead, *ihead;
    void* cub;
    size_t cub_bytes, total_bytes;
};

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static BigLayout big_layout(char* base, size_t nB) {
    BigLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    L.ukeys = (uint64_t*)take(nB * 8); L.ukeys2 = (uint64_t*)take(nB * 8);
    L.ikeys = (uint64_t*)take(2 * nB * 8); L.ikeys2 = (uint64_t*)take(2 * nB * 8);
    // flags / ranks double as the light / heavy flags and ranks of the resolve step (3 nB + 1 entries are enough for both uses)
    L.uflag = (int32_t*)take((3 * nB + 1) * 4); L.urank = (int32_t*)take((3 * nB + 1) * 4);
    L.iflag = (int32_t*)take((3 * nB + 1) * 4); L.irank = (int32_t*)take((3 * nB + 1) * 4);
    L.uhead = (int32_t*)take(nB * 4); L.ihead = (int32_t*)take(2 * nB * 4);
    size_t s1 = 0, s2 = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, s1, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)(2 * nB), 0, 60, (hipStream_t)0);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, s2, (const int32_t*)nullptr, (int32_t*)nullptr, (int)(3 * nB + 1), (hipStream_t)0);
    L.cub_bytes = align256(s1 > s2 ? s1 : s2);
    L.cub = take(L.cub_bytes);
    L.total_bytes = off;
    return L;
}

}  // namespace tkr

extern "C" int tkr_plan_team(int32_t batch_size);
extern "C" int tkr_plan_max_blocks(int32_t batch_size);

extern "C" __attribute__((visibility("hidden"))) int64_t tkr_plan_workspace_bytes_for(int32_t batch_size, int32_t n_batches) {      // (a helper between translation units, not an entry point)
    if (batch_size <= 0 || n_batches <= 0) return 0;
    return (int64_t)tkr::big_layout(nullptr, (size_t)batch_size * n_batches).total_bytes;
}
extern "C" __attribute__((visibility("hidden"))) int64_t tkr_plan_mid_workspace_bytes(int32_t batch_size, int32_t n_batches);     // csrc/planner_mid.hip
extern "C" int64_t tkr_plan_workspace_bytes(int32_t batch_size, int32_t n_batches) {
    static const int big_from = [] { const char* e = getenv("TKR_PLAN_BIG_FROM"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4096; }();
    const int64_t big = batch_size >= big_from ? tkr_plan_workspace_bytes_for(batch_size, n_batches) : 0;
    const int64_t mid = tkr_plan_mid_workspace_bytes(batch_size, n_batches);          // (a few KB: the range sums of every batch)
    return big > mid ? big : mid;
}

// called by tkr_sample_plan for batch_size > 8192
extern "C" __attribute__((visibility("hidden"))) int tkr_sample_plan_big(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr, const int32_t* pos_cols,
                                   const int32_t* cols_sorted, int32_t n_users, int32_t n_items, uint64_t seed,
                                   uint64_t first_triplet, const int64_t* ctl, int32_t n_batches, int32_t B, int32_t* ucnt,
                                   int32_t* icnt, uint32_t* touch_u, uint32_t* touch_i, int32_t* out_u, int32_t* out_i,
                                   int32_t* out_j, int32_t* task, int32_t* occ, int32_t* rec, int32_t* hdr, int32_t* occt,
                                   int32_t* tpar, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace tkr;
    if (B > (1 << 20)) return TKR_EUNSUPPORTED;                       // 9 + 30 + 21 key bits at most
    const size_t nB = (size_t)B * n_batches;
    if (3 * nB + 1 >= (1ull << 31)) return TKR_EUNSUPPORTED;
    if (!workspace || workspace_bytes < tkr_plan_workspace_bytes_for(B, n_batches)) return TKR_EINVAL;
    const BigLayout L = big_layout((char*)workspace, nB);
    hipStream_t s = (hipStream_t)stream;
    const int T = 256;
    auto blocks = [&](size_t n) { return dim3((unsigned)((n + T - 1) / T)); };
    int4* task4 = reinterpret_cast<int4*>(task);
    int2* occ2 = reinterpret_cast<int2*>(occ);

    const KeyBits ku = {bits_for((uint64_t)B), bits_for((uint64_t)n_users)}, ki = {bits_for(2 * (uint64_t)B), bits_for((uint64_t)n_items)};
    const int end_u = ku.ob + ku.rb + bits_for((uint64_t)n_batches), end_i = ki.ob + ki.rb + bits_for((uint64_t)n_batches);
    hipLaunchKernelGGL(big_draw_kernel, blocks(nB), dim3(T), 0, s, tr_users, (uint32_t)n_tr, row_ptr, pos_cols, cols_sorted,
                       (uint32_t)n_items, seed, first_triplet, ctl, B, nB, out_u, out_i, out_j, L.ukeys, L.ikeys, ku, ki);
    TKR_LAUNCH_CHECK();
    size_t cb = L.cub_bytes;
    // The keys leave big_draw_kernel in (batch, occurrence) order and the radix sort is STABLE: sorting on the batch | row bits alone
    // leaves the occurrences of a (batch, row) group in ascending order -- the order of the full key -- in 3 passes instead of 5 at
    // batch 8192 (users 7 + 17 bits instead of 37, items 7 + 14 instead of 35); TKR_PLAN_FULL_SORT=1: the old way.
    static const bool full_sort = getenv("TKR_PLAN_FULL_SORT") && getenv("TKR_PLAN_FULL_SORT")[0] == '1';
    TKR_CHECK(hipcub::DeviceRadixSort::SortKeys(L.cub, cb, (const uint64_t*)L.ukeys, L.ukeys2, (int)nB, full_sort ? 0 : ku.ob, end_u, s));
    cb = L.cub_bytes;
    TKR_CHECK(hipcub::DeviceRadixSort::SortKeys(L.cub, cb, (const uint64_t*)L.ikeys, L.ikeys2, (int)(2 * nB), full_sort ? 0 : ki.ob, end_i, s));
    hipLaunchKernelGGL(big_flag_kernel, blocks(nB + 1), dim3(T), 0, s, L.ukeys2, nB, B, L.uflag, ku);
    hipLaunchKernelGGL(big_flag_kernel, blocks(2 * nB + 1), dim3(T), 0, s, L.ikeys2, 2 * nB, 2 * B, L.iflag, ki);
    TKR_LAUNCH_CHECK();
    cb = L.cub_bytes;
    TKR_CHECK(hipcub::DeviceScan::ExclusiveSum(L.cub, cb, (const int32_t*)L.uflag, L.urank, (int)(nB + 1), s));
    cb = L.cub_bytes;
    TKR_CHECK(hipcub::DeviceScan::ExclusiveSum(L.cub, cb, (const int32_t*)L.iflag, L.irank, (int)(2 * nB + 1), s));
    hipLaunchKernelGGL((big_emit_kernel<false>), blocks(nB), dim3(T), 0, s, L.ukeys2, L.uflag, L.urank, L.urank, nB, B, out_u, out_i,
                       out_j, task4, occ2, occt, L.uhead, touch_u, ku);
    hipLaunchKernelGGL((big_emit_kernel<true>), blocks(2 * nB), dim3(T), 0, s, L.ikeys2, L.iflag, L.irank, L.urank, 2 * nB, B, out_u,
                       out_i, out_j, task4, occ2, occt, L.ihead, touch_i, ki);
    TKR_LAUNCH_CHECK();
    hipLaunchKernelGGL((big_count_kernel<false>), blocks(nB), dim3(T), 0, s, L.uflag, L.urank, L.urank, nB, B, L.uhead, task4);
    hipLaunchKernelGGL((big_count_kernel<true>), blocks(2 * nB), dim3(T), 0, s, L.iflag, L.irank, L.urank, 2 * nB, B, L.ihead, task4);
    hipLaunchKernelGGL(big_fill_kernel, blocks(3 * nB), dim3(T), 0, s, L.urank, L.irank, n_batches, B, task4);
    TKR_LAUNCH_CHECK();
    // resolve: the flag / rank arrays are free again
    const int team = tkr_plan_team(B), lmax = B <= 4096 ? 4 : 16;
    int32_t *is_light = L.uflag, *lrank = L.urank, *is_heavy = L.iflag, *hrank = L.irank;
    hipLaunchKernelGGL(big_parity_kernel, blocks(3 * nB + 1), dim3(T), 0, s, n_batches, B, lmax, task4, occ2, occt, out_u, ucnt, icnt,
                       touch_u, touch_i, tpar, is_light, is_heavy);
    TKR_LAUNCH_CHECK();
    cb = L.cub_bytes;
    TKR_CHECK(hipcub::DeviceScan::ExclusiveSum(L.cub, cb, (const int32_t*)is_light, lrank, (int)(3 * nB + 1), s));
    cb = L.cub_bytes;
    TKR_CHECK(hipcub::DeviceScan::ExclusiveSum(L.cub, cb, (const int32_t*)is_heavy, hrank, (int)(3 * nB + 1), s));
    hipLaunchKernelGGL(big_record_kernel, blocks(3 * nB), dim3(T), 0, s, n_batches, B, team, lmax, tkr_plan_max_blocks(B) * team,
                       task4, occ2, occt, lrank, hrank, rec, reinterpret_cast<int4*>(hdr));
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}
