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
//   plan   : task[3B] = (row|kind<<31, occ_start, occ_count, parity), occ[3B] = per-occurrence
//            partner ids in group order (users first, then items; i-roles before j-roles)
//   parity : which of the two table buffers holds each row at the start of the batch
//            (= number of earlier updates of the row, mod 2), resolved here so that the step
//            kernel needs no dependent lookup: K1a marks (row, batch) in a per-row bitmap,
//            K1b turns prefix popcounts into parity bits and writes the 64-byte per-wave
//            launch records, K1c folds the bitmap into the running update counters.
//
// HBM traffic per triplet: 8 B row_ptr pair + 4 B positive + ~4*log2(deg) B membership
// probes + 12 B triplet + 24 B occ + <=48 B task  ~= 0.1 KB; latency-bound, not on the
// critical path (the step kernels of earlier batches run while later batches are planned).
#include <stdlib.h>

#include "tkr_common.h"

#ifdef TKR_K1_PROF            // scripts/probe_short.py: s_memtime at the phase boundaries of workgroup 0 (100 MHz ticks)
namespace tkr { __device__ unsigned long long k1_prof[32]; }
#define K1_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) tkr::k1_prof[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "plan_parts.h"      // the sorts, task heads, versions and record assembly: shared with the prologue of csrc/bpr_own.hip

namespace tkr {

template <int T>
__global__ __launch_bounds__(T) void sample_plan_kernel(
    const int32_t* __restrict__ tr_users, uint32_t n_tr, const int32_t* __restrict__ row_ptr,
    const int32_t* __restrict__ pos_cols, const int32_t* __restrict__ cols_sorted, uint32_t n_items,
    uint64_t seed, uint64_t first_triplet, const int64_t* __restrict__ ctl, int B, int npad_items,
    int32_t* __restrict__ out_u, int32_t* __restrict__ out_i, int32_t* __restrict__ out_j,
    int4* __restrict__ task_all, int2* __restrict__ occ_all, int32_t* __restrict__ occt_all,
    uint32_t* __restrict__ touch_u, uint32_t* __restrict__ touch_i, bool reg_sort_ok /*row ids leave room for the occurrence bits in 32-bit keys*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const uint64_t batch0 = ctl ? (uint64_t)ctl[0] : 0ull;                    // device-side chunk base
    const uint64_t g0 = first_triplet + (batch0 + (uint64_t)b) * (uint64_t)B;
    plan_phase_a<T, T>(smem, b, tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_items, seed, g0, B, npad_items, out_u + (size_t)b * B,
                       out_i + (size_t)b * B, out_j + (size_t)b * B, task_all + (size_t)b * 3 * B, occ_all + (size_t)b * 3 * B,
                       occt_all + (size_t)b * 3 * B, touch_u, touch_i, reg_sort_ok);
}

// ---- K1b: parities + per-wave launch records ------------------------------------------------
__device__ __forceinline__ int block_exclusive_scan2(int a, int b, int* scan /*LDS [2*(T+1)]*/, int& tot_a,
                                                      int& tot_b, int& ex_b) {
    // wave scans + the four wave totals (thread 0 used to walk all 256 entries)
    const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x / TKR_WAVE;
    int ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < TKR_WAVE; d <<= 1) {
        const int ua = __shfl_up(ia, d), ub = __shfl_up(ib, d);
        if (lane >= d) { ia += ua; ib += ub; }
    }
    __syncthreads();
    if (lane == TKR_WAVE - 1) { scan[wave] = ia; scan[kPlanThreads / TKR_WAVE + wave] = ib; }
    __syncthreads();
    int ea = ia - a;
    ex_b = ib - b;
    tot_a = 0;
    tot_b = 0;
#pragma unroll
    for (int w = 0; w < kPlanThreads / TKR_WAVE; ++w) {
        const int ta = scan[w], tb = scan[kPlanThreads / TKR_WAVE + w];
        if (w < wave) { ea += ta; ex_b += tb; }
        tot_a += ta;
        tot_b += tb;
    }
    return ea;
}

__global__ __launch_bounds__(kPlanThreads) void resolve_kernel(
    int B, int rec_stride /*records per batch*/, int4* __restrict__ task_all, int2* __restrict__ occ_all,
    const int32_t* __restrict__ occt_all,
    const int32_t* __restrict__ ucnt, const int32_t* __restrict__ icnt, const uint32_t* __restrict__ touch_u,
    const uint32_t* __restrict__ touch_i, int32_t* __restrict__ rec_all, int4* __restrict__ hdr_all,
    const int32_t* __restrict__ out_u_all, int32_t* __restrict__ tpar_all /*nullable: per-triplet parities*/) {
    __shared__ int scan[2 * (kPlanThreads + 1)];
    const int kTeam = team_for(B);
    const int lmax = light_max(B);
    const int b = blockIdx.x;
    int4* task = task_all + (size_t)b * 3 * B;
    int2* occ = occ_all + (size_t)b * 3 * B;
    const int32_t* occt = occt_all + (size_t)b * 3 * B;
    int32_t* rec = rec_all + (size_t)b * rec_stride * 16;

    for (int s = threadIdx.x; s < 3 * B; s += kPlanThreads) {
        int4 t = task[s];
        if (t.x != -1) {
            const int row = t.x & 0x7fffffff;
            t.w = (t.x < 0) ? parity_of(icnt, touch_i, row, b) : parity_of(ucnt, touch_u, row, b);
            task[s] = t;
        }
    }
    for (int p = threadIdx.x; p < B; p += kPlanThreads) {           // user occurrences: (i, j)
        int2 o = occ[p];
        const int pi = parity_of(icnt, touch_i, o.x, b), pj = parity_of(icnt, touch_i, o.y, b);
        o.x |= pi << 30;
        o.y |= pj << 30;
        occ[p] = o;
        if (tpar_all) {                                             // every triplet is exactly one user occurrence
            const int t = occt[p];
            const int u = out_u_all[(size_t)b * B + t];
            tpar_all[(size_t)b * B + t] = parity_of(ucnt, touch_u, u, b) | (pi << 1) | (pj << 2);
        }
    }
    for (int p = threadIdx.x; p < 2 * B; p += kPlanThreads) {       // item occurrences: (u, other|role<<31)
        int2 o = occ[B + p];
        o.x |= parity_of(ucnt, touch_u, o.x, b) << 30;
        o.y |= parity_of(icnt, touch_i, o.y & 0x3fffffff, b) << 30;
        occ[B + p] = o;
    }
    __threadfence_block();
    __syncthreads();

    // classify: light (one wave) / heavy (a team); slots by prefix rank in task order
    const int per = (3 * B + kPlanThreads - 1) / kPlanThreads;
    const int beg = min((int)threadIdx.x * per, 3 * B), end = min(beg + per, 3 * B);
    int nl = 0, nh = 0;
    for (int s = beg; s < end; ++s) {
        const int4 t = task[s];
        if (t.x != -1) { if (t.z <= lmax) ++nl; else ++nh; }
    }
    int tot_l, tot_h, hi;
    int li = block_exclusive_scan2(nl, nh, scan, tot_l, tot_h, hi);
    const int lpb = light_per_block(B);
    const int nlb = (tot_l + lpb - 1) / lpb;
    for (int s = beg; s < end; ++s) {
        const int4 t = task[s];
        if (t.x == -1) continue;
        if (t.z <= lmax) {
            int32_t* r = rec + ((size_t)(li / lpb) * kTeam + li % lpb) * 16;
            r[0] = t.x; r[1] = t.w | (1 << 8); r[2] = t.z; r[3] = t.y;
            int tt[4];
            for (int q = 0; q < 4; ++q) {
                const int2 o = (q < t.z) ? occ[t.y + q] : make_int2(0, 0);
                r[4 + 2 * q] = o.x; r[5 + 2 * q] = o.y;
                tt[q] = (q < t.z) ? occt[t.y + q] : 0;
            }
            r[12] = t.z; r[13] = tt[0] | (tt[1] << 16); r[14] = tt[2] | (tt[3] << 16); r[15] = 0;
            ++li;
        } else {
            for (int w = 0; w < kTeam; ++w) {
                int32_t* r = rec + ((size_t)(nlb + hi) * kTeam + w) * 16;
                const int mine = (t.z > w) ? (t.z - w + kTeam - 1) / kTeam : 0;
                r[0] = t.x; r[1] = t.w | (kTeam << 8) | (w << 16); r[2] = mine; r[3] = t.y + w;
                int tt[4];
                for (int q = 0; q < 4; ++q) {
                    const int2 o = (q < mine) ? occ[t.y + w + q * kTeam] : make_int2(0, 0);
                    r[4 + 2 * q] = o.x; r[5 + 2 * q] = o.y;
                    tt[q] = (q < mine) ? occt[t.y + w + q * kTeam] : 0;
                }
                r[12] = t.z; r[13] = tt[0] | (tt[1] << 16); r[14] = tt[2] | (tt[3] << 16); r[15] = 0;
            }
            ++hi;
        }
    }
    for (int s = threadIdx.x; s < nlb * kTeam; s += kPlanThreads) {      // idle wave slots of the light groups
        const int li_of = (s / kTeam) * lpb + (s % kTeam);
        if ((s % kTeam) >= lpb || li_of >= tot_l) {
            int32_t* r = rec + (size_t)s * 16;
            r[0] = -1;
            for (int q = 1; q < 16; ++q) r[q] = 0;
        }
    }
    if (threadIdx.x == 0) hdr_all[b] = make_int4(nlb + tot_h, nlb, tot_h, tot_l + tot_h);
}

// ---- K1b', dataflow form: full versions + one 128-byte record per task -----------------------------------
// The persistent step kernels (csrc/bpr_flow.hip, csrc/bpr_own.hip) run the batches of a chunk inside ONE launch; what orders
// them is data: every row carries the number of updates it has seen (its version) in-band, and a task names the exact version
// of its own row and of every partner row.  Output per batch (oracle/plan_np.py flow_records):
//   pocc[3B] int4   per sorted occurrence: (a, version of a, b | role<<31, version of b); user occurrence: a = i, b = j;
//                   item occurrence: a = u, b = the other item
//   prec[3B][32]    per task slot: [0] row | kind<<31 (-1 = unused slot)  [1] version of the row  [2] occurrences
//                   [3] index of its first occurrence in pocc, counted from batch 0 of this call  [4] batch
//                   [5] the last batch < [4] of this call that updated the row, -1 = none  [6..7] 0
//                   [8+4q .. 11+4q] = pocc of occurrence q < min(4, occurrences)  [24+q] = index of that occurrence's triplet in the
//                   batch (occt)  [28..31] 0
// With n_owner > 0 (K2o: item row r is served by workgroup r % n_owner, which keeps the row in its LDS) the records of a batch's
// ITEM tasks are laid out in (owner, row) order instead of row order -- still the slots [users, users + items) of the batch, the
// order of tasks inside a batch means nothing to the step -- and ohdr[owner * ohdr_stride + batch] = first slot | tasks << 16
// names every owner's run.
template <int T>
__global__ __launch_bounds__(T) void resolve_flow_kernel(
    int B, const int4* __restrict__ task_all, const int2* __restrict__ occ_all, const int32_t* __restrict__ ucnt,
    const int32_t* __restrict__ icnt, const uint32_t* __restrict__ touch_u, const uint32_t* __restrict__ touch_i,
    int4* __restrict__ pocc_all, int4* __restrict__ prec_all, int n_owner, int32_t* __restrict__ ohdr, int ohdr_stride,
    const int32_t* __restrict__ occt_all, int own_words /*32-row words of an owner's bitmap: ceil(ceil(n_items / n_owner) / 32)*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_first_item, s_wave[T / TKR_WAVE];
    const int b = blockIdx.x;
    const int4* task = task_all + (size_t)b * 3 * B;
    const int2* occ = occ_all + (size_t)b * 3 * B;
    int4* pocc = pocc_all + (size_t)b * 3 * B;
    int4* prec = prec_all + (size_t)b * 3 * B * 8;
    K1_STAMP(8);

    for (int p = threadIdx.x; p < B; p += T) {           // user occurrences: (i, j)
        const int2 o = occ[p];
        pocc[p] = make_int4(o.x, version_of(icnt, touch_i, o.x, b), o.y, version_of(icnt, touch_i, o.y, b));
    }
    for (int p = threadIdx.x; p < 2 * B; p += T) {       // item occurrences: (u, other|role<<31)
        const int2 o = occ[B + p];
        pocc[B + p] = make_int4(o.x, version_of(ucnt, touch_u, o.x, b), o.y,
                                version_of(icnt, touch_i, o.y & 0x3fffffff, b));
    }
    // ---- owner order of the item tasks: slot of an item task = first item slot + its rank by (owner, row).  Every owner keeps a
    // bitmap of the rows it meets in this batch (row = owner + n_owner * bit): the rank inside an owner is a popcount below the
    // row's bit -- the same whatever the order in which the tasks set their bits -- and the owners' first ranks a scan of the counts.
    uint32_t* own_mask = reinterpret_cast<uint32_t*>(smem);                            // [n_owner][own_words]
    uint32_t* own_start = own_mask + (size_t)n_owner * own_words;                      // [n_owner]
    int first_item = 0;
#ifdef TKR_K1_PROF
    __syncthreads();
    K1_STAMP(9);
#endif
    if (n_owner > 0) {
        if (threadIdx.x == 0) s_first_item = 3 * B;
        for (int w = threadIdx.x; w < n_owner * own_words; w += T) own_mask[w] = 0u;
        __syncthreads();
        for (int s = threadIdx.x; s < 3 * B; s += T) {   // tasks are [users][items][-1 ...]
            const int x = task[s].x;
            if (x < 0 && x != -1) {
                if (s == 0 || task[s - 1].x >= 0) s_first_item = s;
                const int row = x & 0x7fffffff, bit = row / n_owner;
                atomicOr(&own_mask[(size_t)(row % n_owner) * own_words + (bit >> 5)], 1u << (bit & 31));
            }
        }
        __syncthreads();
        first_item = s_first_item;
        // exclusive scan of the counts over the owners (each thread a run of consecutive owners), header words on the way
        const int per = (n_owner + T - 1) / T;
        const int w0 = min((int)threadIdx.x * per, n_owner), w1 = min(w0 + per, n_owner);
        int mine = 0;
        for (int w = w0; w < w1; ++w)
            for (int j = 0; j < own_words; ++j) mine += __popc(own_mask[(size_t)w * own_words + j]);
        const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x / TKR_WAVE;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < TKR_WAVE; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == TKR_WAVE - 1) s_wave[wave] = incl;
        __syncthreads();
        int run = incl - mine;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
        for (int w = w0; w < w1; ++w) {
            int c = 0;
            for (int j = 0; j < own_words; ++j) c += __popc(own_mask[(size_t)w * own_words + j]);
            ohdr[(size_t)w * ohdr_stride + b] = (first_item + run) | (c << 16);
            own_start[w] = (uint32_t)run;
            run += c;
        }
    }
    __threadfence_block();
    __syncthreads();
    K1_STAMP(10);
    for (int s = threadIdx.x; s < 3 * B; s += T) {
        const int4 t = task[s];
        int dst = s;
        if (n_owner > 0 && t.x < 0 && t.x != -1) {                   // rank inside its owner: the rows of that owner below this one
            const int row = t.x & 0x7fffffff, w = row % n_owner, bit = row / n_owner;
            const uint32_t* m = own_mask + (size_t)w * own_words;
            int before = __popc(m[bit >> 5] & ((1u << (bit & 31)) - 1u));
            for (int j = 0; j < (bit >> 5); ++j) before += __popc(m[j]);
            dst = first_item + (int)own_start[w] + before;
        }
        int4* r = prec + (size_t)dst * 8;
        if (t.x == -1) {
            r[0] = make_int4(-1, 0, 0, 0);
#pragma unroll
            for (int q = 1; q < 8; ++q) r[q] = make_int4(0, 0, 0, 0);
            continue;
        }
        const int row = t.x & 0x7fffffff;
        int ver, prev;
        if (t.x < 0) row_history(icnt, touch_i, row, b, ver, prev);
        else row_history(ucnt, touch_u, row, b, ver, prev);
        r[0] = make_int4(t.x, ver, t.z, b * 3 * B + t.y);
        r[1] = make_int4(b, prev, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) r[2 + q] = (q < t.z) ? pocc[t.y + q] : make_int4(0, 0, 0, 0);
        const int32_t* tt = occt_all + (size_t)b * 3 * B + t.y;          // the triplets of those occurrences (K2o: the slots of their scalars)
        r[6] = make_int4(tt[0], t.z > 1 ? tt[1] : 0, t.z > 2 ? tt[2] : 0, t.z > 3 ? tt[3] : 0);
        r[7] = make_int4(0, 0, 0, 0);
    }
#ifdef TKR_K1_PROF
    __syncthreads();
    K1_STAMP(11);
#endif
}

// ---- K1b'' : the same records for a SHORT call (B <= 256, one thread per task slot and per occurrence) ---------------------------
// resolve_flow_kernel is written to run BESIDE the persistent step (256 threads, ~50 registers, a few KB of LDS) and walks three
// slots per thread through chains of dependent loads: occurrence -> its versions, then task -> its history -> its occurrences.
// In front of a short call nothing runs beside the planner and those ~14 us are all exposed.  Here every thread takes ONE slot:
// the occurrence and the task of its slot are loaded together, their version words together, the occurrences of a task come from
// LDS -- two trips through memory instead of five.  Same output, bit for bit.
__global__ __launch_bounds__(kWideThreads) void resolve_flow_wide_kernel(
    int B, const int4* __restrict__ task_all, const int2* __restrict__ occ_all, const int32_t* __restrict__ ucnt,
    const int32_t* __restrict__ icnt, const uint32_t* __restrict__ touch_u, const uint32_t* __restrict__ touch_i,
    int4* __restrict__ pocc_all, int4* __restrict__ prec_all, int n_owner, int32_t* __restrict__ ohdr, int ohdr_stride,
    const int32_t* __restrict__ occt_all, int own_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, n = 3 * B;
    int4 t;
    int prev, total;
    plan_phase_b_wide<false>(smem, b, B, task_all + (size_t)b * n, occ_all + (size_t)b * n, occt_all + (size_t)b * n, ucnt, icnt, touch_u, touch_i,
                             pocc_all + (size_t)b * n, prec_all + (size_t)b * n * 8, n_owner, ohdr, ohdr_stride, own_words, t, prev, total);
#ifdef TKR_K1_PROF
    __syncthreads();
    K1_STAMP(11);
#endif
}

// ---- K1c: fold the chunk's touch bitmap into the update counters and clear it --------------
__global__ void commit_kernel(int n_users, int n_items, int32_t* __restrict__ ucnt, int32_t* __restrict__ icnt,
                              uint32_t* __restrict__ touch_u, uint32_t* __restrict__ touch_i) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_users + n_items) return;
    int32_t* cnt = (r < n_users) ? ucnt + r : icnt + (r - n_users);
    uint4* w = reinterpret_cast<uint4*>((r < n_users) ? touch_u + (size_t)r * kTouchWords
                                                       : touch_i + (size_t)(r - n_users) * kTouchWords);
    int c = 0;
#pragma unroll
    for (int q = 0; q < kTouchWords / 4; ++q) {
        const uint4 v = w[q];
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    if (c) {
        *cnt += c;
#pragma unroll
        for (int q = 0; q < kTouchWords / 4; ++q) w[q] = make_uint4(0, 0, 0, 0);
    }
}

// ---- K1d: take planned-but-never-run batches out of the update counters ----------------------------
// A plan is consumed incrementally (BPR.train may stop inside a chunk; bench warm-up and run share one chunk).
// When the rest of a plan is dropped, the counters -- already advanced by commit_kernel for every planned
// batch -- must again equal the number of updates that really ran: one decrement per task of a dropped batch.
__global__ void rollback_kernel(const int4* __restrict__ task, size_t n_slots, int32_t* __restrict__ ucnt,
                                int32_t* __restrict__ icnt) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int rowk = task[s].x;
    if (rowk == -1) return;
    atomicSub((rowk < 0) ? icnt + (rowk & 0x7fffffff) : ucnt + rowk, 1);
}

}  // namespace tkr

extern "C" int tkr_plan_team(int32_t batch_size) { return tkr::team_for(batch_size); }

extern "C" __attribute__((visibility("hidden"))) int tkr_plan_commit(int32_t n_users, int32_t n_items, int32_t* ucnt, int32_t* icnt, uint32_t* touch_u,
                                                                     uint32_t* touch_i, void* stream) {      // (for csrc/planner_mid.hip)
    const int rows = n_users + n_items;
    hipLaunchKernelGGL(tkr::commit_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_users, n_items, ucnt, icnt, touch_u, touch_i);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_plan_max_blocks(int32_t batch_size) {
    const int lpb = tkr::light_per_block(batch_size);
    return (3 * batch_size + lpb - 1) / lpb + (3 * batch_size) / (tkr::light_max(batch_size) + 1);
}

extern "C" __attribute__((visibility("hidden"))) int64_t tkr_plan_workspace_bytes_for(int32_t batch_size, int32_t n_batches);       // csrc/planner_big.hip, any batch size
extern "C" __attribute__((visibility("hidden"))) int tkr_plan_mid_ok(int32_t n_users, int32_t n_items, int32_t B);                  // csrc/planner_mid.hip
extern "C" __attribute__((visibility("hidden"))) int64_t tkr_plan_mid_workspace_bytes(int32_t batch_size, int32_t n_batches);
extern "C" __attribute__((visibility("hidden"))) int tkr_sample_plan_mid(
    const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr, const int32_t* pos_cols, const int32_t* cols_sorted, int32_t n_users,
    int32_t n_items, uint64_t seed, uint64_t first_triplet, const int64_t* ctl, int32_t n_batches, int32_t B, int32_t* ucnt, int32_t* icnt,
    uint32_t* touch_u, uint32_t* touch_i, int32_t* out_u, int32_t* out_i, int32_t* out_j, int32_t* task, int32_t* occ, int32_t* rec,
    int32_t* hdr, int32_t* occt, int32_t* tpar, void* workspace, int64_t workspace_bytes, void* stream);
extern "C" __attribute__((visibility("hidden"))) int tkr_sample_plan_big(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr, const int32_t* pos_cols,
                                   const int32_t* cols_sorted, int32_t n_users, int32_t n_items, uint64_t seed,
                                   uint64_t first_triplet, const int64_t* ctl, int32_t n_batches, int32_t B, int32_t* ucnt,
                                   int32_t* icnt, uint32_t* touch_u, uint32_t* touch_i, int32_t* out_u, int32_t* out_i,
                                   int32_t* out_j, int32_t* task, int32_t* occ, int32_t* rec, int32_t* hdr, int32_t* occt,
                                   int32_t* tpar, void* workspace, int64_t workspace_bytes, void* stream);      // csrc/planner_big.hip

static int sample_plan_impl(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr,
                               const int32_t* pos_cols, const int32_t* cols_sorted, int32_t n_users,
                               int32_t n_items, uint64_t seed, uint64_t first_triplet, const int64_t* ctl,
                               int32_t n_batches, int32_t batch_size, int32_t* ucnt, int32_t* icnt,
                               uint32_t* touch_u, uint32_t* touch_i, int32_t* out_u, int32_t* out_i,
                               int32_t* out_j, int32_t* task, int32_t* occ, int32_t* rec, int32_t* hdr,
                               int32_t* occt, int32_t* tpar, int32_t* prec, int32_t* pocc, void* workspace,
                               int64_t workspace_bytes, int32_t n_owner, int32_t* ohdr, int32_t ohdr_stride, void* stream) {
    if (n_owner < 0 || n_owner > 65535 || (n_owner > 0 && (!prec || !ohdr || ohdr_stride < n_batches))) return TKR_EINVAL;
    const int own_words = n_owner > 0 ? ((n_items + n_owner - 1) / n_owner + 31) / 32 : 0;
    if (n_owner > 0 && (size_t)4 * n_owner * (own_words + 1) > 60 * 1024) return TKR_EUNSUPPORTED;      // the owners' row bitmaps live in LDS
    if (n_tr <= 0 || n_items <= 0 || n_users <= 0 || batch_size <= 0 || n_batches < 0) return TKR_EINVAL;
    if (n_users >= (1 << 30) || n_items >= (1 << 30)) return TKR_EUNSUPPORTED;   // id bits 30/31 carry flags
    if (n_batches > 32 * tkr::kTouchWords) return TKR_EUNSUPPORTED;
    if (n_batches == 0) return TKR_OK;
    if (!ucnt || !icnt || !touch_u || !touch_i || !occt) return TKR_EINVAL;
    const bool flow = prec != nullptr;                      // dataflow form of the plan (csrc/bpr_flow.hip)
    if (flow ? !pocc : (!rec || !hdr)) return TKR_EINVAL;
    if (!flow && tkr_plan_mid_ok(n_users, n_items, batch_size) && workspace &&
        workspace_bytes >= tkr_plan_mid_workspace_bytes(batch_size, n_batches)) {
        // 1024 < batch <= 16,384: the counting planner of csrc/planner_mid.hip (four launches, no library sort; TKR_PLAN_MID_FROM moves
        // its lower end, a value above 16,384 switches it off)
        const int rc = tkr_sample_plan_mid(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_users, n_items, seed, first_triplet, ctl,
                                           n_batches, batch_size, ucnt, icnt, touch_u, touch_i, out_u, out_i, out_j, task, occ, rec,
                                           hdr, occt, tpar, workspace, workspace_bytes, stream);
        return rc;                                          // (the counters advance inside: its own fold of batch-major touch maps, or K1c)
    }
    static const int big_from = [] { const char* e = getenv("TKR_PLAN_BIG_FROM"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4096; }();
    if (batch_size > 8192 || (batch_size >= big_from && !flow && workspace &&
                              workspace_bytes >= tkr_plan_workspace_bytes_for(batch_size, n_batches))) {
        // above 8192 the 2B 64-bit keys no longer fit one workgroup's LDS; from 4096 (TKR_PLAN_BIG_FROM) the grid-wide planner is
        // faster alone (per batch: 4.7 vs 6.3 us at 4096, 7.6 vs 10.5 us at 8192).  Beside the steps neither hides (round 5: the
        // steps slow down by what the planner takes): in ONE long call the per-batch planner's three launches come out ahead (22.9 vs
        // 24.1 us per batch at 8192, scripts/probe_plan_host.py), in epoch-sized calls (BPR.train, bench.py: 122 batches per call at
        // 8192) the faster planner does (24.7 vs 28.8) -- that is the default.  The two produce identical plans
        if (flow) return TKR_EUNSUPPORTED;                  // the dataflow step is for small batches
        const int rc = tkr_sample_plan_big(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_users, n_items, seed, first_triplet, ctl,
                                           n_batches, batch_size, ucnt, icnt, touch_u, touch_i, out_u, out_i, out_j, task, occ, rec,
                                           hdr, occt, tpar, workspace, workspace_bytes, stream);
        if (rc != TKR_OK) return rc;
        const int rows_ = n_users + n_items;
        hipLaunchKernelGGL(tkr::commit_kernel, dim3((rows_ + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_users, n_items, ucnt,
                           icnt, touch_u, touch_i);
        TKR_LAUNCH_CHECK();
        return TKR_OK;
    }
    int npad = 1, npad_bits = 0;
    while (npad < 2 * batch_size) { npad <<= 1; ++npad_bits; }
    // 32-bit sort keys (row << bits | occurrence) when the row ids leave the room; TKR_PLAN_LDS_SORT=1: the LDS sort always
    static const bool lds_sort = [] { const char* e = getenv("TKR_PLAN_LDS_SORT"); return e && atoi(e) != 0; }();
    const bool reg_sort_ok = !lds_sort && npad_bits < 31 && (uint64_t)(n_users > n_items ? n_users : n_items) < (1ull << (32 - npad_bits)) - 1ull;
    hipStream_t s = (hipStream_t)stream;
    if (batch_size <= 1024) {
        const size_t lds = (size_t)npad * 8 + (tkr::kPlanThreads + 1) * sizeof(int);
        hipLaunchKernelGGL(tkr::sample_plan_kernel<tkr::kPlanThreads>, dim3(n_batches), dim3(tkr::kPlanThreads), lds, s,
                           tr_users, (uint32_t)n_tr, row_ptr, pos_cols, cols_sorted, (uint32_t)n_items, seed,
                           first_triplet, ctl, batch_size, npad, out_u, out_i, out_j, reinterpret_cast<int4*>(task),
                           reinterpret_cast<int2*>(occ), occt, touch_u, touch_i, reg_sort_ok);
    } else {
        const size_t lds = (size_t)npad * 8 + (tkr::kPlanThreadsBig + 1) * sizeof(int);
        static bool attr_set[64] = {};                 // per device: the attribute belongs to the device's code object
        int dev = 0;
        TKR_CHECK(hipGetDevice(&dev));
        if (lds > 64 * 1024 && !(dev >= 0 && dev < 64 && attr_set[dev])) {
            TKR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tkr::sample_plan_kernel<tkr::kPlanThreadsBig>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(tkr::sample_plan_kernel<tkr::kPlanThreadsBig>, dim3(n_batches), dim3(tkr::kPlanThreadsBig), lds,
                           s, tr_users, (uint32_t)n_tr, row_ptr, pos_cols, cols_sorted, (uint32_t)n_items, seed,
                           first_triplet, ctl, batch_size, npad, out_u, out_i, out_j, reinterpret_cast<int4*>(task),
                           reinterpret_cast<int2*>(occ), occt, touch_u, touch_i, reg_sort_ok);
    }
    TKR_LAUNCH_CHECK();
    // (256 threads: a 1024-thread workgroup does not fit beside the persistent step's 12 waves per CU -- 4 waves per SIMD against the
    // one that its registers leave -- and the planner of the next chunk then waits for the running chunk to END: measured 1.06 ms per
    // launch of this kernel in the profiled bench, -2 % on the steady state)
    if (flow) {
        // a short call (its plan sits in front of its step, nothing runs beside it): one thread per task instead of three tasks per thread
        static const int wide_upto = [] { const char* e = getenv("TKR_PLAN_WIDE_UPTO"); return e ? atoi(e) : 64; }();
        const size_t lds_r = n_owner > 0 ? (size_t)4 * n_owner * (own_words + 1) : 0;
        const size_t lds_w = tkr::plan_phase_b_wide_lds(batch_size, n_owner, own_words);
        if (n_batches <= wide_upto && 3 * batch_size <= tkr::kWideThreads && lds_w <= 64 * 1024)
            hipLaunchKernelGGL(tkr::resolve_flow_wide_kernel, dim3(n_batches), dim3(tkr::kWideThreads), lds_w, s,
                               batch_size, reinterpret_cast<const int4*>(task), reinterpret_cast<const int2*>(occ), ucnt, icnt, touch_u,
                               touch_i, reinterpret_cast<int4*>(pocc), reinterpret_cast<int4*>(prec), n_owner, ohdr, ohdr_stride, occt, own_words);
        else
            hipLaunchKernelGGL(tkr::resolve_flow_kernel<tkr::kPlanThreads>, dim3(n_batches), dim3(tkr::kPlanThreads), lds_r, s, batch_size,
                               reinterpret_cast<const int4*>(task), reinterpret_cast<const int2*>(occ), ucnt, icnt, touch_u,
                               touch_i, reinterpret_cast<int4*>(pocc), reinterpret_cast<int4*>(prec), n_owner, ohdr, ohdr_stride, occt, own_words);
    }
    else
        hipLaunchKernelGGL(tkr::resolve_kernel, dim3(n_batches), dim3(tkr::kPlanThreads), 0, s, batch_size,
                           tkr_plan_max_blocks(batch_size) * tkr::team_for(batch_size), reinterpret_cast<int4*>(task),
                           reinterpret_cast<int2*>(occ), occt, ucnt, icnt, touch_u, touch_i, rec,
                           reinterpret_cast<int4*>(hdr), out_u, tpar);
    TKR_LAUNCH_CHECK();
    const int rows = n_users + n_items;
    hipLaunchKernelGGL(tkr::commit_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, n_users, n_items, ucnt, icnt,
                       touch_u, touch_i);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_sample_plan(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr,
                               const int32_t* pos_cols, const int32_t* cols_sorted, int32_t n_users,
                               int32_t n_items, uint64_t seed, uint64_t first_triplet, const int64_t* ctl,
                               int32_t n_batches, int32_t batch_size, int32_t* ucnt, int32_t* icnt,
                               uint32_t* touch_u, uint32_t* touch_i, int32_t* out_u, int32_t* out_i,
                               int32_t* out_j, int32_t* task, int32_t* occ, int32_t* rec, int32_t* hdr,
                               int32_t* occt, int32_t* tpar, int32_t* prec, int32_t* pocc, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return sample_plan_impl(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_users, n_items, seed, first_triplet, ctl, n_batches,
                            batch_size, ucnt, icnt, touch_u, touch_i, out_u, out_i, out_j, task, occ, rec, hdr, occt, tpar, prec, pocc,
                            workspace, workspace_bytes, 0, nullptr, 0, stream);
}

extern "C" int tkr_sample_plan_owned(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr,
                                     const int32_t* pos_cols, const int32_t* cols_sorted, int32_t n_users,
                                     int32_t n_items, uint64_t seed, uint64_t first_triplet, int32_t n_batches,
                                     int32_t batch_size, int32_t* ucnt, int32_t* icnt, uint32_t* touch_u, uint32_t* touch_i,
                                     int32_t* out_u, int32_t* out_i, int32_t* out_j, int32_t* task, int32_t* occ, int32_t* occt,
                                     int32_t* prec, int32_t* pocc, int32_t n_owner, int32_t* ohdr, int32_t ohdr_stride, void* stream) {
    if (n_owner <= 0) return TKR_EINVAL;
    return sample_plan_impl(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_users, n_items, seed, first_triplet, nullptr, n_batches,
                            batch_size, ucnt, icnt, touch_u, touch_i, out_u, out_i, out_j, task, occ, nullptr, nullptr, occt, nullptr, prec,
                            pocc, nullptr, 0, n_owner, ohdr, ohdr_stride, stream);
}

extern "C" int tkr_plan_rollback(const int32_t* task, int32_t batch_size, int32_t first_batch, int32_t n_batches,
                                 int32_t* ucnt, int32_t* icnt, void* stream) {
    if (!task || !ucnt || !icnt || batch_size <= 0 || first_batch < 0 || n_batches < 0) return TKR_EINVAL;
    if (n_batches == 0) return TKR_OK;
    const size_t per = (size_t)3 * batch_size, n_slots = per * n_batches;
    hipLaunchKernelGGL(tkr::rollback_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const int4*>(task) + per * first_batch, n_slots, ucnt, icnt);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

#ifdef TKR_K1_PROF
extern "C" int tkr_debug_k1_prof(unsigned long long* out /*[32] host*/) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tkr::k1_prof), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -100;
}
#endif
