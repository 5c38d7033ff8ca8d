// The pieces of K1 (the (u,i,j) draw + batch planner, csrc/sampler.hip) as device functions, so that two kernels can run them:
// the planner kernels of sampler.hip (one workgroup per batch, beside or in front of the step) and the PROLOGUE of the persistent
// step csrc/bpr_own.hip, which plans the batches of a short call inside its own launch (one launch instead of four: VERDICT r4 #2).
// Replaces BPR._uniform_user_sampling (single/bpr.py:155-165) and the duplicate bookkeeping of TF's sparse optimizer
// (single/bpr.py:100); every output word is defined by oracle/plan_np.py.
#pragma once
#include "tkr_common.h"
#include "sampler_draw.h"

#ifndef K1_STAMP
#define K1_STAMP(i) do { } while (0)
#endif
#ifndef K1_STAMP_ITEMS
#define K1_STAMP_ITEMS(i) do { } while (0)
#endif

namespace tkr {

constexpr int kPlanThreads = 256;      // resolve/commit kernels, and sample_plan for B <= 1024
constexpr int kPlanThreadsBig = 1024;  // sample_plan for larger batches (the LDS sort dominates there)
constexpr int kLightMax = 4;     // oracle/plan_np.py LIGHT_MAX: occurrences one wave handles, B <= 4096
constexpr int kLightMaxBig = 16; // ... LIGHT_MAX_BIG for larger batches
__host__ __device__ inline int light_max(int B) { return B <= 4096 ? kLightMax : kLightMaxBig; }
constexpr int kTeamBig = 16;     // oracle/plan_np.py TEAM: waves per workgroup / heavy task, B > 16384
constexpr int kTeamMid = 8;      // ... TEAM_MID, 1024 < B <= 16384: at ~72 registers a CU holds three 8-wave workgroups (24 waves) but one of
                                 // 16 waves -- step kernel per batch, 16 vs 8 (round 6): 9.5 / 7.9 us at 2048, 13.0 / 11.5 at 4096, 19.3 / 17.1 at
                                 // 8192, 30.7 / 29.5 at 16,384; 50.4 / 55.4 at 32,768 and 133 / 159 at 65,536 (a popular item's heavy team is
                                 // half as wide); 6 and 4 waves lose to 8 everywhere
constexpr int kTeamSmall = 4;    // ... TEAM_SMALL for B <= 1024 (spreads a small batch over many CUs)
__host__ __device__ inline int team_for(int B) { return B <= 1024 ? kTeamSmall : B <= 16384 ? kTeamMid : kTeamBig; }
// light tasks per workgroup (oracle light_per_block): every wave slot (half-filled groups measured slower)
__host__ __device__ inline int light_per_block(int B) { return team_for(B); }
constexpr int kTouchWords = 16;  // bitmap words per row -> at most 512 batches per call

// In-LDS bitonic sort of n (power of two) 64-bit keys, ascending.
template <int T>
__device__ __forceinline__ void bitonic_sort(uint64_t* keys, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int p = threadIdx.x; p < (n >> 1); p += T) {
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

// ---- the same sort in registers ------------------------------------------------------------------------------------------------
// The LDS sort above pays one barrier + one LDS round trip per compare-exchange step (45 steps for 512 keys: 8.3 us of a 24 us
// kernel that sits in front of every short call).  Here thread t holds keys R*t .. R*t + R - 1 (32-bit: row << OB | occurrence):
// strides below R exchange registers of one thread, strides below 64 R lanes of one wave -- DPP moves and v_permlane{16,32}_swap,
// vector-ALU instructions, no LDS, no barrier -- and only the strides from 64 R on (3 steps of 45 at 512 keys over 4 waves) go
// through LDS.  Same network, same result.
typedef uint32_t lane_pair __attribute__((ext_vector_type(2)));
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "xor mask inside a wave");
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);          // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {                                                                                     // i -> 7 - i -> its quad reversed = i ^ 4
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);                               // row_half_mirror
        return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1b, 0xf, 0xf, false);                                  // quad_perm [3,2,1,0]
    } else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
    else if constexpr (M == 16) {
        const lane_pair r = __builtin_amdgcn_permlane16_swap(v, v, false, false);     // x = rows [0,0,2,2] of v, y = rows [1,1,3,3]
        return (lane & 16) ? r.x : r.y;
    } else {
        const lane_pair r = __builtin_amdgcn_permlane32_swap(v, v, false, false);     // x = [lower half, lower half], y = [upper, upper]
        return (lane & 32) ? r.x : r.y;
    }
}
template <int R, int M>
__device__ __forceinline__ void lane_step(uint32_t (&k)[R], int lane, bool keep_min) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t o = lane_xor<M>(k[r], lane);
        k[r] = keep_min ? min(k[r], o) : max(k[r], o);
    }
}
// n = R * T keys, ascending over e = R * thread + r; `xbuf`: n words of LDS, free on entry (barrier inside before its first use)
template <int T, int R>
__device__ __forceinline__ void register_sort(uint32_t (&k)[R], uint32_t* xbuf, bool active = true, int tid_role = -1) {
    // `active` (wave-uniform): false for the waves of a LARGER workgroup than the T sorting threads (the planner prologue of
    // csrc/bpr_own.hip runs on 768): they only meet the others at the barriers
    const int tid = tid_role >= 0 ? tid_role : (int)threadIdx.x, lane = tid & (TKR_WAVE - 1);      // (tid_role: the thread's number inside a GROUP of T threads of a larger workgroup)
    constexpr int n = R * T;
    constexpr int LOGN = __builtin_ctz(n);
    static_assert((n & (n - 1)) == 0, "power of two");
    // fully unrolled: sizes, strides and register indices are compile-time constants (a register array indexed by a run-time
    // stride would live in scratch memory)
#pragma unroll
    for (int ls = 1; ls <= LOGN; ++ls) {
#pragma unroll
        for (int lj = ls - 1; lj >= 0; --lj) {
            const int size = 1 << ls, stride = 1 << lj;
            if (stride < R) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int q = r ^ stride;
                    if (q > r) {
                        const bool asc = (((tid * R + r) & size) == 0);
                        const uint32_t a = k[r], b = k[q];
                        const bool sw = (a > b) == asc;
                        k[r] = sw ? b : a;
                        k[q] = sw ? a : b;
                    }
                }
            } else {
                const int m = stride / R;                            // lane / thread distance
                const bool asc = (((tid * R) & size) == 0);
                const bool lower = (tid & m) == 0;
                const bool keep_min = lower == asc;
                if (m == 1) lane_step<R, 1>(k, lane, keep_min);
                else if (m == 2) lane_step<R, 2>(k, lane, keep_min);
                else if (m == 4) lane_step<R, 4>(k, lane, keep_min);
                else if (m == 8) lane_step<R, 8>(k, lane, keep_min);
                else if (m == 16) lane_step<R, 16>(k, lane, keep_min);
                else if (m == 32) lane_step<R, 32>(k, lane, keep_min);
                else {                                               // across waves: through LDS
                    __syncthreads();
                    if (active) {
#pragma unroll
                        for (int r = 0; r < R; ++r) xbuf[tid * R + r] = k[r];
                    }
                    __syncthreads();
                    if (active) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const uint32_t o = xbuf[(tid ^ m) * R + r];
                            k[r] = keep_min ? min(k[r], o) : max(k[r], o);
                        }
                    }
                }
            }
        }
    }
}
// a whole sort of this kernel through registers: keys of element e from `make(e)` (row << ob | occurrence, ~0 = padding), result
// into keys[] in the 64-bit form the rest of the kernel reads (row << 32 | occurrence)
template <int T, int R, class Make>
__device__ __forceinline__ void sort_via_registers(uint64_t* keys, int ob, Make make) {
    const bool active = (int)threadIdx.x < T;                        // (a workgroup of more than T threads: the others meet the barriers only)
    uint32_t k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) k[r] = active ? make((int)threadIdx.x * R + r) : 0xffffffffu;
    // the exchange buffer: the upper half of keys[] (n 64-bit slots = 2n words; the lower n words stay clear of the 64-bit result
    // only after the barrier below)
    register_sort<T, R>(k, reinterpret_cast<uint32_t*>(keys), active);
    __syncthreads();
    if (active) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t v = k[r];
            keys[(int)threadIdx.x * R + r] = v == 0xffffffffu ? ~0ull : (((uint64_t)(v >> ob) << 32) | (uint64_t)(v & ((1u << ob) - 1u)));
        }
    }
    __syncthreads();
}

// Turn sorted keys[0..n) (row<<32 | occurrence) into task heads + counts.  Returns the
// number of groups (uniform across the block).  `slot0` = first task slot to fill,
// `occ0` = occ offset of sorted position 0, `kind` = 0 users / 1 items.
template <int T>
__device__ __forceinline__ int emit_tasks(const uint64_t* keys, int n, int4* task, int slot0, int occ0,
                                          int kind, int* scan /*LDS [T+1]*/,
                                          uint32_t* __restrict__ touch, int batch) {
    const int per = (n + T - 1) / T;
    const int beg = min((int)threadIdx.x * per, n), end = min(beg + per, n);
    int cnt = 0;
    for (int p = beg; p < end; ++p)
        cnt += (p == 0) || ((uint32_t)(keys[p] >> 32) != (uint32_t)(keys[p - 1] >> 32));
    // exclusive scan of the per-thread head counts: inside a wave by DPP-free shuffles, the <= 16 wave totals by every thread
    // (thread 0 used to walk all T entries of the LDS array: 256 dependent read-modify-writes, ~7 us of a 35 us kernel, twice)
    const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x / TKR_WAVE;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < TKR_WAVE; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    __syncthreads();                                  // `scan` may still be read from the call before
    if (lane == TKR_WAVE - 1) scan[wave] = incl;
    __syncthreads();
    int s = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < T / TKR_WAVE; ++w) {
        const int t = scan[w];
        if (w < wave) s += t;
        total += t;
    }
    for (int p = beg; p < end; ++p) {
        const uint32_t row = (uint32_t)(keys[p] >> 32);
        if ((p == 0) || (row != (uint32_t)(keys[p - 1] >> 32))) {
            // length of this group: scan forward to the next head (groups are short on
            // average; long ones cost O(len) once)
            int q = p + 1;
            while (q < n && (uint32_t)(keys[q] >> 32) == row) ++q;
            task[slot0 + s] = make_int4((int)(row | ((uint32_t)kind << 31)), occ0 + p, q - p, 0);
            atomicOr(&touch[(size_t)row * kTouchWords + (batch >> 5)], 1u << (batch & 31));
            ++s;
        }
    }
    return total;
}


// ---- phase A of a batch: draw, the two sorts, task heads, occurrence lists, touch bits ------------------------------------------
// T threads run it, TS <= T of them hold the register sorts (TS = 256 in both users: the sizes of a batch up to 1024 fit; the
// prologue of csrc/bpr_own.hip runs it on the 768 threads of a step workgroup).  `smem`: npad_items * 8 + (T / 64 + 1) * 4 bytes.
template <int T, int TS>
__device__ __forceinline__ void plan_phase_a(unsigned char* smem, int b, const int32_t* __restrict__ tr_users, uint32_t n_tr,
                                             const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ pos_cols,
                                             const int32_t* __restrict__ cols_sorted, uint32_t n_items, uint64_t seed, uint64_t g0, int B,
                                             int npad_items, int32_t* __restrict__ bu, int32_t* __restrict__ bi, int32_t* __restrict__ bj,
                                             int4* __restrict__ task, int2* __restrict__ occ, int32_t* __restrict__ occt,
                                             uint32_t* __restrict__ touch_u, uint32_t* __restrict__ touch_i, bool reg_sort_ok) {
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                       // [npad_items]
    int* scan = reinterpret_cast<int*>(smem + (size_t)npad_items * 8);        // [T / 64 + 1]
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    K1_STAMP(0);

    // ---- draw; item keys go to LDS, triplets to HBM ---------------------------------
    for (int t = threadIdx.x; t < B; t += T) {
        int u, i, j;
        draw_triplet(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_items, k0, k1, g0 + t, u, i, j);
        bu[t] = u; bi[t] = i; bj[t] = j;
    }
    __threadfence_block();
    __syncthreads();
    K1_STAMP(1);

    // ---- users: sort (u<<32 | t) ----------------------------------------------------------
    int npad_u = 1;
    while (npad_u < B) npad_u <<= 1;
    // registers when the sizes allow (npad_u = TS or 2TS or 4TS keys, 32-bit keys: row ids and occurrence numbers fit one word)
    int ob = 1;
    while ((1 << ob) < npad_items) ++ob;
    const bool in_regs = reg_sort_ok && npad_items == 2 * npad_u && (npad_u == TS || npad_u == 2 * TS || npad_u == 4 * TS || npad_u == 8 * TS);
    if (in_regs) {
        auto make = [&](int t) { return t < B ? (((uint32_t)bu[t] << ob) | (uint32_t)t) : 0xffffffffu; };
        if (npad_u == TS) sort_via_registers<TS, 1>(keys, ob, make);
        else if (npad_u == 2 * TS) sort_via_registers<TS, 2>(keys, ob, make);
        else if (npad_u == 4 * TS) sort_via_registers<TS, 4>(keys, ob, make);
        else sort_via_registers<TS, 8>(keys, ob, make);
    } else {
        for (int t = threadIdx.x; t < npad_u; t += T)
            keys[t] = (t < B) ? (((uint64_t)(uint32_t)bu[t] << 32) | (uint32_t)t) : ~0ull;
        bitonic_sort<T>(keys, npad_u);
    }
    K1_STAMP(2);
    const int n_uq = emit_tasks<T>(keys, B, task, 0, 0, 0, scan, touch_u, b);
    K1_STAMP(3);
    for (int p = threadIdx.x; p < B; p += T) {
        const int t = (int)(uint32_t)keys[p];
        occ[p] = make_int2(bi[t], bj[t]);
        occt[p] = t;
    }
    __syncthreads();
    K1_STAMP(4);

    // ---- items: sort (item<<32 | o), o<B: i-role of triplet o, else j-role of o-B ---------
    if (in_regs) {
        auto make = [&](int o) {
            return o < B ? (((uint32_t)bi[o] << ob) | (uint32_t)o) : o < 2 * B ? (((uint32_t)bj[o - B] << ob) | (uint32_t)o) : 0xffffffffu;
        };
        if (npad_u == TS) sort_via_registers<TS, 2>(keys, ob, make);
        else if (npad_u == 2 * TS) sort_via_registers<TS, 4>(keys, ob, make);
        else if (npad_u == 4 * TS) sort_via_registers<TS, 8>(keys, ob, make);
        else sort_via_registers<TS, 16>(keys, ob, make);
    } else {
        for (int o = threadIdx.x; o < npad_items; o += T) {
            uint64_t key = ~0ull;
            if (o < B) key = ((uint64_t)(uint32_t)bi[o] << 32) | (uint32_t)o;
            else if (o < 2 * B) key = ((uint64_t)(uint32_t)bj[o - B] << 32) | (uint32_t)o;
            keys[o] = key;
        }
        bitonic_sort<T>(keys, npad_items);
    }
    K1_STAMP(5);
    const int n_iq = emit_tasks<T>(keys, 2 * B, task, n_uq, B, 1, scan, touch_i, b);
    K1_STAMP(6);
    for (int p = threadIdx.x; p < 2 * B; p += T) {
        const int o = (int)(uint32_t)keys[p];
        const bool role = o >= B;
        const int t = role ? o - B : o;
        const uint32_t other = (uint32_t)(role ? bi[t] : bj[t]);
        occ[B + p] = make_int2(bu[t], (int)(other | ((uint32_t)role << 31)));
        occt[B + p] = t;
    }
    for (int s = n_uq + n_iq + threadIdx.x; s < 3 * B; s += T) task[s] = make_int4(-1, 0, 0, 0);
    __syncthreads();
    K1_STAMP(7);
}

// ---- what phase A of a batch leaves in LDS for phase B of the SAME workgroup (the planner prologue of csrc/bpr_own.hip): the draw, the
// task heads and the occurrence lists.  Phase B then starts from LDS instead of a round trip to the words this CU has just stored,
// and phase A itself builds its sort keys and occurrence lists from the LDS copy of the draw (two more round trips).
struct PlanMirror {
    int4* task;        // [3B]
    int2* occ;         // [3B]
    int32_t* occt;     // [3B]
    int32_t *u, *i, *j;        // [B] each
};
__device__ __forceinline__ PlanMirror plan_mirror(unsigned char* p /*16-byte aligned*/, int B) {
    PlanMirror m;
    m.task = reinterpret_cast<int4*>(p);
    m.occ = reinterpret_cast<int2*>(m.task + 3 * B);
    m.occt = reinterpret_cast<int32_t*>(m.occ + 3 * B);
    m.u = m.occt + 3 * B;
    m.i = m.u + B;
    m.j = m.i + B;
    return m;
}
static inline size_t plan_mirror_lds(int B) { return (size_t)3 * B * (16 + 8 + 4) + (size_t)3 * B * 4; }

// the LDS side of a workgroup barrier only: LDS traffic drained (lgkmcnt), global loads of the wave stay in flight across it
// (__syncthreads waits for them too)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- phase A with the user side and the item side of a batch on TWO wave groups at once (the planner prologue of csrc/bpr_own.hip,
// 129 <= B <= 256): threads 0..255 sort the users and emit their task heads and occurrence lists while threads 256..511 do the same
// for the items (one after the other the two sides are 4.3 + 5.1 us of a 17 us phase); further threads only meet the barriers.  Both
// sides pass the same barriers in the same order: a register sort of R x 256 keys crosses waves in 3 steps whatever R is, the head
// count of either side is one scan.  Same words as plan_phase_a.  `smem`: (256 + 512) * 8 + 64 bytes.
template <int T>
__device__ __forceinline__ void plan_phase_a_split(unsigned char* smem, int b, const int32_t* __restrict__ tr_users, uint32_t n_tr,
                                                   const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ pos_cols,
                                                   const int32_t* __restrict__ cols_sorted, uint32_t n_items, uint64_t seed, uint64_t g0, int B,
                                                   int32_t* __restrict__ bu, int32_t* __restrict__ bi, int32_t* __restrict__ bj,
                                                   int4* __restrict__ task, int2* __restrict__ occ, int32_t* __restrict__ occt,
                                                   uint32_t* __restrict__ touch_u, uint32_t* __restrict__ touch_i, const PlanMirror mir) {
    static_assert(T >= 512 && T % 64 == 0, "two groups of 256 threads");
    constexpr int TS = 256;
    uint64_t* keys_u = reinterpret_cast<uint64_t*>(smem);                     // [256]
    uint64_t* keys_i = keys_u + 256;                                          // [512]
    int* scan = reinterpret_cast<int*>(keys_i + 512);                         // [2][4] head counts of the waves of either side
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int tid = threadIdx.x, role = tid >> 8, rt = tid & 255, lane = tid & 63, rw = rt >> 6;
    K1_STAMP(0);
    for (int t = tid; t < B; t += T) {
        int u, i, j;
        draw_triplet(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_items, k0, k1, g0 + t, u, i, j);
        bu[t] = u; bi[t] = i; bj[t] = j;
        mir.u[t] = u; mir.i[t] = i; mir.j[t] = j;
    }
    __syncthreads();
    bu = mir.u; bi = mir.i; bj = mir.j;                                       // from here on the draw is read from LDS
    K1_STAMP(1);
    constexpr int ob = 9;                                                     // occurrence bits of the 32-bit keys: 512 item occurrences
    const bool active = role < 2;
    uint64_t* keys = role == 0 ? keys_u : keys_i;
    // ---- the sorts (eight barriers either way)
    if (role == 1) {
        uint32_t k[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int o = rt * 2 + r;
            k[r] = o < B ? (((uint32_t)bi[o] << ob) | (uint32_t)o) : o < 2 * B ? (((uint32_t)bj[o - B] << ob) | (uint32_t)o) : 0xffffffffu;
        }
        register_sort<TS, 2>(k, reinterpret_cast<uint32_t*>(keys_i), true, rt);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t v = k[r];
            keys_i[rt * 2 + r] = v == 0xffffffffu ? ~0ull : (((uint64_t)(v >> ob) << 32) | (uint64_t)(v & ((1u << ob) - 1u)));
        }
        __syncthreads();
    } else {
        uint32_t k[1];
        k[0] = (role == 0 && rt < B) ? (((uint32_t)bu[rt] << ob) | (uint32_t)rt) : 0xffffffffu;
        register_sort<TS, 1>(k, reinterpret_cast<uint32_t*>(keys_u), role == 0, rt);
        __syncthreads();
        if (role == 0) {
            const uint32_t v = k[0];
            keys_u[rt] = v == 0xffffffffu ? ~0ull : (((uint64_t)(v >> ob) << 32) | (uint64_t)(v & ((1u << ob) - 1u)));
        }
        __syncthreads();
    }
    K1_STAMP(2);
    // ---- task heads: a thread's share of the sorted keys, the head counts scanned per side
    const int n = role == 0 ? B : 2 * B;
    const int per = role == 0 ? 1 : 2;
    const int beg = active ? min(rt * per, n) : 0, end = active ? min(beg + per, n) : 0;
    int cnt = 0;
    for (int p = beg; p < end; ++p) cnt += (p == 0) || ((uint32_t)(keys[p] >> 32) != (uint32_t)(keys[p - 1] >> 32));
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < TKR_WAVE; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (active && lane == TKR_WAVE - 1) scan[role * 4 + rw] = incl;
    __syncthreads();
    K1_STAMP(3); K1_STAMP_ITEMS(16);
    int tot_u = 0, tot_i = 0, s = incl - cnt;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int cu = scan[w], ci = scan[4 + w];
        tot_u += cu; tot_i += ci;
        if (w < rw) s += role == 0 ? cu : ci;
    }
    if (active) {
        const int slot0 = role == 0 ? 0 : tot_u, occ0 = role == 0 ? 0 : B;
        uint32_t* touch = role == 0 ? touch_u : touch_i;
        for (int p = beg; p < end; ++p) {
            const uint32_t row = (uint32_t)(keys[p] >> 32);
            if ((p == 0) || (row != (uint32_t)(keys[p - 1] >> 32))) {
                int q = p + 1;
                while (q < n && (uint32_t)(keys[q] >> 32) == row) ++q;
                const int4 head = make_int4((int)(row | ((uint32_t)role << 31)), occ0 + p, q - p, 0);
                task[slot0 + s] = head;
                mir.task[slot0 + s] = head;
                atomicOr(&touch[(size_t)row * kTouchWords + (b >> 5)], 1u << (b & 31));
                ++s;
            }
        }
        K1_STAMP(4); K1_STAMP_ITEMS(17);
        // ---- occurrence lists
        if (role == 0) {
            if (rt < B) {
                const int t = (int)(uint32_t)keys_u[rt];
                const int2 o2 = make_int2(bi[t], bj[t]);
                occ[rt] = o2; occt[rt] = t;
                mir.occ[rt] = o2; mir.occt[rt] = t;
            }
        } else {
            for (int p = rt; p < 2 * B; p += 256) {
                const int o = (int)(uint32_t)keys_i[p];
                const bool rj = o >= B;
                const int t = rj ? o - B : o;
                const uint32_t other = (uint32_t)(rj ? bi[t] : bj[t]);
                const int2 o2 = make_int2(bu[t], (int)(other | ((uint32_t)rj << 31)));
                occ[B + p] = o2; occt[B + p] = t;
                mir.occ[B + p] = o2; mir.occt[B + p] = t;
            }
        }
    }
    K1_STAMP(5); K1_STAMP_ITEMS(18);
    for (int q = tot_u + tot_i + tid; q < 3 * B; q += T) { task[q] = make_int4(-1, 0, 0, 0); mir.task[q] = make_int4(-1, 0, 0, 0); }
    __syncthreads();
    K1_STAMP(7);
}

// ---- versions from the touch bitmap -----------------------------------------------------------------------------------------------
// FRESH: the bitmap words were set by OTHER workgroups of this very launch (the prologue of csrc/bpr_own.hip, behind its arrival
// barrier): loaded past this CU's L1 (agent-scope relaxed loads; the L2s see each other's atomics).  Otherwise plain loads: the
// words come from an earlier launch.
template <bool FRESH>
__device__ __forceinline__ uint32_t touch_word(const uint32_t* p) {
    if constexpr (FRESH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
// number of updates of `row` before batch `batch` of this call = the VERSION of the row that batch reads
template <bool FRESH = false>
__device__ __forceinline__ int version_of(const int32_t* __restrict__ cnt, const uint32_t* touch, int row, int batch) {
    const uint32_t* w = touch + (size_t)row * kTouchWords;
    int c = cnt[row];
    const int full = batch >> 5;
    for (int q = 0; q < full; ++q) c += __popc(touch_word<FRESH>(w + q));
    c += __popc(touch_word<FRESH>(w + full) & ((1u << (batch & 31)) - 1u));
    return c;
}
__device__ __forceinline__ int parity_of(const int32_t* __restrict__ cnt, const uint32_t* __restrict__ touch, int row, int batch) {
    return version_of<false>(cnt, touch, row, batch) & 1;
}

// version of `row` at `batch` (as version_of) and the last batch < `batch` of this call that touched it (-1: none), from ONE
// round trip: the row's 16 bitmap words as four 16-byte loads (a walk down the words was up to 16 DEPENDENT loads per task: the
// planner of a 512-batch chunk took 1.08 ms beside the persistent step instead of 0.07); `total`: all the batches of the call
// that touch the row (what K1c adds to the row's counter)
typedef uint32_t touch_v4 __attribute__((ext_vector_type(4)));
template <bool FRESH = false>
__device__ __forceinline__ void row_history(const int32_t* __restrict__ cnt, const uint32_t* touch, int row, int batch, int& ver, int& prev,
                                            int& total) {
    uint32_t w[kTouchWords];
    static_assert(kTouchWords == 16, "four 16-byte loads per row");
    if constexpr (FRESH) {
        // ONE descriptor for the whole bitmap (wave-uniform base) + a per-lane byte offset: a descriptor built from each lane's own
        // row pointer makes hipcc wrap the loads in a waterfall loop -- one pass per distinct row of the wave, 64 serial round trips
        // (measured: 28 us for this function)
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(touch), 0, 0xfffffff0u, 0x00020000);
        const int off = row * (kTouchWords * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const touch_v4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off + q * 16, 0, 16 /*sc1*/);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    } else {
        const uint4* w4 = reinterpret_cast<const uint4*>(touch + (size_t)row * kTouchWords);
        const uint4 a = w4[0], b = w4[1], c = w4[2], d = w4[3];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w; w[12] = d.x; w[13] = d.y; w[14] = d.z; w[15] = d.w;
    }
    const int full = batch >> 5;
    const uint32_t below = (1u << (batch & 31)) - 1u;
    int v = cnt[row], p = -1, all = 0;
#pragma unroll
    for (int q = 0; q < kTouchWords; ++q) {
        const uint32_t bits = q < full ? w[q] : q == full ? (w[q] & below) : 0u;
        v += __popc(bits);
        all += __popc(w[q]);
        if (bits) p = q * 32 + 31 - __clz(bits);
    }
    ver = v;
    prev = p;
    total = all;
}
template <bool FRESH = false>
__device__ __forceinline__ void row_history(const int32_t* __restrict__ cnt, const uint32_t* touch, int row, int batch, int& ver, int& prev) {
    int total;
    row_history<FRESH>(cnt, touch, row, batch, ver, prev, total);
}

// a row's counter and the first 64 batches of its bitmap, as loads that are only ISSUED here (FRESH form of phase B: a call of at
// most 64 batches inside the step's launch; the words were set by other workgroups of the launch: past the L1)
struct RowBits { int cnt; unsigned long long bits; };
__device__ __forceinline__ RowBits row_bits(const int32_t* __restrict__ cnt, const uint32_t* touch, int row) {
    RowBits h;
    h.cnt = cnt[row];
    h.bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(touch + (size_t)row * kTouchWords), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return h;
}
// ... the version of the row at `batch` (< 64); prev = its last batch before that (-1: none), total = its batches in the call
__device__ __forceinline__ int row_bits_version(const RowBits& h, int batch, int& prev, int& total) {
    const unsigned long long below = h.bits & ((1ull << batch) - 1ull);
    prev = below ? 63 - __clzll((long long)below) : -1;
    total = __popcll(h.bits);
    return h.cnt + __popcll(below);
}

// ---- phase B of a batch, one thread per task slot and per occurrence (3B <= 768 threads): full versions + one 128-byte record
// per task (the dataflow form of the plan: csrc/sampler.hip has the layout).  FRESH: running inside the step's launch -- bitmap
// words past the L1 (above), and everything OTHER workgroups of the launch will read (pocc, prec, ohdr) stored write-through
// (sc1: in memory when the storing wave's vmcnt drains, no release fence over ~100 KB of dirty lines per workgroup).
// `smem` (16-byte aligned): plan_phase_b_wide_lds() bytes.  Returns this thread's task, its row's last earlier batch and the
// row's touches in the whole call (K1c's increment).
constexpr int kWideThreads = 768;
template <bool FRESH, bool MIRRORED = false>
__device__ __forceinline__ void plan_phase_b_wide(unsigned char* smem, int b, int B, const int4* task_b, const int2* occ_b, const int32_t* occt_b,
                                                  const int32_t* __restrict__ ucnt, const int32_t* __restrict__ icnt, const uint32_t* touch_u,
                                                  const uint32_t* touch_i, int4* pocc /*of batch b*/, int4* prec /*of batch b*/, int n_owner,
                                                  int32_t* ohdr, int ohdr_stride, int own_words, int4& t_out, int& prev_out, int& total_out,
                                                  const PlanMirror mir = PlanMirror{}) {
    constexpr int T = kWideThreads;
    const int n = 3 * B, s = threadIdx.x;
    int4* lp = reinterpret_cast<int4*>(smem);                                          // [3B] the batch's pocc
    int32_t* lt = reinterpret_cast<int32_t*>(lp + n);                                  // [3B] its occt
    uint32_t* own_mask = reinterpret_cast<uint32_t*>(lt + n);                          // [n_owner][own_words]
    uint32_t* own_start = own_mask + (size_t)n_owner * own_words;                      // [n_owner]
    int* s_wave = reinterpret_cast<int*>(own_start + n_owner);                         // [T / 64]
    int* s_first_item = s_wave + T / TKR_WAVE;
    // [3B][2] words 0..7 of every record, by destination slot (the offset is rounded up as an integer, not the pointer: a pointer that
    // has been through uintptr_t is a GENERIC pointer to hipcc, and every access through it a flat instruction)
    const size_t hdr_off = ((size_t)n * 20 + (size_t)4 * n_owner * (own_words + 1) + (T / TKR_WAVE + 1) * 4 + 15) & ~(size_t)15;
    int4* hdr = reinterpret_cast<int4*>(smem + hdr_off);
    const __amdgpu_buffer_rsrc_t pocc_r = __builtin_amdgcn_make_buffer_rsrc(pocc, 0, n * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t prec_r = __builtin_amdgcn_make_buffer_rsrc(prec, 0, n * 128, 0x00020000);
    auto put = [&](const __amdgpu_buffer_rsrc_t& r, int4* base, int idx16, const int4 v) {      // int4 number idx16 of a batch's array
        if constexpr (FRESH) {
            touch_v4 x;
            x.x = (uint32_t)v.x; x.y = (uint32_t)v.y; x.z = (uint32_t)v.z; x.w = (uint32_t)v.w;
            __builtin_amdgcn_raw_buffer_store_b128(x, r, idx16 * 16, 0, 16 /*sc1*/);
        } else {
            base[idx16] = v;
        }
    };
    K1_STAMP(8);
    if (s == 0) *s_first_item = n;
    for (int w = s; w < n_owner * own_words; w += T) own_mask[w] = 0u;
    int4 t = make_int4(-1, 0, 0, 0);
    int ver = 0, prev = -1, total = 0;
    // FRESH (a call planned inside the step's launch: at most 64 batches, so a row's history is its counter and bitmap words 0..1):
    // the three histories of a slot are ISSUED here and only looked at behind the owner order below -- LDS work between barriers that
    // wait for LDS alone (lds_barrier), ~3.6 us that used to follow the loads' round trip instead of hiding it
    RowBits ha = {0, 0ull}, hb = {0, 0ull}, ht = {0, 0ull};
    int2 o = make_int2(0, 0);
    int tt = 0;
    auto bar = [&]() { if constexpr (FRESH) lds_barrier(); else __syncthreads(); };
    if (s < n) {
        if constexpr (MIRRORED) { t = mir.task[s]; o = mir.occ[s]; tt = mir.occt[s]; }          // phase A of this workgroup left them in LDS
        else { t = task_b[s]; o = occ_b[s]; tt = occt_b[s]; }
        const bool user_occ = s < B;                                                   // user occurrences: (i, j); item occurrences: (u, other | role << 31)
        if constexpr (FRESH) {
            ha = user_occ ? row_bits(icnt, touch_i, o.x) : row_bits(ucnt, touch_u, o.x);
            hb = row_bits(icnt, touch_i, o.y & 0x3fffffff);
            if (t.x != -1) ht = t.x < 0 ? row_bits(icnt, touch_i, t.x & 0x7fffffff) : row_bits(ucnt, touch_u, t.x);
        } else {
            const int va = user_occ ? version_of<FRESH>(icnt, touch_i, o.x, b) : version_of<FRESH>(ucnt, touch_u, o.x, b);
            const int vb = version_of<FRESH>(icnt, touch_i, o.y & 0x3fffffff, b);
            if (t.x != -1) {
                if (t.x < 0) row_history<FRESH>(icnt, touch_i, t.x & 0x7fffffff, b, ver, prev, total);
                else row_history<FRESH>(ucnt, touch_u, t.x, b, ver, prev, total);
            }
            const int4 po = make_int4(o.x, va, o.y, vb);
            put(pocc_r, pocc, s, po);
            lp[s] = po;
            lt[s] = tt;
        }
    }
    bar();
    K1_STAMP(9);
    const bool item_task = t.x < 0 && t.x != -1;
    int first_item = 0;
    if (n_owner > 0) {
        {   // the first item task of the batch: one LDS atomic per wave (every item task's own atomicMin queued ~500 deep on one word)
            const unsigned long long im = __ballot(item_task);
            if (im != 0ull && (s & (TKR_WAVE - 1)) == 0) atomicMin(s_first_item, (s & ~(TKR_WAVE - 1)) + __ffsll((long long)im) - 1);
        }
        if (item_task) {
            const int row = t.x & 0x7fffffff, bit = row / n_owner;
            atomicOr(&own_mask[(size_t)(row % n_owner) * own_words + (bit >> 5)], 1u << (bit & 31));
        }
        bar();
        first_item = *s_first_item;
        const int per = (n_owner + T - 1) / T;
        const int w0 = min(s * per, n_owner), w1 = min(w0 + per, n_owner);
        int mine = 0;
        for (int w = w0; w < w1; ++w)
            for (int j = 0; j < own_words; ++j) mine += __popc(own_mask[(size_t)w * own_words + j]);
        const int lane = s & (TKR_WAVE - 1), wave = s / TKR_WAVE;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < TKR_WAVE; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == TKR_WAVE - 1) s_wave[wave] = incl;
        bar();
        int run = incl - mine;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
        for (int w = w0; w < w1; ++w) {
            int c = 0;
            for (int j = 0; j < own_words; ++j) c += __popc(own_mask[(size_t)w * own_words + j]);
            const int hw = (first_item + run) | (c << 16);
            if constexpr (FRESH) __hip_atomic_store(&ohdr[(size_t)w * ohdr_stride + b], hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ohdr[(size_t)w * ohdr_stride + b] = hw;
            own_start[w] = (uint32_t)run;
            run += c;
        }
        bar();
    }
    K1_STAMP(10);
    if constexpr (FRESH) {
        if (s < n) {
            int pa_, ta_;
            const int4 po = make_int4(o.x, row_bits_version(ha, b, pa_, ta_), o.y, row_bits_version(hb, b, pa_, ta_));
            if (t.x != -1) ver = row_bits_version(ht, b, prev, total);
            put(pocc_r, pocc, s, po);
            lp[s] = po;
            lt[s] = tt;
        }
    }
    // the 128-byte records, written as WHOLE cache lines: a task's two header words go to LDS at its destination slot, then eight
    // lanes assemble one record each pass (a wave: eight consecutive records = 1 KB contiguous).  A thread that writes its own record
    // piece by piece puts eight 16-byte stores on eight different lines per instruction: 6,144 partial-line writes per batch, and as
    // write-through stores (FRESH) each is a fabric transaction of its own -- 34 us of a 16 us phase.
    if (s < n) {
        int dst = s;
        if (n_owner > 0 && item_task) {
            const int row = t.x & 0x7fffffff, w = row % n_owner, bit = row / n_owner;
            const uint32_t* m = own_mask + (size_t)w * own_words;
            int before = __popc(m[bit >> 5] & ((1u << (bit & 31)) - 1u));
            for (int j = 0; j < (bit >> 5); ++j) before += __popc(m[j]);
            dst = first_item + (int)own_start[w] + before;
        }
        hdr[2 * dst] = t.x == -1 ? make_int4(-1, 0, 0, 0) : make_int4(t.x, ver, t.z, b * n + t.y);
        hdr[2 * dst + 1] = t.x == -1 ? make_int4(0, 0, 0, 0) : make_int4(b, prev, 0, 0);
    }
    __syncthreads();
    K1_STAMP(12);
    for (int base = 0; base < n; base += T / 8) {
        const int rec = base + (s >> 3), q = s & 7;
        if (rec < n) {
            const int4 h0 = hdr[2 * rec];
            int4 v = make_int4(0, 0, 0, 0);
            if (q == 0) v = h0;
            else if (q == 1) v = hdr[2 * rec + 1];
            else if (h0.x != -1 && q < 7) {
                const int occ0 = h0.w - b * n, cnt_occ = h0.z;                         // first occurrence inside the batch, occurrences
                if (q < 6) { if (q - 2 < cnt_occ) v = lp[occ0 + q - 2]; }
                else v = make_int4(lt[occ0], cnt_occ > 1 ? lt[occ0 + 1] : 0, cnt_occ > 2 ? lt[occ0 + 2] : 0, cnt_occ > 3 ? lt[occ0 + 3] : 0);
            }
            put(prec_r, prec, rec * 8 + q, v);
        }
    }
    K1_STAMP(11);
    t_out = t;
    prev_out = prev;
    total_out = total;
}
static inline size_t plan_phase_b_wide_lds(int B, int n_owner, int own_words) {
    return (size_t)3 * B * 20 + (size_t)4 * n_owner * (own_words + 1) + (kWideThreads / TKR_WAVE + 1) * 4 + 16 + (size_t)3 * B * 32;
}

// ---- K1c for the rows whose FIRST task of the call is this thread's: the row's touches of the whole call go into its update counter,
// its bitmap words back to zero.  (Every touched row has exactly one such task; csrc/sampler.hip commit_kernel walks ALL rows
// instead -- 5 MB at the ML-10M shape -- because it runs behind planners that did not keep the history.)  Call it only once every
// reader of the bitmap is done (the prologue's second barrier).
__device__ __forceinline__ void plan_commit_first_touch(const int4 t, int prev, int total, int32_t* ucnt, int32_t* icnt, uint32_t* touch_u,
                                                        uint32_t* touch_i) {
    if (t.x == -1 || prev != -1) return;
    const bool item = t.x < 0;
    const int row = t.x & 0x7fffffff;
    int32_t* cnt = item ? icnt + row : ucnt + row;
    atomicAdd(cnt, total);               // (no return value: nothing waits for the round trip)
    uint4* w = reinterpret_cast<uint4*>((item ? touch_i : touch_u) + (size_t)row * kTouchWords);
#pragma unroll
    for (int q = 0; q < kTouchWords / 4; ++q) w[q] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace tkr
