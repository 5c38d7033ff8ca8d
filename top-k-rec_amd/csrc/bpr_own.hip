// K2o -- the persistent BPR step with OWNED item rows (round 4): the same arithmetic, tables, versions and acknowledge protocol as
// K2f (csrc/bpr_flow.hip; sess.run([solver, obj]) of single/bpr.py:141 inside the loop of single/bpr.py:139-147, batch t+1 reads
// what batch t wrote), but the chain that set K2f's pace is taken out of memory.
//
// What bounded K2f at batch 256: a popular item is updated in (nearly) every batch, and the task of batch t+1 can only start its
// arithmetic once the row written by the task of batch t has made the trip write-through store -> memory -> polling load of another
// CU: 1.65-1.9 us per link against 1.5 us per batch of task throughput.  Here every item row r has an OWNER: the
// workgroup with arrival number r % n_owner (one workgroup per CU, all resident).  K1 lays the item tasks of a batch out in
// (owner, row) order and names every owner's run (tkr_sample_plan_owned: `ohdr`), so
//   * the owner's waves take the item tasks of their rows in plan order from a queue in LDS (one LDS atomic per task instead of a
//     device-wide ticket),
//   * the row, its RMSProp slot, bias and acknowledge totals live in the owner's LDS from their first update of a launch on:
//     the task of batch t+1 finds what the task of batch t left there ~0.1 us after it was computed -- BEFORE the acknowledge wait and
//     the write-through stores of batch t, which now only serve the row's PARTNERS (user tasks and the other item of a triplet read
//     the granule tables exactly as in K2f),
//   * user tasks are handed out by tickets as before, to the remaining waves of every workgroup.
// The granule tables are written through at every update exactly as by K2f, so a launch leaves them complete: both kernels, the
// exchange (csrc/sync.hip) and get / set work on the same state, and a chunk may be cut into launches anywhere (a task whose row
// was not yet updated in THIS launch -- prec[5] < first batch of the launch -- loads it from the tables like K2f does).
//
// Progress: every producer of a task sits in an earlier batch.  An owner queue is taken in plan order by its own waves only (at
// least one per workgroup serves nothing else), tickets in plan order by the ticket waves; a wave holds at most its current and its
// next task.  The lowest unfinished task is therefore always held by a wave, or next in line for one that holds only lower tasks,
// and waits on nothing unfinished.  Needs every workgroup resident (grid = n_owner <= CUs x workgroups per CU); every spin is bounded.
#include "flow_task.h"

namespace tkr {

constexpr uint32_t kOwnInvalid = 0xffffffffu;

struct OwnQueue {                        // head of the workgroup's LDS block
    uint32_t head;                       // next position of the owner queue
    uint32_t total;                      // item tasks of this owner in the launch
    uint32_t arrival;
    uint32_t pad;
};

// the owner queue: position -> (batch, slot) through the prefix sums of the per-batch run lengths
struct OwnerSrc {
    OwnQueue* q;
    const uint32_t* pre;                 // [nb + 1 + 64]: pre[0] = 0, pre[b + 1] = tasks up to and including batch b; padding 0xffffffff
    const uint32_t* start;               // [nb]: first slot of the owner's run in batch b
    const int4* __restrict__ prec;       // record 0 of the launch's first batch
    uint32_t slots_per_batch;            // 3B
    uint32_t cur;                        // batch of the last task taken (positions only grow)
    __device__ __forceinline__ void prefetch(NextTask& nx, int lane) {
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&q->head, 1u);
        pos = (uint32_t)bcast_i((int)pos, 0);
        nx.have = true;
        nx.w = make_int4(0, 0, 0, 0);
        if (pos >= q->total) { nx.idx = 0xffffffffu; return; }
        uint32_t before;
        for (;;) {
            const uint32_t v = pre[cur + 1 + lane];
            const unsigned long long m = __ballot(v > pos);
            if (m) {
                const int f = __ffsll((long long)m) - 1;
                before = f ? (uint32_t)bcast_i((int)v, f - 1) : pre[cur];
                cur += f;
                break;
            }
            cur += TKR_WAVE;
        }
        nx.idx = cur * slots_per_batch + start[cur] + (pos - before);
        if (lane < 8) nx.w = prec[(size_t)nx.idx * 8 + lane];
    }
};

// user tasks: tickets over the slots [0, B) of every batch (a batch's user tasks come first; what else sits there is skipped)
struct UserTicketSrc {
    uint32_t ticket;
    int home, queues;                    // queues = min(32, ticket waves of the grid): every queue has a wave
    uint32_t total;                      // nb * B
    uint32_t B;
    const int4* __restrict__ prec;       // record 0 of the launch's first batch
    __device__ __forceinline__ void prefetch(NextTask& nx, int lane) {
        const u64 i64 = (u64)(uint32_t)bcast_i((int)ticket, 0) * (uint32_t)queues + (uint32_t)home;
        const uint32_t idx = i64 < total ? (uint32_t)i64 : 0xffffffffu;
        nx.have = true;
        nx.w = make_int4(0, 0, 0, 0);
        if (idx == 0xffffffffu) { nx.idx = idx; return; }
        nx.idx = idx + 2u * B * (idx / B);
        if (lane < 8) nx.w = prec[(size_t)nx.idx * 8 + lane];
    }
};

// the own row of an item task: from the owner's LDS when an earlier task of this launch left it there, else from the tables
template <int NP>
struct LdsOwn {
    const FlowTables& T;
    int lane;
    const u64 *own_p, *own_ms, *own_tail;
    const uint32_t* own_rd;
    uint32_t ver;
    bool sgd;
    uint32_t* ctl;
    uint32_t& spins;
    bool from_lds;
    const volatile uint32_t* tag;        // the row's version word in LDS
    const float* row;                    // [kp] values, [kp] slots, {bias, its slot, expect[0..3], -, -}
    int own_halves;
    __device__ __forceinline__ bool operator()(float (&own)[2 * NP], float (&ms)[2 * NP], Own& o) {
#ifdef TKR_OWN_PROF
        o.t_own0 = __builtin_amdgcn_s_memtime();
#endif
        if (from_lds) {
            uint32_t waited = 0;
            while (*tag != ver) {
                if (spin_fail(waited, ctl, 0)) {
                    if (waited >= kSpinLimit && lane == 0 && atomicCAS(ctl + kCtlDebug, 0u, 4u) == 0u) {
                        ctl[kCtlDebug + 1] = *tag; ctl[kCtlDebug + 3] = ver;
                    }
                    return false;
                }
            }
            spins += waited;
            asm volatile("" ::: "memory");                  // the row is read AFTER its tag
            constexpr int KP = NP * 128;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const float2 a = *reinterpret_cast<const float2*>(row + q * 128 + 2 * lane);
                own[2 * q] = a.x; own[2 * q + 1] = a.y;
                if (!sgd) {
                    const float2 m = *reinterpret_cast<const float2*>(row + KP + q * 128 + 2 * lane);
                    ms[2 * q] = m.x; ms[2 * q + 1] = m.y;
                }
            }
            const float4 t = *reinterpret_cast<const float4*>(row + 2 * KP);
            const float2 t2 = *reinterpret_cast<const float2*>(row + 2 * KP + 4);
            o.b = t.x; o.msb = t.y;
            o.exp[0] = __float_as_uint(t.z); o.exp[1] = __float_as_uint(t.w);
            o.exp[2] = __float_as_uint(t2.x); o.exp[3] = __float_as_uint(t2.y);
        }
        const bool ok = flow_own<NP>(T, lane, own_p, own_ms, own_tail, own_rd, ver, own, ms, o, sgd, ctl, spins, own_halves);     // o.ok: only the acknowledge word is loaded
#ifdef TKR_OWN_PROF
        o.t_own1 = __builtin_amdgcn_s_memtime();
#endif
        return ok;
    }
};

// the new row from the old one and the gradient (TF SparseApplyRMSProp, momentum 0: single/bpr.py:100; or old/methods/bpr.py:57-61)
template <int NP>
__device__ __forceinline__ void own_update(const tkr_flow_state& st, bool sgd, const float (&own)[2 * NP], const float (&ms)[2 * NP],
                                           const Own& o, const float (&g)[2 * NP], float gb, float (&pn)[2 * NP], float (&mn)[2 * NP],
                                           float& bn, float& mbn) {
    constexpr int NE = 2 * NP;
    if (sgd) {
#pragma unroll
        for (int e = 0; e < NE; ++e) { pn[e] = own[e] - st.lr * g[e]; mn[e] = 0.f; }
        bn = o.b - st.lr * gb;
        mbn = o.msb;
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            mn[e] = st.rho * ms[e] + (1.f - st.rho) * g[e] * g[e];
            pn[e] = own[e] - st.lr * g[e] * __builtin_amdgcn_rsqf(mn[e] + st.eps);
        }
        mbn = st.rho * o.msb + (1.f - st.rho) * gb * gb;
        bn = o.b - st.lr * gb * __builtin_amdgcn_rsqf(mbn + st.eps);
    }
}

// version ver+1 lands on the buffer that held ver-1: wait until every reader of ver-1 has acknowledged, then write through
template <int NP>
__device__ __forceinline__ bool own_publish(int lane, bool is_item, uint32_t bmask, bool sgd, u64* tabP, u64* tabM, u64* tabT, size_t woff,
                                            size_t n_rows, int row, int rowk, uint32_t ver, int n_occ, Own& o, const uint32_t* own_rd,
                                            const float (&pn)[2 * NP],
                                            const float (&mn)[2 * NP], float bn, float mbn, uint32_t* ctl, uint32_t& spins, NextTask& nx,
                                            bool skip_ack = false) {
    const uint32_t nv = ver + 1u;
    const uint32_t expect = pick_exp(o, (ver + 1u) & bmask);            // readers of the version that buffer holds now
#ifdef TKR_OWN_PROF
    if (skip_ack) o.rd = expect;                // timing experiment only (unsafe): no acknowledge wait
#endif
    uint32_t waited = 0;
    while ((int32_t)(o.rd - expect) < 0) {
        if (spin_fail(waited, ctl)) {
            if (waited >= kSpinLimit && lane == 0 && atomicCAS(ctl + kCtlDebug, 0u, 2u) == 0u) {
                ctl[kCtlDebug + 1] = o.rd; ctl[kCtlDebug + 2] = expect; ctl[kCtlDebug + 3] = ver; ctl[kCtlDebug + 4] = (uint32_t)rowk;
            }
            return false;
        }
        o.rd = ld_u32(own_rd);
    }
    spins += waited;
#ifdef TKR_OWN_PROF
    o.t_ack = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("" : "+v"(nx.w.x), "+v"(nx.w.y), "+v"(nx.w.z), "+v"(nx.w.w));      // the next record is consumed before the stores go out
    store_row<NP>(tabP + woff, lane, pn, nv);
    if (!sgd) store_row<NP>(tabM + woff, lane, mn, nv);
    store_tail(tabT, n_rows, row, bmask, lane, is_item, bn, mbn, o, ver, n_occ);
    return true;
}

constexpr int own_min_waves(int np, int tpb) { return tpb >= 512 ? 2 : (np == 1 ? 2 : 1); }      // per SIMD: <= 256 registers at 8 waves per CU

template <int NP, int TPB>
__global__ __launch_bounds__(TPB, own_min_waves(NP, TPB)) void bpr_own_kernel(
    tkr_flow_state st, const int4* __restrict__ prec /*record 0 of the first batch to run*/, const int4* __restrict__ pocc,
    const int32_t* __restrict__ ohdr /*[n_owner][ohdr_stride], at the first batch to run*/, int ohdr_stride, int first_batch, int nb,
    int B, int n_owner, int owner_waves, uint32_t tune, uint32_t* __restrict__ ctl, float* __restrict__ loss_out) {
    constexpr int NE = 2 * NP;
    constexpr int KP = NP * 128;
    constexpr int ROWF = 2 * KP + 8;                                  // floats per resident row
    const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x / TKR_WAVE;
    FlowTables T;
    T.tune = 0u;
    T.kp = KP;
    T.ustride = (size_t)st.n_users * T.kp;
    T.istride = (size_t)st.n_items * T.kp;
    T.imask = st.item_bufs == 4 ? 3u : 1u;
    const int item_halves = tail_halves(T.imask);
    T.U = reinterpret_cast<u64*>(st.U); T.msU = reinterpret_cast<u64*>(st.msU); T.tailU = reinterpret_cast<u64*>(st.tailU);
    T.V = reinterpret_cast<u64*>(st.V); T.msV = reinterpret_cast<u64*>(st.msV); T.tailV = reinterpret_cast<u64*>(st.tailV);
    T.rdU = st.rdU; T.rdV = st.rdV;
    const bool sgd = st.opt == 1;
    const bool want_loss = loss_out != nullptr;

    // ---- the workgroup's LDS: queue head | prefix sums | run starts | row tags | rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    OwnQueue* q = reinterpret_cast<OwnQueue*>(smem);
    uint32_t* pre = reinterpret_cast<uint32_t*>(smem + sizeof(OwnQueue));           // [nb + 1 + 64]
    uint32_t* start = pre + nb + 1 + TKR_WAVE;                                         // [nb]
    const int rows_here = (st.n_items + n_owner - 1) / n_owner;
    uint32_t* tags = start + nb;                                                       // [rows_here]
    float* rows = reinterpret_cast<float*>(smem + ((sizeof(OwnQueue) + (size_t)4 * (2 * nb + 1 + TKR_WAVE + rows_here) + 15) & ~(size_t)15));

    if (threadIdx.x == 0) q->arrival = __hip_atomic_fetch_add(ctl + kCtlArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t me = q->arrival;                                                    // owner number: arrival order, whatever the placement
    for (int b = threadIdx.x; b < nb; b += TPB) {
        const uint32_t h = me < (uint32_t)n_owner ? (uint32_t)ohdr[(size_t)me * ohdr_stride + b] : 0u;
        start[b] = h & 0xffffu;
        pre[b + 1] = h >> 16;
    }
    for (int b = threadIdx.x; b < TKR_WAVE; b += TPB) pre[nb + 1 + b] = 0xffffffffu;
    for (int s = threadIdx.x; s < rows_here; s += TPB) tags[s] = kOwnInvalid;
    __syncthreads();
    if (wave == 0) {                                                                   // inclusive scan of the run lengths (nb <= 512)
        const int per = (nb + TKR_WAVE - 1) / TKR_WAVE;
        const int b0 = min(lane * per, nb), b1 = min(b0 + per, nb);
        uint32_t mine = 0;
        for (int b = b0; b < b1; ++b) mine += pre[b + 1];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < TKR_WAVE; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += up;
        }
        uint32_t run = incl - mine;
        for (int b = b0; b < b1; ++b) { run += pre[b + 1]; pre[b + 1] = run; }
        if (lane == TKR_WAVE - 1) { q->total = incl; q->head = 0u; }
        if (lane == 0) pre[0] = 0u;
    }
    __syncthreads();

    uint32_t spins = 0;
    bool alive = true;
#ifdef TKR_OWN_PROF
    u64 prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 tprev = __builtin_amdgcn_s_memtime();
#endif
    NextTask nx;
    nx.idx = 0u; nx.w = make_int4(0, 0, 0, 0); nx.have = false;

    if (wave < owner_waves) {
        // ================= item tasks of the rows this workgroup owns =================
        OwnerSrc feed{q, pre, start, prec, 3u * (uint32_t)B, 0u};
        feed.prefetch(nx, lane);
        while (alive) {
            const uint32_t idx = nx.idx;
            const int4 w = nx.w;
            if (idx == 0xffffffffu) break;
            nx.have = false;
            const int rowk = bcast_i(w.x, 0);
#ifdef TKR_OWN_PROF
            const u64 t0 = __builtin_amdgcn_s_memtime();
#endif
            const uint32_t ver = (uint32_t)bcast_i(w.y, 0);
            const int n_occ = bcast_i(w.z, 0);
            const int first = bcast_i(w.w, 0);
            const int prev = bcast_i(w.y, 1);
            const int row = rowk & 0x7fffffff;
            const int slot = row / n_owner;
            const bool from_lds = prev >= first_batch;                 // an earlier task of THIS launch updated the row: it is (or will be) in LDS

            const size_t roff = (size_t)(ver & T.imask) * T.istride + (size_t)row * T.kp;
            const size_t woff = (size_t)((ver + 1u) & T.imask) * T.istride + (size_t)row * T.kp;
            const uint32_t* own_rd = T.rdV + (T.imask + 1u) * (size_t)row + ((ver + 1u) & T.imask);
            const u64* own_tail = T.tailV + ((size_t)(ver & T.imask) * st.n_items + row) * (2 * item_halves);
            float* lrow = rows + (size_t)slot * ROWF;

            float own[NE], ms[NE], g[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { g[e] = 0.f; ms[e] = 0.f; }
            Own o = {};
            o.ok = from_lds;                                            // flow_fetch then leaves the own row alone
            const bool lazy = (tune & 1u) != 0u;                        // experiment: the next task is only taken once this one is done
            if (lazy) nx.have = true;
            float gb = 0.f, loss_lane = 0.f;
            LdsOwn<NP> own_step{T, lane, T.V + roff, T.msV + roff, own_tail, own_rd, ver, sgd, ctl, spins, from_lds, tags + slot, lrow, item_halves};
            alive = run_task<NP, true>(st, T, lane, n_occ, first, w, pocc, T.V + roff, T.msV + roff, own_tail, ver, own, ms, o, g, gb,
                                       loss_lane, false, sgd, ctl, spins, nx, feed, own_step);
            if (!alive) break;
#ifdef TKR_OWN_PROF
            const u64 t1 = __builtin_amdgcn_s_memtime();
#endif
            float pn[NE], mn[NE], bn, mbn;
            own_update<NP>(st, sgd, own, ms, o, g, gb, pn, mn, bn, mbn);

            // the row's next task finds the new version HERE, now -- not behind the acknowledge wait and the trip through memory
#pragma unroll
            for (int qq = 0; qq < NP; ++qq) {
                *reinterpret_cast<float2*>(lrow + qq * 128 + 2 * lane) = make_float2(pn[2 * qq], pn[2 * qq + 1]);
                if (!sgd) *reinterpret_cast<float2*>(lrow + KP + qq * 128 + 2 * lane) = make_float2(mn[2 * qq], mn[2 * qq + 1]);
            }
            if (lane == 0) {
                const uint32_t rb = ver & T.imask, add = 2u * (uint32_t)n_occ;
                *reinterpret_cast<float4*>(lrow + 2 * KP) = make_float4(bn, mbn, __uint_as_float(o.exp[0] + (rb == 0u ? add : 0u)),
                                                                        __uint_as_float(o.exp[1] + (rb == 1u ? add : 0u)));
                *reinterpret_cast<float2*>(lrow + 2 * KP + 4) =
                    make_float2(__uint_as_float(o.exp[2] + (rb == 2u ? add : 0u)), __uint_as_float(o.exp[3] + (rb == 3u ? add : 0u)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the row is in LDS before its tag says so
            if (lane == 0) *reinterpret_cast<volatile uint32_t*>(tags + slot) = ver + 1u;

            alive = own_publish<NP>(lane, true, T.imask, sgd, T.V, T.msV, T.tailV, woff, (size_t)st.n_items, row, rowk, ver, n_occ, o, own_rd, pn, mn,
                                    bn, mbn, ctl, spins, nx, (tune & 2u) != 0u);
#ifdef TKR_OWN_PROF
            const u64 t4 = __builtin_amdgcn_s_memtime();
#endif
            if (lazy && alive) feed.prefetch(nx, lane);
#ifdef TKR_OWN_PROF
            {
                asm volatile("" : "+v"(nx.w.x) :: "memory");
                const u64 t5 = __builtin_amdgcn_s_memtime();
                prof[0] += o.t_own0 - t0; prof[1] += o.t_own1 - o.t_own0; prof[2] += t1 - o.t_own1; prof[3] += o.t_ack - t1;
                prof[4] += t4 - o.t_ack; prof[5] += t5 - t4; prof[6] += 1; prof[7] += from_lds ? 1 : 0;
            }
#endif
        }
    } else {
        // ================= user tasks, by ticket =================
        const int tw = wave - owner_waves, n_tw = TPB / TKR_WAVE - owner_waves;
        const int queues = min(kQueues, n_tw * (int)gridDim.x);
        const int home = (int)((me * (uint32_t)n_tw + (uint32_t)tw) % (uint32_t)queues);
        const uint32_t total = (uint32_t)nb * (uint32_t)B;
        uint32_t ticket = grab_issue(ctl, lane, home);
        while (alive) {
            if (!nx.have) {
                UserTicketSrc first_feed{ticket, home, queues, total, (uint32_t)B, prec};
                first_feed.prefetch(nx, lane);
                asm volatile("" : "+v"(nx.w.x), "+v"(nx.w.y), "+v"(nx.w.z), "+v"(nx.w.w) :: "memory");
            }
            const uint32_t idx = nx.idx;
            const int4 w = nx.w;
            if (idx == 0xffffffffu) break;
            nx.have = false;
            ticket = grab_issue(ctl, lane, home);                      // the ticket of the task after this one
            const int rowk = bcast_i(w.x, 0);
#ifdef TKR_OWN_PROF
            const u64 t0 = __builtin_amdgcn_s_memtime();
            prof[8] += t0 - tprev;
#endif
            if (rowk < 0) {                                             // an item task (its owner runs it) or an unused slot
#ifdef TKR_OWN_PROF
                tprev = t0;
#endif
                continue;
            }
            const uint32_t ver = (uint32_t)bcast_i(w.y, 0);
            const int n_occ = bcast_i(w.z, 0);
            const int first = bcast_i(w.w, 0);
            const int batch = bcast_i(w.x, 1);
            const int row = rowk;

            const size_t roff = (size_t)(ver & 1u) * T.ustride + (size_t)row * T.kp;
            const size_t woff = (size_t)((ver + 1u) & 1u) * T.ustride + (size_t)row * T.kp;
            const uint32_t* own_rd = T.rdU + 2 * (size_t)row + ((ver + 1u) & 1u);
            const u64* own_tail = T.tailU + ((size_t)(ver & 1u) * st.n_users + row) * 4;

            float own[NE], ms[NE], g[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { g[e] = 0.f; ms[e] = 0.f; }
            Own o = {};
            float gb = 0.f, loss_lane = 0.f;
            UserTicketSrc feed{ticket, home, queues, total, (uint32_t)B, prec};
            GlobalOwn<NP> own_step{T, lane, T.U + roff, T.msU + roff, own_tail, own_rd, ver, sgd, ctl, spins, 2};
            alive = run_task<NP, false>(st, T, lane, n_occ, first, w, pocc, T.U + roff, T.msU + roff, own_tail, ver, own, ms, o, g, gb,
                                        loss_lane, want_loss, sgd, ctl, spins, nx, feed, own_step);
            if (!alive) break;
#ifdef TKR_OWN_PROF
            const u64 t1 = __builtin_amdgcn_s_memtime();
#endif
            if (want_loss) {
                const float tot = wave_sum(loss_lane);
                if (lane == 0) atomicAdd(loss_out + batch, tot);
            }
            float pn[NE], mn[NE], bn, mbn;
            own_update<NP>(st, sgd, own, ms, o, g, gb, pn, mn, bn, mbn);
            alive = own_publish<NP>(lane, false, 1u, sgd, T.U, T.msU, T.tailU, woff, (size_t)st.n_users, row, rowk, ver, n_occ, o, own_rd, pn, mn,
                                    bn, mbn, ctl, spins, nx);
#ifdef TKR_OWN_PROF
            tprev = __builtin_amdgcn_s_memtime();
            prof[9] += t1 - t0; prof[11] += o.t_ack - t1; prof[12] += tprev - o.t_ack; prof[13] += 1;
#endif
        }
    }

    if (lane == 0 && spins) atomicAdd(ctl + kCtlSpins, spins);
#ifdef TKR_OWN_PROF
    if (lane == 0)
        for (int qq = 0; qq < 16; ++qq) atomicAdd(reinterpret_cast<u64*>(ctl + kCtlProf) + qq, prof[qq]);
#endif
    // the last workgroup out puts the ticket words back to zero (as K2f: the next launch needs no memset)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(ctl + kCtlLeave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
        for (int qq = 0; qq < kQueues; ++qq) ctl[qq * kQueueStride] = 0u;
        ctl[kCtlArrive] = 0u;
        ctl[kCtlLeave] = 0u;
    }
}

static size_t own_lds_bytes(int np, int nb, int n_items, int n_owner) {
    const int rows_here = (n_items + n_owner - 1) / n_owner;
    const size_t head = (sizeof(OwnQueue) + (size_t)4 * (2 * nb + 1 + TKR_WAVE + rows_here) + 15) & ~(size_t)15;
    return head + (size_t)rows_here * (2 * np * 128 + 8) * 4;
}

}  // namespace tkr

// workgroups (= owners) a device runs at once for factor width k, or 0 when the item rows do not fit their owners' LDS
extern "C" int32_t tkr_bpr_own_owners(int32_t n_items, int32_t k) {
    if (n_items <= 0 || k <= 0 || k > 256) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    const int np = (k + 127) / 128;
    if (tkr::own_lds_bytes(np, 512, n_items, cus) > 160 * 1024) return 0;
    return cus;
}

extern "C" int tkr_bpr_own_run(const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc, const int32_t* ohdr,
                               int32_t ohdr_stride, int32_t n_owner, int32_t batch_size, int32_t first_batch, int32_t n_batches,
                               uint32_t* ctl, float* loss_out, int32_t owner_waves, void* stream) {
    if (!st || !st->U || !st->V || !st->tailU || !st->tailV || !st->rdU || !st->rdV) return TKR_EINVAL;
    if (st->opt != 0 && st->opt != 1) return TKR_EINVAL;
    if (st->opt == 0 && (!st->msU || !st->msV)) return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->k <= 0) return TKR_EINVAL;
    if (st->k > 256) return TKR_EUNSUPPORTED;
    if (st->item_bufs != 0 && st->item_bufs != 2 && st->item_bufs != 4) return TKR_EINVAL;
    if (!prec || !pocc || !ohdr || !ctl || batch_size <= 0 || n_batches < 0 || first_batch < 0 || n_owner <= 0) return TKR_EINVAL;
    if (ohdr_stride < first_batch + n_batches || n_batches > 512) return TKR_EINVAL;
    if (n_batches == 0) return TKR_OK;
    if ((uint64_t)(first_batch + n_batches) * 3u * (uint64_t)batch_size >= 0xffffffffull / 8) return TKR_EUNSUPPORTED;
    const int np = (st->k + 127) / 128;
    int dev = 0;
    TKR_CHECK(hipGetDevice(&dev));
    static int cached_cus[64];
    static bool attr_set[64][2];
    int cus;
    if (dev >= 0 && dev < 64 && cached_cus[dev] > 0) cus = cached_cus[dev];
    else {
        TKR_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (dev >= 0 && dev < 64) cached_cus[dev] = cus;
    }
    if (n_owner > cus) return TKR_EUNSUPPORTED;                    // every owner must be resident: one workgroup per CU
    const size_t lds = tkr::own_lds_bytes(np, n_batches, st->n_items, n_owner);
    if (lds > 160 * 1024) return TKR_EUNSUPPORTED;
    const int tpb = np == 1 ? 512 : 256;
    const int waves = tpb / TKR_WAVE;
    // default split: item tasks are ~64 % of a batch's tasks at the ML-10M shape and cheaper than user tasks (no own-row trip)
    const uint32_t tune = ((uint32_t)owner_waves >> 8) & 0xffu;     // experiment switches ride in bits 8..15
    owner_waves &= 0xff;
    int ow = owner_waves > 0 ? owner_waves : (waves == 8 ? 5 : 3);
    if (ow < 1) ow = 1;
    if (ow > waves - 1) ow = waves - 1;
    const void* fn = np == 1 ? (const void*)tkr::bpr_own_kernel<1, 512> : (const void*)tkr::bpr_own_kernel<2, 256>;
    if (lds > 64 * 1024 && !(dev >= 0 && dev < 64 && attr_set[dev][np - 1])) {
        TKR_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) attr_set[dev][np - 1] = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const int4* r4 = reinterpret_cast<const int4*>(prec) + (size_t)first_batch * 3 * batch_size * 8;      // record 0 of the first batch to run
    const int4* o4 = reinterpret_cast<const int4*>(pocc);
    ohdr += first_batch;
    if (np == 1)
        hipLaunchKernelGGL((tkr::bpr_own_kernel<1, 512>), dim3(n_owner), dim3(512), lds, s, *st, r4, o4, ohdr, ohdr_stride, first_batch, n_batches,
                           batch_size, n_owner, ow, tune, ctl, loss_out);
    else
        hipLaunchKernelGGL((tkr::bpr_own_kernel<2, 256>), dim3(n_owner), dim3(256), lds, s, *st, r4, o4, ohdr, ohdr_stride, first_batch, n_batches,
                           batch_size, n_owner, ow, tune, ctl, loss_out);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}
