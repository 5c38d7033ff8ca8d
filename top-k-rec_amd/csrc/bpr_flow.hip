// K2f -- the BPR step as ONE persistent launch per chunk of mini-batches (dataflow form of csrc/bpr_step.hip).
//
// Same arithmetic and the same semantics as K2 (sess.run([solver, obj]) of single/bpr.py:141 on the graph of
// single/bpr.py:81-100; batch t+1 reads what batch t wrote, bpr.py:139-147), but the order between batches is no
// longer a kernel boundary (1.5 us of dependent-launch gap + two cold memory levels per 256-triplet batch = 96 % idle,
// profiles/README.md): it is carried by the data.
//
//   * Every table element is an 8-byte GRANULE {fp32 value, uint32 tag}; the tag is the VERSION of its row = the number
//     of updates the row has seen.  Granules are written and read as aligned 16-byte PAIRS by write-through stores and
//     L1-bypassing loads (buffer_store/load_dwordx4 sc1); every aligned 8-byte half arrives whole, so a reader that sees
//     tag == v holds a value of version v: the data is its own flag, no fence and no separate ready word
//     (MI355X_MICROARCH.md, R2 form of the hand-off recipe; per-XCD L2s are not coherent, L1s are never refreshed: plain
//     loads of another workgroup's stores would be stale).
//   * K1 (csrc/sampler.hip resolve_flow_kernel) names, for every task, the exact version of its own row and of every
//     partner row.  A wave takes a task, loads own row + slot + partner rows and re-loads what does not carry the
//     wanted tags yet; rows that the preceding batches did not touch are ready at once, so batch t+1 streams its cold
//     rows while batch t is still computing, and only the chains through rows updated in consecutive batches wait --
//     each link for one store-to-load hand-off instead of a kernel boundary.
//   * Tables stay double-buffered: version v of a row lives in buffer v & 1, so a partner that wants version v still
//     finds it after the row's own task has published v+1.  What must not happen is v+2 landing on v while somebody
//     still reads v.  Readers of a row are exactly the tasks of the triplets it occurs in (2 per occurrence).  Per row
//     and per PARITY of the version read: the row's own task carries the running total expect[p] = 2 x occurrences in the
//     batches that read a version of parity p (tail granules 2, 3), every partner read of version v is acknowledged by
//     an atomic add on the row's rd[v & 1] word, and the task that turns v into v+1 (landing on the buffer of v-1)
//     stores only once rd[(v+1) & 1] >= expect[(v+1) & 1] as carried by version v -- normally long true: the readers of
//     batch t finished while batch t+1 was still loading.  (One counter for both parities would not do: early reads of
//     v would stand in for a straggling read of v-1.)
//   * Tasks are handed out in plan order (batch-major) by 32 ticket counters (one word saturates at ~90 tickets/us,
//     a 256-batch needs ~500/us): task index = 32 * ticket + queue.  A wave's queue is its ARRIVAL number & 31 (an
//     atomic at start, not its block index), so the first 32 waves that actually run cover every queue whatever the
//     dispatch order, placement or residency, and the queues advance at the same rate.  Every producer of a task sits in
//     an earlier batch, i.e. holds a lower task index: the lowest untaken task is the next one its queue's waves take, and
//     the lowest unfinished task never waits on anything unfinished -- no deadlock.  Every spin is bounded (status word).
//
// Run-to-run results are bitwise identical (each task reads exact versions; sums run in plan order); the only float
// atomic is the reported loss.  Against K2 the sums differ in the last bits (lane -> element mapping, heavy rows are
// summed by one wave instead of a team): both are held to the same tolerance against the oracle.
//
// Roofline: the same algorithmic bytes as K2 (48k + 56 per triplet); granules double the bytes that
// really move, which is irrelevant where this kernel is used (B <= 1024: latency-bound) -- large batches keep the
// plain tables and K2.
#include "flow_task.h"

namespace tkr {

#ifdef TKR_FLOW_TRACE
// timing builds only (scripts/probe_flow_timeline.py; a stamp that is stored at once waits ~0.2 us for s_memrealtime, so the segments
// are upper bounds and a build with one more stamp runs slower): per batch, the 100 MHz time stamps of the task of ITEM 0 --
// [0] record in hand, [1] rows valid + gradients done, [2] readers acknowledged, [3] stores issued
__device__ unsigned long long* g_flow_trace;
#define TKR_TRACE(i) if (trace_buf && is_item && row == 0 && lane == 0) trace_buf[(size_t)batch * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define TKR_TRACE(i)
#endif

template <int NP, bool PROF = false>
__global__ __launch_bounds__(256, (NP == 1 ? 2 : 1)) void bpr_flow_kernel(tkr_flow_state st, const int4* __restrict__ prec,
                                                       const int4* __restrict__ pocc, uint32_t total,
                                                       uint32_t* __restrict__ ctl, float* __restrict__ loss_out, uint32_t tune,
                                                       uint32_t per_batch /*records of a batch: 3 * batch_size*/) {
    constexpr int NE = 2 * NP;
    const int lane = threadIdx.x & (TKR_WAVE - 1);
    FlowTables T;
    T.tune = tune;
    T.kp = NP * 128;
    T.ustride = (size_t)st.n_users * T.kp;
    T.istride = (size_t)st.n_items * T.kp;
    T.imask = st.item_bufs == 4 ? 3u : 1u;
    T.U = reinterpret_cast<u64*>(st.U); T.msU = reinterpret_cast<u64*>(st.msU); T.tailU = reinterpret_cast<u64*>(st.tailU);
    T.V = reinterpret_cast<u64*>(st.V); T.msV = reinterpret_cast<u64*>(st.msV); T.tailV = reinterpret_cast<u64*>(st.tailV);
    T.rdU = st.rdU; T.rdV = st.rdV;
    const bool sgd = st.opt == 1;
    const bool want_loss = loss_out != nullptr;
#ifdef TKR_FLOW_TRACE
    unsigned long long* const trace_buf = g_flow_trace;           // read once: a load per mark would drain the memory pipe before the stamp
#endif

    // queue of this wave = (arrival number of its workgroup * 4 + wave) & 31: one atomic per workgroup (a word takes ~12 ns
    // per atomic: per wave that was 12 us of start-up), and still independent of block index, placement and residency
    __shared__ uint32_t wg_arrival;
    // A batch's loss is the sum over its ~250 user tasks.  As atomics on loss_out[batch] -- one word, one memory channel -- they cost
    // 0.6 us of a 2.8 us batch: each is a memory operation the next task's first load waits behind.  The workgroup adds its tasks'
    // sums up in LDS (ds_add_f32: not a memory operation) and hands loss_out ONE atomic per batch it met, on its way out.
    constexpr int kLossSlots = 512;
    __shared__ float wg_loss[kLossSlots];
    __shared__ int wg_loss_base;                                    // index in loss_out of the first batch of this launch
    if (threadIdx.x == 0) wg_arrival = __hip_atomic_fetch_add(ctl + kCtlArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) wg_loss_base = -1;
    for (int q = threadIdx.x; q < kLossSlots; q += blockDim.x) wg_loss[q] = 0.f;
    __syncthreads();
    const int home = (int)((wg_arrival * (blockDim.x / TKR_WAVE) + (threadIdx.x / TKR_WAVE)) & (kQueues - 1));
    uint32_t spins = 0;
    bool alive = true;

    u64 prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 tq = 0;
#define TKR_PROF_MARK(slot)                                   \
    if constexpr (PROF) {                                     \
        const u64 now_ = __builtin_amdgcn_s_memtime();        \
        prof[slot] += now_ - tq;                              \
        tq = now_;                                            \
    }
    if constexpr (PROF) tq = __builtin_amdgcn_s_memtime();
    // The ticket of the NEXT task is taken while the current one runs (its round trip is off the wave's cycle).  Safe: a
    // wave's next ticket is higher than its current one and a task only ever waits for lower ones.
    uint32_t ticket = grab_issue(ctl, lane, home);
    NextTask nx;
    nx.idx = 0u; nx.w = make_int4(0, 0, 0, 0); nx.have = false;
    while (alive) {
        uint32_t idx;
        int4 w;
        if (nx.have) {                                                      // fetched while the previous task ran (flow_group)
            idx = nx.idx;
            w = nx.w;
            TKR_PROF_MARK(0)
            if (idx == 0xffffffffu) break;
        } else {                                                            // first task of the wave, or the previous slot was unused
            idx = grab_index(ticket, home, total);
            TKR_PROF_MARK(0)
            if (idx == 0xffffffffu) break;
            const int4* r = prec + (size_t)idx * 8;
            w = make_int4(0, 0, 0, 0);
            if (lane < 8) w = r[lane];                                      // 128-byte record, one int4 per lane (an L2 hit: K1 just wrote it)
            asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w) :: "memory");   // the record lands BEFORE the atomic is issued: memory
        }                                                                   // returns in order, and the record must not queue behind its round trip
        nx.have = false;
        ticket = grab_issue(ctl, lane, home);                               // the ticket of the task after this one
        const int rowk = bcast_i(w.x, 0);
        TKR_PROF_MARK(1)
        if (rowk == -1) { if constexpr (PROF) prof[6] += 1; continue; }    // unused slot of its batch
        const uint32_t ver = (uint32_t)bcast_i(w.y, 0);
        const int n_occ = bcast_i(w.z, 0);
        const int first = bcast_i(w.w, 0);
        const int batch = bcast_i(w.x, 1);
        const bool is_item = rowk < 0;
        const int row = rowk & 0x7fffffff;

        TKR_TRACE(0)
        const size_t n_rows = is_item ? st.n_items : st.n_users;
        const uint32_t bmask = is_item ? T.imask : 1u;                         // buffers of the own row's table - 1
        const int own_halves = tail_halves(bmask);
        const size_t roff = (size_t)(ver & bmask) * (is_item ? T.istride : T.ustride) + (size_t)row * T.kp;
        const size_t woff = (size_t)((ver + 1u) & bmask) * (is_item ? T.istride : T.ustride) + (size_t)row * T.kp;
        u64* tabP = is_item ? T.V : T.U;
        u64* tabM = is_item ? T.msV : T.msU;
        u64* tabT = is_item ? T.tailV : T.tailU;
        const uint32_t* own_rd = (is_item ? T.rdV : T.rdU) + (bmask + 1u) * (size_t)row + ((ver + 1u) & bmask);

        float own[NE], ms[NE], g[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) { g[e] = 0.f; ms[e] = 0.f; }
        Own o = {};
        float gb = 0.f, loss_lane = 0.f, loss_x = 0.f;

        const u64* own_tail = tabT + ((size_t)(ver & bmask) * n_rows + row) * (2 * own_halves);
        TicketSrc src{ticket, home, total, prec};
        GlobalOwn<NP> own_step{T, lane, tabP + roff, tabM + roff, own_tail, own_rd, ver, sgd, ctl, spins, own_halves};
        alive = is_item ? run_task<NP, true>(st, T, lane, n_occ, first, w, pocc, tabP + roff, tabM + roff, own_tail, ver, own, ms, o,
                                             g, gb, loss_lane, false, sgd, ctl, spins, nx, src, own_step)
                        : run_task<NP, false>(st, T, lane, n_occ, first, w, pocc, tabP + roff, tabM + roff, own_tail, ver, own, ms,
                                              o, g, gb, loss_lane, want_loss, sgd, ctl, spins, nx, src, own_step);
        if (!alive) break;
        TKR_PROF_MARK(2)
        TKR_TRACE(1)
#ifdef TKR_FLOW_TRACE
        if (trace_buf && is_item && row == 0 && lane == 0) { trace_buf[(size_t)batch * 8 + 4] = o.t_valid; trace_buf[(size_t)batch * 8 + 5] = o.t_part; trace_buf[(size_t)batch * 8 + 6] = spins; }
#endif

        if (!is_item && want_loss) {
            const float tot = wave_sum(loss_lane) + loss_x;
            const uint32_t lb = idx / per_batch;                     // batch of this task, counted from the first of the launch
            if (lane == 0) {
                if (lb < (uint32_t)kLossSlots) {
                    atomicAdd(&wg_loss[lb], tot);
                    wg_loss_base = batch - (int)lb;                 // (every task writes the same number)
                } else {
                    loss_add(loss_out + batch, tot);
                }
            }
        }

        // the new row first (nothing in it depends on the acknowledgements): behind the wait only the stores are left
        const uint32_t nv = ver + 1u;
        float pn[NE], mn[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) mn[e] = 0.f;
        float bn, mbn = 0.f;
#ifdef TKR_FLOW_TRACE
        if (T.tune & 16u) {                         // experiment: the row goes back unchanged
#pragma unroll
            for (int e = 0; e < NE; ++e) { pn[e] = own[e]; mn[e] = ms[e]; }
            bn = o.b; mbn = o.msb;
        } else
#endif
        if (sgd) {                                  // old/methods/bpr.py:57-61: P <- P - lr * dcost/dP
#pragma unroll
            for (int e = 0; e < NE; ++e) pn[e] = own[e] - st.lr * g[e];
            bn = o.b - st.lr * gb;
            mbn = o.msb;
        } else {                                    // TF SparseApplyRMSProp, momentum 0 (single/bpr.py:100)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                mn[e] = st.rho * ms[e] + (1.f - st.rho) * g[e] * g[e];
                pn[e] = own[e] - st.lr * g[e] * __builtin_amdgcn_rsqf(mn[e] + st.eps);
            }
            mbn = st.rho * o.msb + (1.f - st.rho) * gb * gb;
            bn = o.b - st.lr * gb * __builtin_amdgcn_rsqf(mbn + st.eps);
        }

        // version ver+1 lands on the buffer that held ver-1: wait until every reader of ver-1 has acknowledged
        const uint32_t expect = pick_exp(o, (ver + 1u) & bmask);            // readers of the version that buffer holds now (ver-1, or ver-3 with four buffers)
        uint32_t waited = 0;
#ifdef TKR_FLOW_TRACE
        if (T.tune & 8u) o.rd = expect;             // experiment: no acknowledge wait
#endif
        while ((int32_t)(o.rd - expect) < 0) {
            if (spin_fail(waited, ctl)) {
                if (waited >= kSpinLimit && lane == 0 && atomicCAS(ctl + kCtlDebug, 0u, 2u) == 0u) {
                    ctl[kCtlDebug + 1] = o.rd; ctl[kCtlDebug + 2] = expect; ctl[kCtlDebug + 3] = ver; ctl[kCtlDebug + 4] = (uint32_t)rowk;
                }
                alive = false;
                break;
            }
            o.rd = ld_u32(own_rd);
        }
        spins += waited;
        if (!alive) break;
        if constexpr (PROF) { if (is_item) prof[7] += __builtin_amdgcn_s_memtime() - tq; }      // the item tasks' share of the acknowledge wait
        TKR_PROF_MARK(3)
        TKR_TRACE(2)

        // the next task's record is consumed BEFORE the stores go out: a wait behind them would include their write-through
        asm volatile("" : "+v"(nx.w.x), "+v"(nx.w.y), "+v"(nx.w.z), "+v"(nx.w.w));
        store_row<NP>(tabP + woff, lane, pn, nv);
        if (!sgd) store_row<NP>(tabM + woff, lane, mn, nv);
        store_tail(tabT, n_rows, row, bmask, lane, is_item, bn, mbn, o, ver, n_occ);
        TKR_TRACE(3)
        if constexpr (PROF) prof[5] += 1;
        TKR_PROF_MARK(4)
    }
#undef TKR_PROF_MARK
    if constexpr (PROF) {
        if (lane == 0)
            for (int q = 0; q < 8; ++q) atomicAdd(reinterpret_cast<u64*>(ctl + kCtlProf) + q, prof[q]);
    }

    if (lane == 0 && spins) atomicAdd(ctl + kCtlSpins, spins);

    // The last workgroup out puts the ticket words back to zero: the next launch starts from a clean ctl without a memset in
    // front of it (a launch of its own: ~8 us of a 20-batch call).  Every wave has its last ticket back by now -- it broke
    // out on the value (the vmcnt covers the give-up path, where the prefetched ticket is never looked at).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (want_loss && wg_loss_base >= 0) {
        const int n_here = (int)min((total + per_batch - 1u) / per_batch, (uint32_t)kLossSlots);
        for (int q = threadIdx.x; q < n_here; q += blockDim.x) {
            const float v = wg_loss[q];
            if (v != 0.f) loss_add(loss_out + wg_loss_base + q, v);
        }
    }
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(ctl + kCtlLeave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
        for (int q = 0; q < kQueues; ++q) ctl[q * kQueueStride] = 0u;
        ctl[kCtlArrive] = 0u;
        ctl[kCtlLeave] = 0u;
    }
}

}  // namespace tkr

#ifdef TKR_FLOW_TRACE
extern "C" int tkr_flow_trace_buffer(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(tkr::g_flow_trace), &p, sizeof(p));
}
#endif

extern "C" int32_t tkr_flow_row_granules(int32_t k) { return (k + 127) / 128 * 128; }
extern "C" int32_t tkr_flow_ctl_words(void) { return tkr::kCtlArrive + 128; }     // (+32 .. +95: profiling builds)

extern "C" int tkr_bpr_flow_run(const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc, int32_t batch_size,
                                int32_t n_batches, uint32_t* ctl, float* loss_out, int32_t waves_per_cu, void* stream) {
    if (!st || !st->U || !st->V || !st->tailU || !st->tailV || !st->rdU || !st->rdV) return TKR_EINVAL;
    if (st->opt != 0 && st->opt != 1) return TKR_EINVAL;
    if (st->opt == 0 && (!st->msU || !st->msV)) return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->k <= 0) return TKR_EINVAL;
    if (st->k > 256) return TKR_EUNSUPPORTED;
    if (st->item_bufs != 0 && st->item_bufs != 2 && st->item_bufs != 4) return TKR_EINVAL;
    if (!prec || !pocc || !ctl || batch_size <= 0 || n_batches < 0) return TKR_EINVAL;
    if (n_batches == 0) return TKR_OK;
    const uint64_t total64 = (uint64_t)n_batches * 3u * (uint64_t)batch_size;
    if (total64 >= 0xffffffffull / 8) return TKR_EUNSUPPORTED;
    const uint32_t total = (uint32_t)total64;
    int dev = 0;
    TKR_CHECK(hipGetDevice(&dev));
    const int np = (st->k + 127) / 128;
    static int cached_cus[64], cached_per_cu[64][2];              // per device and kernel variant: the queries cost more than a short launch
    int cus, per_cu;
    if (dev >= 0 && dev < 64 && cached_per_cu[dev][np - 1] > 0) {
        cus = cached_cus[dev];
        per_cu = cached_per_cu[dev][np - 1];
    } else {
        const void* fn = np == 1 ? (const void*)tkr::bpr_flow_kernel<1, false> : (const void*)tkr::bpr_flow_kernel<2, false>;
        TKR_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        TKR_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0));
        if (per_cu < 1) return TKR_EUNSUPPORTED;
        if (dev >= 0 && dev < 64) { cached_cus[dev] = cus; cached_per_cu[dev][np - 1] = per_cu; }
    }
    const uint32_t tune = ((uint32_t)waves_per_cu >> 8) & 0xffu;     // experiment switches ride in bits 8..15
    waves_per_cu &= 0xff;
    // default: one 4-wave workgroup per CU = 1024 waves, about 1.3 batches of 256 in flight; two per CU from batch 320 on.
    // Measured at the ML-10M shape (round 3, us per batch with 4 / 8 waves per CU): batch 128 1.80 / 1.81, 256 2.44 / 2.45 (bound
    // by the chains through the popular items: more waves only wait more, 1.0 -> 4.0 poll passes per task), 320 2.81 / 2.78,
    // 384 3.30 / 3.13, 448 3.85 / 3.50, 512 4.35 / 3.98 (task throughput starts to matter), 12 per CU no better than 8.
    int want = (waves_per_cu > 0 ? waves_per_cu : (batch_size >= 320 ? 8 : 4)) / 4;            // 256-thread workgroups
    if (want < 1) want = 1;
    if (want > per_cu - (per_cu > 2 ? 1 : 0)) want = per_cu - (per_cu > 2 ? 1 : 0);   // stay inside what is resident at once
    uint32_t grid = (uint32_t)(want * cus);
    const uint32_t need = (total + 3) / 4;                           // never more waves than tasks
    if (grid > need) grid = need;
    if (grid < 8) grid = 8;                                          // >= 32 waves: every queue has a wave
    hipStream_t s = (hipStream_t)stream;
    const int4* r4 = reinterpret_cast<const int4*>(prec);
    const int4* o4 = reinterpret_cast<const int4*>(pocc);
    static const bool prof = getenv("TKR_FLOW_PROFILE") && getenv("TKR_FLOW_PROFILE")[0] == '1';     // cycle sums into ctl (scripts/probe_flow_bench.py)
    if (prof) {
        if (np == 1) hipLaunchKernelGGL((tkr::bpr_flow_kernel<1, true>), dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out, tune, 3u * (uint32_t)batch_size);
        else hipLaunchKernelGGL((tkr::bpr_flow_kernel<2, true>), dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out, tune, 3u * (uint32_t)batch_size);
    } else if (np == 1) hipLaunchKernelGGL((tkr::bpr_flow_kernel<1, false>), dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out, tune, 3u * (uint32_t)batch_size);
    else hipLaunchKernelGGL((tkr::bpr_flow_kernel<2, false>), dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out, tune, 3u * (uint32_t)batch_size);
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

static_assert(tkr::kCtlStatus == TKR_FLOW_CTL_STATUS && tkr::kCtlSpins == TKR_FLOW_CTL_SPINS, "ctl layout of include/tkr.h");
