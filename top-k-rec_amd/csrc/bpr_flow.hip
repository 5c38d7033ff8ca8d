// K2f -- the BPR step as ONE persistent launch per chunk of mini-batches (dataflow form of csrc/bpr_step.hip).
//
// Same arithmetic and the same semantics as K2 (sess.run([solver, obj]) of single/bpr.py:141 on the graph of
// single/bpr.py:81-100; batch t+1 reads what batch t wrote, bpr.py:139-147), but the order between batches is no
// longer a kernel boundary (1.5 us of dependent-launch gap + two cold memory levels per 256-triplet batch = 96 % idle,
// profiles/README.md): it is carried by the data.
//
//   * Every table element is an 8-byte GRANULE {fp32 value, uint32 tag}; the tag is the VERSION of its row = the number
//     of updates the row has seen.  A granule is written by one aligned 8-byte agent-scope store and read by one 8-byte
//     agent-scope load (global_store/load_dwordx2 sc1), so a reader that sees tag == v holds a value of version v: the
//     data is its own flag, no fence and no separate ready word (MI355X_MICROARCH.md, R2 form of the hand-off recipe;
//     per-XCD L2s are not coherent, L1s are never refreshed: plain loads of another workgroup's stores would be stale).
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
//   * Tasks are handed out in plan order (batch-major) by 8 ticket counters (one word saturates at ~90 tickets/us,
//     a 256-batch needs ~500/us): task index = 8 * ticket + queue.  A wave's home queue is its arrival number & 7; it
//     takes from home unless home runs ahead of the slowest queue, then from that one.  Every producer of a task sits
//     in an earlier batch, i.e. holds a lower task index: the lowest untaken task is always taken next by a wave of its
//     home queue (or a thief), and the lowest unfinished task never waits on anything unfinished: no deadlock with
//     >= 8 running waves, whatever the dispatch order, placement or residency.  Every spin is bounded (status word).
//
// Run-to-run results are bitwise identical (each task reads exact versions; sums run in plan order); the only float
// atomic is the reported loss.  Against K2 the sums differ in the last bits (lane -> element mapping, heavy rows are
// summed by one wave instead of a team): both are held to the same tolerance against the oracle.
//
// Roofline: the same algorithmic bytes as K2 (48k + 56 per triplet); granules double the bytes that
// really move, which is irrelevant where this kernel is used (B <= 1024: latency-bound) -- large batches keep the
// plain tables and K2.
#include "tkr_common.h"
#include "../../include/tkr.h"

namespace tkr {

typedef unsigned long long u64;

constexpr int kQueues = 8;
constexpr int kQueueStride = 32;          // uint32 words between ticket counters (one 128-byte line each)
constexpr int kCtlArrive = kQueues * kQueueStride;
constexpr int kCtlExit = kCtlArrive + 1;
constexpr int kCtlStatus = kCtlArrive + 2;
constexpr int kCtlSpins = kCtlArrive + 3;     // diagnostics: spin passes taken
constexpr int kCtlDebug = kCtlArrive + 8;     // 16 words: what the first wave that gave up was waiting for
constexpr uint32_t kSpinLimit = 1u << 20;     // passes of ONE wait (each >= ~0.3 us) before a wave gives up
constexpr int kStealSlack = 4;                // tickets a home queue may lead the slowest queue

__device__ __forceinline__ u64 ld_gran(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_gran_bits(u64* p, uint32_t bits, uint32_t tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_gran(u64* p, float v, uint32_t tag) { st_gran_bits(p, __float_as_uint(v), tag); }
__device__ __forceinline__ float gran_val(u64 g) { return __uint_as_float((uint32_t)g); }
__device__ __forceinline__ uint32_t gran_tag(u64 g) { return (uint32_t)(g >> 32); }
__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane l holds elements q*64 + l of a row (512 contiguous bytes per wave instruction)
template <int NE>
__device__ __forceinline__ void issue_row(const u64* __restrict__ row, int lane, u64 (&x)[NE]) {
#pragma unroll
    for (int q = 0; q < NE; ++q) x[q] = ld_gran(row + q * TKR_WAVE + lane);
}
template <int NE>
__device__ __forceinline__ bool row_tagged(const u64 (&x)[NE], uint32_t tag) {
    bool ok = true;
#pragma unroll
    for (int q = 0; q < NE; ++q) ok &= gran_tag(x[q]) == tag;
    return ok;
}
template <int NE>
__device__ __forceinline__ void row_values(const u64 (&x)[NE], float (&v)[NE]) {
#pragma unroll
    for (int q = 0; q < NE; ++q) v[q] = gran_val(x[q]);
}

template <int NE>
__device__ __forceinline__ void dotf2(const float (&a)[NE], const float (&b1)[NE], const float (&b2)[NE], float& d1,
                                      float& d2) {
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        p1 = fmaf(a[q], b1[q], p1);
        p2 = fmaf(a[q], b2[q], p2);
    }
    d1 = wave_sum(p1);
    d2 = wave_sum(p2);
}

struct FlowTables {                       // device view of tkr_flow_state
    u64 *U, *msU, *tailU, *V, *msV, *tailV;
    uint32_t *rdU, *rdV;
    size_t ustride, istride;              // granules per buffer
    int kp;
};

struct Own {                              // a task's own row while it is processed
    float b, msb;
    uint32_t exp_even, exp_odd, rd;       // expect[0], expect[1] as carried by the version read; rd[(version + 1) & 1]
    bool ok;
};

__device__ __forceinline__ bool spin_fail(uint32_t& spins, uint32_t* ctl) {
    __builtin_amdgcn_s_sleep(4);
    ++spins;
    if ((spins & 255u) == 0u && ld_u32(ctl + kCtlStatus) != 0u) return true;      // somebody else gave up
    if (spins >= kSpinLimit) {
        atomicOr(ctl + kCtlStatus, 1u);
        return true;
    }
    return false;
}

// One group of G <= 4 occurrences of a task.  ITEM = false: the row is a user; a = positive item, b = negative item.
// ITEM = true: the row is an item; a = user, b = the other item (bit 31 of its id: this row is the NEGATIVE item).
// Returns false when a spin ran out.
template <int NE, int G, bool ITEM>
__device__ __forceinline__ bool flow_group(const tkr_flow_state& st, const FlowTables& T, int lane, const int4 (&oc)[4],
                                           const u64* own_p, const u64* own_ms, const u64* own_tail, const uint32_t* own_rd,
                                           uint32_t own_ver, float (&own)[NE], float (&ms)[NE], Own& o, float (&g)[NE],
                                           float& gb, float& loss_lane, float& loss_x, bool want_loss, bool sgd,
                                           uint32_t* ctl, uint32_t& spins) {
    // Every pass first ISSUES all loads it still needs (own row, slot, tail, rd; both partner rows and the item tails of
    // every occurrence) and only then looks at tags: one memory round trip per pass, not one per row.
    u64 xo[NE], xm[NE], xt = 0;
    u64 xa[G][NE], xb[G][NE], xta[G], xtb[G];
    bool part_ok = false;
    uint32_t waited = 0;
    for (;;) {
        if (!o.ok) {
            issue_row<NE>(own_p, lane, xo);
            if (!sgd) issue_row<NE>(own_ms, lane, xm);
            xt = ld_gran(own_tail + (lane & 3));
            o.rd = ld_u32(own_rd);
        }
        if (!part_ok) {
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const int a = oc[q].x, b = oc[q].z & 0x3fffffff;
                const uint32_t va = (uint32_t)oc[q].y, vb = (uint32_t)oc[q].w;
                if constexpr (ITEM) {
                    issue_row<NE>(T.U + (va & 1u) * T.ustride + (size_t)a * T.kp, lane, xa[q]);
                    xta[q] = 0;
                } else {
                    issue_row<NE>(T.V + (va & 1u) * T.istride + (size_t)a * T.kp, lane, xa[q]);
                    xta[q] = ld_gran(T.tailV + ((size_t)(va & 1u) * st.n_items + a) * 4);
                }
                issue_row<NE>(T.V + (vb & 1u) * T.istride + (size_t)b * T.kp, lane, xb[q]);
                xtb[q] = ld_gran(T.tailV + ((size_t)(vb & 1u) * st.n_items + b) * 4);
            }
        }
        if (!o.ok) {
            bool lane_own = row_tagged<NE>(xo, own_ver) && gran_tag(xt) == own_ver;
            if (!sgd) lane_own = lane_own && row_tagged<NE>(xm, own_ver);
            if (__all(lane_own)) {
                o.ok = true;
                row_values<NE>(xo, own);
                if (!sgd) row_values<NE>(xm, ms);
                o.b = bcast_f(gran_val(xt), 0);
                o.msb = bcast_f(gran_val(xt), 1);
                o.exp_even = (uint32_t)bcast_i((int)(uint32_t)xt, 2);
                o.exp_odd = (uint32_t)bcast_i((int)(uint32_t)xt, 3);
            }
        }
        if (!part_ok) {
            bool lane_part = true;
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const uint32_t va = (uint32_t)oc[q].y, vb = (uint32_t)oc[q].w;
                lane_part = lane_part && row_tagged<NE>(xa[q], va) && row_tagged<NE>(xb[q], vb) && gran_tag(xtb[q]) == vb;
                if constexpr (!ITEM) lane_part = lane_part && gran_tag(xta[q]) == va;
            }
            part_ok = __all(lane_part);
        }
        if (o.ok && part_ok) break;
        if (spin_fail(waited, ctl)) {
            if (waited >= kSpinLimit && lane == 0 &&                     // post-mortem of the first wave that gave up
                atomicCAS(ctl + kCtlDebug, 0u, 1u) == 0u) {
                ctl[kCtlDebug + 1] = o.ok;
                ctl[kCtlDebug + 2] = part_ok;
                ctl[kCtlDebug + 3] = own_ver;
                ctl[kCtlDebug + 4] = gran_tag(xo[0]);
                ctl[kCtlDebug + 5] = gran_tag(xt);
                ctl[kCtlDebug + 6] = (uint32_t)oc[0].x;
                ctl[kCtlDebug + 7] = (uint32_t)oc[0].y;
                ctl[kCtlDebug + 8] = gran_tag(xa[0][0]);
                ctl[kCtlDebug + 9] = (uint32_t)oc[0].z;
                ctl[kCtlDebug + 10] = (uint32_t)oc[0].w;
                ctl[kCtlDebug + 11] = gran_tag(xb[0][0]);
                ctl[kCtlDebug + 12] = gran_tag(xtb[0]);
                ctl[kCtlDebug + 13] = ITEM;
                ctl[kCtlDebug + 14] = G;
                ctl[kCtlDebug + 15] = sgd ? 0u : gran_tag(xm[0]);
            }
            return false;
        }
    }
    spins += waited;
    float pa[G][NE], pb[G][NE], ta[G], tb[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        row_values<NE>(xa[q], pa[q]);
        row_values<NE>(xb[q], pb[q]);
        ta[q] = gran_val(xta[q]);
        tb[q] = gran_val(xtb[q]);
    }

    const bool l2 = (st.mode == 0);
#pragma unroll
    for (int q = 0; q < G; ++q) {
        if constexpr (!ITEM) {
            // x = b_i - b_j + <u, v_i> - <u, v_j>     (single/bpr.py:87-89)
            float xui, xuj;
            dotf2<NE>(own, pa[q], pb[q], xui, xuj);
            const float x = ta[q] - tb[q] + xui - xuj;
            const float s = sigmoid_neg(x);
            if (want_loss) loss_x += softplus_neg(x);
            if (l2) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    g[e] += -s * (pa[q][e] - pb[q][e]) + st.lu * own[e];
                    if (want_loss)
                        loss_lane += 0.5f * (own[e] * own[e] * st.lu + pa[q][e] * pa[q][e] * st.li + pb[q][e] * pb[q][e] * st.lj);
                }
                if (want_loss) loss_x += 0.5f * (ta[q] * ta[q] + tb[q] * tb[q]) * st.lb;
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    g[e] += -s * (pa[q][e] - pb[q][e]) + st.lu * sgn(own[e]);
                    if (want_loss) loss_lane += fabsf(own[e]) * st.lu + fabsf(pa[q][e]) * st.li + fabsf(pb[q][e]) * st.lj;
                }
                if (want_loss) loss_x += (fabsf(ta[q]) + fabsf(tb[q])) * st.lb;
            }
        } else {
            const bool role_j = oc[q].z < 0;
            float dr, dn;                                  // <u, v_row>, <u, v_other>
            dotf2<NE>(pa[q], own, pb[q], dr, dn);
            const float br = o.b, bo = tb[q];
            const float x = role_j ? (bo - br + dn - dr) : (br - bo + dr - dn);
            const float s = sigmoid_neg(x);
            const float sg = role_j ? s : -s;
            const float lam = role_j ? st.lj : st.li;
            if (l2) {
#pragma unroll
                for (int e = 0; e < NE; ++e) g[e] += sg * pa[q][e] + lam * own[e];
                gb += sg + st.lb * br;
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) g[e] += sg * pa[q][e] + lam * sgn(own[e]);
                gb += sg + st.lb * sgn(br);
            }
        }
    }

    // acknowledge the partner reads of this group: one add per (occurrence, partner) on the partner row's rd word
    uint32_t* ack = nullptr;
#pragma unroll
    for (int q = 0; q < G; ++q) {
        uint32_t* pa_rd = (ITEM ? T.rdU : T.rdV) + 2 * (size_t)oc[q].x + (oc[q].y & 1);
        uint32_t* pb_rd = T.rdV + 2 * (size_t)(oc[q].z & 0x3fffffff) + (oc[q].w & 1);
        if (lane == 2 * q) ack = pa_rd;
        if (lane == 2 * q + 1) ack = pb_rd;
    }
    if (lane < 2 * G) __hip_atomic_fetch_add(ack, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <int NE, bool ITEM>
__device__ __forceinline__ bool flow_groups(const tkr_flow_state& st, const FlowTables& T, int lane, int n,
                                            const int4 (&oc)[4], const u64* own_p, const u64* own_ms, const u64* own_tail,
                                            const uint32_t* own_rd, uint32_t own_ver, float (&own)[NE], float (&ms)[NE],
                                            Own& o, float (&g)[NE], float& gb, float& loss_lane, float& loss_x,
                                            bool want_loss, bool sgd, uint32_t* ctl, uint32_t& spins) {
    switch (n) {
        case 1: return flow_group<NE, 1, ITEM>(st, T, lane, oc, own_p, own_ms, own_tail, own_rd, own_ver, own, ms, o, g, gb, loss_lane, loss_x, want_loss, sgd, ctl, spins);
        case 2: return flow_group<NE, 2, ITEM>(st, T, lane, oc, own_p, own_ms, own_tail, own_rd, own_ver, own, ms, o, g, gb, loss_lane, loss_x, want_loss, sgd, ctl, spins);
        case 3: return flow_group<NE, 3, ITEM>(st, T, lane, oc, own_p, own_ms, own_tail, own_rd, own_ver, own, ms, o, g, gb, loss_lane, loss_x, want_loss, sgd, ctl, spins);
        default: return flow_group<NE, 4, ITEM>(st, T, lane, oc, own_p, own_ms, own_tail, own_rd, own_ver, own, ms, o, g, gb, loss_lane, loss_x, want_loss, sgd, ctl, spins);
    }
}

// next task index of this wave, or 0xffffffff when every queue is exhausted
__device__ __forceinline__ uint32_t grab(uint32_t* ctl, int lane, int home, uint32_t total) {
    for (;;) {
        const uint32_t h = (lane < kQueues) ? ld_u32(ctl + lane * kQueueStride) : 0xffffffffu;
        uint32_t best = 0xffffffffu, mine = 0xffffffffu;
        int bq = -1;
#pragma unroll
        for (int q = 0; q < kQueues; ++q) {
            const uint32_t hq = (uint32_t)bcast_i((int)h, q);
            const bool live = (u64)hq * kQueues + q < total;
            if (q == home && live) mine = hq;
            if (live && hq < best) { best = hq; bq = q; }             // lowest queue index among equal heads
        }
        if (bq < 0) return 0xffffffffu;
        const int q = (mine != 0xffffffffu && mine <= best + kStealSlack) ? home : bq;
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(ctl + q * kQueueStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (uint32_t)bcast_i((int)t, 0);
        const u64 idx = (u64)t * kQueues + q;
        if (idx < total) return (uint32_t)idx;
        // this queue ran dry between the look and the take: look again (its head now shows it)
    }
}

template <int NE>
__global__ __launch_bounds__(256) void bpr_flow_kernel(tkr_flow_state st, const int4* __restrict__ prec,
                                                       const int4* __restrict__ pocc, uint32_t total,
                                                       uint32_t* __restrict__ ctl, float* __restrict__ loss_out) {
    const int lane = threadIdx.x & (TKR_WAVE - 1);
    FlowTables T;
    T.kp = NE * TKR_WAVE;
    T.ustride = (size_t)st.n_users * T.kp;
    T.istride = (size_t)st.n_items * T.kp;
    T.U = reinterpret_cast<u64*>(st.U); T.msU = reinterpret_cast<u64*>(st.msU); T.tailU = reinterpret_cast<u64*>(st.tailU);
    T.V = reinterpret_cast<u64*>(st.V); T.msV = reinterpret_cast<u64*>(st.msV); T.tailV = reinterpret_cast<u64*>(st.tailV);
    T.rdU = st.rdU; T.rdV = st.rdV;
    const bool sgd = st.opt == 1;
    const bool want_loss = loss_out != nullptr;

    uint32_t arrive = 0;
    if (lane == 0) arrive = __hip_atomic_fetch_add(ctl + kCtlArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int home = bcast_i((int)arrive, 0) & (kQueues - 1);
    uint32_t spins = 0;
    bool alive = true;

    while (alive) {
        const uint32_t idx = grab(ctl, lane, home, total);
        if (idx == 0xffffffffu) break;
        const int4* r = prec + (size_t)idx * 8;
        const int4 w = (lane < 8) ? r[lane] : make_int4(0, 0, 0, 0);      // 128-byte record, one int4 per lane
        const int rowk = bcast_i(w.x, 0);
        if (rowk == -1) continue;                                          // unused slot of its batch
        const uint32_t ver = (uint32_t)bcast_i(w.y, 0);
        const int n_occ = bcast_i(w.z, 0);
        const int first = bcast_i(w.w, 0);
        const int batch = bcast_i(w.x, 1);
        const bool is_item = rowk < 0;
        const int row = rowk & 0x7fffffff;

        const size_t n_rows = is_item ? st.n_items : st.n_users;
        const size_t roff = (size_t)(ver & 1u) * (is_item ? T.istride : T.ustride) + (size_t)row * T.kp;
        const size_t woff = (size_t)((ver + 1u) & 1u) * (is_item ? T.istride : T.ustride) + (size_t)row * T.kp;
        u64* tabP = is_item ? T.V : T.U;
        u64* tabM = is_item ? T.msV : T.msU;
        u64* tabT = is_item ? T.tailV : T.tailU;
        const uint32_t* own_rd = (is_item ? T.rdV : T.rdU) + 2 * (size_t)row + ((ver + 1u) & 1u);

        float own[NE], ms[NE], g[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) { g[e] = 0.f; ms[e] = 0.f; }
        Own o = {0.f, 0.f, 0u, 0u, 0u, false};
        float gb = 0.f, loss_lane = 0.f, loss_x = 0.f;

        for (int done = 0; done < n_occ && alive; done += 4) {
            const int n = min(4, n_occ - done);
            int4 oc[4];
            if (done == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    oc[q] = make_int4(bcast_i(w.x, 2 + q), bcast_i(w.y, 2 + q), bcast_i(w.z, 2 + q), bcast_i(w.w, 2 + q));
            } else {                              // rows with more than 4 occurrences: the next 4 from the occurrence list
                int4 x = make_int4(0, 0, 0, 0);
                if (lane < n) x = pocc[first + done + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) oc[q] = make_int4(bcast_i(x.x, q), bcast_i(x.y, q), bcast_i(x.z, q), bcast_i(x.w, q));
            }
            const bool okg = is_item
                ? flow_groups<NE, true>(st, T, lane, n, oc, tabP + roff, tabM + roff, tabT + ((size_t)(ver & 1u) * n_rows + row) * 4,
                                        own_rd, ver, own, ms, o, g, gb, loss_lane, loss_x, false, sgd, ctl, spins)
                : flow_groups<NE, false>(st, T, lane, n, oc, tabP + roff, tabM + roff, tabT + ((size_t)(ver & 1u) * n_rows + row) * 4,
                                         own_rd, ver, own, ms, o, g, gb, loss_lane, loss_x, want_loss, sgd, ctl, spins);
            if (!okg) alive = false;
        }
        if (!alive) break;

        if (!is_item && want_loss) {
            const float tot = wave_sum(loss_lane) + loss_x;
            if (lane == 0) atomicAdd(loss_out + batch, tot);
        }

        // version ver+1 lands on the buffer that held ver-1: wait until every reader of ver-1 has acknowledged
        const uint32_t expect = (ver & 1u) ? o.exp_even : o.exp_odd;        // readers of version ver-1
        uint32_t waited = 0;
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

        const uint32_t nv = ver + 1u;
        float pn[NE], mn[NE];
        float bn, mbn = 0.f;
        if (sgd) {                                  // old/methods/bpr.py:57-61: P <- P - lr * dcost/dP
#pragma unroll
            for (int e = 0; e < NE; ++e) pn[e] = own[e] - st.lr * g[e];
            bn = o.b - st.lr * gb;
            mbn = o.msb;
        } else {                                    // TF SparseApplyRMSProp, momentum 0 (single/bpr.py:100)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                mn[e] = st.rho * ms[e] + (1.f - st.rho) * g[e] * g[e];
                pn[e] = own[e] - st.lr * g[e] / sqrtf(mn[e] + st.eps);
            }
            mbn = st.rho * o.msb + (1.f - st.rho) * gb * gb;
            bn = o.b - st.lr * gb / sqrtf(mbn + st.eps);
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) st_gran(tabP + woff + e * TKR_WAVE + lane, pn[e], nv);
        if (!sgd) {
#pragma unroll
            for (int e = 0; e < NE; ++e) st_gran(tabM + woff + e * TKR_WAVE + lane, mn[e], nv);
        }
        if (lane < 4) {
            uint32_t tv = 0u;                       // tail = {bias, its slot, expect[0], expect[1]}
            if (lane == 0) tv = is_item ? __float_as_uint(bn) : 0u;
            if (lane == 1) tv = is_item ? __float_as_uint(mbn) : 0u;
            if (lane == 2) tv = o.exp_even + ((ver & 1u) ? 0u : 2u * (uint32_t)n_occ);      // this batch read version ver
            if (lane == 3) tv = o.exp_odd + ((ver & 1u) ? 2u * (uint32_t)n_occ : 0u);
            st_gran_bits(tabT + ((size_t)(nv & 1u) * n_rows + row) * 4 + lane, tv, nv);
        }
    }

    if (lane == 0) {
        if (spins) atomicAdd(ctl + kCtlSpins, spins);
        const uint32_t waves = gridDim.x * (blockDim.x / TKR_WAVE);
        const uint32_t e = __hip_atomic_fetch_add(ctl + kCtlExit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e + 1u == waves) {                      // last wave out: counters ready for the next launch (status stays)
            for (int q = 0; q < kQueues; ++q) __hip_atomic_store(ctl + q * kQueueStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + kCtlArrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + kCtlExit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace tkr

extern "C" int32_t tkr_flow_row_granules(int32_t k) { return (k + TKR_WAVE - 1) / TKR_WAVE * TKR_WAVE; }
extern "C" int32_t tkr_flow_ctl_words(void) { return tkr::kCtlArrive + 32; }

extern "C" int tkr_bpr_flow_run(const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc, int32_t batch_size,
                                int32_t n_batches, uint32_t* ctl, float* loss_out, int32_t waves_per_cu, void* stream) {
    if (!st || !st->U || !st->V || !st->tailU || !st->tailV || !st->rdU || !st->rdV) return TKR_EINVAL;
    if (st->opt != 0 && st->opt != 1) return TKR_EINVAL;
    if (st->opt == 0 && (!st->msU || !st->msV)) return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->k <= 0) return TKR_EINVAL;
    if (st->k > 256) return TKR_EUNSUPPORTED;
    if (!prec || !pocc || !ctl || batch_size <= 0 || n_batches < 0) return TKR_EINVAL;
    if (n_batches == 0) return TKR_OK;
    const uint64_t total64 = (uint64_t)n_batches * 3u * (uint64_t)batch_size;
    if (total64 >= 0xffffffffull / 8) return TKR_EUNSUPPORTED;
    const uint32_t total = (uint32_t)total64;
    int dev = 0, cus = 0;
    TKR_CHECK(hipGetDevice(&dev));
    TKR_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int ne = (st->k + TKR_WAVE - 1) / TKR_WAVE;
    const void* fn = ne == 1 ? (const void*)tkr::bpr_flow_kernel<1> : ne == 2 ? (const void*)tkr::bpr_flow_kernel<2>
                   : ne == 3 ? (const void*)tkr::bpr_flow_kernel<3> : (const void*)tkr::bpr_flow_kernel<4>;
    int per_cu = 0;
    TKR_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0));
    if (per_cu < 1) return TKR_EUNSUPPORTED;
    int want = (waves_per_cu > 0 ? waves_per_cu : 8) / 4;            // 256-thread workgroups
    if (want < 1) want = 1;
    if (want > per_cu - (per_cu > 2 ? 1 : 0)) want = per_cu - (per_cu > 2 ? 1 : 0);   // stay inside what is resident at once
    uint32_t grid = (uint32_t)(want * cus);
    const uint32_t need = (total + 3) / 4;                           // never more waves than tasks
    if (grid > need) grid = need < 2 ? 2 : need;                     // >= 8 waves: every home queue has a wave
    hipStream_t s = (hipStream_t)stream;
    const int4* r4 = reinterpret_cast<const int4*>(prec);
    const int4* o4 = reinterpret_cast<const int4*>(pocc);
    switch (ne) {
        case 1: hipLaunchKernelGGL(tkr::bpr_flow_kernel<1>, dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out); break;
        case 2: hipLaunchKernelGGL(tkr::bpr_flow_kernel<2>, dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out); break;
        case 3: hipLaunchKernelGGL(tkr::bpr_flow_kernel<3>, dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out); break;
        default: hipLaunchKernelGGL(tkr::bpr_flow_kernel<4>, dim3(grid), dim3(256), 0, s, *st, r4, o4, total, ctl, loss_out); break;
    }
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

static_assert(tkr::kCtlStatus == TKR_FLOW_CTL_STATUS && tkr::kCtlSpins == TKR_FLOW_CTL_SPINS, "ctl layout of include/tkr.h");
