// K2o -- the persistent BPR step with OWNED item rows (round 4): the tables, versions and acknowledge protocol of K2f
// (csrc/bpr_flow.hip; sess.run([solver, obj]) of single/bpr.py:141 inside the loop of single/bpr.py:139-147, batch t+1 reads what
// batch t wrote), with the own-row chain that set K2f's pace taken out of memory.
//
// The own-row chain.  A popular item is updated in (nearly) every batch; under K2f the task of batch t+1 starts its arithmetic
// once the row written by the task of batch t has made the trip write-through store -> memory -> polling load of another CU.  Here
// every item row r has an OWNER: workgroup r % n_owner (one workgroup per CU, all resident).  K1 lays the
// item tasks of a batch out in (owner, row) order and names every owner's run (tkr_sample_plan_owned: `ohdr`); the owner's waves
// take them in plan order from a queue in LDS (one LDS atomic per task), and the row, its RMSProp slot, bias and acknowledge totals
// stay in the owner's LDS from the row's first update of a launch on: the task of batch t+1 finds what the task of batch t left
// there -- before that task's acknowledge wait and write-through stores, which now only serve the row's PARTNERS.
//
// Two forms of an item task (template parameter SCALAR; `owner_waves` bit 15 of tkr_bpr_own_run):
//   * row-read (the default): the partner rows of an occurrence -- the user row and the OTHER item's row -- are read from the granule
//     tables at their exact versions, as in K2f (flow_task.h run_task).  Measured 2.18 us per batch at the ML-10M shape against
//     2.75 for K2f on the same box.  What bounds it: throughput -- two to three memory round trips per task in series at 2-3 us
//     apiece, ~700 tasks per batch over 3,072 waves; the dependency structure alone forces only 0.38 hand-offs per batch against
//     K2f's 1.14 (scripts/chain_model.py).
//   * scalar exchange: x_uij = (<u, v_i> + b_i) - (<u, v_j> + b_j) (single/bpr.py:87-89), so the task of item i needs the other item of
//     a triplet only through the scalar <u, v_j> + b_j, which the task of item j computes anyway.  Each publishes d = <u, v_own> +
//     b_own as ONE 8-byte granule {d, epoch} in the slot of (batch, triplet, role) and reads the partner's slot (asked for together
//     with the user rows: the partner has usually published before this task has its own row); the two never read each other's rows
//     (half the partner traffic; one acknowledged reader per occurrence of an item row: the user task).  One wave of every workgroup,
//     the "scout", runs ahead of the queue's head and publishes the scalars of every task whose rows are final, without waiting for
//     anything: that is also what keeps the same-batch rendezvous free of deadlock (below).  Measured 2.74 us per batch: one more
//     serial round trip per item task (publish, then the partner's slot where the early load missed it) and one wave of twelve
//     spent on the scout.  Kept selectable; every test of tests/test_gpu_flow.py runs both forms.
//
// User tasks are handed out by tickets as in K2f, to the remaining waves of every workgroup.  The granule tables are written
// through at every update exactly as by K2f, so a launch leaves them complete (get / set, the exchange of csrc/sync.hip, a K2f
// launch all work on the same state), and a chunk may be cut into launches anywhere: a task whose row was not yet updated in THIS
// launch (prec[5] < first batch of the launch) loads it from the tables like K2f does.
//
// Progress.  Producers of a task sit in earlier batches -- in the scalar form except the partner scalars of the SAME batch.  Owner
// queues are taken in plan order by their own waves, tickets in plan order by the ticket waves, one task per wave at a time (no
// claim-ahead: a claimed task that nobody works on blocked the waves behind it); the scout never waits.  Scalar form: take the oldest
// batch with an unfinished task: all rows its tasks read are final (their writers sit in earlier batches, which are done), so the
// scouts -- which scan from their queue's head on and skip only what is not final yet -- publish every scalar of that batch that a
// taken task has not published itself (a task of several rounds announces ALL its scalars before it waits for any); the tasks of
// that batch that hold a wave then finish, the heads move.  Needs every workgroup resident (grid = n_owner <= CUs); every spin is
// bounded (status word), and the engine steps down to K2f when one runs out.
//
// Results are bitwise reproducible run to run (scout and task run the same code on the same versions; the row-read form sums like
// K2f); the scalar form differs from K2f in the last bits (x is the difference of two rounded dots): same tolerance to the oracle.
#include "flow_task.h"
// s_sleep between two polls of an LDS word (a row's tag, a ring slot's mark).  The polls are ds_read now (flow_task.h lds_peek; as
// flat loads each took a trip down the vector-memory path, which throttled them by accident): a wave that spins on LDS at full rate
// takes issue cycles from the two working waves of its SIMD.  Measured per batch in steady state / per 20-batch call, ML-10M shape:
// nap 0: 2.03 us / 112 us, 1: 2.00 / 105, 2: 1.98 / 101, 4: 1.95 / 99, 8: 1.98 / 98, 16: 2.01 / 99, 32: 2.12 / 102 (flat polls: 2.2 / 110).
#ifndef TKR_TAG_NAP
#define TKR_TAG_NAP 4
#endif
#ifdef TKR_PLAN_STAMP         // profiling build (scripts/probe_short.py): the phases of the planner prologue, workgroup 0
namespace tkr { __device__ unsigned long long own_k1_prof[32]; }
#define K1_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) tkr::own_k1_prof[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define K1_STAMP_ITEMS(i) do { if (blockIdx.x == 0 && threadIdx.x == 256) tkr::own_k1_prof[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "plan_parts.h"

namespace tkr {

// ---- the planner prologue (PLAN = true): K1 of a SHORT call inside the step's own launch -----------------------------------------
// A short call (the driver's `--steps 20`: one launch of 20 batches) used to be five launches -- sample_plan, resolve_flow_wide,
// commit, this kernel, own_loss -- and on a cold one-shot process the HOST's launch path (8-14 us per launch), not the device, was
// what the call waited for.  Here workgroups 0 .. n_plan-1 each plan one batch with the very device functions of csrc/sampler.hip
// (csrc/plan_parts.h; same words, bit for bit: tests/test_gpu_flow.py test_flow_plan_bit_exact runs this form too):
//   phase A   draw, sorts, task heads, touch bits (atomics)          -> arrive at kCtlPlanA, wait for all n_plan
//   phase B   versions from the complete bitmap, records (sc1 stores) -> arrive at kCtlPlanB, wait
//   commit    the rows whose first task of the call is mine: counter += touches, bitmap words back to zero
//   done      arrive at kCtlPlanC; EVERY workgroup waits for it, drops its L1 (one agent-scope acquire) and runs the step.
// All workgroups of the launch are resident (the step needs that anyway), so the barriers complete; every spin is bounded.
struct PlanArgs {
    const int32_t *tr_users, *row_ptr, *pos_cols, *cols_sorted;
    int32_t *ucnt, *icnt;
    uint32_t *touch_u, *touch_i;
    int32_t *out_u, *out_i, *out_j;
    int4* task;
    int2* occ;
    int32_t* occt;
    int4 *prec, *pocc;
    int32_t* ohdr;
    uint64_t seed, first_triplet;
    uint32_t n_tr, n_items;
    int n_plan, npad_items, own_words, reg_sort_ok;
};

// thread 0 adds the workgroup's arrival (if `arrive`) and polls until n have arrived; -> false when a bounded spin ran out (uniform)
template <int NAP>
__device__ __forceinline__ bool plan_rendezvous(uint32_t* counter, uint32_t n, bool arrive, uint32_t* ctl, volatile uint32_t* flag) {
    if (threadIdx.x == 0) {
        if (arrive) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t waited = 0, ok = 1u;
        while (ld_u32(counter) < n) {
            __builtin_amdgcn_s_sleep(NAP);                 // NAP x 64 clocks between polls (up to 256 pollers of one word: not a hot loop)
            if (spin_fail(waited, ctl, 0)) { ok = 0u; break; }
        }
        lds_poke(flag, ok);
    }
    __syncthreads();
    const bool ok = lds_peek(flag) != 0u;
    __syncthreads();                                   // (the flag word is reused by the next rendezvous)
    return ok;
}

constexpr uint32_t kOwnInvalid = 0xffffffffu;

// LDS of the planner prologue's two phases (they follow each other in the same bytes); the mirror of csrc/plan_parts.h lies behind
__host__ __device__ inline size_t plan_scratch_bytes(int B, int npad_items, int n_owner, int own_words) {
    size_t a = (size_t)npad_items * 8 + (kWideThreads / TKR_WAVE + 1) * 4;
    if (a < (size_t)(256 + 512) * 8 + 64) a = (size_t)(256 + 512) * 8 + 64;                  // plan_phase_a_split
    const size_t pb = (size_t)3 * B * 20 + (size_t)4 * n_owner * (own_words + 1) + (kWideThreads / TKR_WAVE + 1) * 4 + 16 + (size_t)3 * B * 32;    // plan_phase_b_wide_lds
    return ((a > pb ? a : pb) + 15) & ~(size_t)15;
}

// int4 number `idx16` of the plan's records.  PLAN (the records were written by other workgroups of THIS launch): past the L1
template <bool PLAN>
__device__ __forceinline__ int4 plan_ld(const int4* __restrict__ prec, const __amdgpu_buffer_rsrc_t& prec_r, size_t idx16) {
    if constexpr (PLAN) {
        const v4u x = __builtin_amdgcn_raw_buffer_load_b128(prec_r, (int)(idx16 * 16), 0, kAuxLoad);
        return make_int4((int)x.x, (int)x.y, (int)x.z, (int)x.w);
    } else {
        return prec[idx16];
    }
}
constexpr int kDotWin = 1024;            // ring of "scalars of queue position p are out" marks (power of two)
constexpr int kScoutAhead = 48;          // queue positions beyond the head the scout looks at

struct OwnQueue {                        // head of the workgroup's LDS block
    uint32_t head;                       // next position of the owner queue
    uint32_t total;                      // item tasks of this owner in the launch
    uint32_t arrival;
    uint32_t head_batch;                 // a batch <= the batch of position `head` (where a scan may start)
};

// the owner queue: position -> (batch, slot) through the prefix sums of the per-batch run lengths
struct QueueMap {
    const uint32_t* pre;                 // [nb + 1 + 64]: pre[0] = 0, pre[b + 1] = tasks up to and including batch b; padding 0xffffffff
    const uint32_t* start;               // [nb]: first slot of the owner's run in batch b
    uint32_t slots_per_batch;            // 3B
    // record index of position pos (counted from the launch's first batch); cur: a batch <= the position's, moved to it
    __device__ __forceinline__ uint32_t locate(uint32_t pos, uint32_t& cur, int lane) const {
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
        return cur * slots_per_batch + start[cur] + (pos - before);
    }
};

// user tasks: tickets over the slots [0, B) of every batch (a batch's user tasks come first; what else sits there is skipped)
template <bool PLAN>
struct UserTicketSrc {
    uint32_t ticket;
    int home, queues;                    // queues = min(32, ticket waves of the grid): every queue has a wave
    uint32_t total;                      // nb * B
    uint32_t B;
    const int4* __restrict__ prec;       // record 0 of the launch's first batch
    __amdgpu_buffer_rsrc_t prec_r;       // ... as a buffer (PLAN)
    __device__ __forceinline__ void prefetch(NextTask& nx, int lane) {
        const u64 i64 = (u64)(uint32_t)bcast_i((int)ticket, 0) * (uint32_t)queues + (uint32_t)home;
        const uint32_t idx = i64 < total ? (uint32_t)i64 : 0xffffffffu;
        nx.have = true;
        nx.w = make_int4(0, 0, 0, 0);
        if (idx == 0xffffffffu) { nx.idx = idx; return; }
        nx.idx = idx + 2u * B * (idx / B);
        if (lane < 8) nx.w = plan_ld<PLAN>(prec, prec_r, (size_t)nx.idx * 8 + lane);
    }
};

// ---- the own row of an item task ------------------------------------------------------------------------------------------------
// LDS layout of a resident row: [kp] values, [kp] slots, {bias, its slot, expect[0..3], -, -}
template <int NP>
__device__ __forceinline__ void lds_row_read(const float* row, int lane, bool sgd, float (&own)[2 * NP], float (&ms)[2 * NP], Own& o) {
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

struct ItemRow {                         // where the own row of an item task is
    const u64 *own_p, *own_ms, *own_tail;     // the tables, at the version read
    const uint32_t* own_rd;
    uint32_t ver;
    bool from_lds;                       // an earlier task of this launch updated the row: it is (or will be) in LDS
    const volatile uint32_t* tag;        // the row's version word in LDS
    const float* row;                    // the row's slot in LDS
    int halves;                          // 16-byte halves of a tail in the tables
    bool rd_staged;                      // LOADER: the acknowledge word was copied beside the partner rows ...
    uint32_t rd_val;                     // ... this is it (a lower bound of the word now: acknowledgements only grow)
};

// the task's way: wait for the row (LDS: the task before it in the chain; tables: only at a launch's first touch), then the acknowledge word
template <int NP>
__device__ __forceinline__ bool item_own_wait(const FlowTables& T, int lane, const ItemRow& r, bool sgd, float (&own)[2 * NP], float (&ms)[2 * NP],
                                              Own& o, uint32_t* ctl, uint32_t& spins) {
    if (r.from_lds) {
        uint32_t waited = 0;
        while (lds_peek(r.tag) != r.ver) {
#if TKR_TAG_NAP > 0
            __builtin_amdgcn_s_sleep(TKR_TAG_NAP);
#endif
            if (spin_fail(waited, ctl, 0)) {
                if (waited >= kSpinLimit && lane == 0 && atomicCAS(ctl + kCtlDebug, 0u, 4u) == 0u) {
                    ctl[kCtlDebug + 1] = lds_peek(r.tag); ctl[kCtlDebug + 3] = r.ver;
                }
                return false;
            }
        }
        spins += waited;
        asm volatile("" ::: "memory");                  // the row is read AFTER its tag
        lds_row_read<NP>(r.row, lane, sgd, own, ms, o);
        o.ok = true;
        if (r.rd_staged) {                              // no trip to memory for the acknowledge word either: own_publish polls it if this copy falls short
            o.rd = r.rd_val;
            return true;
        }
    }
    return flow_own<NP>(T, lane, r.own_p, r.own_ms, r.own_tail, r.own_rd, r.ver, own, ms, o, sgd, ctl, spins, r.halves);   // o.ok: only rd is loaded
}

// the scout's way: the row if it is final NOW (values and bias only), else false
template <int NP>
__device__ __forceinline__ bool item_own_peek(int lane, const ItemRow& r, float (&own)[2 * NP], Own& o) {
    if (r.from_lds) {
        if (lds_peek(r.tag) != r.ver) return false;
        asm volatile("" ::: "memory");
        float ms[2 * NP];
        lds_row_read<NP>(r.row, lane, true, own, ms, o);
        return true;
    }
    v4u xo[NP];
    issue_row<NP>(r.own_p, lane, xo);
    const v2u xb = issue_bias(r.own_tail);
    if (!__all(row_tagged<NP>(xo, r.ver) && xb.y == r.ver)) return false;
    row_values<NP>(xo, own);
    o.b = bcast_f(__uint_as_float(xb.x), 0);
    return true;
}

// the row-read form of an item task (SCALAR = false: csrc/flow_task.h run_task, both partner rows of an occurrence read from the
// tables like K2f does) takes its own row the same way and never prefetches a record
template <int NP>
struct LdsOwnStep {
    const FlowTables& T;
    int lane;
    const ItemRow& r;
    bool sgd;
    uint32_t* ctl;
    uint32_t& spins;
    __device__ __forceinline__ bool operator()(float (&own)[2 * NP], float (&ms)[2 * NP], Own& o) {
#ifdef TKR_OWN_PROF
        o.t_own0 = __builtin_amdgcn_s_memtime();
        const bool ok = item_own_wait<NP>(T, lane, r, sgd, own, ms, o, ctl, spins);
        o.t_own1 = __builtin_amdgcn_s_memtime();
        return ok;
#else
        return item_own_wait<NP>(T, lane, r, sgd, own, ms, o, ctl, spins);
#endif
    }
};
struct NoFeed {
    __device__ __forceinline__ void prefetch(NextTask& nx, int) { nx.have = true; }
};

// ---- a round of up to G occurrences of an item task -----------------------------------------------------------------------------
// Lane q < n of `d` holds occurrence q = (user, version of the user row, other | role<<31, version of the other item [unused]);
// `tq` its triplet's index in the batch.  MODE kFull: the task's round.  kScout: one pass, nothing acknowledged, nothing waited
// for, scalars only; false = not final yet.  kAnnounce: the user rows are waited for, the scalars published, nothing else (a task
// of several rounds announces ALL its scalars before it waits for any partner's: two such tasks would otherwise wait for each
// other's later rounds).
constexpr int kFull = 0, kScout = 1, kAnnounce = 2;
template <int G>
constexpr int own_shift() { return G <= 4 ? 4 : 3; }      // lane L speaks for occurrence L >> shift

template <int NP, int G, int MODE>
__device__ __forceinline__ bool item_round(const tkr_flow_state& st, const FlowTables& T, int lane, int n, const int4 d, int tq, bool first_round,
                                           const ItemRow& r, bool sgd, u64* xch_batch, int xch_bytes, uint32_t epoch, float (&own)[2 * NP],
                                           float (&ms)[2 * NP], Own& o, float (&g)[2 * NP], float& gb, float& lam_sum, uint32_t* ctl,
                                           uint32_t& spins, uint32_t tune) {
    constexpr int SH = own_shift<G>();
    static_assert(G == 4 || G == 8, "occurrences per round");
    v4u xa[G][NP];
    uint32_t waited = 0;
    // the scalars of this round: the first lane of every group speaks for one occurrence
    const int myq = lane >> SH;
    const bool live = (lane & ((1 << SH) - 1)) == 0 && myq < n;
    const bool role = __shfl(d.z, myq) < 0;                        // this row is the NEGATIVE item of the triplet
    const int slot = (2 * __shfl(tq, myq) + (role ? 1 : 0)) * 8;   // bytes: granule (triplet, role)
    const __amdgpu_buffer_rsrc_t xr = row_rsrc(xch_batch, xch_bytes);
    v2u pvo;                                      // the PARTNER's scalar, asked for together with the user rows: a scout (or the partner
    pvo.x = 0u; pvo.y = epoch;                    // itself) has usually published it long before this task has its own row, and the round
    for (;;) {                                    // trip of the poll below would sit on the chain through a popular item.
        // one pass ISSUES every user row of the round and only then looks at tags
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int src = (q < n) ? q : 0;      // straight-line loads: idle slots repeat occurrence 0 (flow_fetch has the reason)
            const int a = bcast_i(d.x, src);
            const uint32_t va = (uint32_t)bcast_i(d.y, src);
            issue_row<NP>(T.U + (va & 1u) * T.ustride + (size_t)a * T.kp, lane, xa[q]);
        }
        if constexpr (MODE == kFull) {
            if (live) pvo = __builtin_amdgcn_raw_buffer_load_b64(xr, slot ^ 8, 0, kAuxLoad);
        }
        bool lane_ok = true, far = false;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int src = (q < n) ? q : 0;
            const uint32_t va = (uint32_t)bcast_i(d.y, src);
            lane_ok = lane_ok && row_tagged<NP>(xa[q], va);
            far = far || (int)(va - (uint32_t)bcast_i((int)xa[q][0].y, 0)) >= 4;
        }
        if (__all(lane_ok)) break;
        if constexpr (MODE == kScout) return false;
        if (far) __builtin_amdgcn_s_sleep(127);
        if (spin_fail(waited, ctl, 4)) {
            if (waited >= kSpinLimit && lane == 0 && atomicCAS(ctl + kCtlDebug, 0u, 5u) == 0u) {
                ctl[kCtlDebug + 1] = (uint32_t)d.x; ctl[kCtlDebug + 2] = (uint32_t)d.y; ctl[kCtlDebug + 3] = xa[0][0].y; ctl[kCtlDebug + 4] = (uint32_t)n;
            }
            return false;
        }
    }
    spins += waited;
    if constexpr (MODE == kScout) {
        if (first_round && !item_own_peek<NP>(lane, r, own, o)) return false;
    } else if constexpr (MODE == kFull) {
        if (lane < n) __hip_atomic_fetch_add(T.rdU + 2 * (size_t)d.x + (d.y & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (first_round && !item_own_wait<NP>(T, lane, r, sgd, own, ms, o, ctl, spins)) return false;
    }

    // d_q = <u_q, v_own> + b_own, all reduced together; lane L holds occurrence L >> SH
    float part[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            acc = fmaf(__uint_as_float(xa[q][c].x), own[2 * c], acc);
            acc = fmaf(__uint_as_float(xa[q][c].z), own[2 * c + 1], acc);
        }
        part[q] = q < n ? acc : 0.f;
    }
    const float dme = reduce_multi<G>(part, lane) + o.b;
    // publish mine; poll the partner's only where the early load did not find it
    if (live) {
        v2u pv;
        pv.x = __float_as_uint(dme); pv.y = epoch;
        __builtin_amdgcn_raw_buffer_store_b64(pv, xr, slot, 0, kAuxStore);
    }
    if constexpr (MODE != kFull) return true;
    float dother = __uint_as_float(pvo.x);
    waited = 0;
    const bool early = __all(pvo.y == epoch);
    while (!early) {
        v2u pv = pvo;
        if (live && pvo.y != epoch) pv = __builtin_amdgcn_raw_buffer_load_b64(xr, slot ^ 8, 0, kAuxLoad);
#ifdef TKR_OWN_PROF
        if (tune & 4u) pv.y = epoch;              // timing experiment only (wrong results): the partner's scalar is never waited for
#endif
        dother = __uint_as_float(pv.x);
        if (__all(pv.y == epoch)) break;
        if (spin_fail(waited, ctl, 1)) {
            if (waited >= kSpinLimit && lane == 0 && atomicCAS(ctl + kCtlDebug, 0u, 6u) == 0u) {
                ctl[kCtlDebug + 1] = (uint32_t)d.x; ctl[kCtlDebug + 2] = (uint32_t)d.z; ctl[kCtlDebug + 3] = (uint32_t)tq; ctl[kCtlDebug + 4] = epoch;
            }
            return false;
        }
    }
    spins += waited;
    const float x = role ? dother - dme : dme - dother;            // x_uij = x_ui - x_uj
    const float sme = fast_sigmoid_neg(x);
    const u64 roles = __ballot(lane < n && d.z < 0);
#pragma unroll
    for (int q = 0; q < G; ++q) {
        if (q < n) {
            const float sq = bcast_f(sme, q << SH);
            const bool rj = (roles >> q) & 1ull;
            const float sg = rj ? sq : -sq;
            gb += sg;
            lam_sum += rj ? st.lj : st.li;
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                g[2 * c] = fmaf(sg, __uint_as_float(xa[q][c].x), g[2 * c]);
                g[2 * c + 1] = fmaf(sg, __uint_as_float(xa[q][c].z), g[2 * c + 1]);
            }
        }
    }
    return true;
}

// all the occurrences of an item task: up to four sit in the record (their triplet indices in its words 24..27), more come from
// pocc / occt in rounds of eight.  SCOUT: scalars only; false = some row is not final yet (nothing is lost: the task publishes them too)
template <int NP, bool SCOUT>
__device__ __forceinline__ bool item_task(const tkr_flow_state& st, const FlowTables& T, int lane, int n_occ, int first, const int4 w,
                                          const int4* __restrict__ pocc, const int32_t* __restrict__ occt, const ItemRow& r, bool sgd,
                                          u64* xch_batch, int xch_bytes, uint32_t epoch, float (&own)[2 * NP], float (&ms)[2 * NP], Own& o,
                                          float (&g)[2 * NP], float& gb, uint32_t* ctl, uint32_t& spins, uint32_t tune) {
    constexpr int NE = 2 * NP;
    constexpr int M = SCOUT ? kScout : kFull;
    float lam_sum = 0.f;
    if (n_occ <= 4) {                             // the common case: its occurrences sit in the record (lanes 2..5), their triplets in lane 6
        const int src = (lane + 2) & 7;
        const int4 d = make_int4(__shfl(w.x, src), __shfl(w.y, src), __shfl(w.z, src), __shfl(w.w, src));
        const int t0 = bcast_i(w.x, 6), t1 = bcast_i(w.y, 6), t2 = bcast_i(w.z, 6), t3 = bcast_i(w.w, 6);
        const int tq = lane == 0 ? t0 : lane == 1 ? t1 : lane == 2 ? t2 : t3;
        if (!item_round<NP, 4, M>(st, T, lane, n_occ, d, tq, true, r, sgd, xch_batch, xch_bytes, epoch, own, ms, o, g, gb, lam_sum, ctl, spins, tune))
            return false;
    } else {
        if constexpr (!SCOUT) {
            if (n_occ > 8) {                      // several rounds: the own row first, then every round's scalars, and only then the waiting
                if (!item_own_wait<NP>(T, lane, r, sgd, own, ms, o, ctl, spins)) return false;
                for (int done = 0; done < n_occ; done += 8) {
                    const int n = min(8, n_occ - done);
                    int4 d = make_int4(0, 0, 0, 0);
                    int tq = 0;
                    if (lane < n) { d = pocc[first + done + lane]; tq = occt[first + done + lane]; }
                    if (!item_round<NP, 8, kAnnounce>(st, T, lane, n, d, tq, false, r, sgd, xch_batch, xch_bytes, epoch, own, ms, o, g, gb, lam_sum, ctl,
                                                      spins, tune))
                        return false;
                }
            }
        }
        for (int done = 0; done < n_occ; done += 8) {
            const int n = min(8, n_occ - done);
            int4 d = make_int4(0, 0, 0, 0);
            int tq = 0;
            if (lane < n) { d = pocc[first + done + lane]; tq = occt[first + done + lane]; }
            if (!item_round<NP, 8, M>(st, T, lane, n, d, tq, done == 0 && (SCOUT || n_occ <= 8), r, sgd, xch_batch, xch_bytes, epoch, own, ms, o, g, gb,
                                      lam_sum, ctl, spins, tune))
                return false;
        }
    }
    if constexpr (!SCOUT) {
        const bool l2 = (st.mode == 0);
#pragma unroll
        for (int e = 0; e < NE; ++e) g[e] = fmaf(lam_sum, l2 ? own[e] : sgn(own[e]), g[e]);
        gb = fmaf((float)n_occ * st.lb, l2 ? o.b : sgn(o.b), gb);
    }
    return true;
}

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

// version ver+1 lands on the buffer that held ver-1 (ver-3 with four buffers): wait until every reader of that version has
// acknowledged, then write through
template <int NP>
__device__ __forceinline__ bool own_publish(int lane, bool is_item, uint32_t bmask, bool sgd, u64* tabP, u64* tabM, u64* tabT, size_t woff,
                                            size_t n_rows, int row, int rowk, uint32_t ver, int n_occ, Own& o, const uint32_t* own_rd,
                                            const float (&pn)[2 * NP], const float (&mn)[2 * NP], float bn, float mbn, uint32_t* ctl,
                                            uint32_t& spins, uint32_t readers_per_occ) {
    const uint32_t nv = ver + 1u;
    const uint32_t expect = pick_exp(o, (ver + 1u) & bmask);            // readers of the version that buffer holds now
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
    store_row<NP>(tabP + woff, lane, pn, nv);
    if (!sgd) store_row<NP>(tabM + woff, lane, mn, nv);
    store_tail(tabT, n_rows, row, bmask, lane, is_item, bn, mbn, o, ver, n_occ, readers_per_occ);
    return true;
}

// ---- the loader / consumer form (LOADER = true, round 5) -----------------------------------------------------------------------------
// One wave of the workgroup -- the loader -- walks the owner queue ahead of the waves that run its tasks and copies every task's
// record and PARTNER rows (an occurrence's user row, the other item's row and bias granule) into a ring of LDS slots by direct-to-LDS
// loads (global_load_lds: no registers, a dozen in flight); the task's wave validates the tags it finds there (flow_task.h LdsStage)
// and only goes to memory itself for what was not final yet when the loader asked.  The loader stays at most kLoadAhead batches in
// front of the queue's head: further ahead, more of what it stages is stale by the time it is read.
constexpr int kLoadSlotRows = 8;                                  // 4 occurrences x (user row, other item's row)
constexpr int kLoadAhead = 2;
constexpr int kLoaders = 3;                                       // loader waves per workgroup
template <int NP>
constexpr int load_slot_bytes() { return 128 + 64 + kLoadSlotRows * NP * 1024; }        // record | bias granules | rows (KB-aligned behind the first 192 bytes + pad)
template <int NP>
constexpr int load_slot_stride() { return (load_slot_bytes<NP>() + 1023) & ~1023; }

__device__ __forceinline__ void wait_vmcnt_at_most(int n) {       // s_waitcnt takes an immediate: the loads of a slot are counted at run time
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;       // (a slot is at most 2 + 3 * 4 = 14 loads at k <= 128)
    }
}
// 16 bytes per lane of the first `lanes` lanes, global -> LDS at dst + 16 * lane, past the L1 (sc1: the tags must be memory's)
__device__ __forceinline__ void dma16(const void* src_lane, unsigned char* dst_wave, int lane, int lanes) {
    if (lane < lanes)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lane, (__attribute__((address_space(3))) void*)dst_wave, 16, 0, 16);
}

constexpr int own_min_waves(int np, int tpb) { return tpb / 256; }      // per SIMD: 16 waves per CU at <= 128 registers (k <= 128), 8 at <= 256

// SCALAR: the item tasks of a triplet exchange scalars (one reader per occurrence of an item row: the user task; wave 0 of a
// workgroup is the scout); else they read each other's rows like K2f (two readers; wave 0 is one more owner wave)
template <int NP, int TPB, bool SCALAR, bool PLAN = false, bool LOADER = false>
__global__ __launch_bounds__(TPB, own_min_waves(NP, TPB)) void bpr_own_kernel(
    tkr_flow_state st, const int4* __restrict__ prec /*record 0 of the first batch to run*/, const int4* __restrict__ pocc,
    const int32_t* __restrict__ occt, const int32_t* __restrict__ ohdr /*[n_owner][ohdr_stride], at the first batch to run*/, int ohdr_stride,
    int first_batch, int nb, int B, int n_owner, int owner_waves, uint32_t tune, uint32_t* __restrict__ ctl, float* __restrict__ loss_out,
    u64* __restrict__ xch /*scalar slots [batches of the plan][B][2], at batch 0 of the plan*/, uint32_t epoch, PlanArgs pa) {
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

    // ---- the workgroup's LDS: queue head | prefix sums | run starts | row tags | scout marks | rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    OwnQueue* q = reinterpret_cast<OwnQueue*>(smem);
    uint32_t* pre = reinterpret_cast<uint32_t*>(smem + sizeof(OwnQueue));           // [nb + 1 + 64]
    uint32_t* start = pre + nb + 1 + TKR_WAVE;                                         // [nb]
    const int rows_here = (st.n_items + n_owner - 1) / n_owner;
    uint32_t* tags = start + nb;                                                       // [rows_here]
    uint32_t* dotmark = tags + rows_here;                                              // [kDotWin]: position + 1 of the task whose scalars are out
    float* rows = reinterpret_cast<float*>(smem + ((sizeof(OwnQueue) + (size_t)4 * (2 * nb + 1 + TKR_WAVE + rows_here + kDotWin) + 15) & ~(size_t)15));
    // LOADER: ring_slots slots of partner rows behind the resident rows (KB-aligned), their ready / freed words behind the ring
    unsigned char* ring = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(rows + (size_t)rows_here * ROWF) + 1023) & ~(uintptr_t)1023);
    const int ring_slots = LOADER ? (int)((tune >> 8) & 0xffu) : 0;                     // (the host sizes the LDS: own_lds_bytes_loader)
    const uint32_t load_ahead = LOADER ? (tune >> 16) : 0u;                             // batches the loaders may run in front of the queue's head
    const int n_loaders = LOADER ? min(kLoaders, owner_waves) : 0;                      // of the owner_waves + 1 waves on the owner queue: at least one runs tasks
    volatile uint32_t* ring_ready = reinterpret_cast<volatile uint32_t*>(ring + (size_t)ring_slots * load_slot_stride<NP>());   // position + 1 staged in the slot
    volatile uint32_t* ring_freed = ring_ready + ring_slots;                            // position + 1 whose task has taken the slot's contents

    // owner number = workgroup number (every workgroup is resident: which CU it sits on means nothing).  The first ticket of a user-task
    // wave goes out NOW, beside the loads of the queue header: a launch used to start with three trips in series -- an arrival
    // counter, the header, the first ticket -- in front of every short call.
    const uint32_t me = blockIdx.x;
    bool alive = true;
    int commit_key = -1, commit_total = 0;       // PLAN: the row this thread commits at the end of the kernel (a task's row word), its touches
#ifdef TKR_PLAN_STAMP        // scripts/probe_short.py with a -DTKR_PLAN_STAMP build: s_memrealtime (100 MHz) of workgroups 0 and 255 at the prologue's phases
#define PLAN_STAMP(i) do { if ((me == 0 || me == gridDim.x - 1) && threadIdx.x == 0) reinterpret_cast<u64*>(ctl + kCtlProf)[(me ? 16 : 0) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PLAN_STAMP(i) do { } while (0)
#endif
    PLAN_STAMP(0);
    // what of the queue header does not come from the plan (before the prologue: a workgroup that plans nothing only waits there)
    for (int b = threadIdx.x; b < TKR_WAVE; b += TPB) pre[nb + 1 + b] = 0xffffffffu;
    for (int s = threadIdx.x; s < rows_here; s += TPB) tags[s] = kOwnInvalid;
    for (int s = threadIdx.x; s < kDotWin; s += TPB) dotmark[s] = 0u;
    if constexpr (LOADER) {
        for (int s = threadIdx.x; s < 2 * ring_slots; s += TPB) lds_poke(ring_ready + s, 0u);
    }
    if constexpr (PLAN) {
        static_assert(TPB == kWideThreads, "phase B of the planner is one thread per task slot");
        unsigned char* scratch = reinterpret_cast<unsigned char*>(rows);                 // the rows' region: no row is resident yet
        volatile uint32_t* flag = &q->arrival;
        const uint32_t n_plan = (uint32_t)pa.n_plan;
        int4 ct = make_int4(-1, 0, 0, 0);                   // this thread's task of the batch, for K1c below
        int cprev = 0, ctotal = 0;
        if (me < n_plan) {
            const int b = (int)me, n3 = 3 * B;
            int4* task_b = pa.task + (size_t)b * n3;
            int2* occ_b = pa.occ + (size_t)b * n3;
            int32_t* occt_b = pa.occt + (size_t)b * n3;
            const bool split = pa.reg_sort_ok != 0 && pa.npad_items == 512 && B > 128;
            const PlanMirror mir = plan_mirror(scratch + plan_scratch_bytes(B, pa.npad_items, n_owner, pa.own_words), B);      // behind both phases' own scratch
            if (split)                                       // the common short call (batch 129 .. 256): users and items side by side
                plan_phase_a_split<TPB>(scratch, b, pa.tr_users, pa.n_tr, pa.row_ptr, pa.pos_cols, pa.cols_sorted, pa.n_items, pa.seed,
                                        pa.first_triplet + (uint64_t)b * (uint64_t)B, B, pa.out_u + (size_t)b * B, pa.out_i + (size_t)b * B,
                                        pa.out_j + (size_t)b * B, task_b, occ_b, occt_b, pa.touch_u, pa.touch_i, mir);
            else
                plan_phase_a<TPB, 256>(scratch, b, pa.tr_users, pa.n_tr, pa.row_ptr, pa.pos_cols, pa.cols_sorted, pa.n_items, pa.seed,
                                       pa.first_triplet + (uint64_t)b * (uint64_t)B, B, pa.npad_items, pa.out_u + (size_t)b * B, pa.out_i + (size_t)b * B,
                                       pa.out_j + (size_t)b * B, task_b, occ_b, occt_b, pa.touch_u, pa.touch_i, pa.reg_sort_ok != 0);
            PLAN_STAMP(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // the touch atomics of every wave are out
            __syncthreads();
            PLAN_STAMP(2);
            alive = plan_rendezvous<4>(ctl + kCtlPlanA, n_plan, true, ctl, flag);
            PLAN_STAMP(3);
            if (alive) {
                if (split)
                    plan_phase_b_wide<true, true>(scratch, b, B, task_b, occ_b, occt_b, pa.ucnt, pa.icnt, pa.touch_u, pa.touch_i, pa.pocc + (size_t)b * n3,
                                                  pa.prec + (size_t)b * n3 * 8, n_owner, pa.ohdr, ohdr_stride, pa.own_words, ct, cprev, ctotal, mir);
                else
                    plan_phase_b_wide<true, false>(scratch, b, B, task_b, occ_b, occt_b, pa.ucnt, pa.icnt, pa.touch_u, pa.touch_i, pa.pocc + (size_t)b * n3,
                                                   pa.prec + (size_t)b * n3 * 8, n_owner, pa.ohdr, ohdr_stride, pa.own_words, ct, cprev, ctotal);
                PLAN_STAMP(4);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // bitmap reads returned, record stores in memory
            __syncthreads();
            PLAN_STAMP(5);
        }
        // (a planner that gave up still arrives: the others then see the status word instead of spinning out one by one)
        alive = plan_rendezvous<12>(ctl + kCtlPlanC, n_plan, me < n_plan, ctl, flag) && alive;
        PLAN_STAMP(8);
        // every planner is past its reads of the bitmap: K1c (the rows whose first task of the call is this thread's get the call's
        // touches added to their counters, their bitmap words zeroed) may go out -- atomics and stores that nobody in this launch
        // reads.  They go out at the END of the kernel: issued here they sat in front of the queue set-up's loads in the planners'
        // memory queues (set-up 3.0 us on a planner against 1.2 elsewhere, and the planners own rows like everybody else).
        commit_key = (alive && me < n_plan && (int)threadIdx.x < 3 * B && ct.x != -1 && cprev == -1) ? ct.x : -1;
        commit_total = ctotal;
        // No acquire fence here (measured 4.5 us for one lane's buffer_inv, 14 for every wave's): the plan words other CUs wrote went
        // out write-through before they arrived (sc1 stores + vmcnt(0)), the L2s see each other's writes, and THIS CU's L1 -- dropped
        // at the start of the launch -- has never held a line of prec / pocc / ohdr: nothing in front of this point loads from
        // them.  The records and owner runs are read past the L1 all the same (plan_ld below).
        PLAN_STAMP(9);
    }
    const bool ticket_wave = wave > owner_waves;
    const int tw = wave - owner_waves - 1, n_tw = TPB / TKR_WAVE - owner_waves - 1;
    const int queues = min(kQueues, max(n_tw, 1) * (int)gridDim.x);
    const int home = ticket_wave ? (int)((me * (uint32_t)n_tw + (uint32_t)tw) % (uint32_t)queues) : 0;
    uint32_t ticket = 0;
    if (ticket_wave) ticket = grab_issue(ctl, lane, home);
    for (int b = threadIdx.x; b < nb; b += TPB) {
        uint32_t h = 0u;
        if (me < (uint32_t)n_owner) {
            if constexpr (PLAN) h = (uint32_t)__hip_atomic_load(ohdr + (size_t)me * ohdr_stride + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else h = (uint32_t)ohdr[(size_t)me * ohdr_stride + b];
        }
        start[b] = h & 0xffffu;
        pre[b + 1] = h >> 16;
    }
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
        if (lane == TKR_WAVE - 1) { q->total = incl; q->head = 0u; q->head_batch = 0u; }
        if (lane == 0) pre[0] = 0u;
    }
    __syncthreads();

    PLAN_STAMP(10);
    uint32_t spins = 0;
    uint32_t ld_ready_spins = 0, ld_tasks = 0, ld_block_spins = 0, ld_fallback = 0;      // LOADER: diagnostics (scripts/probe_own.py)
#ifdef TKR_OWN_PROF
    u64 prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 tprev = __builtin_amdgcn_s_memtime();
#endif
    const QueueMap qm{pre, start, 3u * (uint32_t)B};
    const __amdgpu_buffer_rsrc_t prec_r = row_rsrc(prec, PLAN ? nb * 3 * B * 128 : 16);
    const int xch_bytes = B * 16;
    const uint32_t q_total = q->total;

    // the own row of the item task whose record is w
    auto item_row = [&](const int4 w, ItemRow& r, int& row, int& slot) {
        const int rowk = bcast_i(w.x, 0);
        const uint32_t ver = (uint32_t)bcast_i(w.y, 0);
        const int prev = bcast_i(w.y, 1);
        row = rowk & 0x7fffffff;
        slot = row / n_owner;
        const size_t roff = (size_t)(ver & T.imask) * T.istride + (size_t)row * T.kp;
        r.own_p = T.V + roff; r.own_ms = T.msV + roff;
        r.own_tail = T.tailV + ((size_t)(ver & T.imask) * st.n_items + row) * (2 * item_halves);
        r.own_rd = T.rdV + (T.imask + 1u) * (size_t)row + ((ver + 1u) & T.imask);
        r.ver = ver;
        r.from_lds = prev >= first_batch;
        r.tag = tags + slot;
        r.row = rows + (size_t)slot * ROWF;
        r.halves = item_halves;
        r.rd_staged = false;
        r.rd_val = 0u;
    };

    constexpr uint32_t kItemReaders = SCALAR ? 1u : 2u;
    if (SCALAR && wave == 0) {
        // ================= the scout: scalars of the tasks ahead of the queue's head, as soon as their rows are final =================
        uint32_t hpos = 0, hcur = 0;
        int idle = 0;
        const uint32_t ahead = (uint32_t)kScoutAhead;
        for (;;) {
            const uint32_t head = lds_peek(&q->head);
            if (head >= q_total) break;
            if ((idle & 63) == 63 && ld_u32(ctl + kCtlStatus) != 0u) break;          // somebody gave up
            if (hpos < head || hpos >= q_total || hpos >= head + ahead) {              // (re)start at the head: what was skipped may be final now
                if (hpos >= head && idle) __builtin_amdgcn_s_sleep(16);               // a whole window without work
                hpos = head;
                hcur = lds_peek(&q->head_batch);
                idle = idle < (1 << 20) ? idle + 1 : idle;
                if (hpos >= q_total) continue;
            }
            const uint32_t pos = hpos++;
            if (lds_peek(&dotmark[pos & (kDotWin - 1)]) == pos + 1u) continue;
            const uint32_t idx = qm.locate(pos, hcur, lane);
            int4 w = make_int4(0, 0, 0, 0);
            if (lane < 8) w = prec[(size_t)idx * 8 + lane];
            ItemRow r;
            int row, slot;
            item_row(w, r, row, slot);
            const int n_occ = bcast_i(w.z, 0), first = bcast_i(w.w, 0), batch = bcast_i(w.x, 1);
            float own[NE], ms[NE], g[NE];
            Own o = {};
            float gb = 0.f;
            if (item_task<NP, true>(st, T, lane, n_occ, first, w, pocc, occt, r, sgd, xch + (size_t)batch * B * 2, xch_bytes, epoch, own, ms, o, g, gb,
                                    ctl, spins, tune)) {
                if (lane == 0) dotmark[pos & (kDotWin - 1)] = pos + 1u;
                idle = 0;
            }
        }
    } else if (LOADER && wave > owner_waves - n_loaders && wave <= owner_waves) {
        // ================= the loaders: partner rows of the queue's next tasks -> LDS =================
        // (kLoaders waves, positions dealt round robin: one wave's turn is a record through the scalar cache -- a trip of its own --
        // before it can address the rows; a single loader staged 0.4 tasks per us where the queue wants 0.8)
        constexpr int SLOTB = load_slot_stride<NP>();
        uint32_t lcur = 0;
        int prev_slot = -1;
        uint32_t prev_pos = 0;
        uint32_t waited = 0;
        for (uint32_t pos = (uint32_t)(owner_waves - wave); pos < q_total; pos += (uint32_t)n_loaders) {
            const int slot = (int)(pos % (uint32_t)ring_slots);
            const uint32_t idx = qm.locate(pos, lcur, lane);
            // the slot is free (the task ring_slots positions back took its contents), and the head is not more than kLoadAhead batches behind
            bool gave_up = false;
            // (... unless a wave already waits for this very position: an owner whose tasks are batches apart must not wait for a head
            // that waits for it)
            auto blocked = [&]() {
                return (pos >= (uint32_t)ring_slots && lds_peek(ring_freed + slot) != pos - (uint32_t)ring_slots + 1u) ||
                       (lcur > lds_peek(&q->head_batch) + load_ahead &&
                        pos >= lds_peek(&q->head));
            };
            if (blocked()) {
                // before this wave waits for the queue, the slot it filled last goes out: the task that waits for THAT slot may be the
                // one the queue is waiting for (first version: published only behind the next slot's loads -- with three task waves the
                // workgroup locked up, with seven it merely stalled)
                if (prev_slot >= 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) lds_poke(ring_ready + prev_slot, prev_pos + 1u);
                    prev_slot = -1;
                }
                while (blocked()) {
                    __builtin_amdgcn_s_sleep(2);
                    if (spin_fail(waited, ctl, 0)) { gave_up = true; break; }
                }
            }
            if (gave_up) break;
            unsigned char* sl = ring + (size_t)slot * SLOTB;
            // the record: to the slot for the task's wave, and through the scalar cache for this wave's own addressing
            const int4* rp = prec + (size_t)__builtin_amdgcn_readfirstlane((int)idx) * 8;
            const int4 h0 = rp[0];
            int loads = 1;
            dma16(reinterpret_cast<const unsigned char*>(rp) + lane * 16, sl, lane, 8);
            const int n_occ = h0.z;
            {   // the acknowledge word of the task's own row (the buffer version ver + 1 will take): 4 bytes, lane 0
                const int row_o = h0.x & 0x7fffffff;
                const uint32_t ver_o = (uint32_t)h0.y;
                const uint32_t* rdw = T.rdV + (T.imask + 1u) * (size_t)row_o + ((ver_o + 1u) & T.imask);
                if (lane == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)rdw, (__attribute__((address_space(3))) void*)(sl + 192), 4, 0, 16);
                loads += 1;
            }
            if (n_occ >= 1 && n_occ <= 4) {
                const int4 oc[4] = {rp[2], rp[3], rp[4], rp[5]};
#pragma unroll
                for (int qo = 0; qo < 4; ++qo) {
                    if (qo < n_occ) {                              // (a, version of a, b | role << 31, version of b): a = user, b = the other item
                        const int a = oc[qo].x, b = oc[qo].z & 0x3fffffff;
                        const uint32_t va = (uint32_t)oc[qo].y, vb = (uint32_t)oc[qo].w;
                        const u64* ua = T.U + (va & 1u) * T.ustride + (size_t)a * T.kp;
                        const u64* vbp = T.V + (vb & T.imask) * T.istride + (size_t)b * T.kp;
                        const u64* tb = T.tailV + ((size_t)(vb & T.imask) * st.n_items + b) * (2 * (size_t)tail_halves(T.imask));
#pragma unroll
                        for (int c = 0; c < NP; ++c) {
                            dma16(reinterpret_cast<const unsigned char*>(ua) + c * 1024 + lane * 16, sl + 1024 + ((2 * qo) * NP + c) * 1024, lane, 64);
                            dma16(reinterpret_cast<const unsigned char*>(vbp) + c * 1024 + lane * 16, sl + 1024 + ((2 * qo + 1) * NP + c) * 1024, lane, 64);
                        }
                        dma16(reinterpret_cast<const unsigned char*>(tb), sl + 128 + 16 * qo, lane, 1);     // {bias, tag, its slot, tag}: the first granule is what a task reads
                        loads += 2 * NP + 1;
                    }
                }
            }
            // the slot BEFORE this one is complete once at most this slot's loads are outstanding (loads return in order)
            if (prev_slot >= 0) {
                wait_vmcnt_at_most(loads);
                if (lane == 0) lds_poke(ring_ready + prev_slot, prev_pos + 1u);
            }
            prev_slot = slot;
            prev_pos = pos;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (prev_slot >= 0 && lane == 0) lds_poke(ring_ready + prev_slot, prev_pos + 1u);
        spins += waited;
        ld_block_spins += waited;
    } else if (wave <= owner_waves) {                                                  // (SCALAR: waves 1 .. owner_waves; else 0 .. owner_waves; LOADER: 0 .. owner_waves - kLoaders)
        // ================= item tasks of the rows this workgroup owns =================
        uint32_t cur = 0;
        while (alive) {
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&q->head, 1u);
            pos = (uint32_t)bcast_i((int)pos, 0);
            if (pos >= q_total) break;
            const uint32_t idx = qm.locate(pos, cur, lane);
            if (lane == 0) atomicMax(&q->head_batch, cur);
            int4 w = make_int4(0, 0, 0, 0);
            LdsStage stg = {nullptr, nullptr, nullptr, 0u};
            uint32_t staged_rd = 0u;
            if constexpr (LOADER) {
                // the loader has (or will have) this position's record and partner rows in slot pos % ring_slots
                const int rslot = (int)(pos % (uint32_t)ring_slots);
                unsigned char* sl = ring + (size_t)rslot * load_slot_stride<NP>();
                uint32_t waited = 0;
                while (lds_peek(ring_ready + rslot) != pos + 1u) {
#if TKR_TAG_NAP > 0
                    __builtin_amdgcn_s_sleep(TKR_TAG_NAP);
#endif
                    if (spin_fail(waited, ctl, 0)) { alive = false; break; }
                }
                if (!alive) break;
                spins += waited;
                ld_ready_spins += waited;
                ld_tasks += 1u;
                asm volatile("" ::: "memory");
                if (lane < 8) { const v4u rw = *(const lds_v4u_t*)(sl + lane * 16); w = make_int4((int)rw.x, (int)rw.y, (int)rw.z, (int)rw.w); }
                staged_rd = lds_peek(reinterpret_cast<const volatile uint32_t*>(sl + 192));
                const int n_st = bcast_i(w.z, 0);
                if (n_st >= 1 && n_st <= 4) {
                    stg = LdsStage{sl + 1024, sl + 128, ring_freed + rslot, pos + 1u};
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) lds_poke(ring_freed + rslot, pos + 1u);   // nothing staged beyond the record
                }
            } else {
                if (lane < 8) w = plan_ld<PLAN>(prec, prec_r, (size_t)idx * 8 + lane);
            }
#ifdef TKR_OWN_PROF
            asm volatile("" : "+v"(w.x) :: "memory");
            const u64 t0 = __builtin_amdgcn_s_memtime();
            prof[5] += t0 - tprev;
#endif
            ItemRow r;
            int row, slot;
            item_row(w, r, row, slot);
            if constexpr (LOADER) { r.rd_staged = true; r.rd_val = staged_rd; }
            const int rowk = bcast_i(w.x, 0);
            const uint32_t ver = r.ver;
            const int n_occ = bcast_i(w.z, 0), first = bcast_i(w.w, 0), batch = bcast_i(w.x, 1);
            const size_t woff = (size_t)((ver + 1u) & T.imask) * T.istride + (size_t)row * T.kp;
            float* lrow = rows + (size_t)slot * ROWF;

            float own[NE], ms[NE], g[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { g[e] = 0.f; ms[e] = 0.f; }
            Own o = {};
            float gb = 0.f;
            const uint32_t spins_before = spins;
            if constexpr (SCALAR) {
                alive = item_task<NP, false>(st, T, lane, n_occ, first, w, pocc, occt, r, sgd, xch + (size_t)batch * B * 2, xch_bytes, epoch, own, ms, o,
                                             g, gb, ctl, spins, tune);
            } else {
                o.ok = r.from_lds;                                       // flow_fetch then leaves the own row alone
                NextTask nx;
                nx.idx = 0u; nx.w = make_int4(0, 0, 0, 0); nx.have = true;
                NoFeed feed;
                LdsOwnStep<NP> own_step{T, lane, r, sgd, ctl, spins};
                float loss_lane = 0.f;
                if constexpr (LOADER)
                    alive = run_task<NP, true, NoFeed, LdsOwnStep<NP>, 4, LdsStage>(st, T, lane, n_occ, first, w, pocc, r.own_p, r.own_ms, r.own_tail, ver,
                                                                                    own, ms, o, g, gb, loss_lane, false, sgd, ctl, spins, nx, feed, own_step, stg);
                else
                    alive = run_task<NP, true, NoFeed, LdsOwnStep<NP>, 4>(st, T, lane, n_occ, first, w, pocc, r.own_p, r.own_ms, r.own_tail, ver, own, ms, o,
                                                                          g, gb, loss_lane, false, sgd, ctl, spins, nx, feed, own_step);
            }
            if (!alive) break;
            if constexpr (LOADER && !SCALAR) { if (spins != spins_before) ld_fallback += 1u; }
            if (lane == 0) dotmark[pos & (kDotWin - 1)] = pos + 1u;
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
                const uint32_t rb = ver & T.imask, add = kItemReaders * (uint32_t)n_occ;
                *reinterpret_cast<float4*>(lrow + 2 * KP) = make_float4(bn, mbn, __uint_as_float(o.exp[0] + (rb == 0u ? add : 0u)),
                                                                        __uint_as_float(o.exp[1] + (rb == 1u ? add : 0u)));
                *reinterpret_cast<float2*>(lrow + 2 * KP + 4) =
                    make_float2(__uint_as_float(o.exp[2] + (rb == 2u ? add : 0u)), __uint_as_float(o.exp[3] + (rb == 3u ? add : 0u)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the row is in LDS before its tag says so
            if (lane == 0) lds_poke(tags + slot, ver + 1u);

            alive = own_publish<NP>(lane, true, T.imask, sgd, T.V, T.msV, T.tailV, woff, (size_t)st.n_items, row, rowk, ver, n_occ, o, r.own_rd, pn,
                                    mn, bn, mbn, ctl, spins, kItemReaders);
#ifdef TKR_OWN_PROF
            tprev = __builtin_amdgcn_s_memtime();
            if constexpr (SCALAR) prof[0] += t1 - t0;
            else { prof[0] += o.t_own0 - t0; prof[1] += o.t_own1 - o.t_own0; prof[2] += t1 - o.t_own1; }
            prof[3] += o.t_ack - t1; prof[4] += tprev - o.t_ack; prof[6] += 1; prof[7] += r.from_lds ? 1 : 0;
#endif
        }
    } else {
        // ================= user tasks, by ticket =================
        const uint32_t total = (uint32_t)nb * (uint32_t)B;
        NextTask nx;
        nx.idx = 0u; nx.w = make_int4(0, 0, 0, 0); nx.have = false;
        while (alive) {
            if (!nx.have) {
                UserTicketSrc<PLAN> first_feed{ticket, home, queues, total, (uint32_t)B, prec, prec_r};
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
            UserTicketSrc<PLAN> feed{ticket, home, queues, total, (uint32_t)B, prec, prec_r};
            GlobalOwn<NP> own_step{T, lane, T.U + roff, T.msU + roff, own_tail, own_rd, ver, sgd, ctl, spins, 2};
            alive = run_task<NP, false, UserTicketSrc<PLAN>, GlobalOwn<NP>, 4>(st, T, lane, n_occ, first, w, pocc, T.U + roff, T.msU + roff, own_tail, ver,
                                                                         own, ms, o, g, gb, loss_lane, want_loss, sgd, ctl, spins, nx, feed, own_step);
            if (!alive) break;
#ifdef TKR_OWN_PROF
            const u64 t1 = __builtin_amdgcn_s_memtime();
#endif
            if (want_loss) {
                // A batch's loss is the sum over its ~250 user tasks.  As atomics on loss_out[batch] -- one word, one memory channel --
                // they cost 0.9 us of a 2.2 us batch (hardware fp32 atomics or a compare-and-swap loop alike: each is a memory
                // operation that the next task's first load waits behind).  The row-read form leaves `xch` alone: the task's sum goes
                // there as ONE plain 8-byte store {sum, epoch} in the slot of its record, and own_loss_kernel adds the slots of a batch
                // up after the launch (fixed order: the losses are bitwise reproducible).
                const float tot = wave_sum(loss_lane);
                if constexpr (SCALAR) {
                    if (lane == 0) loss_add(loss_out + batch, tot);
                } else if (lane == 0) {
                    const uint32_t slot = idx - (uint32_t)(batch - first_batch) * 3u * (uint32_t)B;     // user tasks: the first slots of a batch's 3B
                    // (write-through: the workgroup that adds the slots up -- own_loss_kernel, or the LAST one of this launch -- may sit on another XCD)
                    __hip_atomic_store(xch + ((size_t)batch * B + slot) * 2, ((u64)epoch << 32) | (u64)__float_as_uint(tot), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            float pn[NE], mn[NE], bn, mbn;
            own_update<NP>(st, sgd, own, ms, o, g, gb, pn, mn, bn, mbn);
            asm volatile("" : "+v"(nx.w.x), "+v"(nx.w.y), "+v"(nx.w.z), "+v"(nx.w.w));      // the next record is consumed before the stores go out
            alive = own_publish<NP>(lane, false, 1u, sgd, T.U, T.msU, T.tailU, woff, (size_t)st.n_users, row, rowk, ver, n_occ, o, own_rd, pn, mn,
                                    bn, mbn, ctl, spins, 2u);
#ifdef TKR_OWN_PROF
            tprev = __builtin_amdgcn_s_memtime();
            prof[9] += t1 - t0; prof[11] += o.t_ack - t1; prof[12] += tprev - o.t_ack; prof[13] += 1;
#endif
        }
    }

    if (lane == 0 && spins) atomicAdd(ctl + kCtlSpins, spins);
    if constexpr (LOADER) {
        if (lane == 0) {
            atomicAdd(ctl + kCtlProf + 64, ld_tasks); atomicAdd(ctl + kCtlProf + 65, ld_ready_spins);
            atomicAdd(ctl + kCtlProf + 66, ld_block_spins); atomicAdd(ctl + kCtlProf + 67, ld_fallback);
        }
    }
#ifdef TKR_OWN_PROF
    if (lane == 0)
        for (int qq = 0; qq < 16; ++qq) atomicAdd(reinterpret_cast<u64*>(ctl + kCtlProf) + qq, prof[qq]);
#endif
    // the last workgroup out puts the ticket words back to zero (as K2f: the next launch needs no memset)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PLAN_STAMP(11);
    if constexpr (PLAN) {
        if (commit_key != -1) plan_commit_first_touch(make_int4(commit_key, 0, 0, 0), -1, commit_total, pa.ucnt, pa.icnt, pa.touch_u, pa.touch_i);
    }
    const bool fold_loss = !SCALAR && want_loss && (tune & 8u) != 0u;       // tune bit 3 (set by the host for short launches): the last workgroup out adds the losses up
    if (threadIdx.x == 0) {
        const bool last = __hip_atomic_fetch_add(ctl + kCtlLeave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
        if (last) {
            for (int qq = 0; qq < kQueues; ++qq) ctl[qq * kQueueStride] = 0u;
            ctl[kCtlArrive] = 0u;
            ctl[kCtlLeave] = 0u;
            if constexpr (PLAN) { ctl[kCtlPlanA] = 0u; ctl[kCtlPlanB] = 0u; ctl[kCtlPlanC] = 0u; }
        }
        q->arrival = last ? 1u : 0u;
    }
    if (!fold_loss) return;
    // ---- the losses of a SHORT launch: the last workgroup out adds the per-task sums up, instead of a launch of own_loss_kernel behind
    // this one (4.4 us + a kernel boundary + a host launch, of a ~100 us call).  Every other workgroup has drained its stores (above)
    // before it arrived at kCtlLeave, and the slots are write-through.  Same summation order as own_loss_kernel, bit for bit: slot t
    // belongs to "thread" t & 255, a thread adds its slots in order, waves of 64 threads by wave_sum, (p0 + p1) + (p2 + p3).
    __syncthreads();
    if (q->arrival == 0u) return;
    for (int b = wave; b < nb; b += TPB / TKR_WAVE) {
        const int batch = first_batch + b;
        float part[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float acc = 0.f;
            for (int t = w * TKR_WAVE + lane; t < B; t += 256) {
                const u64 g = __hip_atomic_load(xch + ((size_t)batch * B + t) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(g >> 32) == epoch) acc += __uint_as_float((uint32_t)g);
            }
            part[w] = wave_sum(acc);
        }
        if (lane == 0) loss_out[batch] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

// the losses of a launch's batches from the per-task sums its user tasks left in xch (row-read form): loss_out[b] = sum of the slots
// of batch b that carry this launch's epoch
__global__ __launch_bounds__(256) void own_loss_kernel(const u64* __restrict__ xch, int first_batch, int B, uint32_t epoch, float* __restrict__ loss_out) {
    __shared__ float part[4];
    const int b = first_batch + (int)blockIdx.x, lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x / TKR_WAVE;
    float acc = 0.f;
    for (int s = threadIdx.x; s < B; s += 256) {
        const u64 g = xch[((size_t)b * B + s) * 2];
        if ((uint32_t)(g >> 32) == epoch) acc += __uint_as_float((uint32_t)g);
    }
    acc = wave_sum(acc);
    if (lane == TKR_WAVE - 1) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[b] = (part[0] + part[1]) + (part[2] + part[3]);       // ASSIGNED: a batch runs once (no fill in front of a short call)
}

static size_t own_lds_bytes(int np, int nb, int n_items, int n_owner) {
    const int rows_here = (n_items + n_owner - 1) / n_owner;
    const size_t head = (sizeof(OwnQueue) + (size_t)4 * (2 * nb + 1 + TKR_WAVE + rows_here + kDotWin) + 15) & ~(size_t)15;
    return head + (size_t)rows_here * (2 * np * 128 + 8) * 4;
}

// ... with the loader's ring of `slots` slots behind the rows (KB-aligned) and the slots' ready / freed words
static size_t own_lds_bytes_loader(int np, int nb, int n_items, int n_owner, int slots) {
    const size_t base = (own_lds_bytes(np, nb, n_items, n_owner) + 1023) & ~(size_t)1023;
    const size_t stride = np == 1 ? load_slot_stride<1>() : load_slot_stride<2>();
    return base + (size_t)slots * stride + (size_t)slots * 8 + 16;
}
static int own_loader_slots(int np, int nb, int n_items, int n_owner) {            // as many as fit, at most 16; fewer than 4: no loader form
    int slots = 16;
    while (slots >= 4 && own_lds_bytes_loader(np, nb, n_items, n_owner, slots) > 160 * 1024) --slots;
    return slots >= 4 ? slots : 0;
}

// ... with the planner prologue: its scratch (phase A: sort keys + scan; phase B: the batch's occurrences, owner bitmaps) lies over the
// rows' region, which may be smaller (test shapes)
static size_t own_lds_bytes_planned(int np, int nb, int n_items, int n_owner, int B, int npad_items, int own_words) {
    const int rows_here = (n_items + n_owner - 1) / n_owner;
    const size_t head = (sizeof(OwnQueue) + (size_t)4 * (2 * nb + 1 + TKR_WAVE + rows_here + kDotWin) + 15) & ~(size_t)15;
    const size_t rows = (size_t)rows_here * (2 * np * 128 + 8) * 4;
    const size_t scratch = plan_scratch_bytes(B, npad_items, n_owner, own_words) + plan_mirror_lds(B);
    return head + (rows > scratch ? rows : scratch);
}

}  // namespace tkr

// workgroups (= owners) a process runs at once for factor width k when `share` processes split the device's CUs between them
// (share = 1: one workgroup per CU), or 0 when the item rows do not fit their owners' LDS
extern "C" int32_t tkr_bpr_own_owners_shared(int32_t n_items, int32_t k, int32_t share) {
    if (n_items <= 0 || k <= 0 || k > 256 || share <= 0) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    const int owners = cus / share;
    if (owners <= 0) return 0;
    const int np = (k + 127) / 128;
    if (tkr::own_lds_bytes(np, 512, n_items, owners) > 160 * 1024) return 0;
    return owners;
}
extern "C" int32_t tkr_bpr_own_owners(int32_t n_items, int32_t k) { return tkr_bpr_own_owners_shared(n_items, k, 1); }

// one launch of the step on batches [first_batch, first_batch + n_batches); `pa` != nullptr: with the planner prologue (the caller has
// checked own_plan_fusable)
static int own_launch(const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc, const int32_t* occt, const int32_t* ohdr,
                      int32_t ohdr_stride, int32_t n_owner, int32_t batch_size, int32_t first_batch, int32_t n_batches, uint32_t* ctl,
                      float* loss_out, int32_t owner_waves, void* xch, uint32_t epoch, const tkr::PlanArgs* pa, void* stream) {
    if (!st || !st->U || !st->V || !st->tailU || !st->tailV || !st->rdU || !st->rdV) return TKR_EINVAL;
    if (st->opt != 0 && st->opt != 1) return TKR_EINVAL;
    if (st->opt == 0 && (!st->msU || !st->msV)) return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->k <= 0) return TKR_EINVAL;
    if (st->k > 256) return TKR_EUNSUPPORTED;
    if (st->item_bufs != 0 && st->item_bufs != 2 && st->item_bufs != 4) return TKR_EINVAL;
    if (!xch || epoch == 0u) return TKR_EINVAL;
    if (!prec || !pocc || !occt || !ohdr || !ctl || batch_size <= 0 || n_batches < 0 || first_batch < 0 || n_owner <= 0) return TKR_EINVAL;
    if (ohdr_stride < first_batch + n_batches || n_batches > 512) return TKR_EINVAL;
    if (n_batches == 0) return TKR_OK;
    if ((uint64_t)(first_batch + n_batches) * 3u * (uint64_t)batch_size >= 0xffffffffull / 8) return TKR_EUNSUPPORTED;
    const int np = (st->k + 127) / 128;
    int dev = 0;
    TKR_CHECK(hipGetDevice(&dev));
    static int cached_cus[64];
    static bool attr_set[64][6][2];
    int cus;
    if (dev >= 0 && dev < 64 && cached_cus[dev] > 0) cus = cached_cus[dev];
    else {
        TKR_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (dev >= 0 && dev < 64) cached_cus[dev] = cus;
    }
    if (n_owner > cus) return TKR_EUNSUPPORTED;                    // every owner must be resident: one workgroup per CU
    size_t lds = tkr::own_lds_bytes(np, n_batches, st->n_items, n_owner);
    if (pa) lds = tkr::own_lds_bytes_planned(np, n_batches, st->n_items, n_owner, batch_size, pa->npad_items, pa->own_words);
    if (lds > 160 * 1024) return TKR_EUNSUPPORTED;
    const uint32_t tune = ((uint32_t)owner_waves >> 8) & 0xffu;     // experiment switches ride in bits 8..15
    owner_waves &= 0xff;
    const bool wide = np == 1 && (tune & 64u) != 0u;                 // tune bit 6: 16 waves per workgroup (k <= 128: <= 128 registers) instead of 8
    const bool mid = np == 1 && !wide && (tune & 32u) == 0u;          // k <= 128: 12 waves (<= 168 registers) unless tune bit 5 asks for 8
    const int tpb = wide ? 1024 : mid ? 768 : 512;
    const int waves = tpb / TKR_WAVE;
    // one scout, `ow` waves on the owner queue, the rest on user tickets
    int ow = owner_waves > 0 ? owner_waves : (waves == 16 ? 9 : waves == 12 ? 9 : 6);
    if (ow < 1) ow = 1;
    if (ow > waves - 2) ow = waves - 2;
    const bool scalar = (tune & 128u) != 0u;                         // tune bit 7: the scalar-exchange form of the item tasks (default: they read rows)
    if (pa && !(mid && !scalar)) return TKR_EINVAL;
    // a short launch adds its losses up itself (its last workgroup out); tune bit 1 (owner_waves bit 9): always own_loss_kernel
    const bool fold = loss_out && !scalar && n_batches <= 64 && !(tune & 2u);
    // tune bit 0 (owner_waves bit 8): the loader / consumer form (k <= 128, 12 waves, row-read item tasks, no planner prologue) where
    // at least four slots of partner rows fit behind the owner's rows
    const int ring_slots = ((tune & 1u) && mid && !scalar && !pa) ? tkr::own_loader_slots(np, n_batches, st->n_items, n_owner) : 0;
    if (ring_slots) lds = tkr::own_lds_bytes_loader(np, n_batches, st->n_items, n_owner, ring_slots);
    static const int ahead_env = getenv("TKR_OWN_AHEAD") ? atoi(getenv("TKR_OWN_AHEAD")) : 0;       // tuning aid
    const uint32_t ktune = (tune & ~8u & 0xffu) | (fold ? 8u : 0u) | ((uint32_t)ring_slots << 8) | ((uint32_t)(ahead_env > 0 ? ahead_env : tkr::kLoadAhead) << 16);
#ifdef TKR_LAB
    const void* fn = pa ? (const void*)tkr::bpr_own_kernel<1, 768, false, true>
                   : ring_slots ? (const void*)tkr::bpr_own_kernel<1, 768, false, false, true>
                   : np == 1 ? (wide ? (scalar ? (const void*)tkr::bpr_own_kernel<1, 1024, true> : (const void*)tkr::bpr_own_kernel<1, 1024, false>)
                                : mid ? (scalar ? (const void*)tkr::bpr_own_kernel<1, 768, true> : (const void*)tkr::bpr_own_kernel<1, 768, false>)
                                     : (scalar ? (const void*)tkr::bpr_own_kernel<1, 512, true> : (const void*)tkr::bpr_own_kernel<1, 512, false>))
                             : (scalar ? (const void*)tkr::bpr_own_kernel<2, 512, true> : (const void*)tkr::bpr_own_kernel<2, 512, false>);
#else
    // the default library holds the forms that run by default -- 12 waves and row-read item tasks at k <= 128 (with and without the planner
    // prologue), 8 waves at k <= 256; the forms that were measured and lost (scalar exchange + scout, 16 and 8 waves at k <= 128, the
    // loader / consumer ring: DESIGN.md section 4, K2o table) are built by `make LAB=1` only
    if (scalar || wide || (np == 1 && !mid) || (tune & 1u)) return TKR_EUNSUPPORTED;
    const void* fn = pa ? (const void*)tkr::bpr_own_kernel<1, 768, false, true>
                   : np == 1 ? (const void*)tkr::bpr_own_kernel<1, 768, false> : (const void*)tkr::bpr_own_kernel<2, 512, false>;
#endif
    const int variant = pa ? 4 : ring_slots ? 5 : np == 2 ? 3 : wide ? 2 : mid ? 1 : 0;
    if (lds > 64 * 1024 && !(dev >= 0 && dev < 64 && attr_set[dev][variant][scalar])) {
        TKR_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) attr_set[dev][variant][scalar] = true;
    }
    // every owner must be RESIDENT (the dataflow inside a launch waits for workgroups that have to be running): asked of the runtime
    // once per kernel form and device, at the largest LDS size a launch may ask for (K2f had the query, K2o did not)
    static signed char fits[64][6][2];
    if (dev >= 0 && dev < 64 && fits[dev][variant][scalar] == 0) {
        int per_cu = 0;
        TKR_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, tpb, 160 * 1024));
        fits[dev][variant][scalar] = per_cu >= 1 ? 1 : -1;
    }
    if (dev >= 0 && dev < 64 && fits[dev][variant][scalar] < 0) return TKR_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int4* r4 = reinterpret_cast<const int4*>(prec) + (size_t)first_batch * 3 * batch_size * 8;      // record 0 of the first batch to run
    const int4* o4 = reinterpret_cast<const int4*>(pocc);
    ohdr += first_batch;
    const tkr::PlanArgs no_plan = {};
#define TKR_OWN_LAUNCH(NPV, TPBV, SC, PL)                                                                                                 \
    hipLaunchKernelGGL((tkr::bpr_own_kernel<NPV, TPBV, SC, PL>), dim3(n_owner), dim3(TPBV), lds, s, *st, r4, o4, occt, ohdr, ohdr_stride, first_batch, \
                       n_batches, batch_size, n_owner, ow, ktune, ctl, loss_out, static_cast<tkr::u64*>(xch), epoch, PL ? *pa : no_plan)
#ifdef TKR_LAB
    if (pa) TKR_OWN_LAUNCH(1, 768, false, true);
    else if (ring_slots)
        hipLaunchKernelGGL((tkr::bpr_own_kernel<1, 768, false, false, true>), dim3(n_owner), dim3(768), lds, s, *st, r4, o4, occt, ohdr, ohdr_stride,
                           first_batch, n_batches, batch_size, n_owner, ow, ktune, ctl, loss_out, static_cast<tkr::u64*>(xch), epoch, no_plan);
    else if (np == 1 && wide) { if (scalar) TKR_OWN_LAUNCH(1, 1024, true, false); else TKR_OWN_LAUNCH(1, 1024, false, false); }
    else if (np == 1 && mid) { if (scalar) TKR_OWN_LAUNCH(1, 768, true, false); else TKR_OWN_LAUNCH(1, 768, false, false); }
    else if (np == 1) { if (scalar) TKR_OWN_LAUNCH(1, 512, true, false); else TKR_OWN_LAUNCH(1, 512, false, false); }
    else { if (scalar) TKR_OWN_LAUNCH(2, 512, true, false); else TKR_OWN_LAUNCH(2, 512, false, false); }
#else
    if (pa) TKR_OWN_LAUNCH(1, 768, false, true);
    else if (np == 1) TKR_OWN_LAUNCH(1, 768, false, false);
    else TKR_OWN_LAUNCH(2, 512, false, false);
#endif
#undef TKR_OWN_LAUNCH
    TKR_LAUNCH_CHECK();
    if (loss_out && !scalar && !fold) {
        hipLaunchKernelGGL(tkr::own_loss_kernel, dim3(n_batches), dim3(256), 0, s, static_cast<const tkr::u64*>(xch), first_batch, batch_size, epoch,
                           loss_out);
        TKR_LAUNCH_CHECK();
    }
    return TKR_OK;
}

extern "C" int tkr_bpr_own_run(const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc, const int32_t* occt, const int32_t* ohdr,
                               int32_t ohdr_stride, int32_t n_owner, int32_t batch_size, int32_t first_batch, int32_t n_batches,
                               uint32_t* ctl, float* loss_out, int32_t owner_waves, void* xch, uint32_t epoch, void* stream) {
    return own_launch(st, prec, pocc, occt, ohdr, ohdr_stride, n_owner, batch_size, first_batch, n_batches, ctl, loss_out, owner_waves, xch, epoch,
                      nullptr, stream);
}

// the same launch between two HIP events of the caller (nullable hipEvent_t, recorded on `stream`): whoever times the launch --
// bench.py's roofline -- does not come back through the host language between K1's launches, the event and this launch (a short
// call is ~130 us of device work; 20 us of interpreter between its launches leave the device idle)
extern "C" int tkr_bpr_own_run_between(void* ev_before, void* ev_after, const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc,
                                       const int32_t* occt, const int32_t* ohdr, int32_t ohdr_stride, int32_t n_owner, int32_t batch_size,
                                       int32_t first_batch, int32_t n_batches, uint32_t* ctl, float* loss_out, int32_t owner_waves, void* xch,
                                       uint32_t epoch, void* stream) {
    if (ev_before) TKR_CHECK(hipEventRecord((hipEvent_t)ev_before, (hipStream_t)stream));
    const int rc = tkr_bpr_own_run(st, prec, pocc, occt, ohdr, ohdr_stride, n_owner, batch_size, first_batch, n_batches, ctl, loss_out, owner_waves,
                                   xch, epoch, stream);
    if (ev_after) TKR_CHECK(hipEventRecord((hipEvent_t)ev_after, (hipStream_t)stream));
    return rc;
}

extern "C" int tkr_bpr_own_plan_run(const tkr_plan_call* plan, const tkr_flow_state* st, int32_t first_batch, int32_t n_batches, uint32_t* ctl,
                                    float* loss_out, int32_t owner_waves, void* xch, uint32_t epoch, void* ev_before, void* ev_after,
                                    void* stream) {
    if (!plan || !st || first_batch < 0 || n_batches < 0 || first_batch + n_batches > plan->n_batches) return TKR_EINVAL;
    // K1 INSIDE the step's launch (the planner prologue of bpr_own_kernel) when the whole plan is what runs, one workgroup can plan
    // a batch with a thread per task slot, and the default form of the step is asked for; owner_waves bit 12: never
    const uint32_t tune = ((uint32_t)owner_waves >> 8) & 0xffu;
    const int B = plan->batch_size, n_owner = plan->n_owner;
    const int np = (st->k + 127) / 128;
    int npad = 1, npad_bits = 0;
    while (npad < 2 * B) { npad <<= 1; ++npad_bits; }
    const int own_words = n_owner > 0 ? ((plan->n_items + n_owner - 1) / n_owner + 31) / 32 : 0;
    const bool fusable = !(tune & 16u) && np == 1 && !(tune & (32u | 64u | 128u)) && first_batch == 0 && n_batches == plan->n_batches && n_batches > 0 &&
                         n_batches <= 64 && n_batches <= n_owner && B > 0 && 3 * B <= tkr::kWideThreads && plan->n_tr > 0 && plan->n_users > 0 &&
                         plan->n_items > 0 && plan->n_users < (1 << 25) && plan->n_items < (1 << 25) /*32-bit byte offsets into the bitmaps*/ && plan->ohdr_stride >= n_batches &&
                         plan->n_users == st->n_users && plan->n_items == st->n_items &&
                         tkr::own_lds_bytes_planned(np, n_batches, st->n_items, n_owner, B, npad, own_words) <= 160 * 1024;
    if (fusable) {
        if (!plan->tr_users || !plan->row_ptr || !plan->pos_cols || !plan->cols_sorted || !plan->ucnt || !plan->icnt || !plan->touch_u ||
            !plan->touch_i || !plan->out_u || !plan->out_i || !plan->out_j || !plan->task || !plan->occ || !plan->occt || !plan->prec ||
            !plan->pocc || !plan->ohdr)
            return TKR_EINVAL;
        tkr::PlanArgs pa;
        pa.tr_users = plan->tr_users; pa.row_ptr = plan->row_ptr; pa.pos_cols = plan->pos_cols; pa.cols_sorted = plan->cols_sorted;
        pa.ucnt = plan->ucnt; pa.icnt = plan->icnt; pa.touch_u = plan->touch_u; pa.touch_i = plan->touch_i;
        pa.out_u = plan->out_u; pa.out_i = plan->out_i; pa.out_j = plan->out_j;
        pa.task = reinterpret_cast<int4*>(plan->task); pa.occ = reinterpret_cast<int2*>(plan->occ); pa.occt = plan->occt;
        pa.prec = reinterpret_cast<int4*>(plan->prec); pa.pocc = reinterpret_cast<int4*>(plan->pocc); pa.ohdr = plan->ohdr;
        pa.seed = plan->seed; pa.first_triplet = plan->first_triplet;
        pa.n_tr = (uint32_t)plan->n_tr; pa.n_items = (uint32_t)plan->n_items;
        pa.n_plan = n_batches; pa.npad_items = npad; pa.own_words = own_words;
        pa.reg_sort_ok = (npad_bits < 31 && (uint64_t)(plan->n_users > plan->n_items ? plan->n_users : plan->n_items) < (1ull << (32 - npad_bits)) - 1ull) ? 1 : 0;
        if (ev_before) TKR_CHECK(hipEventRecord((hipEvent_t)ev_before, (hipStream_t)stream));
        const int rc = own_launch(st, plan->prec, plan->pocc, plan->occt, plan->ohdr, plan->ohdr_stride, n_owner, B, 0, n_batches, ctl, loss_out,
                                  owner_waves, xch, epoch, &pa, stream);
        if (ev_after) TKR_CHECK(hipEventRecord((hipEvent_t)ev_after, (hipStream_t)stream));
        return rc;
    }
    const int rc = tkr_sample_plan_owned(plan->tr_users, plan->n_tr, plan->row_ptr, plan->pos_cols, plan->cols_sorted, plan->n_users,
                                         plan->n_items, plan->seed, plan->first_triplet, plan->n_batches, plan->batch_size, plan->ucnt,
                                         plan->icnt, plan->touch_u, plan->touch_i, plan->out_u, plan->out_i, plan->out_j, plan->task,
                                         plan->occ, plan->occt, plan->prec, plan->pocc, plan->n_owner, plan->ohdr, plan->ohdr_stride, stream);
    if (rc != TKR_OK) return rc;
    return tkr_bpr_own_run_between(ev_before, ev_after, st, plan->prec, plan->pocc, plan->occt, plan->ohdr, plan->ohdr_stride, plan->n_owner,
                                   plan->batch_size, first_batch, n_batches, ctl, loss_out, owner_waves, xch, epoch, stream);
}

#ifdef TKR_PLAN_STAMP
extern "C" int tkr_debug_own_k1_prof(unsigned long long* out /*[32] host*/) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tkr::own_k1_prof), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -100;
}
#endif
