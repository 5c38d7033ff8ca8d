// Device code shared by the persistent BPR step kernels: the granule accessors, the three steps of a task (flow_fetch / flow_own /
// flow_apply) and run_task.  Included by csrc/bpr_flow.hip (K2f: every task handed out by tickets) and csrc/bpr_own.hip (K2o: item
// tasks served by the workgroup that OWNS the row, the row itself resident in LDS).  See the header of bpr_flow.hip for the protocol.
#pragma once
#include <stdlib.h>

#include "tkr_common.h"
#include "../../include/tkr.h"

namespace tkr {

typedef unsigned long long u64;

constexpr int kQueues = 32;
constexpr int kQueueStride = 32;          // uint32 words between ticket counters (one 128-byte line each)
constexpr int kCtlArrive = kQueues * kQueueStride;
constexpr int kCtlLeave = kCtlArrive + 1;     // workgroups that have taken their last ticket
constexpr int kCtlStatus = kCtlArrive + 2;
constexpr int kCtlSpins = kCtlArrive + 3;     // diagnostics: spin passes taken
constexpr int kCtlPlanA = kCtlArrive + 4;     // K2o with the planner prologue: planner workgroups past phase A (touch bits set) / past phase B (bitmap read) /
constexpr int kCtlPlanB = kCtlArrive + 5;     // done (records written, counters folded); back to zero when the launch ends
constexpr int kCtlPlanC = kCtlArrive + 6;
constexpr int kCtlDebug = kCtlArrive + 8;     // 16 words: what the first wave that gave up was waiting for
constexpr int kCtlProf = kCtlArrive + 32;     // TKR_FLOW_PROFILE=1: 8 x uint64 cycle sums (grab, record, rows, war, finish, tasks, idle slots, war of item tasks)
constexpr uint32_t kSpinLimit = 1u << 20;     // passes of ONE wait (each >= ~0.3 us) before a wave gives up

typedef uint32_t v4u __attribute__((ext_vector_type(4)));      // two granules: {value0, tag0, value1, tag1}
typedef uint32_t v2u __attribute__((ext_vector_type(2)));      // one granule: {value, tag}

// Granules travel in PAIRS: one 16-byte access per lane (buffer_load/store_dwordx4 sc1 = past the L1, write-through past
// the L2).  8-byte write-through stores are one fabric write EACH (a 128-wide row + slot = 256 of them; measured: 17 us per
// 256-batch); 16-byte ones go at the plain rate, and every aligned 8-byte half still arrives whole (MI355X_MICROARCH.md).
// aux: bit 4 = sc1 (agent scope).  Not "volatile" (bit 31): that adds sc0 = system scope; the spin loop carries a
// compiler barrier instead so that every pass really re-loads.
constexpr int kAuxLoad = 16;
constexpr int kAuxStore = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane l holds elements 128*q + 2*l + {0,1} of a row: NP 16-byte loads per lane, 1 KiB contiguous per wave instruction
template <int NP>
__device__ __forceinline__ void issue_row(const u64* __restrict__ row, int lane, v4u (&x)[NP]) {
    const __amdgpu_buffer_rsrc_t r = row_rsrc(row, NP * 1024);
#pragma unroll
    for (int q = 0; q < NP; ++q) x[q] = __builtin_amdgcn_raw_buffer_load_b128(r, q * 1024 + lane * 16, 0, kAuxLoad);
}
template <int NP>
__device__ __forceinline__ bool row_tagged(const v4u (&x)[NP], uint32_t tag) {
    bool ok = true;
#pragma unroll
    for (int q = 0; q < NP; ++q) ok &= (x[q].y == tag) & (x[q].w == tag);
    return ok;
}
template <int NP>
__device__ __forceinline__ void row_values(const v4u (&x)[NP], float (&v)[2 * NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        v[2 * q] = __uint_as_float(x[q].x);
        v[2 * q + 1] = __uint_as_float(x[q].z);
    }
}
template <int NP>
__device__ __forceinline__ void store_row(u64* __restrict__ row, int lane, const float (&v)[2 * NP], uint32_t tag) {
    const __amdgpu_buffer_rsrc_t r = row_rsrc(row, NP * 1024);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        v4u x;
        x.x = __float_as_uint(v[2 * q]); x.y = tag; x.z = __float_as_uint(v[2 * q + 1]); x.w = tag;
        __builtin_amdgcn_raw_buffer_store_b128(x, r, q * 1024 + lane * 16, 0, kAuxStore);
    }
}
// granule 0 of an item's tail: {bias, tag}
__device__ __forceinline__ v2u issue_bias(const u64* __restrict__ tail) {
    return __builtin_amdgcn_raw_buffer_load_b64(row_rsrc(tail, 32), 0, 0, kAuxLoad);
}
// a row's tail (4 granules = 32 bytes with two buffers per row, 8 = 64 bytes with four): `half` 0 = {bias, its slot},
// 1 = {expect[0], expect[1]}, 2 = {expect[2], expect[3]}, 3 = padding (tagged like the rest)
__device__ __forceinline__ v4u issue_tail(const u64* __restrict__ tail, int half, int halves) {
    return __builtin_amdgcn_raw_buffer_load_b128(row_rsrc(tail, halves * 16), half * 16, 0, kAuxLoad);
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

// Sums of M per-lane values over the wave by recursive halving: lane L ends up with the total of value L >> SH,
// SH = 6 - log2(M) (M = 4: L >> 4, M = 8: L >> 3).  The reductions of a popular item's occurrences sit on the chain that limits
// the batch, and a cross-lane move through the LDS crossbar (what __shfl_xor compiles to) is ~70 cycles of latency per level,
// six levels deep.  gfx950 exchanges half-waves and 16-lane rows in the vector ALU (v_permlane32_swap / v_permlane16_swap: one
// instruction swaps the upper half of a with the lower half of b, so a' + b' holds value i's partial sums in the lower lanes and
// value i+half's in the upper ones); inside a row DPP does the rest.  Fixed pattern, so results are repeatable.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int M>
__device__ __forceinline__ float reduce_multi(float (&v)[M], int lane) {
    static_assert(M == 4 || M == 8, "4 or 8 values");
#pragma unroll
    for (int i = 0; i < M / 2; ++i) {                               // lanes l and l + 32
        const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + M / 2]), false, false);
        v[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
#pragma unroll
    for (int i = 0; i < M / 4; ++i) {                               // rows r and r + 1
        const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + M / 4]), false, false);
        v[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    float r = v[0];
    if constexpr (M == 8) {                                         // lanes l and l ^ 8 (= a rotation by 8 inside the row)
        const bool upper = (lane & 8) != 0;
        const float send = upper ? v[0] : v[1], keep = upper ? v[1] : v[0];
        r = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xf, 0xf, false));
        r = dpp_add<0x141>(r);            // row_half_mirror: lane i + lane 7 - i of its group of eight
        r = dpp_add<0xb1>(r);             // quad_perm [1,0,3,2]
        r = dpp_add<0x4e>(r);             // quad_perm [2,3,0,1] -> all eight lanes hold the total
    } else {
        r = dpp_add<0xb1>(r);             // the sixteen lanes of a row
        r = dpp_add<0x4e>(r);
        r = dpp_add<0x124>(r);            // row_ror:4
        r = dpp_add<0x128>(r);            // row_ror:8
    }
    return r;
}

// sigma(-x) on the hardware exp / rcp (1 ulp each): this sits on the chain through the most popular rows
__device__ __forceinline__ float fast_sigmoid_neg(float x) {
    const float e = __expf(-fabsf(x));
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    return x >= 0.f ? e * r : r;
}

struct FlowTables {                       // device view of tkr_flow_state
    u64 *U, *msU, *tailU, *V, *msV, *tailV;
    uint32_t *rdU, *rdV;
    size_t ustride, istride;              // granules per buffer
    uint32_t imask;                       // buffers per ITEM row - 1 (1 or 3; user rows: always two buffers): version v lives in buffer v & imask
    int kp;
    uint32_t tune;                        // experiment switches (scripts/probe_flow_bench.py): bit 0 late acks, bit 1 nap while only the own row is missing
};

// buffers of a row's table: items T.imask + 1, users 2; a tail is 2 halves (two buffers) or 4 (four) of 16 bytes
__device__ __forceinline__ int tail_halves(uint32_t mask) { return mask == 1u ? 2 : 4; }

struct NextTask {                         // the task after the current one, fetched while the current one waits for its rows
    uint32_t idx;                         // its index (0xffffffff: the queue is exhausted)
    int4 w;                               // its record (one int4 per lane, lanes 0..7)
    bool have;
};

struct Own {                              // a task's own row while it is processed
    float b, msb;
    uint32_t exp[4], rd;                  // expect[buffer] as carried by the version read; rd[buffer of version + 1]
    bool ok;
#ifdef TKR_FLOW_TRACE
    unsigned long long t_valid, t_part;   // when the own row / the partner rows validated
#endif
#ifdef TKR_OWN_PROF
    unsigned long long t_own0, t_own1, t_ack;      // entering / leaving the own-row step, readers acknowledged (csrc/bpr_own.hip)
#endif
};

__device__ __forceinline__ void own_tail_values(const v4u xt, Own& o, int halves) {
    o.b = bcast_f(__uint_as_float(xt.x), 0);
    o.msb = bcast_f(__uint_as_float(xt.z), 0);
    o.exp[0] = (uint32_t)bcast_i((int)xt.x, 1);
    o.exp[1] = (uint32_t)bcast_i((int)xt.z, 1);
    o.exp[2] = halves > 2 ? (uint32_t)bcast_i((int)xt.x, 2) : 0u;
    o.exp[3] = halves > 2 ? (uint32_t)bcast_i((int)xt.z, 2) : 0u;
}
__device__ __forceinline__ uint32_t pick_exp(const Own& o, uint32_t buf) {
    return buf == 0u ? o.exp[0] : buf == 1u ? o.exp[1] : buf == 2u ? o.exp[2] : o.exp[3];
}

#ifndef TKR_SPIN_NAP
#define TKR_SPIN_NAP 4      // x 64 clocks between two passes of a bounded spin
#endif
#ifndef TKR_PAIR_NAP
#define TKR_PAIR_NAP 12     // ... between two 16-byte polls of a row's first pair (wait_pair)
#endif
__device__ __forceinline__ bool spin_fail(uint32_t& spins, uint32_t* ctl, int nap = 4) {
    asm volatile("" ::: "memory");                 // the next pass re-loads
    if (nap) __builtin_amdgcn_s_sleep(TKR_SPIN_NAP);
    ++spins;
    if ((spins & 255u) == 0u && ld_u32(ctl + kCtlStatus) != 0u) return true;      // somebody else gave up
    if (spins >= kSpinLimit) {
        atomicOr(ctl + kCtlStatus, 1u);
        return true;
    }
    return false;
}

// next task index of this wave, or 0xffffffff when its queue is exhausted: ONE returning atomic per task on the wave's
// home counter.  (A first version also read all counters to steal from a lagging queue: 768 tasks x 8 loads per batch on
// 8 lines that are being atomically updated serialise at the atomic rate -- measured 10 us per grab.)
__device__ __forceinline__ uint32_t grab_issue(uint32_t* ctl, int lane, int home) {     // lane 0 holds the ticket when it lands
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ctl + home * kQueueStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
}
__device__ __forceinline__ uint32_t grab_index(uint32_t ticket, int home, uint32_t total) {
    const u64 idx = (u64)(uint32_t)bcast_i((int)ticket, 0) * kQueues + home;
    return idx < total ? (uint32_t)idx : 0xffffffffu;
}

// Wait until the first granule pair of a row carries `tag`: ONE 16-byte request per poll (every lane the same address) instead of
// the whole pass a waiting task used to re-issue -- 4 to 16 KB per poll, and a waiting wave that streams costs every other wave's
// loads their latency (MI355X_MICROARCH.md: a hand-off is 0.8-1.0 us on an idle chip, 2.3-5 us beside 8-15 streaming waves).  `far`:
// tags that many versions back mean the producer is at least three updates away: sleep through two hand-offs.
__device__ __forceinline__ bool wait_pair(const u64* row, uint32_t tag, int far, uint32_t* ctl, uint32_t& waited) {
    const __amdgpu_buffer_rsrc_t r = row_rsrc(row, 16);
    for (;;) {
        const v4u p = __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, kAuxLoad);
        if (p.y == tag) return true;
        if ((int)(tag - p.y) >= far) __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(TKR_PAIR_NAP);   // ~0.3 us between polls (measured 0 / 12 / 28 / 60: 2.184 / 2.160 / 2.184 / 2.244 us per batch under K2o)
        if (spin_fail(waited, ctl, 4)) return false;
    }
}

// Where a wave's next task comes from.  TicketSrc: the ticket taken before the current task's loads is back by the time their
// first pass has been waited for; its record is asked for now and lands while the current task validates, waits and computes.
struct TicketSrc {
    uint32_t ticket;
    int home;
    uint32_t total;
    const int4* __restrict__ prec;
    __device__ __forceinline__ void prefetch(NextTask& nx, int lane) {
        nx.idx = grab_index(ticket, home, total);
        nx.w = make_int4(0, 0, 0, 0);
        if (nx.idx != 0xffffffffu && lane < 8) nx.w = prec[(size_t)nx.idx * 8 + lane];
        nx.have = true;
    }
};

// ---- a task in three steps --------------------------------------------------------------------------------------------------
// flow_fetch   the partner rows of one group of n <= G occurrences: issue, re-issue until their tags are right, acknowledge, and
//              PACK what the task's arithmetic needs from them (Packed).  The own row is asked for in the same passes.
// flow_own     only the own row is missing -- the state of every task on a chain through a popular row: the bare poll loop.
// flow_apply   gradients of the group from Packed + the own row.
// Everything that does not need the own row happens in flow_fetch, i.e. while a task on a chain waits: what follows the arrival
// of the own row is that chain's period (measured: 1.65 us per link up to four occurrences, 2.2 us with eight when the partner
// rows were only turned into dot products and gradients afterwards, 4.9 us with twelve, whose second group of partner rows was
// not even asked for before the first group was done -- and the most popular item of a 256-batch has 5-8 occurrences in 81 % of
// the batches and more in 12 %).
//
// Lane q < n of `d` holds occurrence q = (a, version of a, b | role<<31, version of b).  User row: a = positive item, b = negative
// item.  Item row: a = user, b = the other item (bit 31: this row is the NEGATIVE item).  All 2G partner rows are in flight at
// once; the loads are STRAIGHT-LINE code: slots q >= n repeat occurrence 0 (hits in L2, masked out of the arithmetic) -- a branch
// per slot makes the compiler drain the memory pipe (s_waitcnt vmcnt(0)) at every join, one round trip per occurrence instead of
// one per group (seen in the ISA, and as 5.3 instead of 3.9 us per batch).
template <int G>
constexpr int group_shift() { return G <= 4 ? 4 : 3; }      // lane L speaks for occurrence L >> shift in the reductions

// Where the FIRST pass of a fetch finds the partner rows.  NoStage: in the granule tables (every pass asks memory).  LdsStage (K2o's
// loader / consumer form, csrc/bpr_own.hip): a loader wave of the workgroup has copied them into an LDS slot ahead of time --
// occurrence q: granule row `a` at base + 2q rows, row `b` at base + (2q + 1) rows, b's bias granule at bias + 16q -- and the pass
// validates the tags it finds THERE: a row that was fresh when the loader asked for it costs no trip through memory at all, a stale
// one is caught like any other (wait_pair, then a pass on the tables).  The slot is given back once the pass holds it in registers.
// Volatile accesses to LDS words (tags, queue heads, ring marks): through a generic pointer hipcc leaves them FLAT instructions --
// address-space inference skips volatile operations -- i.e. a trip down the vector-memory path to reach LDS, returned under vmcnt
// (`s_waitcnt vmcnt(0)` in every poll: the wave's outstanding global loads come back first).  Cast to the LDS address space explicitly.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32_t;
typedef __attribute__((address_space(3))) v4u lds_v4u_t;
typedef __attribute__((address_space(3))) v2u lds_v2u_t;
__device__ __forceinline__ uint32_t lds_peek(const volatile uint32_t* p) { return *(const lds_vu32_t*)(p); }
__device__ __forceinline__ void lds_poke(volatile uint32_t* p, uint32_t v) { *(lds_vu32_t*)(p) = v; }

struct NoStage {
    static constexpr bool on = false;
    __device__ __forceinline__ bool have() const { return false; }
};
struct LdsStage {
    static constexpr bool on = true;
    const unsigned char* base;          // the slot's rows (KB-aligned); nullptr: nothing staged for this task
    const unsigned char* bias;          // the slot's bias granules
    volatile uint32_t* rel;             // word that frees the slot ...
    uint32_t relv;                      // ... when it holds this value
    __device__ __forceinline__ bool have() const { return base != nullptr; }
};
template <int NP>
__device__ __forceinline__ void lds_granule_row(const unsigned char* row, int lane, v4u (&x)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) x[q] = *(const lds_v4u_t*)(row + q * 1024 + lane * 16);          // (ds_read_b128, not a flat load: see lds_peek)
}

template <int NP, int G>
struct Packed {
    float pa[G][2 * NP];     // item row: the user rows u_q;  user row: v_i - v_j of occurrence q  (0 for q >= n)
    float c[G];              // item row: this lane's share of <u_q, v_other>;  user row: 0
    float bias_me;           // of "my" occurrence (lane >> shift): item row: the other item's bias;  user row: b_i - b_j
    bool role_me;            // item row: my occurrence has this row as the NEGATIVE item
    uint32_t roles;          // item row: bit q = occurrence q has this row as the negative item
    float lam_sum;           // the own row's regulariser weight summed over the occurrences
    float loss_part;         // user row, loss wanted: the partner-only regulariser terms of this lane
    int n;
};

template <int NP, int G, bool ITEM, class Src, class Stg = NoStage>
__device__ __forceinline__ bool flow_fetch(const tkr_flow_state& st, const FlowTables& T, int lane, int n, const int4 d,
                                           const u64* own_p, const u64* own_ms, const u64* own_tail, uint32_t own_ver,
                                           float (&own)[2 * NP], float (&ms)[2 * NP], Own& o, bool want_loss, bool sgd,
                                           uint32_t* ctl, uint32_t& spins, NextTask& nx, Src& feed, Packed<NP, G>& pk, Stg stg = Stg()) {
    constexpr int NE = 2 * NP;
    static_assert(G <= 4 || G == 8, "group width");
    v4u xo[NP], xm[NP], xt = {0u, 0u, 0u, 0u};
    v4u xa[G][NP], xb[G][NP];
    v2u xta[G], xtb[G];
    uint32_t waited = 0;
    const int own_halves = tail_halves(ITEM ? T.imask : 1u);
    const size_t itg = 2 * (size_t)tail_halves(T.imask);           // granules of an item row's tail
    bool staged_pass = Stg::on && stg.have();                      // the first pass reads what the loader staged
    for (;;) {
        // a pass first ISSUES every load it still needs and only then looks at tags: one round trip per pass, not per row
        if (!o.ok) {
            issue_row<NP>(own_p, lane, xo);
            if (!sgd) issue_row<NP>(own_ms, lane, xm);
            xt = issue_tail(own_tail, lane & (own_halves - 1), own_halves);
        }
        if constexpr (Stg::on && ITEM) {
            if (staged_pass) {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int src = (q < n) ? q : 0;
                    lds_granule_row<NP>(stg.base + (2 * src) * (NP * 1024), lane, xa[q]);
                    lds_granule_row<NP>(stg.base + (2 * src + 1) * (NP * 1024), lane, xb[q]);
                    xta[q] = v2u{0u, (uint32_t)bcast_i(d.y, src)};
                    xtb[q] = *(const lds_v2u_t*)(stg.bias + 16 * src);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the slot's contents are in registers: the loader may have it back
                if (lane == 0) lds_poke(stg.rel, stg.relv);
            }
        }
#pragma unroll
        for (int q = 0; q < G; ++q) {
            if (Stg::on && staged_pass) break;
            const int src = (q < n) ? q : 0;
            const int a = bcast_i(d.x, src), b = bcast_i(d.z, src) & 0x3fffffff;
            const uint32_t va = (uint32_t)bcast_i(d.y, src), vb = (uint32_t)bcast_i(d.w, src);
            if constexpr (ITEM) {
                issue_row<NP>(T.U + (va & 1u) * T.ustride + (size_t)a * T.kp, lane, xa[q]);
                xta[q] = v2u{0u, va};
            } else {
                issue_row<NP>(T.V + (va & T.imask) * T.istride + (size_t)a * T.kp, lane, xa[q]);
                xta[q] = issue_bias(T.tailV + ((size_t)(va & T.imask) * st.n_items + a) * itg);
            }
            issue_row<NP>(T.V + (vb & T.imask) * T.istride + (size_t)b * T.kp, lane, xb[q]);
            xtb[q] = issue_bias(T.tailV + ((size_t)(vb & T.imask) * st.n_items + b) * itg);
        }
        if (!o.ok) {
            bool lane_own = row_tagged<NP>(xo, own_ver) && xt.y == own_ver && xt.w == own_ver;
            if (!sgd) lane_own = lane_own && row_tagged<NP>(xm, own_ver);
            if (__all(lane_own)) {
                o.ok = true;
#ifdef TKR_FLOW_TRACE
                o.t_valid = __builtin_amdgcn_s_memrealtime();
#endif
                row_values<NP>(xo, own);
                if (!sgd) row_values<NP>(xm, ms);
                own_tail_values(xt, o, own_halves);
            }
        }
        bool lane_part = true;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int src = (q < n) ? q : 0;
            const uint32_t va = (uint32_t)bcast_i(d.y, src), vb = (uint32_t)bcast_i(d.w, src);
            lane_part = lane_part && row_tagged<NP>(xa[q], va) && row_tagged<NP>(xb[q], vb) && xtb[q].y == vb && xta[q].y == va;
        }
        const bool part_ok = __all(lane_part);
        // The first pass has waited for its loads, so the ticket taken before them is back too: the record of the NEXT task is
        // fetched now and lands while this task validates, waits and computes -- a task used to start with the round trip of its
        // ticket behind the previous task's write-through stores (0.4 us) and then the round trip of its record (0.55 us),
        // a quarter of a wave's time per task.
        if (!nx.have) feed.prefetch(nx, lane);
        staged_pass = false;
        if (part_ok) break;
        // How far away is what we wait for?  The buffer of version v holds v, v-2, v-4, ... (two buffers per row; v-4, v-8 with
        // four): the tag of the version before means the producer is one or two updates away (poll), an older one at least three
        // -- two whole hand-offs: sleep through that (a waiting wave that polls costs everybody's loads latency, a sleeping one
        // nothing)
        const int far_items = 2 * (int)(T.imask + 1u);
        if (!(T.tune & 64u)) {                       // (tune bit 6: the old way, every pass re-issues everything)
            // which partner row is not there yet?  Poll ITS first pair until it is, then take the pass again
            int mq = -1;
            bool mb = false;
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const int src = (q < n) ? q : 0;
                const uint32_t va = (uint32_t)bcast_i(d.y, src), vb = (uint32_t)bcast_i(d.w, src);
                const bool ra = __all(row_tagged<NP>(xa[q], va) && xta[q].y == va), rb = __all(row_tagged<NP>(xb[q], vb) && xtb[q].y == vb);
                if (mq < 0 && !ra) { mq = q; mb = false; }
                if (mq < 0 && !rb) { mq = q; mb = true; }
            }
            if (mq >= 0) {
                const uint32_t id = (uint32_t)(mb ? bcast_i(d.z, mq) & 0x3fffffff : bcast_i(d.x, mq));
                const uint32_t ver = (uint32_t)(mb ? bcast_i(d.w, mq) : bcast_i(d.y, mq));
                const bool is_user = ITEM && !mb;
                const u64* row = is_user ? T.U + (ver & 1u) * T.ustride + (size_t)id * T.kp : T.V + (ver & T.imask) * T.istride + (size_t)id * T.kp;
                if (!wait_pair(row, ver, is_user ? 4 : far_items, ctl, waited)) return false;
                continue;
            }
        }
        bool far = false;
        if (!(T.tune & 2u)) {
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const int src = (q < n) ? q : 0;
                far = far || (int)((uint32_t)bcast_i(d.y, src) - (uint32_t)bcast_i((int)xa[q][0].y, 0)) >= (ITEM ? 4 : far_items) ||
                      (int)((uint32_t)bcast_i(d.w, src) - (uint32_t)bcast_i((int)xb[q][0].y, 0)) >= far_items;
            }
        }
        if (far) __builtin_amdgcn_s_sleep(127);      // 127 x 64 clocks = 3.4 us
        if (spin_fail(waited, ctl, 4)) {
            if (waited >= kSpinLimit && lane == 0 &&                     // post-mortem of the first wave that gave up
                atomicCAS(ctl + kCtlDebug, 0u, 1u) == 0u) {
                ctl[kCtlDebug + 1] = o.ok;
                ctl[kCtlDebug + 2] = 0u;
                ctl[kCtlDebug + 3] = own_ver;
                ctl[kCtlDebug + 4] = xo[0].y;
                ctl[kCtlDebug + 5] = xt.y;
                ctl[kCtlDebug + 6] = (uint32_t)d.x;
                ctl[kCtlDebug + 7] = (uint32_t)d.y;
                ctl[kCtlDebug + 8] = xa[0][0].y;
                ctl[kCtlDebug + 9] = (uint32_t)d.z;
                ctl[kCtlDebug + 10] = (uint32_t)d.w;
                ctl[kCtlDebug + 11] = xb[0][0].y;
                ctl[kCtlDebug + 12] = xtb[0].y;
                ctl[kCtlDebug + 13] = ITEM;
                ctl[kCtlDebug + 14] = (uint32_t)n;
                ctl[kCtlDebug + 15] = sgd ? 0u : xm[0].y;
            }
            return false;
        }
    }
    spins += waited;
#ifdef TKR_FLOW_TRACE
    o.t_part = __builtin_amdgcn_s_memrealtime();
#endif
    // The partner rows are in registers: acknowledge the reads NOW (lane q: one add on rd[version & 1] of both partner rows of
    // occurrence q), not after this task's own row has arrived too -- the next writers of those rows are waiting for exactly this.
    if (lane < n) {
        uint32_t* pa_rd = ITEM ? T.rdU + 2 * (size_t)d.x + (d.y & 1) : T.rdV + (T.imask + 1u) * (size_t)d.x + ((uint32_t)d.y & T.imask);
        uint32_t* pb_rd = T.rdV + (T.imask + 1u) * (size_t)(d.z & 0x3fffffff) + ((uint32_t)d.w & T.imask);
        __hip_atomic_fetch_add(pa_rd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(pb_rd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // pack: everything the arithmetic needs from the partner rows, 3 registers per occurrence instead of 12
    constexpr int SH = group_shift<G>();
    const int myq = lane >> SH;
    const bool l2 = (st.mode == 0);
    pk.n = n; pk.roles = 0u; pk.lam_sum = 0.f; pk.loss_part = 0.f; pk.bias_me = 0.f; pk.role_me = false;
#pragma unroll
    for (int q = 0; q < G; ++q) {
        float av[NE], bv[NE];
        row_values<NP>(xa[q], av);
        row_values<NP>(xb[q], bv);
        const bool live = q < n;
        const float ta = __uint_as_float(xta[q].x), tb = __uint_as_float(xtb[q].x);
        if constexpr (ITEM) {
            float c = 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) { c = fmaf(av[e], bv[e], c); pk.pa[q][e] = live ? av[e] : 0.f; }
            pk.c[q] = live ? c : 0.f;
            const bool role_j = live && bcast_i(d.z, live ? q : 0) < 0;
            if (live) { pk.roles |= (role_j ? 1u : 0u) << q; pk.lam_sum += role_j ? st.lj : st.li; }
            if (myq == q) { pk.bias_me = tb; pk.role_me = role_j; }
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) pk.pa[q][e] = live ? av[e] - bv[e] : 0.f;
            pk.c[q] = 0.f;
            if (live) pk.lam_sum += st.lu;
            if (myq == q) pk.bias_me = ta - tb;
            if (want_loss && live) {
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    pk.loss_part += l2 ? 0.5f * (av[e] * av[e] * st.li + bv[e] * bv[e] * st.lj) : fabsf(av[e]) * st.li + fabsf(bv[e]) * st.lj;
                if (myq == q && (lane & ((1 << SH) - 1)) == 0)
                    pk.loss_part += l2 ? 0.5f * (ta * ta + tb * tb) * st.lb : (fabsf(ta) + fabsf(tb)) * st.lb;
            }
        }
    }
    return true;
}

// Only the own row is missing -- the state of every task on a chain through a popular row, and what the chain's period is made
// of.  A pass of the general loop of flow_fetch is ~300 instructions of one wave with a SIMD to itself: ~0.4 us on top of the
// loads' round trip, and a link was 1.7 us even with its arithmetic compiled out, against 0.6-0.8 us for the same three stores
// and loads in a bare ping-pong (scripts/ubench/hop_xcd.hip).  This loop is the bare ping-pong: three loads, the tags, one branch.
template <int NP>
__device__ __forceinline__ bool flow_own(const FlowTables& T, int lane, const u64* own_p, const u64* own_ms, const u64* own_tail,
                                         const uint32_t* own_rd, uint32_t own_ver, float (&own)[2 * NP], float (&ms)[2 * NP], Own& o,
                                         bool sgd, uint32_t* ctl, uint32_t& spins, int own_halves) {
    if (!o.ok) {
        v4u xo[NP], xm[NP], xt = {0u, 0u, 0u, 0u};
        uint32_t waited = 0;
        for (;;) {
            issue_row<NP>(own_p, lane, xo);
            if (!sgd) issue_row<NP>(own_ms, lane, xm);
            xt = issue_tail(own_tail, lane & (own_halves - 1), own_halves);
            bool lane_own = row_tagged<NP>(xo, own_ver) && xt.y == own_ver && xt.w == own_ver;
            if (!sgd) lane_own = lane_own && row_tagged<NP>(xm, own_ver);
            if (__all(lane_own)) break;
            // a tag two versions of this buffer back: the producer is at least three updates away -- sleep through two hand-offs
            if (!(T.tune & 2u) && (int)(own_ver - (uint32_t)bcast_i((int)xt.y, 0)) >= 2 * own_halves) __builtin_amdgcn_s_sleep(127);
            if (spin_fail(waited, ctl, 0)) {
                if (waited >= kSpinLimit && lane == 0 && atomicCAS(ctl + kCtlDebug, 0u, 3u) == 0u) {
                    ctl[kCtlDebug + 1] = 0u;
                    ctl[kCtlDebug + 2] = 1u;
                    ctl[kCtlDebug + 3] = own_ver;
                    ctl[kCtlDebug + 4] = xo[0].y;
                    ctl[kCtlDebug + 5] = xt.y;
                }
                return false;
            }
        }
        spins += waited;
        o.ok = true;
#ifdef TKR_FLOW_TRACE
        o.t_valid = __builtin_amdgcn_s_memrealtime();
#endif
        row_values<NP>(xo, own);
        if (!sgd) row_values<NP>(xm, ms);
        own_tail_values(xt, o, own_halves);
    }
    // the acknowledge count is loaded NOW, once, and returns while the gradients are computed (not in every pass: the word is
    // under atomic update by the readers, and the copy of an earlier pass predates the last acknowledgements anyway)
    o.rd = ld_u32(own_rd);
    return true;
}

// the tail of version ver + 1 of a row (bias, its slot, the acknowledge totals: this batch read version ver, so the total of ITS
// buffer grows by the readers of an occurrence: both other tasks of the triplet, or only the user task for an item row under K2o),
// written by lanes 0 .. halves-1
__device__ __forceinline__ void store_tail(u64* tabT, size_t n_rows, int row, uint32_t mask, int lane, bool is_item, float bn, float mbn,
                                           const Own& o, uint32_t ver, int n_occ, uint32_t readers_per_occ = 2u) {
    const int halves = tail_halves(mask);
    const uint32_t nv = ver + 1u, rb = ver & mask, add = readers_per_occ * (uint32_t)n_occ;
    if (lane < halves) {
        v4u tv;
        tv.y = nv; tv.w = nv;
        if (lane == 0) {
            tv.x = is_item ? __float_as_uint(bn) : 0u;
            tv.z = is_item ? __float_as_uint(mbn) : 0u;
        } else if (lane == 1) {
            tv.x = o.exp[0] + (rb == 0u ? add : 0u);
            tv.z = o.exp[1] + (rb == 1u ? add : 0u);
        } else if (lane == 2) {
            tv.x = o.exp[2] + (rb == 2u ? add : 0u);
            tv.z = o.exp[3] + (rb == 3u ? add : 0u);
        } else {
            tv.x = 0u; tv.z = 0u;
        }
        __builtin_amdgcn_raw_buffer_store_b128(tv, row_rsrc(tabT + ((size_t)(nv & mask) * n_rows + row) * (2 * halves), halves * 16), lane * 16, 0,
                                               kAuxStore);
    }
}

// x_t of every occurrence: one dot product each (user row: <u, v_i - v_j>; item row: <u, v_row - v_other>, single/bpr.py:87-89),
// all reduced together; lane L then holds occurrence L >> shift and the sigmoids run side by side.  Then the data terms of the
// gradient per occurrence and the regulariser of the own row once for all of them (n * lambda * own).
template <int NP, int G, bool ITEM>
__device__ __forceinline__ void flow_apply(const tkr_flow_state& st, const FlowTables& T, int lane, const Packed<NP, G>& pk,
                                           const float (&own)[2 * NP], const Own& o, float (&g)[2 * NP], float& gb,
                                           float& loss_lane, bool want_loss) {
    constexpr int NE = 2 * NP;
    constexpr int SH = group_shift<G>();
#ifdef TKR_FLOW_TRACE
    if (T.tune & 4u) return;                      // experiment: no gradients at all (what is the link without its arithmetic?)
#endif
    const bool l2 = (st.mode == 0);
    float part[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        float acc = -pk.c[q];
#pragma unroll
        for (int e = 0; e < NE; ++e) acc = fmaf(pk.pa[q][e], own[e], acc);
        part[q] = acc;
    }
    float dotv;
    const int myq = lane >> SH;
    if constexpr (G == 1) {
        dotv = wave_sum(part[0]);                // every lane
    } else if constexpr (G <= 4) {               // two to four values ride one reduction of four (lane L: value L >> 4)
        float p4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p4[q] = q < G ? part[q] : 0.f;
        dotv = reduce_multi<4>(p4, lane);
    } else {
        dotv = reduce_multi<G>(part, lane);
    }
    const float bd = ITEM ? o.b - pk.bias_me : pk.bias_me;
    const float x_me = (ITEM && pk.role_me) ? -(bd + dotv) : (bd + dotv);
    const float s_me = fast_sigmoid_neg(x_me);
    if constexpr (!ITEM) {
        if (want_loss) {
            if (myq < pk.n && (lane & ((1 << SH) - 1)) == 0) loss_lane += softplus_neg(x_me);
            float own_part = 0.f;
#pragma unroll
            for (int e = 0; e < NE; ++e) own_part += l2 ? 0.5f * own[e] * own[e] * st.lu : fabsf(own[e]) * st.lu;
            loss_lane += pk.loss_part + (float)pk.n * own_part;
        }
    }
#pragma unroll
    for (int q = 0; q < G; ++q) {
        if (q < pk.n) {
            const float s = bcast_f(s_me, q << SH);
            float sg = -s;
            if constexpr (ITEM) {
                if ((pk.roles >> q) & 1u) sg = s;
                gb += sg;
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] = fmaf(sg, pk.pa[q][e], g[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) g[e] = fmaf(pk.lam_sum, l2 ? own[e] : sgn(own[e]), g[e]);
    if constexpr (ITEM) gb = fmaf((float)pk.n * st.lb, l2 ? o.b : sgn(o.b), gb);
}

// All the occurrences of one task.  Up to four sit in the record; five to sixteen come from the occurrence list in groups of
// eight, and the partner rows of BOTH groups are fetched before the own row is waited for; beyond that, group after group.
// `own_step(own, ms, o)`: whatever is still missing of the own row once the partner rows are packed -- GlobalOwn (flow_own: the poll
// loop on the granule tables) or, in csrc/bpr_own.hip, the row's slot in the LDS of the workgroup that owns it.
template <int NP>
struct GlobalOwn {
    const FlowTables& T;
    int lane;
    const u64 *own_p, *own_ms, *own_tail;
    const uint32_t* own_rd;
    uint32_t ver;
    bool sgd;
    uint32_t* ctl;
    uint32_t& spins;
    int own_halves;
    __device__ __forceinline__ bool operator()(float (&own)[2 * NP], float (&ms)[2 * NP], Own& o) {
        return flow_own<NP>(T, lane, own_p, own_ms, own_tail, own_rd, ver, own, ms, o, sgd, ctl, spins, own_halves);
    }
};

template <int NP, bool ITEM, class Src, class OwnStep, int KBIG = 8, class Stg = NoStage>
__device__ __forceinline__ bool run_task(const tkr_flow_state& st, const FlowTables& T, int lane, int n_occ, int first, const int4 w,
                                         const int4* __restrict__ pocc, const u64* own_p, const u64* own_ms, const u64* own_tail,
                                         uint32_t ver, float (&own)[2 * NP], float (&ms)[2 * NP], Own& o,
                                         float (&g)[2 * NP], float& gb, float& loss_lane, bool want_loss, bool sgd, uint32_t* ctl,
                                         uint32_t& spins, NextTask& nx, Src& feed, OwnStep& own_step, Stg stg = Stg()) {
#define TKR_FETCH(GG, nn, dd, PK)                                                                                                  \
    flow_fetch<NP, GG, ITEM>(st, T, lane, nn, dd, own_p, own_ms, own_tail, ver, own, ms, o, want_loss, sgd, ctl, spins, nx, feed, PK)
#define TKR_ONE(GG, nn, dd)                                                                                    \
    {                                                                                                          \
        Packed<NP, GG> pk;                                                                                     \
        if (!flow_fetch<NP, GG, ITEM, Src, Stg>(st, T, lane, nn, dd, own_p, own_ms, own_tail, ver, own, ms, o, want_loss, sgd, ctl, spins, nx, \
                                                feed, pk, stg))                                                \
            return false;                                                                                      \
        if (!own_step(own, ms, o)) return false;                                                               \
        flow_apply<NP, GG, ITEM>(st, T, lane, pk, own, o, g, gb, loss_lane, want_loss);                          \
        return true;                                                                                           \
    }
    if (n_occ <= 4) {                             // the common case: its occurrences sit in the record (lanes 2..5)
        const int src = (lane + 2) & 7;
        const int4 d = make_int4(__shfl(w.x, src), __shfl(w.y, src), __shfl(w.z, src), __shfl(w.w, src));
        switch (n_occ) {
            case 1: TKR_ONE(1, 1, d)
            case 2: TKR_ONE(2, 2, d)
            case 3: TKR_ONE(3, 3, d)
            default: TKR_ONE(4, 4, d)
        }
    }
    constexpr int kBig = KBIG;                    // occurrences per round beyond the record's four (8; K2o's user tasks: 4, fewer registers)
    static_assert(kBig == 4 || kBig == 8, "round width");
    if (n_occ <= 2 * kBig) {
        const int n0 = min(kBig, n_occ), n1 = n_occ - n0;
        int4 d0 = make_int4(0, 0, 0, 0), d1 = make_int4(0, 0, 0, 0);
        if (lane < n0) d0 = pocc[first + lane];
        if (lane < n1) d1 = pocc[first + kBig + lane];
        Packed<NP, kBig> p0, p1;
        if (!TKR_FETCH(kBig, n0, d0, p0)) return false;
        if (n1 > 0 && !TKR_FETCH(kBig, n1, d1, p1)) return false;
        if (!own_step(own, ms, o)) return false;
        flow_apply<NP, kBig, ITEM>(st, T, lane, p0, own, o, g, gb, loss_lane, want_loss);
        if (n1 > 0) flow_apply<NP, kBig, ITEM>(st, T, lane, p1, own, o, g, gb, loss_lane, want_loss);
        return true;
    }
    for (int done = 0; done < n_occ; done += kBig) {          // (no batch of 256 has such a row; batches of thousands do)
        const int n = min(kBig, n_occ - done);
        int4 d = make_int4(0, 0, 0, 0);
        if (lane < n) d = pocc[first + done + lane];
        Packed<NP, kBig> pk;
        if (!TKR_FETCH(kBig, n, d, pk)) return false;
        if (done == 0 && !own_step(own, ms, o)) return false;
        flow_apply<NP, kBig, ITEM>(st, T, lane, pk, own, o, g, gb, loss_lane, want_loss);
    }
    return true;
#undef TKR_ONE
#undef TKR_FETCH
}

}  // namespace tkr
