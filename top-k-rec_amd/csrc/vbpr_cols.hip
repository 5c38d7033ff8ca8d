// K3, column-plan form -- one VBPR mini-batch (single/vbpr.py:50-73,114) in THREE launches with short dependent chains
// (round 3; batch <= 1024, kh % 4 == 0, CSR view of feat).  Same objective and updates as csrc/vbpr_step.hip (read its head
// for the [B, B] pair objective and the S_t / T_t separation); what changes is who waits for whom.
//
//   L1 vbpr_tproject_kernel   P_t = (f_i - f_j).cem, alpha_t, beta_t, uce_u(t)            [needs cem, U, I of batch t-1]
//   L2 vbpr_pairsum_kernel    S_t, T_t, the pair sum of the loss                           [needs every alpha, beta]
//   L3 vbpr_update_kernel     row tasks (sparse RMSProp on [ure|uce], ire, irb: the launch records of K1) AND column tasks
//                             (TF's dense ApplyRMSProp on cem, icb) side by side in ONE grid  [needs S, T]
// Measured and dropped (MI355X, ML-10M shape, d = 20,000): the pair sums INSIDE the update launch -- 64 pair blocks at the head of
// the grid publish S_t, T_t as tagged 8-byte granules (write-through stores, L1-bypassing loads), row and column blocks stage all
// 2B granules into LDS with one coalesced pass per block, repeated until every tag matches.  Correct (all parity tests), but
// the update launch went from 9.7 to 14.4 us where the separate pair launch costs 2.7 us (a dependent launch of a small grid),
// and every consumer polling the granules it needs itself was worse still (62 us per batch: 120 k scattered 8-byte reads of the
// same 4 KB).  An all-to-all hand-off inside a launch is dearer than a kernel boundary on this chip (MI355X_MICROARCH.md,
// rows allgather / boundary); a done-counter bumped by every block of a 1,500-block grid costs 18 us (atomics on one word).
// Round 4, also dropped: the pair sums by the LAST workgroup of the projection to finish (a one-to-all hand-off: every workgroup
// stores its four values write-through, drains them and bumps one counter; whoever counts B - 1 computes S_t, T_t for the batch
// from LDS) -- two launches per batch instead of three.  Correct on every parity test, but the projection went 6.9 -> 14.2 us
// (one thread per (triplet, side), 256 reciprocals each; 25 us with one wave per triplet and its two reductions) where the
// pair launch costs 4.0 us + a 2.7 us boundary: one workgroup does in 7 us what 64 do in 1.3, and every workgroup first waits
// for its write-through stores (scratch/k3_lastwg.patch in the builder's tree; 28.7 vs 28.3 us per batch under rocprofv3).
// Also dropped: no S, T arrays at all -- every task recomputes the pair sums it needs from e^alpha, e^beta held in registers
// (one fma + one reciprocal per term, 16-lane DPP reduction per column group).  Bit-compatible within tolerance, but a column
// ENTRY then costs 2 * 256 terms instead of two loads: 26 M reciprocals per batch against the pair launch's 130 k, and the
// step went from 23.1 to 30.5 us (d = 20,000; 21.1 -> 26.6 us for the dense d = 128 feat).
//
// The four-launch sparse view walked the static CSC of feat per batch: 1.04 M entries read for the 5 % whose item is in the
// batch, through a byte map (is the item in the batch?), a slot table (where is its sum A = sum +-W_t?) and only then the A row
// -- five dependent memory levels, and a dependence on the row kernel that produces A.  Which (triplet, feature column) pairs a
// batch touches depends on the sampled triplets and on the STRUCTURE of feat only, never on a model value, so it is prepared
// beside K1 (tkr_vbpr_colplan, off the step's critical path):
//   tcnt [B], tent [B][2*row_cap]     per triplet: the nonzeros of f_i (+value) followed by those of f_j (-value) -- L1 reads its
//                                     gather list in ONE level instead of ti/tj -> f_ptr -> f_col/f_val
//   colh [d][8]                       per feature column: (entries, first entry in cent, then its first three entries inline)
//   cent [entries of the batch][2]    (t, +-value) grouped by column, runs in (t, side) order
// A column task is header -> uce rows of its entries (two levels; a third only for the 25 % of columns with more than three
// entries): G_cem[c] = sum sv * (-T_t) * uce_u(t), G_icb[c] = sum sv * (-S_t) -- no per-item sums, no dependence on the row tasks.
//
// Roofline: HBM.  Per batch cem and its slot are read and written once (16*d*kh B: TF's dense optimizer touches every element,
// vbpr.py:65,67,73), the plan is read once (~16 B per nonzero of the 2B feature rows), one cem row is gathered per such nonzero (L2
// hits: cem is 5 MB at d = 20,000); bench.py quotes the fraction on those bytes.
#include <stdlib.h>

#include "tkr_common.h"
#include "../../include/tkr.h"
#include "vbpr_rows.h"

extern "C" int tkr_plan_team(int32_t batch_size);
extern "C" int tkr_plan_max_blocks(int32_t batch_size);
extern "C" int64_t tkr_vbpr_workspace_floats(int32_t batch_size, int32_t kh, int32_t d);
extern "C" __attribute__((visibility("hidden"))) int64_t tkr_vbpr_workspace_core_floats(int32_t batch_size, int32_t kh, int32_t d);

namespace tkr {
// the generic form for any kh (csrc/vbpr_wide.hip)
__attribute__((visibility("hidden"))) void vbpr_wide_project(const tkr_vbpr_state& st, const int32_t* ti, const int32_t* tj, const int32_t* tu, const int32_t* tp,
                                                             const int32_t* tc, const int2* te, int tcap, int B, float* P, float* ab2, float* Wm,
                                                             float* loss, hipStream_t s);
__attribute__((visibility("hidden"))) int vbpr_wide_col_blocks(int d);
__attribute__((visibility("hidden"))) void vbpr_wide_update(const tkr_vbpr_state& st, const int32_t* rec, const int2* occ2, const int32_t* occt, const int4* hdr4,
                                                            const float* s_buf, const float* t_buf, const float* P, const float* Wm, const int4* colh,
                                                            const int2* cent, int B, float* loss, hipStream_t s);

// ------------------------------------------------------------------------------------------------------------------------------
// K1-side: one workgroup per batch, wave w OWNS the columns [w*RW, (w+1)*RW) and their counters in LDS.  Every wave reads every
// row of the batch in (t, side) order and keeps the entries that fall into its range: a row's columns are distinct, so the
// lanes of one load never meet in a counter, and successive rows are successive LDS operations of ONE wave -- the order inside
// a column's run is the row order, deterministic without a sort (LDS atomics from many waves would place entries in arrival
// order; runs are long for common features, so "sort it in the consumer" does not scale either).
constexpr int kColWaves = 16;

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

// One pass over the 2B rows of the batch, in row order; CH = 64-entry chunks per row (rows are at most 64 * CH entries long).
// R rows are in flight -- every load of the group is issued before any is used -- but they are CONSUMED row by row, all chunks
// of a row before the next row: the position of an entry inside its column's run is the number of earlier rows that hold the
// column.  (A first version consumed chunk by chunk across the rows of a group; a column in row A's second chunk and row B's first
// then came out B before A -- tests/test_gpu_vbpr.py::test_column_plan_bit_exact caught it.)
template <bool PLACE, int CH>
__device__ __forceinline__ void colplan_pass(const int2* __restrict__ rowinfo, int rows, const int32_t* __restrict__ f_col,
                                             const float* __restrict__ f_val, uint32_t* __restrict__ mine, int lo, int hi, int lane,
                                             int2* __restrict__ cent) {
    constexpr int R = CH <= 2 ? 8 : (CH <= 4 ? 4 : 1);
    for (int g = 0; g < rows; g += R) {
        int col[R][CH];
        float val[R][CH];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int2 ri = (g + q < rows) ? rowinfo[g + q] : make_int2(0, 0);
#pragma unroll
            for (int h = 0; h < CH; ++h) {
                const bool in = 64 * h + lane < ri.y;
                col[q][h] = in ? f_col[ri.x + 64 * h + lane] : -1;
                if constexpr (PLACE) val[q][h] = in ? f_val[ri.x + 64 * h + lane] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
#pragma unroll
            for (int h = 0; h < CH; ++h) {
                if (col[q][h] >= lo && col[q][h] < hi) {
                    if constexpr (PLACE) {
                        const uint32_t pos = atomicAdd(&mine[col[q][h] - lo], 1u);
                        const int r = g + q;
                        cent[pos] = make_int2(r >> 1, __float_as_int((r & 1) ? -val[q][h] : val[q][h]));
                    } else {
                        atomicAdd(&mine[col[q][h] - lo], 1u);
                    }
                }
            }
        }
    }
}

template <int CH>
__global__ __launch_bounds__(kColWaves * 64) void vbpr_colplan_kernel(
    const int32_t* __restrict__ f_ptr, const int32_t* __restrict__ f_col, const float* __restrict__ f_val, int d,
    const int32_t* __restrict__ ti_all, const int32_t* __restrict__ tj_all, int B, int row_cap, int4* __restrict__ colh_all,
    int2* __restrict__ cent_all, int32_t* __restrict__ tcnt_all, int2* __restrict__ tent_all) {
    extern __shared__ __attribute__((aligned(16))) uint32_t colplan_sm[];
    uint32_t* const sm = colplan_sm;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rows = 2 * B, tcap = 2 * row_cap;
    int2* rowinfo = reinterpret_cast<int2*>(sm);                       // [rows] (first nonzero, count)
    uint32_t* tot = sm + 2 * rows;                                     // [kColWaves]
    uint32_t* cur = tot + kColWaves;                                   // [kColWaves * RW]
    const int RW = (d + kColWaves - 1) / kColWaves;
    const int lo = min(d, w * RW), hi = min(d, lo + RW);
    uint32_t* mine = cur + (size_t)w * RW;
    const int32_t* ti = ti_all + (size_t)b * B;
    const int32_t* tj = tj_all + (size_t)b * B;
    for (int r = tid; r < rows; r += kColWaves * 64) {
        const int item = (r & 1) ? tj[r >> 1] : ti[r >> 1];
        const int s0 = f_ptr[item];
        rowinfo[r] = make_int2(s0, min(f_ptr[item + 1] - s0, row_cap));
    }
    for (int c = lane; c < hi - lo; c += 64) mine[c] = 0u;
    __syncthreads();
    // the gather list of every triplet for L1: f_i's nonzeros (+value), then f_j's (-value)
    int2* tent = tent_all + (size_t)b * B * tcap;
    for (int t = w; t < B; t += kColWaves) {
        const int2 ri = rowinfo[2 * t], rj = rowinfo[2 * t + 1];
        int2* dst = tent + (size_t)t * tcap;
        for (int e = lane; e < ri.y; e += 64) dst[e] = make_int2(f_col[ri.x + e], __float_as_int(f_val[ri.x + e]));
        for (int e = lane; e < rj.y; e += 64) dst[ri.y + e] = make_int2(f_col[rj.x + e], __float_as_int(-f_val[rj.x + e]));
        if (lane == 0) tcnt_all[(size_t)b * B + t] = ri.y + rj.y;
    }
    int2* cent = cent_all + (size_t)b * B * tcap;
    colplan_pass<false, CH>(rowinfo, rows, f_col, f_val, mine, lo, hi, lane, cent);
    // exclusive scan of the own range; the ranges are consecutive, so the base of range w is the sum of the totals before it
    uint32_t carry = 0;
    for (int c0 = 0; c0 < hi - lo; c0 += 64) {
        const int c = c0 + lane;
        const uint32_t v = c < hi - lo ? mine[c] : 0u;
        uint32_t sum;
        const uint32_t ex = wave_excl_scan(v, lane, sum);
        if (c < hi - lo) mine[c] = carry + ex;
        carry += sum;
    }
    if (lane == 0) tot[w] = carry;
    __syncthreads();
    uint32_t base = 0;
    for (int q = 0; q < w; ++q) base += tot[q];
    for (int c = lane; c < hi - lo; c += 64) mine[c] += base;
    colplan_pass<true, CH>(rowinfo, rows, f_col, f_val, mine, lo, hi, lane, cent);
    // headers: after the placement mine[c] is the END of column c's run; its start is the end of the column before it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's entries are written: it reads the first three of each run back
    int4* colh = colh_all + (size_t)b * d * 2;
    for (int c0 = 0; c0 < hi - lo; c0 += 64) {
        const int c = c0 + lane;
        if (c < hi - lo) {
            const int end = (int)mine[c];
            const int beg = c > 0 ? (int)mine[c - 1] : (int)base;
            const int n = end - beg;
            int2 e0 = make_int2(0, 0), e1 = e0, e2 = e0;
            if (n > 0) e0 = cent[beg];
            if (n > 1) e1 = cent[beg + 1];
            if (n > 2) e2 = cent[beg + 2];
            colh[(size_t)(lo + c) * 2] = make_int4(n, beg, e0.x, e0.y);
            colh[(size_t)(lo + c) * 2 + 1] = make_int4(e1.x, e1.y, e2.x, e2.y);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// L1: one workgroup of W waves per triplet.  Entry e of the triplet's gather list goes to wave e % W, lane (e / W) & 63: a
// list of up to 64 * W entries is ONE load per lane, issued together with the triplet's rows ([ure|uce]_u, ire_i, ire_j, biases:
// wave 0; parities from K1's tpar); then every wave gathers one kh-wide cem row (+ icb) per entry, 32 in flight, and the four
// partial sums are added in wave order.  Wave 0 forms alpha_t, beta_t, keeps uce_u for the column tasks and the regularisers'
// share of the loss -- what vbpr_sproject_kernel does, with two dependent memory levels instead of four.
template <int NH, int W>
__global__ __launch_bounds__(W * 64) void vbpr_tproject_kernel(tkr_vbpr_state st, const int32_t* __restrict__ ti,
                                                           const int32_t* __restrict__ tj, const int32_t* __restrict__ tu,
                                                           const int32_t* __restrict__ tpar, const int32_t* __restrict__ tcnt,
                                                           const int2* __restrict__ tent, int tcap, int B, float* __restrict__ P,
                                                           float* __restrict__ ab_out, float* __restrict__ Wraw,
                                                           float* __restrict__ loss_out, int tune) {
    __shared__ float red[W][NH * 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = blockIdx.x;
    const int kh = st.kh;
    const int2* list = tent + (size_t)t * tcap;
    int n = tcnt[t];                                                   // same level as the speculative first round of entries
    if (tune & 1) n = 0;                                               // (TKR_VBPR_TUNE: timing experiments, scripts/probe_vbpr.py)
    int2 ent = list[min(W * lane + wave, tcap - 1)];
    float ure[NH], uce[NH], vi[NH], vj[NH], bi = 0.f, bj = 0.f;
    if (wave == 0 && !(tune & 2)) {
        const int pr = tpar[t], k2 = 2 * kh;
        const int u = tu[t], i = ti[t], j = tj[t];
        const float* urow = st.U + ((size_t)(pr & 1) * st.n_users + u) * k2;
        const float* ri = st.I + ((size_t)((pr >> 1) & 1) * st.n_items + i) * kh;
        const float* rj = st.I + ((size_t)((pr >> 2) & 1) * st.n_items + j) * kh;
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const int c = min(lane + e * 64, kh - 1);
            const bool ok = lane + e * 64 < kh;
            const float a = urow[c], b = urow[kh + c], x = ri[c], y = rj[c];
            ure[e] = ok ? a : 0.f; uce[e] = ok ? b : 0.f; vi[e] = ok ? x : 0.f; vj[e] = ok ? y : 0.f;
        }
        bi = st.irb[(size_t)((pr >> 1) & 1) * st.n_items + i];
        bj = st.irb[(size_t)((pr >> 2) & 1) * st.n_items + j];
    }
    float acc[NH], q = 0.f;
#pragma unroll
    for (int e = 0; e < NH; ++e) acc[e] = 0.f;
    for (int e0 = 0; e0 < n; e0 += 64 * W) {                           // one round for lists of up to 64 * W entries
        if (e0) ent = list[min(e0 + W * lane + wave, tcap - 1)];
        const int mine_n = (min(n - e0, 64 * W) - wave + W - 1) / W;   // entries of this round that fell to this wave
        const int col = lane < mine_n ? ent.x : 0;
        const float val = lane < mine_n ? __int_as_float(ent.y) : 0.f;
        constexpr int UN = NH == 1 ? 32 : 16;
        for (int g0 = 0; g0 < mine_n; g0 += UN) {
            float rowv[UN][NH], bv[UN], vv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int c = __builtin_amdgcn_readlane(col, (g0 + u) & 63);
                vv[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), (g0 + u) & 63));
                const float* crow = st.cem + (size_t)c * kh;
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) rowv[u][hh] = crow[min(lane + hh * 64, kh - 1)];
                bv[u] = st.icb[c];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) acc[hh] = fmaf(vv[u], rowv[u][hh], acc[hh]);
                q = fmaf(vv[u], bv[u], q);
            }
        }
    }
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) red[wave][lane + hh * 64] = acc[hh];
    if (lane == 0) red[wave][NH * 64] = q;
    __syncthreads();
    if (wave != 0) return;
    float p[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        const int n2 = lane + hh * 64;
        float a = red[0][n2];
#pragma unroll
        for (int w = 1; w < W; ++w) a += red[w][n2];                   // wave order: a fixed summation order
        p[hh] = a;
        if (n2 < kh) P[(size_t)t * kh + n2] = p[hh];
        else p[hh] = 0.f;
    }
    float qsum = red[0][NH * 64];
#pragma unroll
    for (int w = 1; w < W; ++w) qsum += red[w][NH * 64];
    float d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int e = 0; e < NH; ++e) {
        d1 = fmaf(ure[e], vi[e] - vj[e], d1);
        d2 = fmaf(uce[e], p[e], d2);
    }
    const float alpha = bi - bj + qsum, beta = wave_sum(d1) + wave_sum(d2);
#pragma unroll
    for (int e = 0; e < NH; ++e) {
        const int c = lane + e * 64;
        if (c < kh) Wraw[(size_t)t * kh + c] = uce[e];
    }
    if (lane == 0) { ab_out[t] = alpha; ab_out[B + t] = beta; ab_out[2 * B + t] = pair_exp(alpha); ab_out[3 * B + t] = pair_exp(beta); }
    if (loss_out) {
        const bool l2 = st.mode == 0;
        float loss = 0.f, loss_lane = 0.f;
        if (l2) {
            loss += 0.5f * (bi * bi + bj * bj) * st.lb;
#pragma unroll
            for (int e = 0; e < NH; ++e)
                loss_lane += 0.5f * ((ure[e] * ure[e] + uce[e] * uce[e]) * st.lu + vi[e] * vi[e] * st.li + vj[e] * vj[e] * st.lj);
        } else {
            loss += (fabsf(bi) + fabsf(bj)) * st.lb;
#pragma unroll
            for (int e = 0; e < NH; ++e)
                loss_lane += (fabsf(ure[e]) + fabsf(uce[e])) * st.lu + fabsf(vi[e]) * st.li + fabsf(vj[e]) * st.lj;
        }
        const float tot = wave_sum(loss_lane) + loss;
        if (lane == 0) loss_out[t] = tot;                            // this triplet's word (vbpr_loss_sum_kernel adds a batch's words up)
    }
}

// L2: the [B, B] pair sums (csrc/vbpr_step.hip, head).  One wave per triplet t, every alpha / beta the wave needs loaded up front
// (B <= 1024: 16 per lane and array):  S_t = sum_b sigma(-(alpha_t + beta_b)),  T_t = sum_a sigma(-(alpha_a + beta_t)).
// FRESH: the sums are read by other blocks of the SAME launch (the pair blocks of vbpr_update_kernel): stored write-through
template <bool FRESH>
__device__ __forceinline__ void pairsum_wave(const float* __restrict__ ab /*[4][B]*/, int B, float* sS, float* sT, float* __restrict__ loss_out,
                                             int t, int lane) {
    if (t >= B) return;
    const float* beta = ab + B;
    const float* ealpha = ab + 2 * B;                                // e^alpha, e^beta (pair_exp, tkr_common.h)
    const float* ebeta = ab + 3 * B;
    const float a_t = ab[t], ea_t = ealpha[t], eb_t = ebeta[t];
    float al[16], be[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        al[r] = 0.f; be[r] = 0.f;
        if (64 * r < B) {                                            // uniform
            const int o = min(lane + 64 * r, B - 1);
            al[r] = ealpha[o];
            be[r] = ebeta[o];
        }
    }
    float s_row = 0.f, s_col = 0.f, loss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (64 * r < B) {                                            // uniform
            const bool in = lane + 64 * r < B;
            s_row += in ? pair_sigmoid(ea_t, be[r]) : 0.f;
            if (loss_out) loss += in ? pair_softplus_neg(ea_t, be[r], a_t + beta[min(lane + 64 * r, B - 1)]) : 0.f;
            s_col += in ? pair_sigmoid(al[r], eb_t) : 0.f;
        }
    }
    s_row = wave_sum(s_row);
    s_col = wave_sum(s_col);
    if (lane == 0) {
        if constexpr (FRESH) {
            __hip_atomic_store(sS + t, s_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sT + t, s_col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            sS[t] = s_row; sT[t] = s_col;
        }
    }
    if (loss_out) {
        loss = wave_sum(loss);
        if (lane == 0) loss_out[B + t] = loss;
    }
}
__global__ __launch_bounds__(256) void vbpr_pairsum_kernel(const float* __restrict__ ab /*[4][B]*/, int B, float* __restrict__ sS,
                                                          float* __restrict__ sT, float* __restrict__ loss_out) {
    pairsum_wave<false>(ab, B, sS, sT, loss_out, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// L3.  Blocks [0, n_row_blocks) run the row tasks of the batch (vbpr_rows_body: the launch records of K1), the others the
// column tasks: LPC lanes hold one cem row as float4 each (kh <= 4 * LPC, kh % 4 == 0), G = 256 / LPC column groups per block,
// of which the first `cpb` own a column each.  A run of up to kLightRun entries is summed by its group alone (the first three
// ride in the header, the rest four in flight); longer runs (a common feature; every column of a narrow dense feat) are split
// over all G groups of the block, the partial sums combined in group order (fixed summation order either way: results are
// run-to-run identical).
constexpr int kLightRun = 32;

// UN entries of a run in flight at a time: 4 behind a light run's header; 16 on the long runs of a narrow dense feat (a column meets
// all 2B triplet rows: 32 entries per group -- at 4 per round that was eight dependent trips, 8 of the update's 11 us at d_c = 128)
template <int LPC, class PS, int UN = 4>
__device__ __forceinline__ void col_accumulate(const int2* __restrict__ cent, int first, int last, const PS& ps,
                                               const float* __restrict__ Wraw, int kh, int gl, bool live, float4& g, float& gi) {
    for (int p = first; p < last; p += UN) {
        int2 e[UN];
        int et[UN];
        float tt[UN], ss[UN];
        float4 wr[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) { e[q] = cent[min(p + q, last - 1)]; et[q] = e[q].x; }
#pragma unroll
        for (int q = 0; q < UN; ++q)
            wr[q] = live ? *reinterpret_cast<const float4*>(Wraw + (size_t)e[q].x * kh + 4 * gl) : make_float4(0.f, 0.f, 0.f, 0.f);
        ps.get(et, UN, ss, tt);
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const float sv = (p + q < last) ? __int_as_float(e[q].y) : 0.f;
            const float a = -sv * tt[q];
            g.x = fmaf(a, wr[q].x, g.x); g.y = fmaf(a, wr[q].y, g.y); g.z = fmaf(a, wr[q].z, g.z); g.w = fmaf(a, wr[q].w, g.w);
            gi = fmaf(-sv, ss[q], gi);
        }
    }
}

template <int LPC, class PS, int UNL = 4>
__device__ __forceinline__ void col_block(const tkr_vbpr_state& st, const PS& ps, const float* __restrict__ Wraw,
                                          const int4* __restrict__ colh, const int2* __restrict__ cent, int cb, int cpb,
                                          float* __restrict__ loss_out, int tune, float* shm) {
    constexpr int G = 256 / LPC;
    const int tid = threadIdx.x, grp = tid / LPC, gl = tid % LPC, lane = tid & 63;
    const int kh = st.kh, d = st.d;
    const int c = cb * cpb + grp;
    const bool own = grp < cpb && c < d;
    const bool live = 4 * gl < kh;                               // lanes past the row's width carry zeros
    float* part = shm;                                           // [G][4 * LPC + 1]
    int* runs = reinterpret_cast<int*>(shm + G * (4 * LPC + 1)); // [G][2]: (first entry, count) of every group's column
    int4 h0 = make_int4(0, 0, 0, 0), h1 = h0;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), pms = pv;
    float bv0 = 0.f, bms0 = 0.f;
    if (own) {
        h0 = colh[(size_t)c * 2];
        h1 = colh[(size_t)c * 2 + 1];
        if (live) {
            pv = *reinterpret_cast<const float4*>(st.cem + (size_t)c * kh + 4 * gl);
            pms = *reinterpret_cast<const float4*>(st.mscem + (size_t)c * kh + 4 * gl);
        }
        bv0 = st.icb[c];
        bms0 = st.msicb[c];
    }
    const int n = (tune & 4) ? 0 : h0.x, beg = h0.y;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float gi = 0.f;
    // ---- light runs: the group alone; entries 0..2 came with the header, the others four in flight
    const bool light = own && n > 0 && n <= kLightRun;
    const int et[3] = {h0.z, h1.x, h1.z};
    const int ev[3] = {h0.w, h1.y, h1.w};
    int2 more[4] = {make_int2(0, 0), make_int2(0, 0), make_int2(0, 0), make_int2(0, 0)};
    float4 wr[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (light) {
        if (n > 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) more[q] = cent[beg + min(3 + q, n - 1)];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int t = q < n ? et[q] : et[0];
            wr[q] = live ? *reinterpret_cast<const float4*>(Wraw + (size_t)t * kh + 4 * gl) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // (pair sums formed by the first blocks of THIS launch: every wave of the block comes by here, with the loads above in flight)
    pair_sums_wait(ps);
    if (light) {
        float tt[3], ss[3];
        ps.get(et, n, ss, tt);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float sv = q < n ? __int_as_float(ev[q]) : 0.f;
            const float a = -sv * tt[q];
            g.x = fmaf(a, wr[q].x, g.x); g.y = fmaf(a, wr[q].y, g.y); g.z = fmaf(a, wr[q].z, g.z); g.w = fmaf(a, wr[q].w, g.w);
            gi = fmaf(-sv, ss[q], gi);
        }
        if (n > 3) {
            float t4[4], s4[4];
            float4 w4[4];
            int mt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mt[q] = more[q].x;
                w4[q] = live ? *reinterpret_cast<const float4*>(Wraw + (size_t)more[q].x * kh + 4 * gl) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            ps.get(mt, 4, s4, t4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sv = (3 + q < n) ? __int_as_float(more[q].y) : 0.f;
                const float a = -sv * t4[q];
                g.x = fmaf(a, w4[q].x, g.x); g.y = fmaf(a, w4[q].y, g.y); g.z = fmaf(a, w4[q].z, g.z); g.w = fmaf(a, w4[q].w, g.w);
                gi = fmaf(-sv, s4[q], gi);
            }
            if (n > 7) col_accumulate<LPC, PS>(cent, beg + 7, beg + n, ps, Wraw, kh, gl, live, g, gi);
        }
    }
    // ---- long runs: all G groups of the block on one column at a time
    if (gl == 0) { runs[2 * grp] = beg; runs[2 * grp + 1] = own ? n : 0; }
    __syncthreads();
    for (int q = 0; q < cpb; ++q) {
        const int nq = runs[2 * q + 1];
        if (nq <= kLightRun) continue;                           // uniform over the block
        const int bq = runs[2 * q];
        const int chunk = (nq + G - 1) / G;
        const int lo = min(nq, grp * chunk), hi = min(nq, lo + chunk);
        float4 pg = make_float4(0.f, 0.f, 0.f, 0.f);
        float pgi = 0.f;
        col_accumulate<LPC, PS, UNL>(cent, bq + lo, bq + hi, ps, Wraw, kh, gl, live, pg, pgi);
        float* mp = part + grp * (4 * LPC + 1);
        mp[4 * gl + 0] = pg.x; mp[4 * gl + 1] = pg.y; mp[4 * gl + 2] = pg.z; mp[4 * gl + 3] = pg.w;
        if (gl == 0) mp[4 * LPC] = pgi;
        __syncthreads();
        if (grp == q) {                                          // the column's own group adds the partial sums in group order
            for (int x = 0; x < G; ++x) {
                const float* sp = part + x * (4 * LPC + 1);
                g.x += sp[4 * gl + 0]; g.y += sp[4 * gl + 1]; g.z += sp[4 * gl + 2]; g.w += sp[4 * gl + 3];
                gi += sp[4 * LPC];
            }
        }
        __syncthreads();
    }
    // ---- TF's dense ApplyRMSProp on cem[c][.] and icb[c] (vbpr.py:65,67,73), exactly as V3 / S3 of csrc/vbpr_step.hip
    const bool l2 = st.mode == 0;
    float lpart = 0.f;
    if (own && live && !(tune & 16)) {
        float gv[4] = {g.x, g.y, g.z, g.w}, v[4] = {pv.x, pv.y, pv.z, pv.w}, ms[4] = {pms.x, pms.y, pms.z, pms.w};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float gg = gv[x] + st.le * (l2 ? v[x] : sgn(v[x]));
            lpart += l2 ? 0.5f * st.le * v[x] * v[x] : st.le * fabsf(v[x]);
            ms[x] += (gg * gg - ms[x]) * (1.f - st.rho);
            v[x] = v[x] - st.lr * gg / sqrtf(ms[x] + st.eps);
        }
        *reinterpret_cast<float4*>(st.mscem + (size_t)c * kh + 4 * gl) = make_float4(ms[0], ms[1], ms[2], ms[3]);
        *reinterpret_cast<float4*>(st.cem + (size_t)c * kh + 4 * gl) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (own && gl == 0) {
        const float v = bv0;
        const float gg = gi + st.lb * (l2 ? v : sgn(v));
        lpart += l2 ? 0.5f * st.lb * v * v : st.lb * fabsf(v);
        float ms = bms0;
        ms += (gg * gg - ms) * (1.f - st.rho);
        st.msicb[c] = ms;
        st.icb[c] = v - st.lr * gg / sqrtf(ms + st.eps);
    }
    if (loss_out) {                                              // the block's word: its four waves in wave order
        lpart = wave_sum(lpart);
        __syncthreads();                                         // (shm: the long runs are done with it)
        if (lane == 0) shm[tid >> 6] = lpart;
        __syncthreads();
        if (tid == 0) loss_out[cb] = (shm[0] + shm[1]) + (shm[2] + shm[3]);
    }
}

// INLINE: no pair-sum launch in front of this one -- rows and columns work S_t / T_t out where they need them (PairSumInline; s_in
// then points at the batch's [alpha | beta | e^alpha | e^beta]), and B / 4 more blocks at the end of the grid add up the pair terms
// of the loss, one wave per triplet (what vbpr_pairsum_kernel did beside its sums).
// FUSED (round 5): no pair-sum launch either, and nobody recomputes anything -- the first (B + 3) / 4 blocks of the grid ARE the pair-sum
// kernel (same code, same order of summation: S_t, T_t and the loss words are bit for bit those of the three-launch form), store
// their sums write-through, drain, and bump `pair_done`; every other block runs its chain (header -> entries -> uce rows; the rows'
// records and occurrences) and only looks at the counter where it first needs a sum (PairSumFresh) -- by then, ~2.5 us into the
// launch, the pair blocks (~3 us) are done or nearly so.  s_in = S, t_in = T as in the three-launch form, ab = the batch's
// [alpha | beta | e^alpha | e^beta].  Two launches per batch.
// MEASURED (MI355X, ML-10M shape, batch 256, per batch without the loss): sparse d = 20,000: 25.4 us against 22.7 with the pair-sum
// launch; dense d_c = 128: 23.6 against 18.0 (first version, every wave polling the counter and reading the sums past the L1: 28.6 /
// 21.8).  The pair blocks are not done at ~3 us but at ~5 (two trips for their inputs, the write-through stores' drain, the
// counter's atomic), the waiters see it a poll later, and the update's own chain behind the sums (sums -> entries 4..7 -> stores) is
// still ahead of them: what overlaps is the 2.5 us of header + first gathers, what is added is the hand-off.  The third way of
// removing the pair launch that lost (after round 4's last-block sums and round 5's inline sums): selectable (tkr_vbpr_set_pairs(2)),
// bit-identical, not the default.
template <int NE, int LPC, bool INLINE, int UNL = 4, bool FUSED = false>
__global__ __launch_bounds__(256) void vbpr_update_kernel(
    tkr_vbpr_state st, const int32_t* __restrict__ rec_all, const int2* __restrict__ occ, const int32_t* __restrict__ occt,
    const int4* __restrict__ hdr, const float* __restrict__ s_in, const float* __restrict__ t_in, const float* __restrict__ P,
    const float* __restrict__ Wraw /*[B][kh]: uce_u(t)*/, const int4* __restrict__ colh, const int2* __restrict__ cent,
    int n_row_blocks, int n_col_blocks, int cpb, float* __restrict__ loss_out /*the batch's loss words: [B] projection | [B] pair sums | [column blocks]*/,
    int ps_B /*batch size*/, int tune, const float* __restrict__ ab = nullptr, uint32_t* pair_done = nullptr, uint32_t pair_target = 0u) {
    constexpr int G = 256 / LPC;
    constexpr int ROWS_LDS = 2 * 4 * (NE * TKR_WAVE + 1);
    constexpr int COLS_LDS = G * (4 * LPC + 1) + 2 * G;
    __shared__ float shm[ROWS_LDS > COLS_LDS ? ROWS_LDS : COLS_LDS];
    typedef float (*red_t)[NE * TKR_WAVE + 1];
    if constexpr (FUSED) {
        static_assert(!INLINE, "one way of getting the pair sums");
        const int n_pair = (ps_B + 3) / 4;
        if ((int)blockIdx.x < n_pair) {
            pairsum_wave<true>(ab, ps_B, const_cast<float*>(s_in), const_cast<float*>(t_in), loss_out, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the write-through stores are in memory
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(pair_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const int bid = (int)blockIdx.x - n_pair;
        const bool past_l1 = (ps_B & 31) != 0;
        if (bid < n_row_blocks) {
            if (tune & 8) return;
            const PairSumFresh ps{s_in, t_in, pair_done, pair_target, nullptr, past_l1};
            vbpr_rows_body<NE, 4>(st, rec_all, occ, occt, hdr, ps, P, nullptr, nullptr, nullptr, reinterpret_cast<red_t>(shm),
                                  reinterpret_cast<red_t>(shm + 4 * (NE * TKR_WAVE + 1)), bid, n_row_blocks);
            return;
        }
        __shared__ uint32_t pairs_here;                                // wave 0 of the block tells the others (PairSumFresh)
        if (threadIdx.x == 0) pairs_here = 0u;
        __syncthreads();
        const PairSumFresh ps{s_in, t_in, pair_done, pair_target, &pairs_here, past_l1};
        col_block<LPC, PairSumFresh, UNL>(st, ps, Wraw, colh, cent, bid - n_row_blocks, cpb, loss_out ? loss_out + 2 * ps_B : nullptr, tune, shm);
    } else if constexpr (INLINE) {
        const float* ab = s_in;
        if ((int)blockIdx.x >= n_row_blocks + n_col_blocks) {        // the pair terms of the loss: wave per triplet
            const int lane = threadIdx.x & 63, t = ((int)blockIdx.x - n_row_blocks - n_col_blocks) * 4 + (threadIdx.x >> 6);
            if (t >= ps_B || !loss_out) return;
            const float a_t = ab[t], ea_t = ab[2 * ps_B + t];
            float loss = 0.f;
            for (int b = lane; b < ps_B; b += 64) loss += pair_softplus_neg(ea_t, ab[3 * ps_B + b], a_t + ab[ps_B + b]);
            loss = wave_sum(loss);
            if (lane == 0) loss_out[ps_B + t] = loss;
            return;
        }
        if ((int)blockIdx.x < n_row_blocks) {
            if (tune & 8) return;
            PairSumInline<64> ps;
            ps.load(ab + 2 * ps_B, ab + 3 * ps_B, ps_B);
            vbpr_rows_body<NE, 4>(st, rec_all, occ, occt, hdr, ps, P, nullptr, nullptr, nullptr, reinterpret_cast<red_t>(shm),
                                  reinterpret_cast<red_t>(shm + 4 * (NE * TKR_WAVE + 1)), blockIdx.x, n_row_blocks);
            return;
        }
        PairSumInline<LPC> ps;
        ps.load(ab + 2 * ps_B, ab + 3 * ps_B, ps_B);
        col_block<LPC>(st, ps, Wraw, colh, cent, (int)blockIdx.x - n_row_blocks, cpb, loss_out ? loss_out + 2 * ps_B : nullptr, tune, shm);
    } else {
        const PairSumArrays ps{s_in, t_in};
        if ((int)blockIdx.x < n_row_blocks) {
            if (tune & 8) return;
            vbpr_rows_body<NE, 4>(st, rec_all, occ, occt, hdr, ps, P, nullptr, nullptr, nullptr, reinterpret_cast<red_t>(shm),
                                  reinterpret_cast<red_t>(shm + 4 * (NE * TKR_WAVE + 1)), blockIdx.x, n_row_blocks);
            return;
        }
        col_block<LPC, PairSumArrays, UNL>(st, ps, Wraw, colh, cent, (int)blockIdx.x - n_row_blocks, cpb, loss_out ? loss_out + 2 * ps_B : nullptr, tune, shm);
    }
}

template <int NE, int LPC>
static void launch_update(const tkr_vbpr_state& st, const int32_t* rec, const int2* occ2, const int32_t* occt, const int4* hdr4,
                          const float* s_buf, const float* t_buf, const float* P, const float* Wm, const int4* colh, const int2* cent,
                          int B, int cpb, float* loss, hipStream_t stream, int tune, const float* ab_inline /*or null: S / T from s_buf / t_buf*/,
                          bool long_runs, const float* ab_fused = nullptr /*the pair sums by the first blocks of this launch*/,
                          uint32_t* pair_done = nullptr, uint32_t pair_target = 0u) {
    constexpr int G = 256 / LPC;
    if (cpb <= 0 || cpb > G) cpb = G;
    const int n_row_blocks = vbpr_grid(B, 4);
    const int n_col_blocks = (st.d + cpb - 1) / cpb;
#ifdef TKR_LAB                                                   // the two placements of the pair sums that lost to a launch of their own: make LAB=1
    if (ab_fused) {
        const dim3 grid((B + 3) / 4 + n_row_blocks + n_col_blocks);
        if (long_runs)
            hipLaunchKernelGGL((vbpr_update_kernel<NE, LPC, false, 16, true>), grid, dim3(256), 0, stream, st, rec, occ2, occt, hdr4, s_buf, t_buf, P, Wm, colh,
                               cent, n_row_blocks, n_col_blocks, cpb, loss, B, tune, ab_fused, pair_done, pair_target);
        else
            hipLaunchKernelGGL((vbpr_update_kernel<NE, LPC, false, 4, true>), grid, dim3(256), 0, stream, st, rec, occ2, occt, hdr4, s_buf, t_buf, P, Wm, colh,
                               cent, n_row_blocks, n_col_blocks, cpb, loss, B, tune, ab_fused, pair_done, pair_target);
    } else if (ab_inline)
        hipLaunchKernelGGL((vbpr_update_kernel<NE, LPC, true>), dim3(n_row_blocks + n_col_blocks + (loss ? (B + 3) / 4 : 0)), dim3(256), 0, stream, st, rec,
                           occ2, occt, hdr4, ab_inline, nullptr, P, Wm, colh, cent, n_row_blocks, n_col_blocks, cpb, loss, B, tune);
    else
#endif
    if (long_runs)           // a narrow dense feat: every column's run is split over the block's groups, 16 entries in flight (168 registers; 32: slower, 21.5 vs 19.7 us)
        hipLaunchKernelGGL((vbpr_update_kernel<NE, LPC, false, 16>), dim3(n_row_blocks + n_col_blocks), dim3(256), 0, stream, st, rec, occ2, occt, hdr4,
                           s_buf, t_buf, P, Wm, colh, cent, n_row_blocks, n_col_blocks, cpb, loss, B, tune);
    else
        hipLaunchKernelGGL((vbpr_update_kernel<NE, LPC, false>), dim3(n_row_blocks + n_col_blocks), dim3(256), 0, stream, st, rec, occ2, occt, hdr4,
                           s_buf, t_buf, P, Wm, colh, cent, n_row_blocks, n_col_blocks, cpb, loss, B, tune);
}

}  // namespace tkr

namespace tkr {
// loss_out[b] += the `count` loss words of batch b: one per triplet from the projection, one per triplet from the pair sums, one per
// column block of the update -- plain stores of their tasks (64 spread atomic slots cost the step 2.9 of 25.7 us: ~5,500 atomics per
// batch, each a memory operation its wave's next load waits behind).  Fixed order: thread t adds words t, t + 256, ..., then the
// wave tree, then the four waves -- the loss of a batch is bitwise reproducible.
__global__ __launch_bounds__(256) void vbpr_loss_sum_kernel(const float* __restrict__ words, size_t stride, int count, float* __restrict__ loss_out) {
    __shared__ float part[4];
    const int b = blockIdx.x;
    const float* w = words + (size_t)b * stride;
    float acc = 0.f;
    for (int q = threadIdx.x; q < count; q += 256) acc += w[q];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[b] += (part[0] + part[1]) + (part[2] + part[3]);
}
}  // namespace tkr

// where the pair sums S_t, T_t of a batch come from: 0 = their own launch between the projection and the update (three launches
// per batch), 1 = every task works out the ones it needs (PairSumInline, batch <= 256), 2 = the first blocks of the update launch
// (vbpr_update_kernel FUSED).  Initial value: TKR_VBPR_PAIRS, else kVbprPairsDefault.
namespace tkr {
constexpr int kVbprPairsDefault = 0;
static int g_vbpr_pairs = -1;
static int vbpr_pairs_mode() {
    if (g_vbpr_pairs < 0) {
        const char* e = getenv("TKR_VBPR_PAIRS");
        const int v = e ? atoi(e) : kVbprPairsDefault;
#ifdef TKR_LAB
        g_vbpr_pairs = (v >= 0 && v <= 2) ? v : kVbprPairsDefault;
#else
        (void)v;
        g_vbpr_pairs = kVbprPairsDefault;
#endif
    }
    return g_vbpr_pairs;
}
}  // namespace tkr
extern "C" int tkr_vbpr_set_pairs(int32_t mode) {
    if (mode < 0 || mode > 2) return TKR_EINVAL;
#ifndef TKR_LAB
    if (mode != 0) return TKR_EUNSUPPORTED;                      // placements 1 and 2 live in the lab library (make LAB=1)
#endif
    tkr::g_vbpr_pairs = mode;
    return TKR_OK;
}

extern "C" int64_t tkr_vbpr_colplan_lds_bytes(int32_t batch_size, int32_t d) {
    const int64_t RW = (d + tkr::kColWaves - 1) / tkr::kColWaves;
    return 4 * (4ll * batch_size + tkr::kColWaves + tkr::kColWaves * RW);
}

extern "C" int tkr_vbpr_colplan(const int32_t* f_ptr, const int32_t* f_col, const float* f_val, int32_t d, const int32_t* tri_i,
                                const int32_t* tri_j, int32_t batch_size, int32_t n_batches, int32_t row_cap, int32_t* colh,
                                int32_t* cent, int32_t* tcnt, int32_t* tent, void* stream) {
    if (!f_ptr || !f_col || !f_val || !tri_i || !tri_j || !colh || !cent || !tcnt || !tent || d <= 0 || batch_size <= 0 || n_batches < 0 ||
        row_cap <= 0)
        return TKR_EINVAL;
    const int64_t lds = tkr_vbpr_colplan_lds_bytes(batch_size, d);
    // one batch's column counters live in one CU's LDS; a row is read as at most 16 chunks of 64 entries
    if (lds > 160 * 1024 || row_cap > 1024 || 2ll * batch_size * row_cap >= (1ll << 30)) return TKR_EUNSUPPORTED;
    if (n_batches == 0) return TKR_OK;
#define TKR_COLPLAN(CH_)                                                                                                                   \
    do {                                                                                                                                   \
        static bool raised = false;                                                                                                        \
        if (!raised) {                                                                                                                     \
            TKR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tkr::vbpr_colplan_kernel<CH_>),                                    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                       \
            raised = true;                                                                                                                 \
        }                                                                                                                                  \
        hipLaunchKernelGGL(tkr::vbpr_colplan_kernel<CH_>, dim3(n_batches), dim3(tkr::kColWaves * 64), (size_t)lds, (hipStream_t)stream,    \
                           f_ptr, f_col, f_val, d, tri_i, tri_j, batch_size, row_cap, reinterpret_cast<int4*>(colh),                      \
                           reinterpret_cast<int2*>(cent), tcnt, reinterpret_cast<int2*>(tent));                                           \
    } while (0)
    if (row_cap <= 64) TKR_COLPLAN(1);
    else if (row_cap <= 128) TKR_COLPLAN(2);
    else if (row_cap <= 256) TKR_COLPLAN(4);
    else TKR_COLPLAN(16);
#undef TKR_COLPLAN
    TKR_LAUNCH_CHECK();
    return TKR_OK;
}

extern "C" int tkr_vbpr_run_cols(const tkr_vbpr_state* st, const int32_t* tri_i, const int32_t* tri_j, const int32_t* rec,
                                 const int32_t* occ, const int32_t* hdr, const int32_t* occt, const int32_t* tri_u, const int32_t* tpar,
                                 const int32_t* colh, const int32_t* cent, const int32_t* tcnt, const int32_t* tent, int32_t row_cap,
                                 int32_t cols_per_block, int32_t batch_size, int32_t n_batches, float* workspace, float* loss_out,
                                 void* stream) {
    if (!st || !st->U || !st->msU || !st->I || !st->msI || !st->irb || !st->msirb || !st->cem || !st->mscem || !st->icb || !st->msicb)
        return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->kh <= 0 || st->d <= 0) return TKR_EINVAL;
    if (!tri_i || !tri_j || !rec || !occ || !hdr || !occt || !tri_u || !tpar || !colh || !cent || !tcnt || !tent || !workspace ||
        batch_size <= 0 || n_batches < 0 || row_cap <= 0)
        return TKR_EINVAL;
    if (batch_size > 1024) return TKR_EUNSUPPORTED;
    const bool wide = st->kh > 128 || (st->kh & 3);               // the generic form (csrc/vbpr_wide.hip): any kh
    const int B = batch_size, kh = st->kh, tcap = 2 * row_cap;
    const size_t stride_r = (size_t)tkr_plan_max_blocks(B) * tkr_plan_team(B) * 16;
    const size_t stride_o = (size_t)3 * B;
    if (tkr_vbpr_workspace_core_floats(B, kh, st->d) < (int64_t)B * (6 + 2 * kh)) return TKR_EINVAL;
    float* s_buf = workspace;                                        // S_t [B] | T_t [B] | alpha, beta, e^alpha, e^beta [4B] | P [B][kh] | uce rows [B][kh]
    float* t_buf = s_buf + B;
    float* ab2 = t_buf + B;
    float* P = ab2 + 4 * (size_t)B;
    float* Wm = P + (size_t)B * kh;
    const int NH = (kh + 63) / 64, NE = (2 * kh + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    static const int tune = getenv("TKR_VBPR_TUNE") ? atoi(getenv("TKR_VBPR_TUNE")) : 0;      // timing experiments only (results invalid)
    static const int pw_env = getenv("TKR_VBPR_PWAVES") ? atoi(getenv("TKR_VBPR_PWAVES")) : 0;
    // waves per triplet of the projection: enough that a triplet's gather list (2 rows of feat) is one round of <= 32 gathers per wave
    if (pw_env && pw_env != 4 && pw_env != 8 && pw_env != 16) return TKR_EINVAL;       // the instantiated team sizes
    int pw = pw_env ? pw_env : (row_cap <= 64 ? 4 : (row_cap <= 128 || B > 256 ? 8 : 16));
    if (NH == 2 && pw > 8) pw = 8;          // kh > 64: two registers per gathered row and wave -- the 16-wave team is not instantiated there:
                                            // two rounds of gathers per wave instead of one (was a silent fall-through)
    // the loss words of the call's batches: behind the step's own scratch, every word written by its task (nothing to zero)
    const int lpc_ = kh <= 16 ? 4 : (kh <= 32 ? 8 : (kh <= 64 ? 16 : 32));
    const int cpb_ = (cols_per_block <= 0 || cols_per_block > 256 / lpc_) ? 256 / lpc_ : cols_per_block;
    const int n_col_blocks = wide ? tkr::vbpr_wide_col_blocks(st->d) : (st->d + cpb_ - 1) / cpb_;
    const size_t loss_stride = (size_t)2 * B + n_col_blocks;
    float* slots = loss_out ? workspace + tkr_vbpr_workspace_core_floats(B, kh, st->d) : nullptr;
    if (slots && (n_batches > 512 || (int64_t)n_batches * (int64_t)loss_stride > tkr_vbpr_workspace_floats(B, kh, st->d) - tkr_vbpr_workspace_core_floats(B, kh, st->d)))
        return TKR_EUNSUPPORTED;
    // The pair sums INSIDE the update launch (two launches per batch: TKR_VBPR_PAIRS=1): rows and column groups work S_t / T_t out for
    // their own entries from e^alpha / e^beta (PairSumInline).  Measured (round 5, B = 256, d = 20,000): 31.6 us per batch against
    // 23.1 with the pair-sum launch -- the 4.5 us launch and its boundary go, but every one of the update's ~1,400 blocks now does
    // 100-250 reciprocals per lane in front of its stores.  Parity-green, not the default.
    // TKR_VBPR_PAIRS=2: the pair sums by the first blocks of the update launch (vbpr_update_kernel FUSED) -- two launches per batch
    // and nothing recomputed; 0: the three-launch form.
    const int pairs_mode = wide ? 0 : tkr::vbpr_pairs_mode();
    const bool inline_pairs = pairs_mode == 1 && B <= 256;
    const bool fused_pairs = pairs_mode == 2;
    uint32_t* pair_done = reinterpret_cast<uint32_t*>(workspace + tkr_vbpr_workspace_floats(B, kh, st->d) - 64);     // the call's counter: last 64 words
    if (fused_pairs) TKR_CHECK(hipMemsetAsync(pair_done, 0, sizeof(uint32_t), (hipStream_t)stream));
    const bool long_runs = 2.0 * B * (double)row_cap / st->d > (double)tkr::kLightRun;      // (row_cap: the longest row of feat)
    for (int b = 0; b < n_batches; ++b) {
        const int32_t* ti = tri_i + (size_t)b * B;
        const int32_t* tj = tri_j + (size_t)b * B;
        const int32_t* tu = tri_u + (size_t)b * B;
        const int32_t* tp = tpar + (size_t)b * B;
        const int32_t* r = rec + b * stride_r;
        const int2* o2 = reinterpret_cast<const int2*>(occ + b * stride_o * 2);
        const int4* h4 = reinterpret_cast<const int4*>(hdr + (size_t)b * 4);
        const int32_t* ot = occt + b * stride_o;
        const int4* ch = reinterpret_cast<const int4*>(colh) + (size_t)b * st->d * 2;
        const int2* ce = reinterpret_cast<const int2*>(cent) + (size_t)b * B * tcap;
        const int32_t* tc = tcnt + (size_t)b * B;
        const int2* te = reinterpret_cast<const int2*>(tent) + (size_t)b * B * tcap;
        float* l = slots ? slots + (size_t)b * loss_stride : nullptr;
        if (wide) {
            tkr::vbpr_wide_project(*st, ti, tj, tu, tp, tc, te, tcap, B, P, ab2, Wm, l, s);
            hipLaunchKernelGGL(tkr::vbpr_pairsum_kernel, dim3((B + 3) / 4), dim3(256), 0, s, ab2, B, s_buf, t_buf, l);
            tkr::vbpr_wide_update(*st, r, o2, ot, h4, s_buf, t_buf, P, Wm, ch, ce, B, l, s);
            TKR_LAUNCH_CHECK();
            continue;
        }
        if (tune & 64) {
        } else if (NH == 1 && pw == 4) hipLaunchKernelGGL((tkr::vbpr_tproject_kernel<1, 4>), dim3(B), dim3(256), 0, s, *st, ti, tj, tu, tp, tc, te, tcap, B, P, ab2, Wm, l, tune);
        else if (NH == 1 && pw == 8) hipLaunchKernelGGL((tkr::vbpr_tproject_kernel<1, 8>), dim3(B), dim3(512), 0, s, *st, ti, tj, tu, tp, tc, te, tcap, B, P, ab2, Wm, l, tune);
        else if (NH == 1) hipLaunchKernelGGL((tkr::vbpr_tproject_kernel<1, 16>), dim3(B), dim3(1024), 0, s, *st, ti, tj, tu, tp, tc, te, tcap, B, P, ab2, Wm, l, tune);
        else if (pw == 4) hipLaunchKernelGGL((tkr::vbpr_tproject_kernel<2, 4>), dim3(B), dim3(256), 0, s, *st, ti, tj, tu, tp, tc, te, tcap, B, P, ab2, Wm, l, tune);
        else hipLaunchKernelGGL((tkr::vbpr_tproject_kernel<2, 8>), dim3(B), dim3(512), 0, s, *st, ti, tj, tu, tp, tc, te, tcap, B, P, ab2, Wm, l, tune);
        if (!(tune & 32) && !inline_pairs && !fused_pairs) hipLaunchKernelGGL(tkr::vbpr_pairsum_kernel, dim3((B + 3) / 4), dim3(256), 0, s, ab2, B, s_buf, t_buf, l);
        if (tune & 128) continue;
        const int lpc = kh <= 16 ? 4 : (kh <= 32 ? 8 : (kh <= 64 ? 16 : 32));
#define TKR_UPD(NE_, LPC_) tkr::launch_update<NE_, LPC_>(*st, r, o2, ot, h4, s_buf, t_buf, P, Wm, ch, ce, B, cols_per_block, l, s, tune, inline_pairs ? ab2 : nullptr, long_runs, \
                                                         fused_pairs ? ab2 : nullptr, pair_done, (uint32_t)(b + 1) * (uint32_t)((B + 3) / 4))
        switch (lpc) {
            case 4: TKR_UPD(1, 4); break;
            case 8: TKR_UPD(1, 8); break;
            case 16: if (NE == 1) TKR_UPD(1, 16); else TKR_UPD(2, 16); break;
            default: if (NE <= 3) TKR_UPD(3, 32); else TKR_UPD(4, 32); break;
        }
#undef TKR_UPD
        TKR_LAUNCH_CHECK();
    }
    if (slots && n_batches > 0) {
        hipLaunchKernelGGL(tkr::vbpr_loss_sum_kernel, dim3(n_batches), dim3(256), 0, s, slots, loss_stride, (int)loss_stride, loss_out);
        TKR_LAUNCH_CHECK();
    }
    return TKR_OK;
}
