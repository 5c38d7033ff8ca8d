// K2 -- one BPR mini-batch: gather -> pairwise logistic loss -> de-duplicated gradient ->
// sparse RMSProp, in ONE launch.
//
// Replaces sess.run([solver, obj]) of the reference (single/bpr.py:141) on the graph of
// single/bpr.py:81-100: 5 embedding_lookups, x_uij, log(1+exp(-x)) + regularisers, autodiff
// to IndexedSlices, unique + segment-sum of duplicate rows, SparseApplyRMSProp on U, V, b.
//
// Work decomposition (plan from K1): one 64-lane wave per *unique touched row* of the batch;
// rows with more than 4 occurrences get a team of 16 waves (one workgroup) whose partial
// gradients are combined through LDS in wave order.  A wave owns its row: it re-derives
// s_t = sigma(-x_t) for each of its occurrences from the partner rows, sums the
// per-occurrence gradients, applies RMSProp once and writes the row back.
//
// Every gather must see PRE-step values while other waves of the same launch write POST-step
// values.  Tables are double-buffered ([2][n][k]); the planner knows how often each row has
// been updated, so the record of every wave carries the parity (= buffer) of its own row and
// of every partner row: read buffer p, write buffer p^1.  No stamp lookup, no inter-workgroup
// ordering inside a launch; visibility between batches comes from the kernel boundary.
//
// Latency structure per wave: [64-B record] -> [all rows of <=4 occurrences + own row + slot]
// -> compute -> store.  Two dependent memory levels.
//
// Roofline: HBM / cache-bandwidth bound gather-scatter.  Algorithmic bytes per triplet
// (no credit for in-batch duplicates): 3 rows x (param+ms) x (read+write)
// = 48k B + 32 B biases + 24 B ids  ->  6,200 B at k = 128.
#include <stdlib.h>

#include "tkr_common.h"
#include "../../include/tkr.h"

extern "C" int tkr_plan_team(int32_t batch_size);

namespace tkr {

// waves per workgroup = per heavy-row team: 4 for B <= 1024, 16 above (oracle/plan_np.py team_for).  A CU
// pulls only ~10 B/clk from the fabric, so a small batch must be spread over many CUs: 4-wave workgroups put
// a 256-batch on ~180 CUs instead of ~47 (measured: the row-gather level drops from 3.4 us to ~1 us).
constexpr int kIdMask = 0x3fffffff;

// Row access: lane l owns the NE = ceil(k/64) CONTIGUOUS elements [l*NE, l*NE+NE) of a row.  VEC (chosen on
// the host) means FULL rows, k == 64*NE: one unpredicated 4/8/16-byte access per lane (256 B / 512 B / 1 KiB
// per wave-instruction), nothing between the load and its first real use.  Otherwise loads come from
// clamped (always valid) addresses and are zeroed by a select: a predicated load would compile to an
// exec-branch whose join forces s_waitcnt vmcnt(0) and serialises the row loads of a wave.
template <int NE, bool VEC>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int k, int lane, float (&r)[NE]) {
    const int e0 = lane * NE;
    if constexpr (VEC && NE == 1) {
        r[0] = base[e0];
    } else if constexpr (VEC && NE == 2) {
        const float2 v = *reinterpret_cast<const float2*>(base + e0);
        r[0] = v.x; r[1] = v.y;
    } else if constexpr (VEC && NE == 4) {
        const float4 v = *reinterpret_cast<const float4*>(base + e0);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const float v = base[min(e0 + q, k - 1)];
            r[q] = (e0 + q < k) ? v : 0.f;
        }
    }
}

template <int NE, bool VEC>
__device__ __forceinline__ void store_row(float* __restrict__ base, int k, int lane, const float (&r)[NE]) {
    const int e0 = lane * NE;
    if constexpr (VEC && NE == 1) {
        base[e0] = r[0];
    } else if constexpr (VEC && NE == 2) {
        *reinterpret_cast<float2*>(base + e0) = make_float2(r[0], r[1]);
    } else if constexpr (VEC && NE == 4) {
        *reinterpret_cast<float4*>(base + e0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
        for (int q = 0; q < NE; ++q)
            if (e0 + q < k) base[e0 + q] = r[q];
    }
}

// (sum a*b1, sum a*b2): the one dot-product routine every task uses, so that s_t comes out
// bit-identical in the user task and in both item tasks of a triplet.
template <int NE>
__device__ __forceinline__ void dot2(const float (&a)[NE], const float (&b1)[NE], const float (&b2)[NE],
                                     float& d1, float& d2) {
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        p1 = fmaf(a[q], b1[q], p1);
        p2 = fmaf(a[q], b2[q], p2);
    }
    wave_sum2(p1, p2);
    d1 = p1;
    d2 = p2;
}

struct Acc {            // per-wave accumulators besides the row gradient
    float gb;           // item tasks: bias gradient
    float loss_lane;    // user tasks: per-lane regulariser partial
    float loss_x;       // user tasks: wave-uniform part (softplus + bias regulariser)
};

// G occurrences of a USER row: oa = i | par<<30, ob = j | par<<30
template <int NE, bool VEC, int G>
__device__ __forceinline__ void user_group(const tkr_bpr_state& st, int lane, const int (&oa)[4], const int (&ob)[4],
                                           const float (&ur_in)[NE], float (&g)[NE], Acc& acc, bool want_loss) {
    const int k = st.k;
    const size_t istride = (size_t)st.n_items * k;
    float vi[G][NE], vj[G][NE], bi[G], bj[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        const int i = oa[q] & kIdMask, pi = (oa[q] >> 30) & 1;
        const int j = ob[q] & kIdMask, pj = (ob[q] >> 30) & 1;
        load_row<NE, VEC>(st.V + pi * istride + (size_t)i * k, k, lane, vi[q]);
        load_row<NE, VEC>(st.V + pj * istride + (size_t)j * k, k, lane, vj[q]);
        bi[q] = st.b[(size_t)pi * st.n_items + i];
        bj[q] = st.b[(size_t)pj * st.n_items + j];
    }
    // First use of the owned row sits HERE, behind the issue of every partner load: without this register
    // barrier the compiler hoists the loop-invariant regulariser terms (lambda*own, sgn(own), own^2) above the
    // loads and has to wait for the own row first -- two memory levels instead of one.
    float ur[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { ur[e] = ur_in[e]; asm volatile("" : "+v"(ur[e])); }
    const bool l2 = (st.mode == 0);
#pragma unroll
    for (int q = 0; q < G; ++q) {
        float xui, xuj;
        dot2<NE>(ur, vi[q], vj[q], xui, xuj);
        const float x = bi[q] - bj[q] + xui - xuj;
        const float s = sigmoid_neg(x);
        if (want_loss) acc.loss_x += softplus_neg(x);
        if (l2) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                g[e] += -s * (vi[q][e] - vj[q][e]) + st.lu * ur[e];
                if (want_loss) acc.loss_lane += 0.5f * (ur[e] * ur[e] * st.lu + vi[q][e] * vi[q][e] * st.li + vj[q][e] * vj[q][e] * st.lj);
            }
            if (want_loss) acc.loss_x += 0.5f * (bi[q] * bi[q] + bj[q] * bj[q]) * st.lb;
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                g[e] += -s * (vi[q][e] - vj[q][e]) + st.lu * sgn(ur[e]);
                if (want_loss) acc.loss_lane += fabsf(ur[e]) * st.lu + fabsf(vi[q][e]) * st.li + fabsf(vj[q][e]) * st.lj;
            }
            if (want_loss) acc.loss_x += (fabsf(bi[q]) + fabsf(bj[q])) * st.lb;
        }
    }
}

// G occurrences of an ITEM row: oa = u | par<<30, ob = other | par<<30 | role<<31
template <int NE, bool VEC, int G>
__device__ __forceinline__ void item_group(const tkr_bpr_state& st, int lane, const int (&oa)[4], const int (&ob)[4],
                                           const float (&vr_in)[NE], float br_in, float (&g)[NE], Acc& acc) {
    const int k = st.k;
    const size_t ustride = (size_t)st.n_users * k, istride = (size_t)st.n_items * k;
    float uu[G][NE], vo[G][NE], bo[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        const int u = oa[q] & kIdMask, pu = (oa[q] >> 30) & 1;
        const int o = ob[q] & kIdMask, po = (ob[q] >> 30) & 1;
        load_row<NE, VEC>(st.U + pu * ustride + (size_t)u * k, k, lane, uu[q]);
        load_row<NE, VEC>(st.V + po * istride + (size_t)o * k, k, lane, vo[q]);
        bo[q] = st.b[(size_t)po * st.n_items + o];
    }
    float vr[NE];                                     // see user_group: first use of the owned row after the loads
#pragma unroll
    for (int e = 0; e < NE; ++e) { vr[e] = vr_in[e]; asm volatile("" : "+v"(vr[e])); }
    float br = br_in;
    asm volatile("" : "+v"(br));
    const bool l2 = (st.mode == 0);
#pragma unroll
    for (int q = 0; q < G; ++q) {
        const bool role_j = ob[q] < 0;
        float dr, dn;                                  // <u, v_row>, <u, v_other>
        dot2<NE>(uu[q], vr, vo[q], dr, dn);
        // role i: row is the positive item  x = b_r - b_o + <u,v_r> - <u,v_o>
        // role j: row is the negative item  x = b_o - b_r + <u,v_o> - <u,v_r>
        const float x = role_j ? (bo[q] - br + dn - dr) : (br - bo[q] + dr - dn);
        const float s = sigmoid_neg(x);
        const float sg = role_j ? s : -s;
        const float lam = role_j ? st.lj : st.li;
        if (l2) {
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] += sg * uu[q][e] + lam * vr[e];
            acc.gb += sg + st.lb * br;
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] += sg * uu[q][e] + lam * sgn(vr[e]);
            acc.gb += sg + st.lb * sgn(br);
        }
    }
}

#ifndef TKR_K2_WAVES
#define TKR_K2_WAVES 1
#endif
#ifndef TKR_K2_GS
#define TKR_K2_GS 2
#endif
template <int NE> constexpr int kGroupOf = NE >= 2 ? TKR_K2_GS : 4;

template <int NE, bool VEC, bool ITEM>
__device__ __forceinline__ void run_group(const tkr_bpr_state& st, int lane, int n, const int (&oa)[4],
                                          const int (&ob)[4], const float (&row)[NE], float br, float (&g)[NE],
                                          Acc& acc, bool want_loss) {
    constexpr int GS = kGroupOf<NE>;
    if constexpr (ITEM) {
        if (n == 1) item_group<NE, VEC, 1>(st, lane, oa, ob, row, br, g, acc);
        else if (GS == 2 || n == 2) item_group<NE, VEC, 2>(st, lane, oa, ob, row, br, g, acc);
        else if constexpr (GS > 2) {
            if (n == 3) item_group<NE, VEC, 3>(st, lane, oa, ob, row, br, g, acc);
            else item_group<NE, VEC, 4>(st, lane, oa, ob, row, br, g, acc);
        }
    } else {
        if (n == 1) user_group<NE, VEC, 1>(st, lane, oa, ob, row, g, acc, want_loss);
        else if (GS == 2 || n == 2) user_group<NE, VEC, 2>(st, lane, oa, ob, row, g, acc, want_loss);
        else if constexpr (GS > 2) {
            if (n == 3) user_group<NE, VEC, 3>(st, lane, oa, ob, row, g, acc, want_loss);
            else user_group<NE, VEC, 4>(st, lane, oa, ob, row, g, acc, want_loss);
        }
    }
}

template <int NE, bool VEC, int kTeam, bool SGD>
__global__ __launch_bounds__((kTeam * TKR_WAVE), (TKR_K2_WAVES)) void bpr_step_kernel(
    tkr_bpr_state st, int32_t* rec_all /*word 15 of a user task's record receives its loss sum: not const, not restrict*/,
    const int2* __restrict__ occ, const int4* __restrict__ hdr, float* __restrict__ loss_out, int reverse) {
    __shared__ float red[kTeam][NE * TKR_WAVE + 1];
    const int lane = threadIdx.x & (TKR_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int4 h = *hdr;                         // (workgroups used, light workgroups, heavy tasks, tasks)
    const int n_blocks = __builtin_amdgcn_readfirstlane(h.x);
    const int nlb = __builtin_amdgcn_readfirstlane(h.y);
    const int k = st.k;
    const size_t ustride = (size_t)st.n_users * k, istride = (size_t)st.n_items * k;

    // The heavy teams sit at the END of the record list (oracle/plan_np.py launch_plan); the grid walks it BACKWARDS, so the
    // longest tasks -- a popular item's hundreds of occurrences summed by one team -- start first and the light workgroups fill in
    // behind them (longest-processing-time order; forwards, the last workgroups to start were the slowest: TKR_K2_ORDER=0).
    for (int it = blockIdx.x; it < n_blocks; it += gridDim.x) {
        const int blk = reverse ? n_blocks - 1 - it : it;
        const bool heavy = blk >= nlb;           // workgroup-uniform
        const int word = (lane < 16) ? rec_all[((size_t)blk * kTeam + wave) * 16 + lane] : 0;
        const int rowk = bcast_i(word, 0);
        if (rowk == -1) continue;                // idle wave of the last light group (never in a heavy group)
        const int meta = bcast_i(word, 1);
        const int n_occ = bcast_i(word, 2);
        const int first = bcast_i(word, 3);
        const int par = meta & 1, team = (meta >> 8) & 0xff;
        const bool is_item = rowk < 0;
        const int row = rowk & 0x7fffffff;

        float own[NE], g[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) g[e] = 0.f;
        Acc acc = {0.f, 0.f, 0.f};
        const size_t roff = (is_item ? par * istride : par * ustride) + (size_t)row * k;
        load_row<NE, VEC>((is_item ? st.V : st.U) + roff, k, lane, own);
        float ms[NE];                                    // RMSProp slot of the owned row: same memory level as the row
        if constexpr (!SGD) load_row<NE, VEC>((is_item ? st.msV : st.msU) + roff, k, lane, ms);
        const size_t boff = is_item ? (size_t)par * st.n_items + row : 0;     // valid either way: no branch
        const float br_raw = st.b[boff];
        float msb_raw = 0.f;
        if constexpr (!SGD) msb_raw = st.msb[boff];
        const float br = is_item ? br_raw : 0.f, msb = is_item ? msb_raw : 0.f;

        // occurrences 4 .. 67 of the task (the first four ride in the record): asked for NOW, in one load beside the first group's rows.
        // Fetched group by group they were a trip of their own in front of every further group's row loads: a 16-occurrence light
        // task (batch 8192) was 7 dependent trips, the tail of the launch.
        int2 omore = make_int2(0, 0);
        if (n_occ > 4 && lane < n_occ - 4) omore = occ[first + (4 + lane) * team];
        // occurrences per round: 4 (k <= 64) or 2 (wider rows: the partner rows of four occurrences are 16 registers + 16 of addresses,
        // 83 in all, and a 16-wave workgroup then has a CU to itself; TKR_K2_GS)
        constexpr int GS = kGroupOf<NE>;
        for (int done = 0; done < n_occ; done += GS) {
            const int n = min(GS, n_occ - done);
            int oa[4] = {0, 0, 0, 0}, ob[4] = {0, 0, 0, 0};
            if (done < 4) {                       // (4 % GS == 0: a round lies inside the record or behind it)
#pragma unroll
                for (int q = 0; q < GS; ++q) { oa[q] = bcast_i(word, (4 + 2 * (done + q)) & 15); ob[q] = bcast_i(word, (5 + 2 * (done + q)) & 15); }
            } else if (done + GS <= 68) {
#pragma unroll
                for (int q = 0; q < GS; ++q) { oa[q] = bcast_i(omore.x, (done - 4 + q) & 63); ob[q] = bcast_i(omore.y, (done - 4 + q) & 63); }
            } else {                              // very heavy rows: fetch the next occurrences
                int2 o = make_int2(0, 0);
                if (lane < n) o = occ[first + (done + lane) * team];
#pragma unroll
                for (int q = 0; q < GS; ++q) { oa[q] = bcast_i(o.x, q); ob[q] = bcast_i(o.y, q); }
            }
            if (is_item) run_group<NE, VEC, true>(st, lane, n, oa, ob, own, br, g, acc, false);
            else run_group<NE, VEC, false>(st, lane, n, oa, ob, own, 0.f, g, acc, loss_out != nullptr);
        }

        if (!is_item && loss_out) {
            // The loss of a batch is the sum over its user tasks -- thousands of waves.  As atomics on one word they cost 10 ns EACH,
            // one after the other at the memory side: 90 us on top of a 22 us launch at batch 8192.  Every wave has a 64-byte launch
            // record of its own whose last word K1 leaves zero: its sum goes THERE (a plain store), and step_loss_kernel adds a
            // call's records up afterwards.
            const float tot = wave_sum(acc.loss_lane) + acc.loss_x;
            if (lane == 0) rec_all[((size_t)blk * kTeam + wave) * 16 + 15] = __float_as_int(tot);
        }

        if (heavy) {                               // combine the team's partial gradients in wave order
#pragma unroll
            for (int e = 0; e < NE; ++e) red[wave][lane + e * TKR_WAVE] = g[e];
            if (lane == 0) red[wave][NE * TKR_WAVE] = acc.gb;
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    float s = 0.f;
                    for (int w = 0; w < kTeam; ++w) s += red[w][lane + e * TKR_WAVE];
                    g[e] = s;
                }
                float s = 0.f;
                for (int w = 0; w < kTeam; ++w) s += red[w][NE * TKR_WAVE];
                acc.gb = s;
            }
            __syncthreads();
            if (wave != 0) continue;
        }

        // ---- RMSProp on the owned row (TF SparseApplyRMSProp, momentum 0), written to buffer par^1
        const size_t woff = (is_item ? (par ^ 1) * istride : (par ^ 1) * ustride) + (size_t)row * k;
        float pn[NE];
        if constexpr (SGD) {                       // old/methods/bpr.py:57-61: P <- P - lr * dcost/dP
#pragma unroll
            for (int e = 0; e < NE; ++e) pn[e] = own[e] - st.lr * g[e];
            store_row<NE, VEC>((is_item ? st.V : st.U) + woff, k, lane, pn);
            if (is_item && lane == 0) st.b[(size_t)(par ^ 1) * st.n_items + row] = br - st.lr * acc.gb;
        } else {
            float mn[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                mn[e] = st.rho * ms[e] + (1.f - st.rho) * g[e] * g[e];
                pn[e] = own[e] - st.lr * g[e] / sqrtf(mn[e] + st.eps);
            }
            store_row<NE, VEC>((is_item ? st.msV : st.msU) + woff, k, lane, mn);
            store_row<NE, VEC>((is_item ? st.V : st.U) + woff, k, lane, pn);
            if (is_item && lane == 0) {
                const float m2 = st.rho * msb + (1.f - st.rho) * acc.gb * acc.gb;
                st.msb[(size_t)(par ^ 1) * st.n_items + row] = m2;
                st.b[(size_t)(par ^ 1) * st.n_items + row] = br - st.lr * acc.gb / sqrtf(m2 + st.eps);
            }
        }
    }
}

static int step_grid(int B, int team) {
    // enough workgroups for every light task plus a handful of teams; the kernel grid-strides
    const int lpb = team;                            // oracle/plan_np.py light_per_block
    const int light = (3 * B + lpb - 1) / lpb;
    int grid = light + 16;
    static const int cap = getenv("TKR_K2_GRID") ? atoi(getenv("TKR_K2_GRID")) : 2048;      // tuning aid
    if (grid > cap) grid = cap;
    return grid;
}

template <int NE, bool VEC, int TEAM>
static int launch_step_t(const tkr_bpr_state& st, int32_t* rec, const int32_t* occ, const int32_t* hdr, int B,
                       float* loss_out, hipStream_t stream) {
    static const int reverse = (getenv("TKR_K2_ORDER") && getenv("TKR_K2_ORDER")[0] == '0') ? 0 : 1;
    if (st.opt == 1)
        hipLaunchKernelGGL((bpr_step_kernel<NE, VEC, TEAM, true>), dim3(step_grid(B, TEAM)), dim3(TEAM * TKR_WAVE), 0, stream,
                           st, rec, reinterpret_cast<const int2*>(occ), reinterpret_cast<const int4*>(hdr), loss_out, reverse);
    else
        hipLaunchKernelGGL((bpr_step_kernel<NE, VEC, TEAM, false>), dim3(step_grid(B, TEAM)), dim3(TEAM * TKR_WAVE), 0, stream,
                           st, rec, reinterpret_cast<const int2*>(occ), reinterpret_cast<const int4*>(hdr), loss_out, reverse);
    return (int)hipGetLastError();
}

template <int NE, bool VEC>
static int launch_step(const tkr_bpr_state& st, int32_t* rec, const int32_t* occ, const int32_t* hdr, int B,
                       float* loss_out, hipStream_t stream) {
    switch (tkr_plan_team(B)) {                    // csrc/plan_parts.h team_for
        case 4: return launch_step_t<NE, VEC, 4>(st, rec, occ, hdr, B, loss_out, stream);
        case 8: return launch_step_t<NE, VEC, 8>(st, rec, occ, hdr, B, loss_out, stream);
        default: return launch_step_t<NE, VEC, 16>(st, rec, occ, hdr, B, loss_out, stream);
    }
}

// ---- any width: the generic row form (k > 512, and 256 < k <= 512 at batch sizes above 1024) -------------------------------------------
// The kernel above holds a row in ceil(k / 64) registers per array and the partner rows of a round beside it: 512 factors is where
// that ends.  Here nothing is resident: a task walks its occurrences one after the other, every occurrence in two passes over the
// k dimension -- (A) the dot products of the triplet, so s_t = sigma(-x_t); (B) the occurrence's gradient, added to the task's
// running sum, which lives in the row the task is going to WRITE (buffer par ^ 1 of its own row: nobody reads it in this launch) --
// and ends with the RMSProp pass over that row.  The waves of a heavy team would each need a running sum of their own: wave 0
// walks the records of the whole team instead, in wave order.  Slow (every partner row is read twice, the sum goes through
// memory) but the same step for every k and batch size: same plan, same double buffering, same loss word per task; sums in
// another order than the register form (lane-strided instead of lane-contiguous), inside the tolerance of the step tests.
__device__ __forceinline__ float ld_fresh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int kTeam, bool SGD>
__global__ __launch_bounds__((kTeam * TKR_WAVE)) void bpr_wide_kernel(tkr_bpr_state st, int32_t* rec_all, const int2* __restrict__ occ,
                                                                       const int4* __restrict__ hdr, float* __restrict__ loss_out, int reverse) {
    const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x >> 6;
    const int4 h = *hdr;
    const int n_blocks = __builtin_amdgcn_readfirstlane(h.x), nlb = __builtin_amdgcn_readfirstlane(h.y);
    const int k = st.k;
    const size_t ustride = (size_t)st.n_users * k, istride = (size_t)st.n_items * k;
    const bool l2 = st.mode == 0, want_loss = loss_out != nullptr;
    for (int it = blockIdx.x; it < n_blocks; it += gridDim.x) {
        const int blk = reverse ? n_blocks - 1 - it : it;
        const bool heavy = blk >= nlb;
        if (heavy && wave != 0) continue;                        // (no barrier in this kernel)
        const int n_rec = heavy ? kTeam : 1;
        const int head = (lane < 16) ? rec_all[((size_t)blk * kTeam + (heavy ? 0 : wave)) * 16 + lane] : 0;
        const int rowk = bcast_i(head, 0);
        if (rowk == -1) continue;
        const int par = bcast_i(head, 1) & 1;
        const bool is_item = rowk < 0;
        const int row = rowk & 0x7fffffff;
        const size_t roff = (is_item ? par * istride : par * ustride) + (size_t)row * k;
        const size_t woff = (is_item ? (par ^ 1) * istride : (par ^ 1) * ustride) + (size_t)row * k;
        const float* own = (is_item ? st.V : st.U) + roff;
        float* gbuf = (is_item ? st.V : st.U) + woff;            // the running gradient sum, then the new row
        const float br = is_item ? st.b[(size_t)par * st.n_items + row] : 0.f;
        float gb = 0.f, loss = 0.f;
        bool first_occ = true;
        for (int w = 0; w < n_rec; ++w) {
            const int word = w == 0 ? head : ((lane < 16) ? rec_all[((size_t)blk * kTeam + w) * 16 + lane] : 0);
            if (bcast_i(word, 0) == -1) continue;
            const int n_occ = bcast_i(word, 2), first = bcast_i(word, 3), team = (bcast_i(word, 1) >> 8) & 0xff;
            for (int q = 0; q < n_occ; ++q) {
                int oa, ob;
                if (q < 4) { oa = bcast_i(word, 4 + 2 * q); ob = bcast_i(word, 5 + 2 * q); }
                else { const int2 o = occ[first + q * team]; oa = o.x; ob = o.y; }
                const int a = oa & kIdMask, pa = (oa >> 30) & 1, b = ob & kIdMask, pb = (ob >> 30) & 1;
                const bool role_j = ob < 0;
                // user task: r1 = v_i, r2 = v_j; item task: r1 = u, r2 = the other item of the triplet
                const float* r1 = is_item ? st.U + pa * ustride + (size_t)a * k : st.V + pa * istride + (size_t)a * k;
                const float* r2 = st.V + pb * istride + (size_t)b * k;
                float d1 = 0.f, d2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;        // pass A
                for (int e = lane; e < k; e += TKR_WAVE) {
                    const float o = own[e], x1 = r1[e], x2 = r2[e];
                    if (is_item) { d1 = fmaf(x1, o, d1); d2 = fmaf(x1, x2, d2); }      // <u, v_row>, <u, v_other>
                    else { d1 = fmaf(o, x1, d1); d2 = fmaf(o, x2, d2); }               // <u, v_i>, <u, v_j>
                    if (want_loss && !is_item) {
                        if (l2) { n0 = fmaf(o, o, n0); n1 = fmaf(x1, x1, n1); n2 = fmaf(x2, x2, n2); }
                        else { n0 += fabsf(o); n1 += fabsf(x1); n2 += fabsf(x2); }
                    }
                }
                d1 = wave_sum(d1); d2 = wave_sum(d2);
                float x, coef, lam;
                if (is_item) {
                    const float bo = st.b[(size_t)pb * st.n_items + b];
                    x = role_j ? (bo - br + d2 - d1) : (br - bo + d1 - d2);
                    const float sg = sigmoid_neg(x);
                    coef = role_j ? sg : -sg;
                    lam = role_j ? st.lj : st.li;
                    gb += coef + st.lb * (l2 ? br : sgn(br));
                } else {
                    const float bi = st.b[(size_t)pa * st.n_items + a], bj = st.b[(size_t)pb * st.n_items + b];
                    x = bi - bj + d1 - d2;
                    coef = -sigmoid_neg(x);
                    lam = st.lu;
                    if (want_loss) {
                        n0 = wave_sum(n0); n1 = wave_sum(n1); n2 = wave_sum(n2);
                        loss += softplus_neg(x) + (l2 ? 0.5f * (n0 * st.lu + n1 * st.li + n2 * st.lj) + 0.5f * (bi * bi + bj * bj) * st.lb
                                                      : (n0 * st.lu + n1 * st.li + n2 * st.lj) + (fabsf(bi) + fabsf(bj)) * st.lb);
                    }
                }
                for (int e = lane; e < k; e += TKR_WAVE) {        // pass B
                    const float o = own[e];
                    const float part = is_item ? r1[e] : (r1[e] - r2[e]);
                    const float prev = first_occ ? 0.f : ld_fresh(gbuf + e);
                    gbuf[e] = prev + coef * part + lam * (l2 ? o : sgn(o));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sums are in memory before the next occurrence reads them back
                first_occ = false;
            }
        }
        if (!is_item && want_loss && lane == 0) rec_all[((size_t)blk * kTeam + (heavy ? 0 : wave)) * 16 + 15] = __float_as_int(loss);
        const float* msrc = (is_item ? st.msV : st.msU) + (SGD ? 0 : roff);
        float* mdst = (is_item ? st.msV : st.msU) + (SGD ? 0 : woff);
        for (int e = lane; e < k; e += TKR_WAVE) {
            const float g = ld_fresh(gbuf + e), o = own[e];
            if constexpr (SGD) {
                gbuf[e] = o - st.lr * g;
            } else {
                const float m2 = st.rho * msrc[e] + (1.f - st.rho) * g * g;
                mdst[e] = m2;
                gbuf[e] = o - st.lr * g / sqrtf(m2 + st.eps);
            }
        }
        if (is_item && lane == 0) {
            if constexpr (SGD) {
                st.b[(size_t)(par ^ 1) * st.n_items + row] = br - st.lr * gb;
            } else {
                const float m2 = st.rho * st.msb[(size_t)par * st.n_items + row] + (1.f - st.rho) * gb * gb;
                st.msb[(size_t)(par ^ 1) * st.n_items + row] = m2;
                st.b[(size_t)(par ^ 1) * st.n_items + row] = br - st.lr * gb / sqrtf(m2 + st.eps);
            }
        }
    }
}

template <int TEAM>
static int launch_wide(const tkr_bpr_state& st, int32_t* rec, const int32_t* occ, const int32_t* hdr, int B, float* loss_out, hipStream_t stream) {
    if (st.opt == 1)
        hipLaunchKernelGGL((bpr_wide_kernel<TEAM, true>), dim3(step_grid(B, TEAM)), dim3(TEAM * TKR_WAVE), 0, stream, st, rec,
                           reinterpret_cast<const int2*>(occ), reinterpret_cast<const int4*>(hdr), loss_out, 1);
    else
        hipLaunchKernelGGL((bpr_wide_kernel<TEAM, false>), dim3(step_grid(B, TEAM)), dim3(TEAM * TKR_WAVE), 0, stream, st, rec,
                           reinterpret_cast<const int2*>(occ), reinterpret_cast<const int4*>(hdr), loss_out, 1);
    return (int)hipGetLastError();
}

static int launch_wide_team(const tkr_bpr_state& st, int32_t* rec, const int32_t* occ, const int32_t* hdr, int B, float* loss_out, hipStream_t stream) {
    switch (tkr_plan_team(B)) {
        case 4: return launch_wide<4>(st, rec, occ, hdr, B, loss_out, stream);
        case 8: return launch_wide<8>(st, rec, occ, hdr, B, loss_out, stream);
        default: return launch_wide<16>(st, rec, occ, hdr, B, loss_out, stream);
    }
}

static int dispatch_step(const tkr_bpr_state& st, int32_t* rec, const int32_t* occ, const int32_t* hdr, int B,
                         float* loss_out, hipStream_t stream) {
    const int ne = (st.k + TKR_WAVE - 1) / TKR_WAVE;
    const bool full = (st.k == ne * TKR_WAVE) && ne != 3;           // k = 64, 128, 256: unpredicated vector rows
    switch (ne) {
        case 1: return full ? launch_step<1, true>(st, rec, occ, hdr, B, loss_out, stream)
                            : launch_step<1, false>(st, rec, occ, hdr, B, loss_out, stream);
        case 2: return full ? launch_step<2, true>(st, rec, occ, hdr, B, loss_out, stream)
                            : launch_step<2, false>(st, rec, occ, hdr, B, loss_out, stream);
        case 3: return launch_step<3, false>(st, rec, occ, hdr, B, loss_out, stream);
        case 4: return full ? launch_step<4, true>(st, rec, occ, hdr, B, loss_out, stream)
                            : launch_step<4, false>(st, rec, occ, hdr, B, loss_out, stream);
        case 5: case 6: case 7: case 8:         // 256 < k <= 512: eight elements per lane, predicated rows; the 4- and 8-wave teams of batches
            // up to 16,384 only (a 16-wave workgroup leaves a wave 128 registers: four partner-row pairs of eight do not fit)
            if (tkr_plan_team(B) == 4) return launch_step_t<8, false, 4>(st, rec, occ, hdr, B, loss_out, stream);
            if (tkr_plan_team(B) == 8) return launch_step_t<8, false, 8>(st, rec, occ, hdr, B, loss_out, stream);
            return launch_wide_team(st, rec, occ, hdr, B, loss_out, stream);
        default:                                // k > 512: the generic row form
            return launch_wide_team(st, rec, occ, hdr, B, loss_out, stream);
    }
}

}  // namespace tkr

namespace tkr {
// loss_out[b] += the sums the user tasks of batch b left in word 15 of their records (zero in every other record); one atomic
// per (batch, slice: 8 to 128 of them by the batch size)
__global__ __launch_bounds__(256) void step_loss_kernel(const int32_t* __restrict__ rec, size_t stride_r, const int4* __restrict__ hdr, int team,
                                                        float* __restrict__ loss_out) {
    __shared__ float part[4];
    const int b = blockIdx.x, sl = blockIdx.y, lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x / TKR_WAVE;
    const int n_rec = hdr[b].x * team;
    const int per = (n_rec + (int)gridDim.y - 1) / (int)gridDim.y;
    const int lo = min(sl * per, n_rec), hi = min(lo + per, n_rec);
    const int32_t* r = rec + (size_t)b * stride_r;
    float acc = 0.f;
    for (int q = lo + (int)threadIdx.x; q < hi; q += 256) acc += __int_as_float(r[(size_t)q * 16 + 15]);
    acc = wave_sum(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (part[0] + part[1]) + (part[2] + part[3]);
        if (tot != 0.f) loss_add(loss_out + b, tot);
    }
}
}  // namespace tkr

extern "C" int tkr_plan_max_blocks(int32_t batch_size);

static int check_state(const tkr_bpr_state* st) {
    if (!st || !st->U || !st->V || !st->b) return TKR_EINVAL;
    if (st->opt != 0 && st->opt != 1) return TKR_EINVAL;
    if (st->opt == 0 && (!st->msU || !st->msV || !st->msb)) return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->k <= 0) return TKR_EINVAL;
    return TKR_OK;
}

// ---- launch path -------------------------------------------------------------------------------------
// One direct launch per batch, in plan order on the caller's stream (the kernel boundary is what makes batch
// t+1 see batch t, single/bpr.py:141).  No hidden state: nothing is captured, cached or allocated here.
extern "C" int tkr_bpr_run(const tkr_bpr_state* st, int32_t* rec, const int32_t* occ, const int32_t* hdr,
                           int32_t batch_size, int32_t n_batches, float* loss_out, void* stream) {
    const int rc = check_state(st);
    if (rc != TKR_OK) return rc;
    if (!rec || !occ || !hdr || batch_size <= 0 || n_batches < 0) return TKR_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t stride_r = (size_t)tkr_plan_max_blocks(batch_size) * tkr_plan_team(batch_size) * 16;
    const size_t stride_o = (size_t)3 * batch_size * 2;
    for (int b = 0; b < n_batches; ++b) {
        const int r = tkr::dispatch_step(*st, rec + b * stride_r, occ + b * stride_o, hdr + (size_t)b * 4, batch_size,
                                         loss_out ? loss_out + b : nullptr, s);
        if (r != 0) return r;
    }
    if (loss_out && n_batches > 0) {
        hipLaunchKernelGGL(tkr::step_loss_kernel, dim3(n_batches, batch_size >= 16384 ? 128 : batch_size >= 2048 ? 32 : 8), dim3(256), 0, s, rec, stride_r,
                           reinterpret_cast<const int4*>(hdr), tkr_plan_team(batch_size), loss_out);
        TKR_LAUNCH_CHECK();
    }
    return TKR_OK;
}
