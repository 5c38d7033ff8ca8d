// K2 -- one BPR mini-batch: gather -> pairwise logistic loss -> de-duplicated gradient ->
// sparse RMSProp, in ONE launch.
//
// Replaces sess.run([solver, obj]) of the reference (single/bpr.py:141) on the graph of
// single/bpr.py:81-100: 5 embedding_lookups, x_uij, log(1+exp(-x)) + regularisers, autodiff
// to IndexedSlices, unique + segment-sum of duplicate rows, SparseApplyRMSProp on U, V, b.
//
// Work decomposition: one 64-lane wave per *unique touched row* of the batch (plan from K1).
// A wave owns its row: it re-derives s_t = sigma(-x_t) for each of the row's occurrences
// from the partner rows, sums the per-occurrence gradients in plan order (the sequential
// segment-sum order of the oracle), applies RMSProp once and writes the row back.
//
// Every gather must see PRE-step values while other waves of the same launch write
// POST-step values.  Parameters are therefore kept in two buffers per table with a per-row
// stamp  (serial<<1 | buffer-holding-the-current-value):  the owner writes the new row to
// the *other* buffer and re-stamps; a reader that finds this launch's serial in the stamp
// takes the other buffer, i.e. the old value.  No inter-workgroup ordering is needed inside
// a launch; visibility between batches comes from the kernel boundary.
//
// Roofline: HBM / cache-bandwidth bound gather-scatter.  Algorithmic bytes per triplet
// (no credit for in-batch duplicates): 3 rows x (param+ms) x (read+write)
// = 48k B + 32 B biases + 24 B ids  ->  6,200 B at k = 128.
#include "tkr_common.h"
#include "../../include/tkr.h"

namespace tkr {

constexpr int kStepThreads = 256;                 // 4 waves per workgroup
constexpr int kWavesPerWG = kStepThreads / TKR_WAVE;

struct RowRef { const float* p; };

template <int NE>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int k, int lane, float (&r)[NE]) {
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = lane + q * TKR_WAVE;
        r[q] = (e < k) ? base[e] : 0.f;
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

// which buffer holds the value of a row as of the START of batch `serial`
__device__ __forceinline__ int pre_step_buffer(int stamp, int serial) {
    return (stamp & 1) ^ (int)((stamp >> 1) == serial);
}

template <int NE>
__global__ __launch_bounds__(kStepThreads) void bpr_step_kernel(
    tkr_bpr_state st, const int4* __restrict__ task_all, const int2* __restrict__ occ,
    int serial, float* __restrict__ loss_out) {
    const int lane = threadIdx.x & (TKR_WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerWG + (threadIdx.x >> 6));
    const int4 tk = task_all[w];
    const int rowk = __builtin_amdgcn_readfirstlane(tk.x);
    if (rowk == -1) return;
    const int start = __builtin_amdgcn_readfirstlane(tk.y);
    const int count = __builtin_amdgcn_readfirstlane(tk.z);
    const bool is_item = rowk < 0;          // kind bit = sign bit
    const int row = rowk & 0x7fffffff;
    const int k = st.k;
    const size_t ustride = (size_t)st.n_users * k, istride = (size_t)st.n_items * k;
    const bool l2 = (st.mode == 0);

    if (!is_item) {
        // ------------------------------------------------ user row: owns U[row]
        const int cb = st.ustamp[row] & 1;           // only this wave re-stamps the row
        float ur[NE], ms[NE], g[NE];
        load_row<NE>(st.U + cb * ustride + (size_t)row * k, k, lane, ur);
        load_row<NE>(st.msU + cb * ustride + (size_t)row * k, k, lane, ms);
#pragma unroll
        for (int q = 0; q < NE; ++q) g[q] = 0.f;
        float loss_lane = 0.f, loss_x = 0.f;
        for (int c0 = 0; c0 < count; c0 += TKR_WAVE) {
            const int nc = min(count - c0, TKR_WAVE);
            int io = 0, jo = 0, ibuf = 0, jbuf = 0;
            float bio = 0.f, bjo = 0.f;
            if (lane < nc) {
                const int2 oc = occ[start + c0 + lane];
                io = oc.x; jo = oc.y;
                ibuf = pre_step_buffer(st.istamp[io], serial);
                jbuf = pre_step_buffer(st.istamp[jo], serial);
                bio = st.b[(size_t)ibuf * st.n_items + io];
                bjo = st.b[(size_t)jbuf * st.n_items + jo];
            }
            for (int m = 0; m < nc; ++m) {
                const int i_m = bcast_i(io, m), j_m = bcast_i(jo, m);
                const float bi = bcast_f(bio, m), bj = bcast_f(bjo, m);
                float vi[NE], vj[NE];
                load_row<NE>(st.V + bcast_i(ibuf, m) * istride + (size_t)i_m * k, k, lane, vi);
                load_row<NE>(st.V + bcast_i(jbuf, m) * istride + (size_t)j_m * k, k, lane, vj);
                float xui, xuj;
                dot2<NE>(ur, vi, vj, xui, xuj);
                const float x = bi - bj + xui - xuj;
                const float s = sigmoid_neg(x);
                loss_x += softplus_neg(x);
                if (l2) {
#pragma unroll
                    for (int q = 0; q < NE; ++q) {
                        g[q] += -s * (vi[q] - vj[q]) + st.lu * ur[q];
                        loss_lane += 0.5f * (ur[q] * ur[q] * st.lu + vi[q] * vi[q] * st.li + vj[q] * vj[q] * st.lj);
                    }
                    loss_x += 0.5f * (bi * bi + bj * bj) * st.lb;
                } else {
#pragma unroll
                    for (int q = 0; q < NE; ++q) {
                        g[q] += -s * (vi[q] - vj[q]) + st.lu * sgn(ur[q]);
                        loss_lane += fabsf(ur[q]) * st.lu + fabsf(vi[q]) * st.li + fabsf(vj[q]) * st.lj;
                    }
                    loss_x += (fabsf(bi) + fabsf(bj)) * st.lb;
                }
            }
        }
        float* Uo = st.U + (cb ^ 1) * ustride + (size_t)row * k;
        float* Mo = st.msU + (cb ^ 1) * ustride + (size_t)row * k;
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = lane + q * TKR_WAVE;
            if (e < k) {
                const float m2 = st.rho * ms[q] + (1.f - st.rho) * g[q] * g[q];
                Mo[e] = m2;
                Uo[e] = ur[q] - st.lr * g[q] / sqrtf(m2 + st.eps);
            }
        }
        if (lane == 0) st.ustamp[row] = (serial << 1) | (cb ^ 1);
        if (loss_out) {
            const float tot = wave_sum(loss_lane) + loss_x;     // loss_x is wave-uniform
            if (lane == 0) atomicAdd(loss_out, tot);
        }
        return;
    }

    // ---------------------------------------------------- item row: owns V[row], b[row]
    const int cb = st.istamp[row] & 1;
    float vr[NE], ms[NE], g[NE];
    load_row<NE>(st.V + cb * istride + (size_t)row * k, k, lane, vr);
    load_row<NE>(st.msV + cb * istride + (size_t)row * k, k, lane, ms);
    const float br = st.b[(size_t)cb * st.n_items + row];
    const float msbr = st.msb[(size_t)cb * st.n_items + row];
    float gb = 0.f;
#pragma unroll
    for (int q = 0; q < NE; ++q) g[q] = 0.f;
    for (int c0 = 0; c0 < count; c0 += TKR_WAVE) {
        const int nc = min(count - c0, TKR_WAVE);
        int uo = 0, oo = 0, ubuf = 0, obuf = 0;
        float boo = 0.f;
        if (lane < nc) {
            const int2 oc = occ[start + c0 + lane];
            uo = oc.x; oo = oc.y;                       // oo: other item | role<<31
            const int other = oo & 0x7fffffff;
            ubuf = pre_step_buffer(st.ustamp[uo], serial);
            obuf = pre_step_buffer(st.istamp[other], serial);
            boo = st.b[(size_t)obuf * st.n_items + other];
        }
        for (int m = 0; m < nc; ++m) {
            const int u_m = bcast_i(uo, m), o_m = bcast_i(oo, m);
            const bool role_j = o_m < 0;
            const int other = o_m & 0x7fffffff;
            const float bo = bcast_f(boo, m);
            float uu[NE], vo[NE];
            load_row<NE>(st.U + bcast_i(ubuf, m) * ustride + (size_t)u_m * k, k, lane, uu);
            load_row<NE>(st.V + bcast_i(obuf, m) * istride + (size_t)other * k, k, lane, vo);
            float dr, dn;                                  // <u, v_row>, <u, v_other>
            dot2<NE>(uu, vr, vo, dr, dn);
            // role i: row is the positive item  x = b_r - b_o + <u,v_r> - <u,v_o>
            // role j: row is the negative item  x = b_o - b_r + <u,v_o> - <u,v_r>
            const float x = role_j ? (bo - br + dn - dr) : (br - bo + dr - dn);
            const float s = sigmoid_neg(x);
            const float sg = role_j ? s : -s;
            const float lam = role_j ? st.lj : st.li;
            if (l2) {
#pragma unroll
                for (int q = 0; q < NE; ++q) g[q] += sg * uu[q] + lam * vr[q];
                gb += sg + st.lb * br;
            } else {
#pragma unroll
                for (int q = 0; q < NE; ++q) g[q] += sg * uu[q] + lam * sgn(vr[q]);
                gb += sg + st.lb * sgn(br);
            }
        }
    }
    float* Vo = st.V + (cb ^ 1) * istride + (size_t)row * k;
    float* Mo = st.msV + (cb ^ 1) * istride + (size_t)row * k;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = lane + q * TKR_WAVE;
        if (e < k) {
            const float m2 = st.rho * ms[q] + (1.f - st.rho) * g[q] * g[q];
            Mo[e] = m2;
            Vo[e] = vr[q] - st.lr * g[q] / sqrtf(m2 + st.eps);
        }
    }
    if (lane == 0) {
        const float m2 = st.rho * msbr + (1.f - st.rho) * gb * gb;
        st.msb[(size_t)(cb ^ 1) * st.n_items + row] = m2;
        st.b[(size_t)(cb ^ 1) * st.n_items + row] = br - st.lr * gb / sqrtf(m2 + st.eps);
        st.istamp[row] = (serial << 1) | (cb ^ 1);
    }
}

template <int NE>
static int launch_step(const tkr_bpr_state& st, const int32_t* task, const int32_t* occ, int B, int serial,
                       float* loss_out, hipStream_t stream) {
    const int n_waves = 3 * B;
    const int grid = (n_waves + kWavesPerWG - 1) / kWavesPerWG;
    hipLaunchKernelGGL(bpr_step_kernel<NE>, dim3(grid), dim3(kStepThreads), 0, stream, st,
                       reinterpret_cast<const int4*>(task), reinterpret_cast<const int2*>(occ), serial, loss_out);
    return (int)hipGetLastError();
}

static int dispatch_step(const tkr_bpr_state& st, const int32_t* task, const int32_t* occ, int B, int serial,
                         float* loss_out, hipStream_t stream) {
    switch ((st.k + TKR_WAVE - 1) / TKR_WAVE) {
        case 1: return launch_step<1>(st, task, occ, B, serial, loss_out, stream);
        case 2: return launch_step<2>(st, task, occ, B, serial, loss_out, stream);
        case 3: return launch_step<3>(st, task, occ, B, serial, loss_out, stream);
        case 4: return launch_step<4>(st, task, occ, B, serial, loss_out, stream);
        default: return TKR_EUNSUPPORTED;       // k > 256
    }
}

}  // namespace tkr

static int check_state(const tkr_bpr_state* st) {
    if (!st || !st->U || !st->msU || !st->V || !st->msV || !st->b || !st->msb || !st->ustamp || !st->istamp)
        return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->k <= 0) return TKR_EINVAL;
    if (st->k > 256) return TKR_EUNSUPPORTED;
    return TKR_OK;
}

extern "C" int tkr_bpr_step(const tkr_bpr_state* st, const int32_t* task, const int32_t* occ,
                            int32_t batch_size, int32_t serial, float* loss_out, void* stream) {
    const int rc = check_state(st);
    if (rc != TKR_OK) return rc;
    if (!task || !occ || batch_size <= 0 || serial <= 0 || serial >= (1 << 30)) return TKR_EINVAL;
    return tkr::dispatch_step(*st, task, occ, batch_size, serial, loss_out, (hipStream_t)stream);
}

extern "C" int tkr_bpr_run(const tkr_bpr_state* st, const int32_t* task, const int32_t* occ,
                           int32_t batch_size, int32_t n_batches, int32_t first_serial, float* loss_out,
                           void* stream) {
    const int rc = check_state(st);
    if (rc != TKR_OK) return rc;
    if (!task || !occ || batch_size <= 0 || n_batches < 0 || first_serial <= 0 ||
        (int64_t)first_serial + n_batches >= (1 << 30))
        return TKR_EINVAL;
    const size_t stride_t = (size_t)3 * batch_size * 4, stride_o = (size_t)3 * batch_size * 2;
    for (int b = 0; b < n_batches; ++b) {
        const int r = tkr::dispatch_step(*st, task + b * stride_t, occ + b * stride_o, batch_size,
                                         first_serial + b, loss_out ? loss_out + b : nullptr,
                                         (hipStream_t)stream);
        if (r != 0) return r;
    }
    return TKR_OK;
}
