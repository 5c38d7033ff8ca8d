// K3 for ANY factor width -- the generic form of the column-plan step (csrc/vbpr_cols.hip; single/vbpr.py:50-73,114).
//
// The kernels of csrc/vbpr_cols.hip hold a half-width row (kh = k // 2 factors) in one or two registers per lane and a cem row in
// the float4s of a column group: kh <= 128, kh % 4 == 0.  single/vbpr.py:18 takes any k.  Here nothing is resident: every kernel
// walks the factor dimension in strides of its thread count, on the SAME plan (K1's launch records, the column plan of
// tkr_vbpr_colplan), the same double buffering, the same loss words and the same pair-sum launch (vbpr_pairsum_kernel works on
// alpha / beta only).  Slow -- a row task re-walks its occurrence list once per 64 factors, a column task its run once per 64 --
// but the same objective and updates; sums in another order than the register form, inside the tolerance of the step tests.
//
//   W1 vbpr_wide_project_kernel   one workgroup per triplet: P_t = (f_i - f_j).cem from the triplet's gather list (staged in LDS),
//                                 alpha_t, beta_t, e^alpha, e^beta, uce_u(t), the regularisers' share of the loss
//   L2 vbpr_pairsum_kernel        (csrc/vbpr_cols.hip)
//   W3 vbpr_wide_update_kernel    row blocks: a wave per launch record (wave 0 of a heavy team walks the whole team), factor by factor;
//                                 column blocks: a wave per feature column, TF's dense ApplyRMSProp on cem[c][.] and icb[c]
#include "tkr_common.h"
#include "../../include/tkr.h"
#include "vbpr_rows.h"

namespace tkr {

__global__ __launch_bounds__(256) void vbpr_wide_project_kernel(tkr_vbpr_state st, const int32_t* __restrict__ ti, const int32_t* __restrict__ tj,
                                                               const int32_t* __restrict__ tu, const int32_t* __restrict__ tpar,
                                                               const int32_t* __restrict__ tcnt, const int2* __restrict__ tent, int tcap, int B,
                                                               float* __restrict__ P, float* __restrict__ ab_out, float* __restrict__ Wraw,
                                                               float* __restrict__ loss_out) {
    extern __shared__ int2 s_ent[];                              // the triplet's gather list: (column, +-value bits)
    __shared__ float s_red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = blockIdx.x;
    const int kh = st.kh, k2 = 2 * kh;
    const int n = tcnt[t];
    for (int e = tid; e < n; e += 256) s_ent[e] = tent[(size_t)t * tcap + e];
    const int pr = tpar[t], u = tu[t], i = ti[t], j = tj[t];
    const float* urow = st.U + ((size_t)(pr & 1) * st.n_users + u) * k2;
    const float* ri = st.I + ((size_t)((pr >> 1) & 1) * st.n_items + i) * kh;
    const float* rj = st.I + ((size_t)((pr >> 2) & 1) * st.n_items + j) * kh;
    const float bi = st.irb[(size_t)((pr >> 1) & 1) * st.n_items + i], bj = st.irb[(size_t)((pr >> 2) & 1) * st.n_items + j];
    __syncthreads();
    const bool l2 = st.mode == 0;
    float d = 0.f, q = 0.f, reg = 0.f;
    for (int c = tid; c < kh; c += 256) {
        float acc = 0.f;
        for (int e = 0; e < n; ++e) acc = fmaf(__int_as_float(s_ent[e].y), st.cem[(size_t)s_ent[e].x * kh + c], acc);
        P[(size_t)t * kh + c] = acc;
        const float a = urow[c], b = urow[kh + c], x = ri[c], y = rj[c];
        Wraw[(size_t)t * kh + c] = b;
        d = fmaf(a, x - y, d);
        d = fmaf(b, acc, d);
        reg += l2 ? 0.5f * ((a * a + b * b) * st.lu + x * x * st.li + y * y * st.lj) : (fabsf(a) + fabsf(b)) * st.lu + fabsf(x) * st.li + fabsf(y) * st.lj;
    }
    for (int e = tid; e < n; e += 256) q = fmaf(__int_as_float(s_ent[e].y), st.icb[s_ent[e].x], q);
    d = wave_sum(d); q = wave_sum(q); reg = wave_sum(reg);
    if (lane == 0) { s_red[wave][0] = d; s_red[wave][1] = q; s_red[wave][2] = reg; }
    __syncthreads();
    if (tid == 0) {
        const float beta = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
        const float alpha = bi - bj + ((s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]));
        ab_out[t] = alpha; ab_out[B + t] = beta; ab_out[2 * B + t] = pair_exp(alpha); ab_out[3 * B + t] = pair_exp(beta);
        if (loss_out)
            loss_out[t] = ((s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2])) + (l2 ? 0.5f * (bi * bi + bj * bj) * st.lb : (fabsf(bi) + fabsf(bj)) * st.lb);
    }
}

// occurrence q of a record: the first four ride in it, the others in the occurrence lists (stride `team`)
__device__ __forceinline__ void wide_occurrence(const WaveRec& r, int q, const int2* __restrict__ occ, const int32_t* __restrict__ occt, int& oa, int& ob,
                                                int& ot) {
    if (q < 4) {
        oa = q == 0 ? r.oa[0] : q == 1 ? r.oa[1] : q == 2 ? r.oa[2] : r.oa[3];
        ob = q == 0 ? r.ob[0] : q == 1 ? r.ob[1] : q == 2 ? r.ob[2] : r.ob[3];
        ot = q == 0 ? r.ot[0] : q == 1 ? r.ot[1] : q == 2 ? r.ot[2] : r.ot[3];
    } else {
        const int2 o = occ[r.first + q * r.team];
        oa = o.x; ob = o.y; ot = occt[r.first + q * r.team];
    }
}

__global__ __launch_bounds__(256) void vbpr_wide_update_kernel(tkr_vbpr_state st, const int32_t* __restrict__ rec_all, const int2* __restrict__ occ,
                                                              const int32_t* __restrict__ occt, const int4* __restrict__ hdr,
                                                              const float* __restrict__ sS, const float* __restrict__ sT, const float* __restrict__ P,
                                                              const float* __restrict__ Wraw, const int4* __restrict__ colh, const int2* __restrict__ cent,
                                                              int n_row_blocks, int n_col_blocks, float* __restrict__ loss_out /*[B] | [B] | [column blocks]*/,
                                                              int B) {
    __shared__ float s_loss[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kh = st.kh, k2 = 2 * kh;
    const bool l2 = st.mode == 0;
    const size_t ustride = (size_t)st.n_users * k2, istride = (size_t)st.n_items * kh;
    if ((int)blockIdx.x < n_row_blocks) {
        // ---- row tasks (vbpr_rows_body, factor by factor)
        const int4 h4 = *hdr;
        const int n_blocks = __builtin_amdgcn_readfirstlane(h4.x), nlb = __builtin_amdgcn_readfirstlane(h4.y);
        for (int it = blockIdx.x; it < n_blocks; it += n_row_blocks) {
            const int blk = n_blocks - 1 - it;
            const bool heavy = blk >= nlb;
            if (heavy && wave != 0) continue;                    // wave 0 walks the records of the whole team, in wave order
            const int n_rec = heavy ? 4 : 1;
            const WaveRec head = read_rec(rec_all, 4, blk, heavy ? 0 : wave, lane);
            if (head.rowk == -1) continue;
            const bool is_item = head.rowk < 0;
            const int row = head.rowk & 0x7fffffff, par = head.par, width = is_item ? kh : k2;
            const float* src = is_item ? st.I + par * istride + (size_t)row * kh : st.U + par * ustride + (size_t)row * k2;
            const float* msrc = is_item ? st.msI + par * istride + (size_t)row * kh : st.msU + par * ustride + (size_t)row * k2;
            float* po = is_item ? st.I + (par ^ 1) * istride + (size_t)row * kh : st.U + (par ^ 1) * ustride + (size_t)row * k2;
            float* mo = is_item ? st.msI + (par ^ 1) * istride + (size_t)row * kh : st.msU + (par ^ 1) * ustride + (size_t)row * k2;
            const float br = is_item ? st.irb[(size_t)par * st.n_items + row] : 0.f;
            float gb = 0.f;
            for (int c0 = 0; c0 < width; c0 += 64) {             // (wave-uniform trip count)
                const int c = c0 + lane;
                const bool in = c < width;
                const float own = in ? src[c] : 0.f;
                float g = 0.f;
                for (int w = 0; w < n_rec; ++w) {
                    const WaveRec r = w == 0 ? head : read_rec(rec_all, 4, blk, w, lane);
                    if (r.rowk == -1) continue;
                    for (int q = 0; q < r.n_occ; ++q) {
                        int oa, ob, ot;
                        wide_occurrence(r, q, occ, occt, oa, ob, ot);
                        const float sa = sS[ot], sg = sT[ot];        // rows under alpha are scaled by S_t, rows under beta by T_t
                        if (is_item) {
                            const int u = oa & kIdMaskV, pu = (oa >> 30) & 1;
                            const bool role_j = ob < 0;
                            const float lam = role_j ? st.lj : st.li;
                            const float ur = in ? st.U[pu * ustride + (size_t)u * k2 + c] : 0.f;
                            g += (role_j ? sg : -sg) * ur + lam * (l2 ? own : sgn(own));
                            if (c0 == 0) gb += (role_j ? sa : -sa) + st.lb * (l2 ? br : sgn(br));
                        } else {
                            const int i = oa & kIdMaskV, pi = (oa >> 30) & 1, j = ob & kIdMaskV, pj = (ob >> 30) & 1;
                            float partner = 0.f;
                            if (in) partner = c < kh ? st.I[pi * istride + (size_t)i * kh + c] - st.I[pj * istride + (size_t)j * kh + c] : P[(size_t)ot * kh + c - kh];
                            g += -sg * partner + st.lu * (l2 ? own : sgn(own));
                        }
                    }
                }
                if (in) {
                    const float m2 = st.rho * msrc[c] + (1.f - st.rho) * g * g;
                    mo[c] = m2;
                    po[c] = own - st.lr * g / sqrtf(m2 + st.eps);
                }
            }
            if (is_item && lane == 0) {
                const float m2 = st.rho * st.msirb[(size_t)par * st.n_items + row] + (1.f - st.rho) * gb * gb;
                st.msirb[(size_t)(par ^ 1) * st.n_items + row] = m2;
                st.irb[(size_t)(par ^ 1) * st.n_items + row] = br - st.lr * gb / sqrtf(m2 + st.eps);
            }
        }
        return;
    }
    // ---- column tasks: four feature columns per block, a wave each
    const int cb = (int)blockIdx.x - n_row_blocks;
    const int c = cb * 4 + wave;
    float lpart = 0.f;
    if (c < st.d) {
        const int4 h0 = colh[(size_t)c * 2];
        const int n = h0.x, beg = h0.y;
        for (int l0 = 0; l0 < kh; l0 += 64) {
            const int l = l0 + lane;
            const bool in = l < kh;
            float g = 0.f;
            for (int e = 0; e < n; ++e) {
                const int2 en = cent[beg + e];                   // (triplet, +-value bits)
                g = fmaf(-__int_as_float(en.y) * sT[en.x], in ? Wraw[(size_t)en.x * kh + l] : 0.f, g);
            }
            if (in) {
                const float v = st.cem[(size_t)c * kh + l];
                const float gg = g + st.le * (l2 ? v : sgn(v));
                lpart += l2 ? 0.5f * st.le * v * v : st.le * fabsf(v);
                float ms = st.mscem[(size_t)c * kh + l];
                ms += (gg * gg - ms) * (1.f - st.rho);
                st.mscem[(size_t)c * kh + l] = ms;
                st.cem[(size_t)c * kh + l] = v - st.lr * gg / sqrtf(ms + st.eps);
            }
        }
        float gi = 0.f;
        for (int e = lane; e < n; e += 64) {
            const int2 en = cent[beg + e];
            gi = fmaf(-__int_as_float(en.y), sS[en.x], gi);
        }
        gi = wave_sum(gi);
        if (lane == 0) {
            const float v = st.icb[c];
            const float gg = gi + st.lb * (l2 ? v : sgn(v));
            lpart += l2 ? 0.5f * st.lb * v * v : st.lb * fabsf(v);
            float ms = st.msicb[c];
            ms += (gg * gg - ms) * (1.f - st.rho);
            st.msicb[c] = ms;
            st.icb[c] = v - st.lr * gg / sqrtf(ms + st.eps);
        }
    }
    if (loss_out) {
        lpart = wave_sum(lpart);
        if (lane == 0) s_loss[wave] = lpart;
        __syncthreads();
        if (threadIdx.x == 0) loss_out[2 * B + cb] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
    }
}

// one batch of the generic form; the pair-sum launch between the two is the caller's (csrc/vbpr_cols.hip)
__attribute__((visibility("hidden"))) void vbpr_wide_project(const tkr_vbpr_state& st, const int32_t* ti, const int32_t* tj, const int32_t* tu, const int32_t* tp,
                                                             const int32_t* tc, const int2* te, int tcap, int B, float* P, float* ab2, float* Wm,
                                                             float* loss, hipStream_t s) {
    hipLaunchKernelGGL(vbpr_wide_project_kernel, dim3(B), dim3(256), (size_t)tcap * sizeof(int2), s, st, ti, tj, tu, tp, tc, te, tcap, B, P, ab2, Wm, loss);
}
__attribute__((visibility("hidden"))) int vbpr_wide_col_blocks(int d) { return (d + 3) / 4; }
__attribute__((visibility("hidden"))) void vbpr_wide_update(const tkr_vbpr_state& st, const int32_t* rec, const int2* occ2, const int32_t* occt, const int4* hdr4,
                                                            const float* s_buf, const float* t_buf, const float* P, const float* Wm, const int4* colh,
                                                            const int2* cent, int B, float* loss, hipStream_t s) {
    const int n_row_blocks = vbpr_grid(B, 4), n_col_blocks = vbpr_wide_col_blocks(st.d);
    hipLaunchKernelGGL(vbpr_wide_update_kernel, dim3(n_row_blocks + n_col_blocks), dim3(256), 0, s, st, rec, occ2, occt, hdr4, s_buf, t_buf, P, Wm, colh, cent,
                       n_row_blocks, n_col_blocks, loss, B);
}

}  // namespace tkr
