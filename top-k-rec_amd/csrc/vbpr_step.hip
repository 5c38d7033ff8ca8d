// K3 -- one VBPR mini-batch (content-aware BPR).
//
// Replaces sess.run([solver, obj]) of single/vbpr.py:114 on the graph of single/vbpr.py:50-73, and
// the per-batch host gather + H2D feed of two dense [B, d] feature slices (vbpr.py:114): the
// feature matrix stays resident in HBM and rows are gathered by the kernels.
//
//   alpha_t = irb_i - irb_j + (f_i - f_j).icb          beta_t = <ure_u, ire_i - ire_j> + <uce_u, (f_i - f_j).cem>
//   obj = sum over ALL PAIRS (a, b) of the batch of log(1 + exp(-(alpha_a + beta_b))) + regularisers
//
// The pair sum is what the reference's graph computes: item_rating_bias is [n_items, 1], so irbb - jrbb is [B, 1], x_ui - x_uj is
// [B] and matmul(ic - jc, icb) is [B, 1] (vbpr.py:54-61) -- TensorFlow broadcasts their sum to [B, B] and reduce_sum (:64)
// runs over all of it.  (BPR's item_bias is 1-D, bpr.py:79: there the sum is the B triplets.)  The objective separates:
// with S_a = sum_b sigma(-(alpha_a + beta_b)) and T_b = sum_a sigma(-(alpha_a + beta_b)), every variable under alpha (irb, icb)
// gets the gradient of one triplet scaled by S_t, every variable under beta (ure, uce, ire, cem) scaled by T_t.
//
// iceb/jceb enter the reference only through x_ui - x_uj and are not regularised, so projecting the
// DIFFERENCE once is exact in real arithmetic and halves the contraction.
//
// Five launches per batch, all on pre-step values:
//   V1  project   P_t = (f_i - f_j).cem, q_t = (f_i - f_j).icb      fp32 MFMA, split over d (K)
//   V1r reduce    P_t, q_t = sum of the split-K partials (slice-parallel, fixed order)
//   V1b occur     per user occurrence: alpha_t, beta_t, uce_u, the regularisers' share of the loss
//                                                                   (plan of K1: parities inline)
//   V1p pair      S_t, T_t (one wave per triplet walks the batch twice), the pair sum of the loss, W_t = -T_t * uce_u
//   V2  rows      sparse RMSProp on [ure|uce] rows, ire rows, irb   (same launch records as K2,
//                 gradients use the stored S_t, T_t, P_t -- nothing is recomputed)
//   V3  dense     G_cem = D^T.W + le*cem, G_icb = D^T.(-S) + lb*icb  fp32 MFMA over the batch,
//                 with TF's DENSE ApplyRMSProp fused into the epilogue (every element of cem/icb is
//                 updated every batch; vbpr.py:65,67,73)
//
// Sparse view (st.f_ptr != NULL; tf-idf-like content features are ~0.5 % dense): V1/V1r are replaced by S1, one wave
// per triplet gathering one cem row per nonzero of f_i and f_j; V2's item tasks additionally store, per unique item
// of the batch, A = sum over its occurrences of (+/-)W_t and a = sum of (-/+)s_t and enter the item into the
// batch's membership bitmap; V3 is replaced by S3, one wave per feature column walking the column's (item, value) list
// (CSC), adding value * A[slot] for the tagged items in ascending item order (deterministic) and applying the same
// dense RMSProp.  Work drops from 4*d*kh flop per triplet to ~2*nnz_row*kh; what remains is the 16*d*kh B of
// dense optimizer traffic per batch plus the 8 B per nonzero of the CSC walk.
//
// Roofline (dense features, d = 20,000, kh = 64, B = 256): 4*d*kh = 5.12 MFLOP per triplet on the
// fp32 MFMA (V1 + V3) against 2*4d B = 160 KB of feature rows per triplet read twice from HBM/MALL
// and 16*d*kh B = 20 MB of dense optimizer traffic per batch: MFMA and HBM ceilings are within 10 %
// of each other, both reported by bench.
#include "tkr_common.h"
#include "../../include/tkr.h"
#include "vbpr_rows.h"

extern "C" int tkr_plan_team(int32_t batch_size);
extern "C" int tkr_plan_max_blocks(int32_t batch_size);
extern "C" int64_t tkr_vbpr_workspace_floats(int32_t batch_size, int32_t kh, int32_t d);
extern "C" __attribute__((visibility("hidden"))) int64_t tkr_vbpr_workspace_core_floats(int32_t batch_size, int32_t kh, int32_t d);

namespace tkr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSlice = 128;                      // d-columns per split-K slice of V1 (64 per MFMA k-slot)

__host__ __device__ inline int vbpr_slices(int d) { return (d + kSlice - 1) / kSlice; }

// ------------------------------------------------------------------------------------------------
// V1: Ppart[s][t][0..kh) = sum_{c in slice s} (f_i[c]-f_j[c]) * cem[c][.],  Ppart[s][t][kh] = same with icb
// one wave per (slice, 32 triplets); MFMA rows = triplets, columns = kh, K = slice columns
template <int NT>
__global__ __launch_bounds__(64) void vbpr_project_kernel(tkr_vbpr_state st, const int32_t* __restrict__ ti,
                                                         const int32_t* __restrict__ tj, int B,
                                                         float* __restrict__ ppart) {
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int s = blockIdx.x, t = blockIdx.y * 32 + m;
    const int d = st.d, kh = st.kh, NP = kh + 1;
    const bool tv = t < B;
    const float* fi = st.feat + (size_t)(tv ? ti[t] : 0) * d;
    const float* fj = st.feat + (size_t)(tv ? tj[t] : 0) * d;
    const int base = s * kSlice + h * (kSlice / 2);
    const bool vec = (d & 3) == 0;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float q = 0.f;
    constexpr int CH = 16;                                       // columns per batch: all loads first, then the MFMAs
    for (int kk = 0; kk < kSlice / 2; kk += CH) {
        const int c = base + kk;
        float a[CH], w[CH], bop[CH][NT];
#pragma unroll
        for (int e = 0; e < CH; ++e) a[e] = 0.f;
        if (tv) {
            if (vec && c + CH - 1 < d) {
#pragma unroll
                for (int g = 0; g < CH / 4; ++g) {
                    const float4 x = *reinterpret_cast<const float4*>(fi + c + 4 * g);
                    const float4 y = *reinterpret_cast<const float4*>(fj + c + 4 * g);
                    a[4 * g + 0] = x.x - y.x; a[4 * g + 1] = x.y - y.y; a[4 * g + 2] = x.z - y.z; a[4 * g + 3] = x.w - y.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < CH; ++e)
                    if (c + e < d) a[e] = fi[c + e] - fj[c + e];
            }
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int ce = c + e;
            const bool cv = ce < d;
            w[e] = cv ? st.icb[ce] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 32 + m;
                bop[e][nt] = (cv && n < kh) ? st.cem[(size_t)ce * kh + n] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            q = fmaf(a[e], w[e], q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bop[e][nt], acc[nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int to = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = nt * 32 + m;
            if (to < B && n < kh) ppart[((size_t)s * B + to) * NP + n] = acc[nt][r];
        }
    }
    q += __shfl_xor(q, 32, 64);
    if (h == 0 && tv) ppart[((size_t)s * B + t) * NP + kh] = q;
}

// ------------------------------------------------------------------------------------------------
// V1r: P[t][n] = sum_s Ppart[s][t][n] (n < kh), Q[t] = sum_s Ppart[s][t][kh].  One workgroup per triplet:
// 16 slice groups x 64 columns, every thread sums its slices in ascending order, groups combined in order.
__global__ __launch_bounds__(1024) void vbpr_reduce_kernel(const float* __restrict__ ppart, int S, int B, int kh,
                                                          float* __restrict__ P, float* __restrict__ Q) {
    constexpr int SG = 16;                                       // slice groups: thread (sg, ln) sums slices sg, sg+16, ...
    __shared__ float red[SG][65];
    const int t = blockIdx.x, sg = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int NP = kh + 1;
    for (int n0 = 0; n0 < NP; n0 += 64) {
        const int n = n0 + ln;
        float a = 0.f;
        if (n < NP) {
#pragma unroll 10
            for (int s = sg; s < S; s += SG) a += ppart[((size_t)s * B + t) * NP + n];
        }
        red[sg][ln] = a;
        __syncthreads();
        if (sg == 0 && n < NP) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < SG; ++g) v += red[g][ln];        // fixed order
            if (n < kh) P[(size_t)t * kh + n] = v; else Q[t] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// V1b: user occurrences -> s_t, P_t, W_t, loss.   NH = ceil(kh / 64)
template <int NH, int kVTeam>
__global__ __launch_bounds__((kVTeam * TKR_WAVE)) void vbpr_occur_kernel(
    tkr_vbpr_state st, const int32_t* __restrict__ rec_all, const int2* __restrict__ occ,
    const int32_t* __restrict__ occt, const int4* __restrict__ hdr, int B, const float* __restrict__ Q,
    float* __restrict__ ab_out /*[4][B]: alpha, beta, e^alpha, e^beta*/, const float* __restrict__ P, float* __restrict__ Wm,
    float* __restrict__ loss_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_blocks = __builtin_amdgcn_readfirstlane((*hdr).x);
    const int kh = st.kh, k2 = 2 * kh;
    const size_t ustride = (size_t)st.n_users * k2, istride = (size_t)st.n_items * kh;
    const bool l2 = st.mode == 0;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const WaveRec r = read_rec(rec_all, kVTeam, blk, wave, lane);
        if (r.rowk < 0) continue;                       // item task or idle wave (-1): nothing to do here
        const int u = r.rowk;
        const float* urow = st.U + r.par * ustride + (size_t)u * k2;
        float ure[NH], uce[NH];
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const int c = lane + e * 64;
            ure[e] = c < kh ? urow[c] : 0.f;
            uce[e] = c < kh ? urow[kh + c] : 0.f;
        }
        float loss = 0.f, loss_lane = 0.f;
        for (int done = 0; done < r.n_occ; done += 4) {
            const int n = min(4, r.n_occ - done);
            int oa[4], ob[4], ot[4];
            next_occ(r, done, n, lane, occ, occt, oa, ob, ot);
            for (int q = 0; q < n; ++q) {
                const int i = oa[q] & kIdMaskV, pi = (oa[q] >> 30) & 1;
                const int j = ob[q] & kIdMaskV, pj = (ob[q] >> 30) & 1;
                const int t = ot[q];
                float p[NH], vi[NH], vj[NH];
                const float qsum = Q[t];
#pragma unroll
                for (int e = 0; e < NH; ++e) {
                    const int c = lane + e * 64;
                    p[e] = c < kh ? P[(size_t)t * kh + c] : 0.f;
                }
                const float* ri = st.I + pi * istride + (size_t)i * kh;
                const float* rj = st.I + pj * istride + (size_t)j * kh;
                float d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int e = 0; e < NH; ++e) {
                    const int c = lane + e * 64;
                    vi[e] = c < kh ? ri[c] : 0.f;
                    vj[e] = c < kh ? rj[c] : 0.f;
                    d1 = fmaf(ure[e], vi[e] - vj[e], d1);
                    d2 = fmaf(uce[e], p[e], d2);
                }
                const float bi = st.irb[(size_t)pi * st.n_items + i], bj = st.irb[(size_t)pj * st.n_items + j];
                const float alpha = bi - bj + qsum, beta = wave_sum(d1) + wave_sum(d2);
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
#pragma unroll
                for (int e = 0; e < NH; ++e) {
                    const int c = lane + e * 64;
                    if (c < kh) Wm[(size_t)t * kh + c] = uce[e];          // scaled by -T_t once the pair kernel knows it
                }
                if (lane == 0) { ab_out[t] = alpha; ab_out[B + t] = beta; ab_out[2 * B + t] = pair_exp(alpha); ab_out[3 * B + t] = pair_exp(beta); }
            }
        }
        if (loss_out) {
            const float tot = wave_sum(loss_lane) + loss;
            if (lane == 0) loss_add_spread(loss_out, tot);
        }
    }
}

// V1p: the [B, B] pair sum of the reference's objective (see the head of this file).  One wave per triplet t:
//   S_t = sum_b sigma(-(alpha_t + beta_b))   scales the gradients of the variables under alpha (irb, icb)
//   T_t = sum_a sigma(-(alpha_a + beta_t))   scales those under beta (ure, uce, ire, cem); W_t = -T_t * uce_u for V3 / S3
// and the loss gets sum_b log(1 + exp(-(alpha_t + beta_b))).  Sums run in index order (lane l takes b = l, l+64, ...; DPP tree).
__global__ __launch_bounds__(256) void vbpr_pair_kernel(const float* __restrict__ ab /*[2][B]*/, int B, int kh,
                                                       float* __restrict__ sS, float* __restrict__ sT, float* __restrict__ Wm,
                                                       float* __restrict__ loss_out) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= B) return;
    const float* beta = ab + B;
    const float* ealpha = ab + 2 * B;                 // e^alpha, e^beta (pair_exp, written beside alpha and beta)
    const float* ebeta = ab + 3 * B;
    const float a_t = ab[t], ea_t = ealpha[t], eb_t = ebeta[t];
    float s_row = 0.f, s_col = 0.f, loss = 0.f;
    // eight partners of the lane in flight (a load, a wait and three transcendentals per trip of the plain loop: 87 us of the 205 of a
    // batch of 8192, where the arithmetic alone is ~30); the sums still run in index order, one accumulator each: same bits
    constexpr int UN = 8;                               // (16: no faster)
    for (int o0 = lane; o0 < B; o0 += 64 * UN) {
        float eb[UN], ea[UN], be[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int o = min(o0 + 64 * q, B - 1);
            eb[q] = ebeta[o]; ea[q] = ealpha[o];
            be[q] = loss_out ? beta[o] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            if (o0 + 64 * q < B) {
                s_row += pair_sigmoid(ea_t, eb[q]);
                if (loss_out) loss += pair_softplus_neg(ea_t, eb[q], a_t + be[q]);
                s_col += pair_sigmoid(ea[q], eb_t);
            }
        }
    }
    s_row = wave_sum(s_row);
    s_col = wave_sum(s_col);
    if (lane == 0) { sS[t] = s_row; sT[t] = s_col; }
    if (Wm)                                           // (the column-plan path scales by T_t where it reads the row)
        for (int c = lane; c < kh; c += 64) Wm[(size_t)t * kh + c] *= -s_col;
    if (loss_out) {
        loss = wave_sum(loss);
        if (lane == 0) loss_add_spread(loss_out, loss);
    }
}

template <int NE, int kVTeam>
__global__ __launch_bounds__((kVTeam * TKR_WAVE)) void vbpr_rows_kernel(
    tkr_vbpr_state st, const int32_t* __restrict__ rec_all, const int2* __restrict__ occ,
    const int32_t* __restrict__ occt, const int4* __restrict__ hdr, const float* __restrict__ s_in /*S_t*/,
    const float* __restrict__ t_in /*T_t*/, const float* __restrict__ P, const float* __restrict__ Wm,
    float* __restrict__ Aw /*[slots][kh] or null*/, float* __restrict__ ab /*[slots]*/) {
    __shared__ float red[kVTeam][NE * TKR_WAVE + 1];
    __shared__ float red2[kVTeam][NE * TKR_WAVE + 1];
    vbpr_rows_body<NE, kVTeam>(st, rec_all, occ, occt, hdr, PairSumArrays{s_in, t_in}, P, Wm, Aw, ab, red, red2, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// V3: 64 rows of cem/icb per workgroup; 4 waves split the batch (K of the contraction); MFMA rows =
// feature columns c (two interleaved m-tiles fed by one float2 load), columns = kh, K = triplets.
template <int NT>
__global__ __launch_bounds__(256) void vbpr_dense_kernel(tkr_vbpr_state st, const int32_t* __restrict__ ti,
                                                        const int32_t* __restrict__ tj, int B,
                                                        const float* __restrict__ s_in, const float* __restrict__ Wm,
                                                        float* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NW = 4, LD = NT * 32 + 1;
    float* red = sm;                                   // [NW][64][LD]
    float* redi = sm + NW * 64 * LD;                   // [NW][2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
    const int d = st.d, kh = st.kh;
    const int c0 = blockIdx.x * 64, cA = c0 + 2 * m;
    const int TB = (((B + NW - 1) / NW) + 1) & ~1, TH = TB / 2;
    f32x16 acc[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[q][nt] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float gi0 = 0.f, gi1 = 0.f;
    const bool pair = ((d & 1) == 0) && (cA + 1 < d);
    constexpr int UN = 8;                                        // triplets per batch of loads
    for (int k0 = 0; k0 < TH; k0 += 32) {
        // this half's next 32 triplet ids / s_t, one per lane, then broadcast inside the half
        const int tl = wave * TB + h * TH + k0 + m;
        const bool lv = (k0 + m < TH) && (tl < B);
        const int my_i = lv ? ti[tl] : 0, my_j = lv ? tj[tl] : 0;
        const float my_s = lv ? s_in[tl] : 0.f;
        const int lim = min(32, TH - k0);
        for (int kb = 0; kb < lim; kb += UN) {
            float a0[UN], a1[UN], sg[UN], bw[UN][NT];
#pragma unroll
            for (int e = 0; e < UN; ++e) {
                const int src = (lane & 32) | ((kb + e) & 31);
                const int it = __shfl(my_i, src, 64), jt = __shfl(my_j, src, 64);
                sg[e] = __shfl(my_s, src, 64);
                const int t = wave * TB + h * TH + k0 + kb + e;
                const bool tv = (kb + e < lim) && (t < B);
                a0[e] = 0.f; a1[e] = 0.f;
                if (tv) {
                    const float* fi = st.feat + (size_t)it * d;
                    const float* fj = st.feat + (size_t)jt * d;
                    if (pair) {
                        const float2 x = *reinterpret_cast<const float2*>(fi + cA);
                        const float2 y = *reinterpret_cast<const float2*>(fj + cA);
                        a0[e] = x.x - y.x; a1[e] = x.y - y.y;
                    } else {
                        if (cA < d) a0[e] = fi[cA] - fj[cA];
                        if (cA + 1 < d) a1[e] = fi[cA + 1] - fj[cA + 1];
                    }
                } else {
                    sg[e] = 0.f;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = nt * 32 + m;
                    bw[e][nt] = (tv && n < kh) ? Wm[(size_t)t * kh + n] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < UN; ++e) {
                gi0 = fmaf(-sg[e], a0[e], gi0);
                gi1 = fmaf(-sg[e], a1[e], gi1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bw[e][nt], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bw[e][nt], acc[1][nt], 0, 0, 0);
                }
            }
        }
    }
    // C layout: column = lane&31 (n), row = (r&3)+8*(r>>2)+4*h = m-index -> feature column c0 + 2*row + q
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mi = (r & 3) + 8 * (r >> 2) + 4 * h;
                red[(wave * 64 + 2 * mi + q) * LD + nt * 32 + m] = acc[q][nt][r];
            }
    redi[(wave * 2 + h) * 64 + 2 * m] = gi0;
    redi[(wave * 2 + h) * 64 + 2 * m + 1] = gi1;
    __syncthreads();
    const bool l2 = st.mode == 0;
    float lpart = 0.f;
    for (int idx = tid; idx < 64 * kh; idx += 256) {
        const int cr = idx / kh, n = idx % kh, c = c0 + cr;
        if (c >= d) continue;
        float g = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) g += red[(w * 64 + cr) * LD + n];
        const size_t o = (size_t)c * kh + n;
        const float v = st.cem[o];
        g += st.le * (l2 ? v : sgn(v));
        lpart += l2 ? 0.5f * st.le * v * v : st.le * fabsf(v);
        float ms = st.mscem[o];
        ms += (g * g - ms) * (1.f - st.rho);                 // TF dense ApplyRMSProp
        st.mscem[o] = ms;
        st.cem[o] = v - st.lr * g / sqrtf(ms + st.eps);
    }
    if (tid < 64 && c0 + tid < d) {
        const int c = c0 + tid;
        float g = 0.f;
#pragma unroll
        for (int w = 0; w < NW * 2; ++w) g += redi[w * 64 + tid];
        const float v = st.icb[c];
        g += st.lb * (l2 ? v : sgn(v));
        lpart += l2 ? 0.5f * st.lb * v * v : st.lb * fabsf(v);
        float ms = st.msicb[c];
        ms += (g * g - ms) * (1.f - st.rho);
        st.msicb[c] = ms;
        st.icb[c] = v - st.lr * g / sqrtf(ms + st.eps);
    }
    if (loss_out) {
        lpart = wave_sum(lpart);
        if (lane == 0 && lpart != 0.f) loss_add_spread(loss_out, lpart);
    }
}

// ------------------------------------------------------------------------------------------------
// S1 (sparse view): P_t = sum_nz(f_i) v * cem[c] - sum_nz(f_j) v * cem[c], q_t likewise with icb.  One workgroup of
// four waves per triplet: every wave takes a contiguous quarter of each item's nonzeros, reads 64 (column, value)
// pairs at a time and gathers one kh-wide cem row per nonzero, 16 gathers in flight; the four partial sums are added
// in wave order (fixed summation order).  The kernel is a chain of dependent gathers, so it wants many short waves.
template <int NH>
__global__ __launch_bounds__(256) void vbpr_sproject_kernel(tkr_vbpr_state st, const int32_t* __restrict__ ti,
                                                           const int32_t* __restrict__ tj, int B, float* __restrict__ P,
                                                           float* __restrict__ Q, const int32_t* __restrict__ tu,
                                                           const int32_t* __restrict__ tpar, float* __restrict__ ab_out,
                                                           float* __restrict__ Wm, float* __restrict__ loss_out) {
    __shared__ float red[4][NH * 64 + 1];
    __shared__ uint32_t s_par;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = blockIdx.x;
    const int kh = st.kh;
    if (t == 0) {                                       // new batch: advance the counter, clear the bitmap of the batch before
        const SparseScratch x = sparse_scratch(st);
        if (threadIdx.x == 0) { const uint32_t c = *x.counter + 1u; *x.counter = c; s_par = c & 1u; }
        __syncthreads();
        uint32_t* other = reinterpret_cast<uint32_t*>(x.member(s_par ^ 1u));
        for (int w = threadIdx.x; w < x.map_words; w += blockDim.x) other[w] = 0u;
    }
    float acc[NH], q = 0.f;
#pragma unroll
    for (int e = 0; e < NH; ++e) acc[e] = 0.f;
    // Fused scoring (K1 handed over the three row parities of the triplet): wave 0 issues the loads of [ure|uce]_u, ire_i,
    // ire_j and the two biases now, so they are in flight while the four waves gather the feature projections.
    const bool fused = tpar != nullptr;
    float ure[NH], uce[NH], vi[NH], vj[NH], bi = 0.f, bj = 0.f;
    if (fused && wave == 0) {
        const int pr = tpar[t], k2 = 2 * kh;
        const float* urow = st.U + ((size_t)(pr & 1) * st.n_users + tu[t]) * k2;
        const float* ri = st.I + ((size_t)((pr >> 1) & 1) * st.n_items + ti[t]) * kh;
        const float* rj = st.I + ((size_t)((pr >> 2) & 1) * st.n_items + tj[t]) * kh;
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const int c = min(lane + e * 64, kh - 1);
            const bool ok = lane + e * 64 < kh;
            const float a = urow[c], b = urow[kh + c], x = ri[c], y = rj[c];
            ure[e] = ok ? a : 0.f; uce[e] = ok ? b : 0.f; vi[e] = ok ? x : 0.f; vj[e] = ok ? y : 0.f;
        }
        bi = st.irb[(size_t)((pr >> 1) & 1) * st.n_items + ti[t]];
        bj = st.irb[(size_t)((pr >> 2) & 1) * st.n_items + tj[t]];
    }
    for (int side = 0; side < 2; ++side) {
        const int item = side ? tj[t] : ti[t];
        const float sign = side ? -1.f : 1.f;
        const int b0 = st.f_ptr[item], e0i = st.f_ptr[item + 1];
        const int chunk = (e0i - b0 + 3) >> 2;
        const int beg = b0 + wave * chunk, end = min(e0i, beg + chunk);
        for (int p0 = beg; p0 < end; p0 += 64) {
            const int p = p0 + lane;
            const int col = p < end ? st.f_col[p] : 0;
            const float val = p < end ? sign * st.f_val[p] : 0.f;
            const int n = min(64, end - p0);
            // groups of 16 nonzeros: the row gathers are issued together; lanes past the end hold (column 0, value 0)
            for (int e0 = 0; e0 < n; e0 += 16) {
                float rowv[16][NH], bv[16], vv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int c = __builtin_amdgcn_readlane(col, (e0 + u) & 63);
                    vv[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), (e0 + u) & 63));
                    const float* crow = st.cem + (size_t)c * kh;
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh) rowv[u][hh] = crow[min(lane + hh * 64, kh - 1)];
                    bv[u] = st.icb[c];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh) acc[hh] = fmaf(vv[u], rowv[u][hh], acc[hh]);
                    q = fmaf(vv[u], bv[u], q);
                }
            }
        }
    }
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) red[wave][lane + hh * 64] = acc[hh];
    if (lane == 0) red[wave][NH * 64] = q;
    __syncthreads();
    if (wave == 0) {
        float p[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const int n2 = lane + hh * 64;
            p[hh] = ((red[0][n2] + red[1][n2]) + red[2][n2]) + red[3][n2];
            if (n2 < kh) P[(size_t)t * kh + n2] = p[hh];
            else p[hh] = 0.f;
        }
        const float qsum = ((red[0][NH * 64] + red[1][NH * 64]) + red[2][NH * 64]) + red[3][NH * 64];
        if (lane == 0) Q[t] = qsum;
        if (fused) {                                     // what V1b does per user occurrence, here per triplet
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
                if (c < kh) Wm[(size_t)t * kh + c] = uce[e];              // scaled by -T_t once the pair kernel knows it
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
                if (lane == 0) loss_add_spread(loss_out, tot);
            }
        }
    }
}

// S3 (sparse view): one wave per feature column c.  G_cem[c] = sum over the column's items that are in the batch
// (bit set in the batch's bitmap) of value * A[slot], G_icb[c] likewise with a[slot], + regulariser, then TF's dense ApplyRMSProp on
// cem[c][.] and icb[c] exactly as V3 does.
template <int NH>
__global__ __launch_bounds__(256) void vbpr_sdense_kernel(tkr_vbpr_state st, const float* __restrict__ Aw,
                                                         const float* __restrict__ ab, float* __restrict__ loss_out) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= st.d) return;
    const int kh = st.kh;
    float g[NH], gi = 0.f;
#pragma unroll
    for (int e = 0; e < NH; ++e) g[e] = 0.f;
    const int beg = st.c_ptr[c], end = st.c_ptr[c + 1];
    const SparseScratch x = sparse_scratch(st);
    const unsigned char* __restrict__ member = x.member(*x.counter);
    // the row of cem / its slot do not depend on the walk: issue their loads first
    float pv[NH], pms[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        const size_t o = (size_t)c * kh + min(lane + hh * 64, kh - 1);
        pv[hh] = st.cem[o];
        pms[hh] = st.mscem[o];
    }
    const float bv0 = st.icb[c], bms0 = st.msicb[c];
    for (int p0 = beg; p0 < end; p0 += 64) {
        const int p = p0 + lane;
        const bool valid = p < end;
        const int item = valid ? st.c_item[p] : 0;
        const float val = valid ? st.c_val[p] : 0.f;
        const bool hit = valid && member[item] != 0;
        const int slot_l = hit ? x.slots[item] : 0;
        uint64_t m = __ballot(hit);
        while (m) {                                     // ascending lane = ascending item: a fixed summation order
            // four hits per round: their A rows are gathered together (one memory round trip per round, not per hit -- a column
            // has ~4 hits at batch 256); a round with fewer hits repeats the first slot with value 0
            int slot[4];
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int l = m ? __ffsll((long long)m) - 1 : 0;
                const bool have = m != 0;
                m &= m - 1;
                slot[u] = have ? __builtin_amdgcn_readlane(slot_l, l) : slot[0];
                v[u] = have ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), l)) : 0.f;
            }
            float arow[4][NH], av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) arow[u][hh] = Aw[(size_t)slot[u] * kh + min(lane + hh * 64, kh - 1)];
                av[u] = ab[slot[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) g[hh] = fmaf(v[u], arow[u][hh], g[hh]);
                gi = fmaf(v[u], av[u], gi);
            }
        }
    }
    const bool l2 = st.mode == 0;
    float lpart = 0.f;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        const int n2 = lane + hh * 64;
        if (n2 < kh) {
            const size_t o = (size_t)c * kh + n2;
            const float v = pv[hh];
            const float gg = g[hh] + st.le * (l2 ? v : sgn(v));
            lpart += l2 ? 0.5f * st.le * v * v : st.le * fabsf(v);
            float ms = pms[hh];
            ms += (gg * gg - ms) * (1.f - st.rho);               // TF dense ApplyRMSProp
            st.mscem[o] = ms;
            st.cem[o] = v - st.lr * gg / sqrtf(ms + st.eps);
        }
    }
    if (lane == 0) {
        const float v = bv0;
        const float gg = gi + st.lb * (l2 ? v : sgn(v));
        lpart += l2 ? 0.5f * st.lb * v * v : st.lb * fabsf(v);
        float ms = bms0;
        ms += (gg * gg - ms) * (1.f - st.rho);
        st.msicb[c] = ms;
        st.icb[c] = v - st.lr * gg / sqrtf(ms + st.eps);
    }
    if (loss_out) {
        lpart = wave_sum(lpart);
        if (lane == 0 && lpart != 0.f) loss_add_spread(loss_out, lpart);
    }
}


template <int NT, int TEAM>
static int launch_vbpr_t(const tkr_vbpr_state& st, const int32_t* ti, const int32_t* tj, const int32_t* rec,
                         const int32_t* occ, const int32_t* hdr, const int32_t* occt, int B, float* ws, float* loss,
                         hipStream_t stream, const int32_t* tu, const int32_t* tpar) {
    const int kh = st.kh, S = vbpr_slices(st.d);
    float* ppart = ws;
    float* s_buf = ppart + (size_t)S * B * (kh + 1);
    float* P = s_buf + B;
    float* Wm = P + (size_t)B * kh;
    float* Q = Wm + (size_t)B * kh;
    float* ab2 = ws + tkr_vbpr_workspace_core_floats(B, kh, st.d) - 5 * (size_t)B;      // alpha, beta, e^alpha, e^beta [B] of the batch
    float* t_buf = ab2 + 4 * (size_t)B;                                            // T_t (s_buf holds S_t)
    const int2* occ2 = reinterpret_cast<const int2*>(occ);
    const int4* hdr4 = reinterpret_cast<const int4*>(hdr);
    const dim3 rgrid(vbpr_grid(B, TEAM)), rblock(TEAM * 64);
    const int NH = (kh + 63) / 64, NE = (2 * kh + 63) / 64;
    const bool sparse = st.f_ptr != nullptr;
    float* Aw = nullptr;
    float* ab = nullptr;
    if (sparse) {
        Aw = Q + B;
        ab = Aw + (size_t)tkr_plan_max_blocks(B) * TEAM * kh;
        // with the per-triplet parities of K1 the projection also scores the triplet: no per-occurrence launch
        if (NH == 1) hipLaunchKernelGGL(vbpr_sproject_kernel<1>, dim3(B), dim3(256), 0, stream, st, ti, tj, B, P, Q, tu, tpar, ab2, Wm, loss);
        else hipLaunchKernelGGL(vbpr_sproject_kernel<2>, dim3(B), dim3(256), 0, stream, st, ti, tj, B, P, Q, tu, tpar, ab2, Wm, loss);
    } else {
        hipLaunchKernelGGL(vbpr_project_kernel<NT>, dim3(S, (B + 31) / 32), dim3(64), 0, stream, st, ti, tj, B, ppart);
        hipLaunchKernelGGL(vbpr_reduce_kernel, dim3(B), dim3(1024), 0, stream, ppart, S, B, kh, P, Q);
    }
    if (!(sparse && tpar)) {
        if (NH == 1) hipLaunchKernelGGL((vbpr_occur_kernel<1, TEAM>), rgrid, rblock, 0, stream, st, rec, occ2, occt, hdr4, B, Q, ab2, P, Wm, loss);
        else hipLaunchKernelGGL((vbpr_occur_kernel<2, TEAM>), rgrid, rblock, 0, stream, st, rec, occ2, occt, hdr4, B, Q, ab2, P, Wm, loss);
    }
    hipLaunchKernelGGL(vbpr_pair_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, ab2, B, kh, s_buf, t_buf, Wm, loss);
    switch (NE) {
        case 1: hipLaunchKernelGGL((vbpr_rows_kernel<1, TEAM>), rgrid, rblock, 0, stream, st, rec, occ2, occt, hdr4, s_buf, t_buf, P, Wm, Aw, ab); break;
        case 2: hipLaunchKernelGGL((vbpr_rows_kernel<2, TEAM>), rgrid, rblock, 0, stream, st, rec, occ2, occt, hdr4, s_buf, t_buf, P, Wm, Aw, ab); break;
        case 3: hipLaunchKernelGGL((vbpr_rows_kernel<3, TEAM>), rgrid, rblock, 0, stream, st, rec, occ2, occt, hdr4, s_buf, t_buf, P, Wm, Aw, ab); break;
        default: hipLaunchKernelGGL((vbpr_rows_kernel<4, TEAM>), rgrid, rblock, 0, stream, st, rec, occ2, occt, hdr4, s_buf, t_buf, P, Wm, Aw, ab); break;
    }
    if (sparse) {
        if (NH == 1) hipLaunchKernelGGL(vbpr_sdense_kernel<1>, dim3((st.d + 3) / 4), dim3(256), 0, stream, st, Aw, ab, loss);
        else hipLaunchKernelGGL(vbpr_sdense_kernel<2>, dim3((st.d + 3) / 4), dim3(256), 0, stream, st, Aw, ab, loss);
        return (int)hipGetLastError();
    }
    const size_t lds = (size_t)(4 * 64 * (NT * 32 + 1) + 4 * 2 * 64) * sizeof(float);
    auto dense = vbpr_dense_kernel<NT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(dense, dim3((st.d + 63) / 64), dim3(256), lds, stream, st, ti, tj, B, s_buf, Wm, loss);
    return (int)hipGetLastError();
}

template <int NT>
static int launch_vbpr(const tkr_vbpr_state& st, const int32_t* ti, const int32_t* tj, const int32_t* rec,
                       const int32_t* occ, const int32_t* hdr, const int32_t* occt, int B, float* ws, float* loss,
                       hipStream_t stream, const int32_t* tu, const int32_t* tpar) {
    switch (tkr_plan_team(B)) {                    // csrc/plan_parts.h team_for
        case 4: return launch_vbpr_t<NT, 4>(st, ti, tj, rec, occ, hdr, occt, B, ws, loss, stream, tu, tpar);
        case 8: return launch_vbpr_t<NT, 8>(st, ti, tj, rec, occ, hdr, occt, B, ws, loss, stream, tu, tpar);
        default: return launch_vbpr_t<NT, 16>(st, ti, tj, rec, occ, hdr, occt, B, ws, loss, stream, tu, tpar);
    }
}

}  // namespace tkr

extern "C" int tkr_plan_max_blocks(int32_t batch_size);

// the step's own scratch ...
extern "C" __attribute__((visibility("hidden"))) int64_t tkr_vbpr_workspace_core_floats(int32_t batch_size, int32_t kh, int32_t d) {
    const int64_t S = tkr::vbpr_slices(d);
    const int64_t slots = (int64_t)tkr_plan_max_blocks(batch_size) * tkr_plan_team(batch_size);     // sparse view: A, a per item task
    return S * batch_size * (kh + 1) + 2ll * batch_size + 2ll * batch_size * kh + slots * (kh + 1) + 5ll * batch_size;
}
// ... and behind it the loss words of up to 512 batches of a call: 64 spread slots per batch for this file's kernels (tkr_common.h
// loss_add_spread), one word per triplet (twice) and per column block for the column-plan step (csrc/vbpr_cols.hip: no atomics)
extern "C" int64_t tkr_vbpr_workspace_floats(int32_t batch_size, int32_t kh, int32_t d) {
    const int64_t spread = (int64_t)tkr::kLossSlots * tkr::kLossSlotStride, per_task = 2ll * batch_size + d;
    return tkr_vbpr_workspace_core_floats(batch_size, kh, d) + 512ll * (spread > per_task ? spread : per_task) + 64;      // (+ the column-plan step's pair counter)
}

namespace tkr {
// loss_out[b] += the 64 slots of batch b
__global__ __launch_bounds__(64) void loss_slots_kernel(const float* __restrict__ slots, float* __restrict__ loss_out) {
    const int b = blockIdx.x;
    const float v = wave_sum(slots[(size_t)b * kLossSlots * kLossSlotStride + threadIdx.x * kLossSlotStride]);
    if (threadIdx.x == 0) loss_out[b] += v;
}
// a call's loss bookkeeping: slots zeroed in front of its batches, added up behind them; batch b's kernels get slots(b) as their loss pointer
struct LossSlots {
    float* base;
    float* out;
    int n;
    hipStream_t s;
    int begin(float* ws_end_of_core, float* loss_out, int n_batches, hipStream_t stream) {
        base = loss_out ? ws_end_of_core : nullptr; out = loss_out; n = n_batches; s = stream;
        if (!base || n <= 0) return TKR_OK;
        if (n > 512) return TKR_EUNSUPPORTED;
        return hipMemsetAsync(base, 0, (size_t)n * kLossSlots * kLossSlotStride * sizeof(float), s) == hipSuccess ? TKR_OK : TKR_EINVAL;
    }
    float* of(int b) const { return base ? base + (size_t)b * kLossSlots * kLossSlotStride : nullptr; }
    int end() const {
        if (!base || n <= 0) return TKR_OK;
        hipLaunchKernelGGL(loss_slots_kernel, dim3(n), dim3(64), 0, s, base, out);
        return (int)hipGetLastError();
    }
};
}  // namespace tkr

extern "C" int tkr_vbpr_run(const tkr_vbpr_state* st, const int32_t* tri_i, const int32_t* tri_j, const int32_t* rec,
                            const int32_t* occ, const int32_t* hdr, const int32_t* occt, const int32_t* tri_u,
                            const int32_t* tpar, int32_t batch_size, int32_t n_batches, float* workspace, float* loss_out,
                            void* stream) {
    if (!st || !st->U || !st->msU || !st->I || !st->msI || !st->irb || !st->msirb || !st->cem || !st->mscem ||
        !st->icb || !st->msicb || !st->feat)
        return TKR_EINVAL;
    if (st->n_users <= 0 || st->n_items <= 0 || st->kh <= 0 || st->d <= 0) return TKR_EINVAL;
    if (!tri_i || !tri_j || !rec || !occ || !hdr || !occt || !workspace || batch_size <= 0 || n_batches < 0) return TKR_EINVAL;
    if (st->kh > 128 || batch_size > 65536) return TKR_EUNSUPPORTED;     // the launch records carry a triplet's index in 16 bits
    if (st->f_ptr && (!st->f_col || !st->f_val || !st->c_ptr || !st->c_item || !st->c_val || !st->item_tag)) return TKR_EINVAL;
    const size_t stride_r = (size_t)tkr_plan_max_blocks(batch_size) * tkr_plan_team(batch_size) * 16;
    const size_t stride_o = (size_t)3 * batch_size;
    const int NT = (st->kh + 31) / 32;
    tkr::LossSlots ls;
    TKR_CHECK_RC(ls.begin(workspace + tkr_vbpr_workspace_core_floats(batch_size, st->kh, st->d), loss_out, n_batches, (hipStream_t)stream));
    for (int b = 0; b < n_batches; ++b) {
        const int32_t* ti = tri_i + (size_t)b * batch_size;
        const int32_t* tj = tri_j + (size_t)b * batch_size;
        const int32_t* r = rec + b * stride_r;
        const int32_t* o = occ + b * stride_o * 2;
        const int32_t* h = hdr + (size_t)b * 4;
        const int32_t* ot = occt + b * stride_o;
        const int32_t* tu = (tri_u && tpar) ? tri_u + (size_t)b * batch_size : nullptr;
        const int32_t* tp = (tri_u && tpar) ? tpar + (size_t)b * batch_size : nullptr;
        float* l = ls.of(b);
        int rc;
        switch (NT) {
            case 1: rc = tkr::launch_vbpr<1>(*st, ti, tj, r, o, h, ot, batch_size, workspace, l, (hipStream_t)stream, tu, tp); break;
            case 2: rc = tkr::launch_vbpr<2>(*st, ti, tj, r, o, h, ot, batch_size, workspace, l, (hipStream_t)stream, tu, tp); break;
            case 3: rc = tkr::launch_vbpr<3>(*st, ti, tj, r, o, h, ot, batch_size, workspace, l, (hipStream_t)stream, tu, tp); break;
            default: rc = tkr::launch_vbpr<4>(*st, ti, tj, r, o, h, ot, batch_size, workspace, l, (hipStream_t)stream, tu, tp); break;
        }
        if (rc != 0) return rc;
    }
    return ls.end();
}
