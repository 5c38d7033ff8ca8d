// Row tasks of the VBPR step (sparse RMSProp on [ure|uce], ire, irb) and what they share with the other kernels of
// csrc/vbpr_step.hip and csrc/vbpr_cols.hip: the launch records of K1 and the scratch layout of the sparse view.
#pragma once
#include "tkr_common.h"
#include "../../include/tkr.h"

namespace tkr {

constexpr int kIdMaskV = 0x3fffffff;

// Scratch of the sparse view inside st.item_tag (n_items x 8 bytes, zeroed once by the caller):
//   slots [n_items] int32 | member 0 [n_items rounded to 4] bytes | member 1 [...] bytes | batch counter [1] uint32
// Membership of an item in the current batch is one BYTE written by the wave that owns the item's task (a plain store:
// global atomicOr on a shared bitmap word cost 25 us per batch) in a 10 KB map that the CUs' L1 holds, instead of 10^6
// random 8-byte reads of a tag table per column walk; the slot is only fetched for the ~5 % of entries that hit.  The
// two maps alternate by batch: S1 advances the counter and clears the map of the batch before; V2 marks items and
// stores slots; S3 reads.  All in stream order, no host state.
struct SparseScratch {
    int32_t* slots;
    unsigned char* member0;     // map of parity p: member0 + p * 4 * map_words (no pointer array: a dynamically indexed
    uint32_t* counter;          // local array would live in scratch memory)
    int map_words;
    __host__ __device__ unsigned char* member(uint32_t parity) const { return member0 + (size_t)(parity & 1u) * 4 * map_words; }
};
__host__ __device__ inline SparseScratch sparse_scratch(const tkr_vbpr_state& st) {
    SparseScratch x;
    x.map_words = (st.n_items + 3) / 4;
    x.slots = reinterpret_cast<int32_t*>(st.item_tag);
    x.member0 = reinterpret_cast<unsigned char*>(x.slots + st.n_items);
    x.counter = reinterpret_cast<uint32_t*>(x.member0 + 8 * x.map_words);
    return x;
}


// record access shared by V1b and V2 (64-byte wave records of K1, see oracle/plan_np.py)
struct WaveRec {
    int rowk, par, team, n_occ, first;
    int oa[4], ob[4], ot[4];
};

__device__ __forceinline__ WaveRec read_rec(const int32_t* __restrict__ rec_all, int team, int blk, int wave, int lane) {
    const int word = (lane < 16) ? rec_all[((size_t)blk * team + wave) * 16 + lane] : 0;
    WaveRec r;
    r.rowk = bcast_i(word, 0);
    const int meta = bcast_i(word, 1);
    r.par = meta & 1;
    r.team = (meta >> 8) & 0xff;
    r.n_occ = bcast_i(word, 2);
    r.first = bcast_i(word, 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) { r.oa[q] = bcast_i(word, 4 + 2 * q); r.ob[q] = bcast_i(word, 5 + 2 * q); }
    const int t01 = bcast_i(word, 13), t23 = bcast_i(word, 14);
    r.ot[0] = t01 & 0xffff; r.ot[1] = (t01 >> 16) & 0xffff; r.ot[2] = t23 & 0xffff; r.ot[3] = (t23 >> 16) & 0xffff;
    return r;
}

// occurrences [done, done+4) of a wave: inline for done == 0, else fetched from occ/occt
__device__ __forceinline__ void next_occ(const WaveRec& r, int done, int n, int lane, const int2* __restrict__ occ,
                                         const int32_t* __restrict__ occt, int (&oa)[4], int (&ob)[4], int (&ot)[4]) {
    if (done == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { oa[q] = r.oa[q]; ob[q] = r.ob[q]; ot[q] = r.ot[q]; }
    } else {
        int2 o = make_int2(0, 0);
        int t = 0;
        if (lane < n) { o = occ[r.first + (done + lane) * r.team]; t = occt[r.first + (done + lane) * r.team]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { oa[q] = bcast_i(o.x, q); ob[q] = bcast_i(o.y, q); ot[q] = bcast_i(t, q); }
    }
}

// V2: sparse RMSProp on the touched [ure|uce] rows (users) and ire rows + irb (items).  NE = ceil(2kh/64)
// where a task finds the pair sums S_t, T_t of up to N triplets at once (entries q >= n are not looked at): two plain arrays
// written by an earlier launch
struct PairSumArrays {
    const float* __restrict__ s;
    const float* __restrict__ t;
    template <int N>
    __device__ __forceinline__ void get(const int (&tri)[N], int n, float (&S)[N], float (&T)[N]) const {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const int x = tri[q < n ? q : 0];
            S[q] = s[x];
            T[q] = t[x];
        }
    }
};

// ... or two arrays that the FIRST blocks of this very launch are writing (csrc/vbpr_cols.hip, the pair blocks of vbpr_update_kernel).
// A task waits where it first needs a sum until `done` says every pair block has stored its sums (the pair blocks have the lowest
// block ids: they are resident before any block that waits).  Who polls the word matters: every wave of a ~1,500-block grid polling
// it through the L2 (6,000 pollers of one channel) cost the update launch 6 us; here ONE wave per block polls (with a nap) and tells
// the block's other waves through an LDS word (`flag`: set up by the block before anybody waits; null: this wave polls for itself --
// the row blocks, where a wave without a task never comes by).  The sums are then ordinary cached loads: this CU's L1 was dropped at
// the start of the launch and nothing of the launch reads the arrays before its wait; `past_l1`: the arrays share a cache line with
// their neighbours (batch size not a multiple of 32), read them past the L1.  A wait that never ends is a broken launch: trap.
struct PairSumFresh {
    const float* s;
    const float* t;
    const uint32_t* done;
    uint32_t target;
    volatile uint32_t* flag;
    bool past_l1;
    mutable bool ready = false;
    __device__ __forceinline__ void wait() const {
        if (ready) return;
        uint32_t spins = 0;
        if (flag == nullptr || (threadIdx.x >> 6) == 0) {
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 22)) __builtin_trap();
            }
            if (flag != nullptr && (threadIdx.x & 63) == 0) *(__attribute__((address_space(3))) volatile uint32_t*)(flag) = 1u;
        } else {
            while (*(const __attribute__((address_space(3))) volatile uint32_t*)(flag) == 0u) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 24)) __builtin_trap();
            }
        }
        asm volatile("" ::: "memory");
        ready = true;
    }
    template <int N>
    __device__ __forceinline__ void get(const int (&tri)[N], int n, float (&S)[N], float (&T)[N]) const {
        wait();
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const int x = tri[q < n ? q : 0];
            if (past_l1) {
                S[q] = __hip_atomic_load(s + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                T[q] = __hip_atomic_load(t + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                S[q] = s[x];
                T[q] = t[x];
            }
        }
    }
};
template <class PS> __device__ __forceinline__ void pair_sums_wait(const PS&) {}
__device__ __forceinline__ void pair_sums_wait(const PairSumFresh& ps) { ps.wait(); }

// ... or nowhere: the TEAM lanes that ask (a wave for a row task, the lanes of one column group) work S_t and T_t out themselves from
// e^alpha, e^beta of the batch ([B] each, 2 KB, cache-hot): 2B / TEAM reciprocals per lane and entry.  What this buys is the launch
// of the pair-sum kernel between the projection and the update and the boundary in front of it (4.5 + 1.7 of the 25 us of a
// 256-batch): csrc/vbpr_cols.hip runs the column-plan step in TWO launches with it.  Fixed order per asker.
template <int TEAM>
struct PairSumInline {
    static constexpr int NB = 256 / TEAM;       // a lane's share of the batch (B <= 256): b = lane-in-team + TEAM * r
    const float* __restrict__ ea;
    const float* __restrict__ eb;
    float la[NB], lb[NB];                       // e^alpha_b, e^beta_b of the lane's share, loaded ONCE (load(): one round trip; a loop
                                                // that fetched them per entry was 16 dependent cache trips per entry: 47 us per batch)
    __device__ __forceinline__ void load(const float* ea_, const float* eb_, int B) {
        ea = ea_; eb = eb_;
        const int l = threadIdx.x & (TEAM - 1);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int b = l + TEAM * r;
            la[r] = b < B ? ea_[b] : INFINITY;  // (past the batch: 1 / (1 + x * inf) = 0)
            lb[r] = b < B ? eb_[b] : INFINITY;
        }
    }
    template <int N>
    __device__ __forceinline__ void get(const int (&tri)[N], int n, float (&S)[N], float (&T)[N]) const {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const int x = tri[q < n ? q : 0];
            const float ea_t = ea[x], eb_t = eb[x];
            float s = 0.f, t = 0.f;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                s += pair_sigmoid(ea_t, lb[r]);
                t += pair_sigmoid(la[r], eb_t);
            }
            if constexpr (TEAM == 64) {
                s = wave_sum(s);
                t = wave_sum(t);
            } else {
#pragma unroll
                for (int o = TEAM / 2; o > 0; o >>= 1) {
                    s += __shfl_xor(s, o, 64);
                    t += __shfl_xor(t, o, 64);
                }
            }
            S[q] = s;
            T[q] = t;
        }
    }
};

template <int NE, int kVTeam, typename PairSums>
__device__ __forceinline__ void vbpr_rows_body(
    const tkr_vbpr_state& st, const int32_t* __restrict__ rec_all, const int2* __restrict__ occ,
    const int32_t* __restrict__ occt, const int4* __restrict__ hdr, const PairSums& ps, const float* __restrict__ P, const float* __restrict__ Wm,
    float* __restrict__ Aw /*[slots][kh] or null*/, float* __restrict__ ab /*[slots]*/,
    float (*red)[NE * TKR_WAVE + 1], float (*red2)[NE * TKR_WAVE + 1], int first_blk, int blk_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int4 h4 = *hdr;
    const int n_blocks = __builtin_amdgcn_readfirstlane(h4.x);
    const int nlb = __builtin_amdgcn_readfirstlane(h4.y);
    const int kh = st.kh, k2 = 2 * kh;
    const size_t ustride = (size_t)st.n_users * k2, istride = (size_t)st.n_items * kh;
    const bool l2 = st.mode == 0;
    for (int it = first_blk; it < n_blocks; it += blk_stride) {
        const int blk = n_blocks - 1 - it;           // backwards: the heavy teams at the end of the record list start first (csrc/bpr_step.hip)
        const bool heavy = blk >= nlb;
        const WaveRec r = read_rec(rec_all, kVTeam, blk, wave, lane);
        if (r.rowk == -1) continue;
        const bool is_item = r.rowk < 0;
        const int row = r.rowk & 0x7fffffff, par = r.par;
        const int width = is_item ? kh : k2;
        const float* src = is_item ? st.I + par * istride + (size_t)row * kh : st.U + par * ustride + (size_t)row * k2;
        float own[NE], g[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + e * 64;
            own[e] = c < width ? src[c] : 0.f;
            g[e] = 0.f;
        }
        const float br = is_item ? st.irb[(size_t)par * st.n_items + row] : 0.f;
        float gb = 0.f;
        float aw[NE], asum = 0.f;                       // sparse view: per-item sums for the column walk of S3
#pragma unroll
        for (int e = 0; e < NE; ++e) aw[e] = 0.f;
        const bool want_a = is_item && Aw != nullptr;
        for (int done = 0; done < r.n_occ; done += 4) {
            const int n = min(4, r.n_occ - done);
            int oa[4], ob[4], ot[4];
            next_occ(r, done, n, lane, occ, occt, oa, ob, ot);
            float sa4[4], sg4[4];
            ps.get(ot, n, sa4, sg4);                             // S_t, T_t of the four occurrences, one round trip
            for (int q = 0; q < n; ++q) {
                const float sa = sa4[q], sg = sg4[q];            // rows under beta: scaled by T_t
                if (is_item) {
                    const int u = oa[q] & kIdMaskV, pu = (oa[q] >> 30) & 1;
                    const bool role_j = ob[q] < 0;
                    const float sgn_s = role_j ? sg : -sg;
                    // (the bias sits under alpha: scaled by S_t)
                    const float sgn_a = role_j ? sa : -sa;
                    const float lam = role_j ? st.lj : st.li;
                    const float* ur = st.U + pu * ustride + (size_t)u * k2;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const int c = lane + e * 64;
                        if (c < kh) g[e] += sgn_s * ur[c] + lam * (l2 ? own[e] : sgn(own[e]));
                    }
                    gb += sgn_a + st.lb * (l2 ? br : sgn(br));
                    if (want_a) {
                        const float* wt = Wm + (size_t)ot[q] * kh;
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            const int c = lane + e * 64;
                            if (c < kh) aw[e] += role_j ? -wt[c] : wt[c];
                        }
                        asum += sgn_a;
                    }
                } else {
                    const int i = oa[q] & kIdMaskV, pi = (oa[q] >> 30) & 1;
                    const int j = ob[q] & kIdMaskV, pj = (ob[q] >> 30) & 1;
                    const float* ri = st.I + pi * istride + (size_t)i * kh;
                    const float* rj = st.I + pj * istride + (size_t)j * kh;
                    const float* pt = P + (size_t)ot[q] * kh;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const int c = lane + e * 64;
                        if (c < k2) {
                            const float partner = c < kh ? (ri[c] - rj[c]) : pt[c - kh];
                            g[e] += -sg * partner + st.lu * (l2 ? own[e] : sgn(own[e]));
                        }
                    }
                }
            }
        }
        if (heavy) {
#pragma unroll
            for (int e = 0; e < NE; ++e) { red[wave][lane + e * 64] = g[e]; red2[wave][lane + e * 64] = aw[e]; }
            if (lane == 0) { red[wave][NE * 64] = gb; red2[wave][NE * 64] = asum; }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    float a = 0.f, a2 = 0.f;
                    for (int w = 0; w < kVTeam; ++w) { a += red[w][lane + e * 64]; a2 += red2[w][lane + e * 64]; }
                    g[e] = a;
                    aw[e] = a2;
                }
                float a = 0.f, a2 = 0.f;
                for (int w = 0; w < kVTeam; ++w) { a += red[w][NE * 64]; a2 += red2[w][NE * 64]; }
                gb = a;
                asum = a2;
            }
            __syncthreads();
            if (wave != 0) continue;
        }
        if (want_a) {                                   // slot = this wave's record index; the item joins the batch's bitmap
            const int slot = blk * kVTeam + wave;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int c = lane + e * 64;
                if (c < kh) Aw[(size_t)slot * kh + c] = aw[e];
            }
            if (lane == 0) {
                ab[slot] = asum;
                const SparseScratch x = sparse_scratch(st);
                x.slots[row] = slot;
                x.member(*x.counter)[row] = 1;
            }
        }
        const float* msrc = is_item ? st.msI + par * istride + (size_t)row * kh : st.msU + par * ustride + (size_t)row * k2;
        float* po = is_item ? st.I + (par ^ 1) * istride + (size_t)row * kh : st.U + (par ^ 1) * ustride + (size_t)row * k2;
        float* mo = is_item ? st.msI + (par ^ 1) * istride + (size_t)row * kh : st.msU + (par ^ 1) * ustride + (size_t)row * k2;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + e * 64;
            if (c < width) {
                const float m2 = st.rho * msrc[c] + (1.f - st.rho) * g[e] * g[e];
                mo[c] = m2;
                po[c] = own[e] - st.lr * g[e] / sqrtf(m2 + st.eps);
            }
        }
        if (is_item && lane == 0) {
            const float m2 = st.rho * st.msirb[(size_t)par * st.n_items + row] + (1.f - st.rho) * gb * gb;
            st.msirb[(size_t)(par ^ 1) * st.n_items + row] = m2;
            st.irb[(size_t)(par ^ 1) * st.n_items + row] = br - st.lr * gb / sqrtf(m2 + st.eps);
        }
    }
}

inline int vbpr_grid(int B, int team) {
    const int lpb = team;                                // oracle/plan_np.py light_per_block
    int grid = (3 * B + lpb - 1) / lpb + 16;
    return grid > 2048 ? 2048 : grid;
}

}  // namespace tkr
