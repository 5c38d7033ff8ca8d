// The (u, i, j) draw of K1, shared by the per-batch planner (sampler.hip) and the grid-wide one (planner_big.hip).
// Replaces BPR._uniform_user_sampling (single/bpr.py:155-165); stream definition in oracle/plan_np.py.
#pragma once
#include "tkr_common.h"

namespace tkr {

constexpr int kMaxRounds = 64;   // oracle/plan_np.py MAX_ROUNDS

// is `item` among cols_sorted[lo, hi) (ascending)?  A 4-ary search: three pivots per step, loaded together -- the draw of a triplet is
// a chain of dependent loads (user -> row bounds -> positive; then this search for every candidate negative), and a binary search is
// log2(degree) trips where this is log4 (a degree of 36: 3 trips instead of 6; the draw is 4.9 us of the 17 us phase A of a batch).
// Wider is NOT faster (round 5, measured inside the planner prologue of csrc/bpr_own.hip: sixteen segments per step + a window of 32
// at the end = 2 trips and 47 loads per candidate took 9.0 us where this takes 5.6): a load whose 64 lanes hit 64 different lines
// occupies the CU's L1 for ~64 cycles whether or not anybody waits for it, so the cost is trips x ~0.5 us + loads x ~0.14 us.
__device__ __forceinline__ bool is_member(const int32_t* __restrict__ cols_sorted, int lo, int hi, int item) {
    int a = lo, b = hi;                              // if present, item sits in [a, b)
    while (b - a > 3) {
        const int q = (b - a) >> 2;
        const int m1 = a + q, m2 = a + 2 * q, m3 = a + 3 * q;
        const int v1 = cols_sorted[m1], v2 = cols_sorted[m2], v3 = cols_sorted[m3];
        if (item == v1 || item == v2 || item == v3) return true;
        if (item < v1) b = m1;
        else if (item < v2) { a = m1 + 1; b = m2; }
        else if (item < v3) { a = m2 + 1; b = m3; }
        else a = m3 + 1;
    }
    const int last = hi > lo ? hi - 1 : lo;
    const int c0 = cols_sorted[min(a, last)], c1 = cols_sorted[min(a + 1, last)], c2 = cols_sorted[min(a + 2, last)];
    return (a < b && c0 == item) || (a + 1 < b && c1 == item) || (a + 2 < b && c2 == item);
}

__device__ __forceinline__ void draw_triplet(const int32_t* __restrict__ tr_users, uint32_t n_tr,
                                             const int32_t* __restrict__ row_ptr,
                                             const int32_t* __restrict__ pos_cols,
                                             const int32_t* __restrict__ cols_sorted, uint32_t n_items,
                                             uint32_t k0, uint32_t k1, uint64_t g, int& u, int& i, int& j) {
    const uint32_t c0 = (uint32_t)g, c1 = (uint32_t)(g >> 32);
    u32x4 w = philox4x32_10(c0, c1, 0u, 0u, k0, k1);
    u = tr_users[mulhi64(w.x, w.y, n_tr)];
    const int lo = row_ptr[u], hi = row_ptr[u + 1];
    i = pos_cols[lo + (int)mulhi64(w.z, w.w, (uint32_t)(hi - lo))];
    int cand = 0;
    bool found = false;
    for (uint32_t r = 1; r <= (uint32_t)kMaxRounds && !found; ++r) {
        w = philox4x32_10(c0, c1, r, 0u, k0, k1);
        cand = (int)mulhi64(w.x, w.y, n_items);
        if (!is_member(cols_sorted, lo, hi, cand)) { found = true; break; }
        cand = (int)mulhi64(w.z, w.w, n_items);
        if (!is_member(cols_sorted, lo, hi, cand)) { found = true; break; }
    }
    if (!found) {   // cyclic scan fallback (user rated almost everything)
        for (uint32_t s = 0; s < n_items && is_member(cols_sorted, lo, hi, cand); ++s)
            cand = (cand + 1 == (int)n_items) ? 0 : cand + 1;
    }
    j = cand;
}

}  // namespace tkr
