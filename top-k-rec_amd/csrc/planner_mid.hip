// K1 for MID-SIZE batches (1024 < batch_size <= 65,536): the plan of csrc/sampler.hip / oracle/plan_np.py, word for word, in SIX
// launches and without a library sort (to 16,384 with the ranges' run lists as 16-bit entries in LDS; above: the wide form of build).
//
// single/bpr.py:103-113 takes any batch_size.  One workgroup per batch (sampler.hip) leaves most of the chip idle from a few
// thousand triplets on and takes a millisecond per batch; the grid-wide planner (planner_big.hip) sorts (batch | row | occurrence)
// keys of the whole call with hipcub::DeviceRadixSort and ranks with DeviceScan: ~45 launches per call, 7 us of device time per batch
// of 8192 in front of 17 us of step (round 6 trace: rocprim 190 us, parity + record 120 us, emit / count / fill / flag 70 us per 61
// batches).  Here the sort is a COUNTING sort over row ranges:
//
//   draw      one thread per triplet (the Philox stream of sampler_draw.h), (u, i, j) out
//   count     one workgroup per (batch, row range): the rows of its range that the batch draws, counted in LDS ->
//             (tasks, occurrences, light tasks, heavy tasks) of the range
//   build     the same workgroups, with the sums over the ranges in front of theirs: counts again, exclusive scan -> every row's
//             run in the range's occurrence list, filled through an LDS word per row {run start, cursor}; a row's run is then put in
//             ascending occurrence order (the order of the oracle's stable sort: up to 4 entries a comparator network by the row's
//             thread, up to 64 one wave ranking by counting through readlane, up to 512 the same over LDS, more through a bitmap over
//             the batch's occurrence indices); tasks, touch bits, occ, occt out
//   prefix    the touch maps are BATCH-MAJOR inside this call (touch[b][row / 32]; the other planners use the same bytes row-major and
//             every planner leaves them zero): a wave per word turns them into the rows' parities at the start of every batch and
//             advances the rows' counters (K1c)
//   resolve   the workgroups of count / build again: parities (one word of a compact per-batch table per lookup), light / heavy ranks
//             from the range sums, the 64-byte wave records, the header
//   zero      the touch maps back to zero
// (When the call's batches do not fit the maps batch-major -- the last few of 512 batches when n_rows is not a multiple of 32 -- the maps
// stay row-major, resolve looks parities up as the other planners do and K1c is csrc/sampler.hip's commit_kernel.)
//
// A range is at most 8192 rows (one LDS word each) and is sized for ~2048 user / ~4096 item occurrences; every workgroup walks the
// batch's draws (32-64 KB from L2 at batch 8192) and keeps what falls into its range -- worth it while a range keeps a fair share of
// them (tkr_plan_mid_ok).  Integer work; every output word is defined by oracle/plan_np.py and must match it bit for bit
// (tests/test_gpu_bpr.py test_sample_plan_bit_exact).
#include <stdlib.h>
#include <utility>

#include "tkr_common.h"
#include "sampler_draw.h"
#include "plan_parts.h"

namespace tkr {

#ifdef TKR_MID_PROF          // profiling build: cycles per phase of the build kernel, summed over its item / user workgroups
__device__ unsigned long long mid_prof[32];
#define MID_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&mid_prof[(r.item ? 16 : 0) + (i)], now_ - t_last_); t_last_ = now_; } } while (0)
#define MID_STAMP_INIT() unsigned long long t_last_ = clock64()
#else
#define MID_STAMP(i) do { } while (0)
#define MID_STAMP_INIT() do { } while (0)
#endif

constexpr int kMidThreads = 1024;
constexpr int kMidWaves = kMidThreads / TKR_WAVE;
constexpr int kMidRows = 8192;            // rows per range at most (kMidThreads * 8)
constexpr int kMidMaxRanges = 128;        // per batch; more (tens of millions of users): the grid-wide planner
constexpr int kMidMaxB = 16384;           // run starts and cursors are 16-bit halves of one LDS word, occurrence indices 16-bit
constexpr int kMidThreadSort = 16;        // a run up to this long is sorted by the row's thread: a compare, a five-comparator network, a bitonic network of
                                          // 16 on registers (an insertion sort of 5 .. 16 entries ON LDS made its thread the one the workgroup waited for)
constexpr int kMidWaveSort = 512;         // ... up to this long: one wave ranks it by counting (up to 64: through readlane; else 8 entries per lane)
__host__ __device__ inline int mid_queue_cap(int B) { return 2 * B / (kMidThreadSort + 1) + 1; }        // longer runs of one range: at most
constexpr int kMidHuge = 2 * kMidMaxB / (kMidWaveSort + 1) + 1;

struct MidGeom { int gu, gi, ru, ri; };   // user / item ranges per batch, rows per range

// batches above kMidMaxB (to kMidWideMaxB): the WIDE form of the build step -- the ranges' occurrence lists live in the workspace (32-bit
// entries at exact offsets: the count step's sums), a row has two LDS words {run start, cursor}, ranges are at most kMidWideRows rows
constexpr int kMidWideMaxB = 65536;
constexpr int kMidWideRows = 4096;
constexpr int kMidWideThreadSort = 32;    // wide form: a run up to this long is sorted in the registers of the row's thread (a bitonic network of 16 or
                                          // 32; a wave per run of 17 .. 64 entries -- one in eight of an item range's runs at batch 65,536 -- was 25 of its 83 us)
__host__ __device__ inline bool mid_wide(int B) { return B > kMidMaxB; }

static MidGeom mid_geom(int n_users, int n_items, int B) {
    const int cap = mid_wide(B) ? kMidWideRows : kMidRows;
    auto ranges = [cap](int n, int by_keys) {
        int g = (n + cap - 1) / cap;
        if (by_keys > g) g = by_keys;
        if (g > n) g = n;
        const int r = (n + g - 1) / g;
        return std::pair<int, int>((n + r - 1) / r, r);
    };
    // (1024 ... 4096 occurrences per range measured alike at batch 8192: 20.7-21.3 us per batch on the line; the wide form walks 4x to
    // 8x the draws per workgroup and takes twice the occurrences per range)
    const int per_u = mid_wide(B) ? 4096 : 2048, per_i = mid_wide(B) ? 8192 : 4096;      // (wide form at 65,536: 2048/4096 ... 16,384/16,384 within 2 %)
    const auto u = ranges(n_users, (B + per_u - 1) / per_u), i = ranges(n_items, (2 * B + per_i - 1) / per_i);
    return MidGeom{u.first, i.first, u.second, i.second};
}

__global__ void mid_draw_kernel(const int32_t* __restrict__ tr_users, uint32_t n_tr, const int32_t* __restrict__ row_ptr,
                                const int32_t* __restrict__ pos_cols, const int32_t* __restrict__ cols_sorted, uint32_t n_items,
                                uint64_t seed, uint64_t first_triplet, const int64_t* __restrict__ ctl, int B, size_t total,
                                int32_t* __restrict__ out_u, int32_t* __restrict__ out_i, int32_t* __restrict__ out_j) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const uint64_t batch0 = ctl ? (uint64_t)ctl[0] : 0ull;
    int u, i, j;
    draw_triplet(tr_users, n_tr, row_ptr, pos_cols, cols_sorted, n_items, (uint32_t)seed, (uint32_t)(seed >> 32),
                 first_triplet + batch0 * (uint64_t)B + g, u, i, j);
    out_u[g] = u; out_i[g] = i; out_j[g] = j;
}

struct MidRange {                          // what a workgroup works on
    int b, c, item, lo, rows, n_occ;       // batch, range, kind, first row, rows in the range, occurrences of the batch of this kind
    const int32_t *a0, *a1;                // users: out_u; items: out_i, out_j (occurrence o >= B is j[o - B])
};

__device__ __forceinline__ MidRange mid_range(const MidGeom& g, int B, int n_users, int n_items, const int32_t* out_u, const int32_t* out_i,
                                              const int32_t* out_j) {
    MidRange r;
    r.b = blockIdx.y; r.c = blockIdx.x;
    r.item = r.c >= g.gu;
    const int ci = r.item ? r.c - g.gu : r.c, per = r.item ? g.ri : g.ru, n = r.item ? n_items : n_users;
    r.lo = ci * per;
    r.rows = min(per, n - r.lo);
    r.n_occ = r.item ? 2 * B : B;
    r.a0 = (r.item ? out_i : out_u) + (size_t)r.b * B;
    r.a1 = out_j + (size_t)r.b * B;
    return r;
}

__device__ __forceinline__ int mid_row_of(const MidRange& r, int B, int o) { return (r.item && o >= B) ? r.a1[o - B] : r.a0[o]; }

// Walk the batch's draws and call hit(occurrence, row - lo) for those of the range.  Sixteen draws per thread are asked for together, as
// four 16-byte loads (one at a time every L2 round trip stood alone: 9 us of an item range's 51 at batch 8192 went into reading 16
// values per thread; as 4-byte loads the walk of 131,072 draws took 45 us per pass at batch 65,536); batch sizes that are not a
// multiple of four take the plain walk.
template <class F>
__device__ __forceinline__ void mid_walk(const MidRange& r, int B, F&& hit) {
    if ((B & 3) == 0) {
        constexpr int UN = 4;
        for (int o0 = threadIdx.x * 4; o0 < r.n_occ; o0 += UN * 4 * kMidThreads) {
            int4 v[UN];
#pragma unroll
            for (int x = 0; x < UN; ++x) {
                const int o = o0 + x * 4 * kMidThreads;
                v[x] = make_int4(-1, -1, -1, -1);             // (-1 - lo is no row of any range)
                if (o < r.n_occ) v[x] = *reinterpret_cast<const int4*>((r.item && o >= B) ? r.a1 + (o - B) : r.a0 + o);
            }
#pragma unroll
            for (int x = 0; x < UN; ++x) {
                const int o = o0 + x * 4 * kMidThreads;
                const unsigned r0 = (unsigned)(v[x].x - r.lo), r1 = (unsigned)(v[x].y - r.lo), r2 = (unsigned)(v[x].z - r.lo), r3 = (unsigned)(v[x].w - r.lo);
                if (r0 < (unsigned)r.rows) hit(o, r0);
                if (r1 < (unsigned)r.rows) hit(o + 1, r1);
                if (r2 < (unsigned)r.rows) hit(o + 2, r2);
                if (r3 < (unsigned)r.rows) hit(o + 3, r3);
            }
        }
    } else {
        for (int o = threadIdx.x; o < r.n_occ; o += kMidThreads) {
            const unsigned rl = (unsigned)(mid_row_of(r, B, o) - r.lo);
            if (rl < (unsigned)r.rows) hit(o, rl);
        }
    }
}

// cnt[row - lo] = occurrences of the row in the batch
__device__ __forceinline__ void mid_count(const MidRange& r, int B, uint32_t* cnt) {
    for (int q = threadIdx.x; q < r.rows; q += kMidThreads) cnt[q] = 0;
    __syncthreads();
    mid_walk(r, B, [&](int, unsigned rl) { atomicAdd(&cnt[rl], 1u); });
    __syncthreads();
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < TKR_WAVE; d <<= 1) {
        const int up = __shfl_up(v, d);
        if (lane >= d) v += up;
    }
    return v;
}

// exclusive prefix of (a, b) over the workgroup's threads, totals out; scr: 2 * kMidWaves ints
__device__ __forceinline__ void mid_scan2(int a, int b, int* scr, int& ex_a, int& ex_b, int& tot_a, int& tot_b) {
    const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x >> 6;
    const int ia = wave_incl_scan(a, lane), ib = wave_incl_scan(b, lane);
    __syncthreads();                                       // scr may still be read from an earlier call
    if (lane == TKR_WAVE - 1) { scr[wave] = ia; scr[kMidWaves + wave] = ib; }
    __syncthreads();
    // every wave scans the kMidWaves wave totals in its lanes 0 .. kMidWaves - 1
    int wa = lane < kMidWaves ? scr[lane] : 0, wb = lane < kMidWaves ? scr[kMidWaves + lane] : 0;
#pragma unroll
    for (int d = 1; d < kMidWaves; d <<= 1) {
        const int ua = __shfl_up(wa, d), ub = __shfl_up(wb, d);
        if (lane >= d) { wa += ua; wb += ub; }
    }
    tot_a = __shfl(wa, kMidWaves - 1); tot_b = __shfl(wb, kMidWaves - 1);
    const int pa = __shfl(wa, max(wave - 1, 0)), pb = __shfl(wb, max(wave - 1, 0));
    ex_a = ia - a + (wave ? pa : 0); ex_b = ib - b + (wave ? pb : 0);
}

// sums of the range records: over the ranges in front of `c` (pre) and over all of the batch (all)
__device__ __forceinline__ void mid_sums(const int4* __restrict__ agg, int G, int c, int4* scr4, int4& pre, int4& all) {
    int4 mine = make_int4(0, 0, 0, 0), mine_pre = make_int4(0, 0, 0, 0);
    if ((int)threadIdx.x < G) {
        mine = agg[threadIdx.x];
        if ((int)threadIdx.x < c) mine_pre = mine;
    }
    const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x >> 6;
    if (wave < 2) {                                        // kMidMaxRanges = 128: two waves hold them all
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mine.x += __shfl_xor(mine.x, d); mine.y += __shfl_xor(mine.y, d); mine.z += __shfl_xor(mine.z, d); mine.w += __shfl_xor(mine.w, d);
            mine_pre.x += __shfl_xor(mine_pre.x, d); mine_pre.y += __shfl_xor(mine_pre.y, d);
            mine_pre.z += __shfl_xor(mine_pre.z, d); mine_pre.w += __shfl_xor(mine_pre.w, d);
        }
        if (lane == 0) { scr4[2 * wave] = mine; scr4[2 * wave + 1] = mine_pre; }
    }
    __syncthreads();
    const int4 a0 = scr4[0], p0 = scr4[1], a1 = scr4[2], p1 = scr4[3];
    all = make_int4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    pre = make_int4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
    __syncthreads();
}

__global__ __launch_bounds__(kMidThreads) void mid_count_kernel(MidGeom g, int B, int lmax, int n_users, int n_items,
                                                                 const int32_t* __restrict__ out_u, const int32_t* __restrict__ out_i,
                                                                 const int32_t* __restrict__ out_j, int4* __restrict__ agg) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mid_lds[];
    __shared__ int red[4 * kMidWaves];
    uint32_t* cnt = mid_lds;
    const MidRange r = mid_range(g, B, n_users, n_items, out_u, out_i, out_j);
    mid_count(r, B, cnt);
    int nt = 0, nk = 0, nl = 0, nh = 0;
    for (int q = threadIdx.x; q < r.rows; q += kMidThreads) {
        const int c = (int)cnt[q];
        nt += c > 0; nk += c; nl += (c > 0 && c <= lmax); nh += c > lmax;
    }
    const int lane = threadIdx.x & (TKR_WAVE - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { nt += __shfl_xor(nt, d); nk += __shfl_xor(nk, d); nl += __shfl_xor(nl, d); nh += __shfl_xor(nh, d); }
    if (lane == 0) { red[wave] = nt; red[kMidWaves + wave] = nk; red[2 * kMidWaves + wave] = nl; red[3 * kMidWaves + wave] = nh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int4 s = make_int4(0, 0, 0, 0);
        for (int w = 0; w < kMidWaves; ++w) { s.x += red[w]; s.y += red[kMidWaves + w]; s.z += red[2 * kMidWaves + w]; s.w += red[3 * kMidWaves + w]; }
        agg[(size_t)r.b * (g.gu + g.gi) + r.c] = s;
    }
}

template <int N>
__device__ __forceinline__ void reg_sort_asc(uint32_t (&v)[N]) {           // bitonic network on registers, N a power of two
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t lo = min(v[i], v[l]), hi = max(v[i], v[l]);
                    const bool asc = (i & k) == 0;
                    v[i] = asc ? lo : hi;
                    v[l] = asc ? hi : lo;
                }
            }
}

// dynamic LDS of the build kernel: cnt[kMidRows] | list[2B] (16-bit) | queue[mid_queue_cap(B)] (16-bit) | huge[kMidHuge] (16-bit) | bitmap[2B / 32]
__host__ __device__ inline size_t mid_build_lds(int B) {
    return (size_t)kMidRows * 4 + (size_t)2 * B * 2 + (size_t)(mid_queue_cap(B) + kMidHuge + 2) / 2 * 4 + (size_t)((2 * B + 31) / 32) * 4 + 16;
}

__global__ __launch_bounds__(kMidThreads, 8) void mid_build_kernel(MidGeom g, int B, int n_users, int n_items,
                                                                 const int32_t* __restrict__ out_u, const int32_t* __restrict__ out_i,
                                                                 const int32_t* __restrict__ out_j, const int4* __restrict__ agg,
                                                                 int4* __restrict__ task_all, int2* __restrict__ occ_all,
                                                                 int32_t* __restrict__ occt_all, uint32_t* __restrict__ touch_u,
                                                                 uint32_t* __restrict__ touch_i, int wu, int wi /*words per batch of the batch-major touch maps; 0: row-major*/) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mid_lds[];
    __shared__ int scr[2 * kMidWaves];
    __shared__ int4 scr4[4];
    __shared__ int qn, hn;
    uint32_t* cnt = mid_lds;
    uint16_t* list = reinterpret_cast<uint16_t*>(cnt + kMidRows);
    uint16_t* queue = list + 2 * B;
    uint16_t* huge = queue + mid_queue_cap(B);
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(mid_lds) + (kMidRows * 4 + 2 * B * 2 + (mid_queue_cap(B) + kMidHuge + 2) / 2 * 4) / 4;
    const int tid = threadIdx.x, lane = tid & (TKR_WAVE - 1), wave = tid >> 6;
    const int G = g.gu + g.gi;
    const MidRange r = mid_range(g, B, n_users, n_items, out_u, out_i, out_j);
    MID_STAMP_INIT();
    int4 pre, all;
    mid_sums(agg + (size_t)r.b * G, G, r.c, scr4, pre, all);
    MID_STAMP(0);
    if (tid == 0) { qn = 0; hn = 0; }
    mid_count(r, B, cnt);
    MID_STAMP(1);

    // every row's run: thread t owns the rows [t * per, t * per + per)
    const int per = (r.rows + kMidThreads - 1) / kMidThreads;
    const int r0 = min(tid * per, r.rows), r1 = min(r0 + per, r.rows);
    int my_k = 0, my_t = 0;
    for (int q = r0; q < r1; ++q) { const int c = (int)cnt[q]; my_k += c; my_t += c > 0; }
    int ex_k, ex_t, tot_k, tot_t;
    mid_scan2(my_k, my_t, scr, ex_k, ex_t, tot_k, tot_t);
    int4* task = task_all + (size_t)r.b * 3 * B;
    uint32_t* touch = r.item ? touch_i : touch_u;
    const int wt = r.item ? wi : wu;
    uint32_t tw = 0;                                       // batch-major touch map: the bits of this thread's rows, word by word
    int tw_at = -1;
    for (int q = r0; q < r1; ++q) {
        const int c = (int)cnt[q];
        if (c > 0) {
            const int row = r.lo + q;
            task[pre.x + ex_t] = make_int4((int)((uint32_t)row | ((uint32_t)r.item << 31)), pre.y + ex_k, c, 0);
            if (wt) {
                if ((row >> 5) != tw_at) {
                    if (tw) atomicOr(&touch[(size_t)r.b * wt + tw_at], tw);
                    tw = 0; tw_at = row >> 5;
                }
                tw |= 1u << (row & 31);
            } else {
                atomicOr(&touch[(size_t)row * kTouchWords + (r.b >> 5)], 1u << (r.b & 31));
            }
            cnt[q] = (uint32_t)ex_k << 16;                // run start | cursor
            ex_k += c;
            ++ex_t;
        }
    }
    if (tw) atomicOr(&touch[(size_t)r.b * wt + tw_at], tw);
    __syncthreads();
    MID_STAMP(2);
    mid_walk(r, B, [&](int o, unsigned rl) {
        const uint32_t old = atomicAdd(&cnt[rl], 1u);
        list[(old >> 16) + (old & 0xffffu)] = (uint16_t)o;
    });
    __syncthreads();
    MID_STAMP(3);
    // ascending occurrence order inside every run (the fill above is in arrival order)
    for (int q = r0; q < r1; ++q) {
        const uint32_t w = cnt[q];
        const int c = (int)(w & 0xffffu), s = (int)(w >> 16);
        if (c == 2) {                                      // (most runs that need anything: one compare)
            const uint16_t a = list[s], b2 = list[s + 1];
            if (a > b2) { list[s] = b2; list[s + 1] = a; }
        } else if (c == 3 || c == 4) {                     // a five-comparator network on registers, the missing fourth entry = +inf
            uint32_t a0 = list[s], a1 = list[s + 1], a2 = list[s + 2], a3 = c == 4 ? list[s + 3] : 0xffffffffu;
            auto cx = [](uint32_t& x, uint32_t& y) { const uint32_t lo = min(x, y), hi = max(x, y); x = lo; y = hi; };
            cx(a0, a1); cx(a2, a3); cx(a0, a2); cx(a1, a3); cx(a1, a2);
            list[s] = (uint16_t)a0; list[s + 1] = (uint16_t)a1; list[s + 2] = (uint16_t)a2;
            if (c == 4) list[s + 3] = (uint16_t)a3;
        } else if (c > 4 && c <= kMidThreadSort) {         // a bitonic network of 16 on registers (one run in forty of an item range at batch 8192;
            uint32_t v[16];                                // a wave each, they were most of the 28 % of the kernel in its wave tier)
#pragma unroll
            for (int x = 0; x < 16; ++x) v[x] = x < c ? (uint32_t)list[s + x] : 0xffffffffu;
            reg_sort_asc<16>(v);
#pragma unroll
            for (int x = 0; x < 16; ++x)
                if (x < c) list[s + x] = (uint16_t)v[x];
        } else if (c > kMidThreadSort) {
            queue[atomicAdd(&qn, 1)] = (uint16_t)q;
        }
    }
    __syncthreads();
    MID_STAMP(4);
    const int n_q = qn;
    for (int qi = wave; qi < n_q; qi += kMidWaves) {          // a wave per longer run: rank by counting
        const int q = queue[qi];
        const uint32_t w = cnt[q];
        const int c = (int)(w & 0xffffu), s = (int)(w >> 16);
        if (c > kMidWaveSort) {
            if (lane == 0) huge[atomicAdd(&hn, 1)] = (uint16_t)q;
            continue;
        }
        if (c <= TKR_WAVE) {                              // one entry per lane: the others come through readlane, not LDS
            const uint32_t e = lane < c ? list[s + lane] : 0xffffffffu;
            int rk = 0;
            for (int y = 0; y < c; ++y) rk += (uint32_t)__builtin_amdgcn_readlane((int)e, y) < e;
            if (lane < c) list[s + rk] = (uint16_t)e;     // (the run was read in full before)
            continue;
        }
        constexpr int E = kMidWaveSort / TKR_WAVE;
        uint32_t e[E];
        int rk[E];
#pragma unroll
        for (int x = 0; x < E; ++x) { const int idx = lane + x * TKR_WAVE; e[x] = idx < c ? list[s + idx] : 0xffffffffu; rk[x] = 0; }
        for (int y0 = 0; y0 < c; y0 += 4) {               // (four reads in flight; past the run: list[] is still inside the LDS block)
            uint32_t v[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) v[z] = y0 + z < c ? list[s + y0 + z] : 0xffffffffu;
#pragma unroll
            for (int z = 0; z < 4; ++z)
#pragma unroll
                for (int x = 0; x < E; ++x) rk[x] += v[z] < e[x];
        }
        __builtin_amdgcn_wave_barrier();                  // every read of the run before its first write
#pragma unroll
        for (int x = 0; x < E; ++x)
            if (lane + x * TKR_WAVE < c) list[s + rk[x]] = (uint16_t)e[x];
    }
    __syncthreads();
    MID_STAMP(5);
    const int n_h = hn, words = (2 * B + 31) / 32;       // <= kMidThreads words at kMidMaxB
    for (int hi = 0; hi < n_h; ++hi) {                    // (only heavily skewed data: one row with > 512 occurrences in a batch)
        const int q = huge[hi];
        const uint32_t w = cnt[q];
        const int c = (int)(w & 0xffffu), s = (int)(w >> 16);
        if (tid < words) bitmap[tid] = 0;
        __syncthreads();
        for (int x = tid; x < c; x += kMidThreads) { const uint32_t o = list[s + x]; atomicOr(&bitmap[o >> 5], 1u << (o & 31)); }
        __syncthreads();
        uint32_t bits = tid < words ? bitmap[tid] : 0;
        int ex, ex2, tot, tot2;
        mid_scan2(__popc(bits), 0, scr, ex, ex2, tot, tot2);
        int pos = s + ex;
        while (bits) {
            const int bit = __ffs(bits) - 1;
            list[pos++] = (uint16_t)(tid * 32 + bit);
            bits &= bits - 1;
        }
        __syncthreads();
    }
    MID_STAMP(6);
    // occurrences out: user occurrence t -> (i[t], j[t]); item occurrence o -> (u[t], the other item | role << 31)
    int2* occ = occ_all + (size_t)r.b * 3 * B + pre.y;
    int32_t* occt = occt_all + (size_t)r.b * 3 * B + pre.y;
    const int32_t* bu = out_u + (size_t)r.b * B;
    const int32_t* bi = out_i + (size_t)r.b * B;
    const int32_t* bj = out_j + (size_t)r.b * B;
    for (int p0 = tid; p0 < tot_k; p0 += 4 * kMidThreads) {      // (four at a time: the gathers of one entry waited alone)
        int o[4], a[4], c2[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) o[x] = p0 + x * kMidThreads < tot_k ? (int)list[p0 + x * kMidThreads] : 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (r.item) {
                const bool role = o[x] >= B;
                const int t = role ? o[x] - B : o[x];
                a[x] = bu[t];
                c2[x] = (int)((uint32_t)(role ? bi[t] : bj[t]) | ((uint32_t)role << 31));
                o[x] = t;
            } else {
                a[x] = bi[o[x]];
                c2[x] = bj[o[x]];
            }
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int p = p0 + x * kMidThreads;
            if (p < tot_k) { occ[p] = make_int2(a[x], c2[x]); occt[p] = o[x]; }
        }
    }
    MID_STAMP(7);
    if (r.c == G - 1)                                     // slots past the batch's last task
        for (int q = all.x + tid; q < 3 * B; q += kMidThreads) task[q] = make_int4(-1, 0, 0, 0);
}

// ---- the WIDE form of build (kMidMaxB < batch <= kMidWideMaxB) ---------------------------------------------------------------------
// dynamic LDS: start[kMidWideRows] | cur[kMidWideRows] | list[kMidWideList] | stage[kMidWaves][kMidWaveSort] | bitmap[2B / 32] | queue (16-bit) |
// huge (16-bit).  A range's occurrence list lives in LDS while it has at most kMidWideList entries (its count is known: the range sums) and
// in the workspace otherwise (heavily skewed data: scattered 4-byte stores to memory are partial-line writes, 16 of an item range's 83 us at
// batch 65,536 before the list moved into LDS); the code is the same, through a generic pointer.
constexpr int kMidWideList = 16384;
__host__ __device__ inline int mid_wide_queue_cap(int B) { return 2 * B / (kMidWideThreadSort + 1) + 1; }
__host__ __device__ inline int mid_wide_huge_cap(int B) { return 2 * B / (kMidWaveSort + 1) + 1; }
__host__ __device__ inline size_t mid_build_wide_lds(int B) {
    return (size_t)4 * (2 * kMidWideRows + kMidWideList + kMidWaves * kMidWaveSort + (2 * B + 31) / 32) + (size_t)(mid_wide_queue_cap(B) + mid_wide_huge_cap(B) + 2) / 2 * 4 + 16;
}

__global__ __launch_bounds__(kMidThreads) void mid_build_wide_kernel(MidGeom g, int B, int n_users, int n_items,
                                                                      const int32_t* __restrict__ out_u, const int32_t* __restrict__ out_i,
                                                                      const int32_t* __restrict__ out_j, const int4* __restrict__ agg,
                                                                      int4* __restrict__ task_all, int2* __restrict__ occ_all,
                                                                      int32_t* __restrict__ occt_all, uint32_t* __restrict__ touch_u,
                                                                      uint32_t* __restrict__ touch_i, int wu, int wi, uint32_t* olist_all /*[n_batches][3B]*/) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mid_lds[];
    __shared__ int scr[2 * kMidWaves];
    __shared__ int4 scr4[4];
    __shared__ int qn, hn;
    const int words = (2 * B + 31) / 32;
    uint32_t* start = mid_lds;
    uint32_t* cur = start + kMidWideRows;
    uint32_t* lds_list = cur + kMidWideRows;
    uint32_t* stage = lds_list + kMidWideList;
    uint32_t* bitmap = stage + kMidWaves * kMidWaveSort;
    uint16_t* queue = reinterpret_cast<uint16_t*>(bitmap + words);
    uint16_t* huge = queue + mid_wide_queue_cap(B);
    const int tid = threadIdx.x, lane = tid & (TKR_WAVE - 1), wave = tid >> 6;
    const int G = g.gu + g.gi;
    const MidRange r = mid_range(g, B, n_users, n_items, out_u, out_i, out_j);
    MID_STAMP_INIT();
    int4 pre, all;
    mid_sums(agg + (size_t)r.b * G, G, r.c, scr4, pre, all);
    MID_STAMP(0);
    if (tid == 0) { qn = 0; hn = 0; }
    mid_count(r, B, start);
    MID_STAMP(1);                                // start[] holds the counts for now

    const int per = (r.rows + kMidThreads - 1) / kMidThreads;
    const int r0 = min(tid * per, r.rows), r1 = min(r0 + per, r.rows);
    int my_k = 0, my_t = 0;
    for (int q = r0; q < r1; ++q) { const int c = (int)start[q]; my_k += c; my_t += c > 0; }
    int ex_k, ex_t, tot_k, tot_t;
    mid_scan2(my_k, my_t, scr, ex_k, ex_t, tot_k, tot_t);
    int4* task = task_all + (size_t)r.b * 3 * B;
    uint32_t* touch = r.item ? touch_i : touch_u;
    const int wt = r.item ? wi : wu;
    // this range's runs, in the order of its rows
    uint32_t* olist = (tot_k <= kMidWideList) ? lds_list : olist_all + (size_t)r.b * 3 * B + pre.y;
    uint32_t tw = 0;
    int tw_at = -1;
    for (int q = r0; q < r1; ++q) {
        const int c = (int)start[q];
        if (c > 0) {
            const int row = r.lo + q;
            task[pre.x + ex_t] = make_int4((int)((uint32_t)row | ((uint32_t)r.item << 31)), pre.y + ex_k, c, 0);
            if (wt) {
                if ((row >> 5) != tw_at) {
                    if (tw) atomicOr(&touch[(size_t)r.b * wt + tw_at], tw);
                    tw = 0; tw_at = row >> 5;
                }
                tw |= 1u << (row & 31);
            } else {
                atomicOr(&touch[(size_t)row * kTouchWords + (r.b >> 5)], 1u << (r.b & 31));
            }
            ++ex_t;
        }
        start[q] = (uint32_t)ex_k;
        cur[q] = 0;
        ex_k += c;
    }
    if (tw) atomicOr(&touch[(size_t)r.b * wt + tw_at], tw);
    __syncthreads();
    MID_STAMP(2);
    // (A draw in sixteen is a hit: the hits of eight slots are compacted through the wave's stage and then taken a lane each, instead of
    // every slot running the cursor's round trip for three or four lanes.  What the fill costs at batch 65,536 is the WALK, though --
    // 25 of its 31 % of the kernel wait for the draws: every one of a batch's 32 ranges reads all of them, 110 MB per pass and call.)
    {
        uint32_t* hits = stage + wave * kMidWaveSort;               // <= 8 slots x 64 lanes
        const auto take = [&](int n) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = lane; i < n; i += TKR_WAVE) {
                const uint32_t e = hits[i];
                olist[start[e >> 17] + atomicAdd(&cur[e >> 17], 1u)] = e & 0x1ffffu;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        int n = 0;
        const auto put = [&](bool pred, int o, unsigned rl) {
            const unsigned long long m = __ballot(pred);
            if (pred) hits[n + __popcll(m & ((1ull << lane) - 1ull))] = (rl << 17) | (uint32_t)o;
            n += __popcll(m);
        };
        if ((B & 3) == 0) {
            for (int o0 = tid * 4; o0 < ((r.n_occ + 4 * kMidThreads - 1) / (4 * kMidThreads)) * 4 * kMidThreads; o0 += 4 * 4 * kMidThreads) {     // (wave-uniform trip count)
                int4 v[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int o = o0 + x * 4 * kMidThreads;
                    v[x] = make_int4(-1, -1, -1, -1);
                    if (o < r.n_occ) v[x] = *reinterpret_cast<const int4*>((r.item && o >= B) ? r.a1 + (o - B) : r.a0 + o);
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int o = o0 + x * 4 * kMidThreads;
                    const unsigned q0 = (unsigned)(v[x].x - r.lo), q1 = (unsigned)(v[x].y - r.lo), q2 = (unsigned)(v[x].z - r.lo), q3 = (unsigned)(v[x].w - r.lo);
                    put(q0 < (unsigned)r.rows, o, q0);
                    put(q1 < (unsigned)r.rows, o + 1, q1);
                    put(q2 < (unsigned)r.rows, o + 2, q2);
                    put(q3 < (unsigned)r.rows, o + 3, q3);
                    if (x & 1) { take(n); n = 0; }
                }
            }
        } else {
            for (int o0 = 0; o0 < r.n_occ; o0 += kMidThreads) {
                const int o = o0 + tid;
                const unsigned rl = o < r.n_occ ? (unsigned)(mid_row_of(r, B, o) - r.lo) : 0xffffffffu;
                put(rl < (unsigned)r.rows, o, rl);
                take(n); n = 0;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    MID_STAMP(3);
    // ascending occurrence order inside every run
    for (int q = r0; q < r1; ++q) {
        const int c = (int)cur[q], s = (int)start[q];
        if (c >= 2 && c <= 16) {
            uint32_t v[16];
#pragma unroll
            for (int x = 0; x < 16; ++x) v[x] = x < c ? olist[s + x] : 0xffffffffu;
            reg_sort_asc<16>(v);
#pragma unroll
            for (int x = 0; x < 16; ++x)
                if (x < c) olist[s + x] = v[x];
        } else if (c > 16 && c <= kMidWideThreadSort) {
            uint32_t v[kMidWideThreadSort];
#pragma unroll
            for (int x = 0; x < kMidWideThreadSort; ++x) v[x] = x < c ? olist[s + x] : 0xffffffffu;
            reg_sort_asc<kMidWideThreadSort>(v);
#pragma unroll
            for (int x = 0; x < kMidWideThreadSort; ++x)
                if (x < c) olist[s + x] = v[x];
        } else if (c > kMidWideThreadSort) {
            queue[atomicAdd(&qn, 1)] = (uint16_t)q;
        }
    }
    __syncthreads();
    MID_STAMP(4);
    const int n_q = qn;
    uint32_t* mine_stage = stage + wave * kMidWaveSort;
    for (int qi = wave; qi < n_q; qi += kMidWaves) {
        const int q = queue[qi];
        const int c = (int)cur[q], s = (int)start[q];
        if (c > kMidWaveSort) {
            if (lane == 0) huge[atomicAdd(&hn, 1)] = (uint16_t)q;
            continue;
        }
        if (c <= TKR_WAVE) {
            const uint32_t e = lane < c ? olist[s + lane] : 0xffffffffu;
            int rk = 0;
            for (int y = 0; y < c; ++y) rk += (uint32_t)__builtin_amdgcn_readlane((int)e, y) < e;
            if (lane < c) olist[s + rk] = e;
            continue;
        }
        constexpr int E = kMidWaveSort / TKR_WAVE;
        uint32_t e[E];
        int rk[E];
#pragma unroll
        for (int x = 0; x < E; ++x) {
            const int idx = lane + x * TKR_WAVE;
            e[x] = idx < c ? olist[s + idx] : 0xffffffffu;
            mine_stage[idx] = e[x];
            rk[x] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int y0 = 0; y0 < c; y0 += 4) {
            uint32_t v[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) v[z] = mine_stage[y0 + z];          // (past the run: the padding stored above, never below an entry)
#pragma unroll
            for (int z = 0; z < 4; ++z)
#pragma unroll
                for (int x = 0; x < E; ++x) rk[x] += v[z] < e[x];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int x = 0; x < E; ++x)
            if (lane + x * TKR_WAVE < c) olist[s + rk[x]] = e[x];
    }
    __syncthreads();
    MID_STAMP(5);
    const int n_h = hn, wpt = (words + kMidThreads - 1) / kMidThreads;       // bitmap words per thread
    for (int hi = 0; hi < n_h; ++hi) {
        const int q = huge[hi];
        const int c = (int)cur[q], s = (int)start[q];
        for (int w = tid; w < words; w += kMidThreads) bitmap[w] = 0;
        __syncthreads();
        for (int x = tid; x < c; x += kMidThreads) { const uint32_t o = olist[s + x]; atomicOr(&bitmap[o >> 5], 1u << (o & 31)); }
        __syncthreads();
        const int w0 = min(tid * wpt, words), w1 = min(w0 + wpt, words);
        int pc = 0;
        for (int w = w0; w < w1; ++w) pc += __popc(bitmap[w]);
        int ex, ex2, tot, tot2;
        mid_scan2(pc, 0, scr, ex, ex2, tot, tot2);
        int pos = s + ex;
        for (int w = w0; w < w1; ++w) {
            uint32_t bits = bitmap[w];
            while (bits) {
                const int bit = __ffs(bits) - 1;
                olist[pos++] = (uint32_t)(w * 32 + bit);
                bits &= bits - 1;
            }
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    MID_STAMP(6);
    int2* occ = occ_all + (size_t)r.b * 3 * B + pre.y;
    int32_t* occt = occt_all + (size_t)r.b * 3 * B + pre.y;
    const int32_t* bu = out_u + (size_t)r.b * B;
    const int32_t* bi = out_i + (size_t)r.b * B;
    const int32_t* bj = out_j + (size_t)r.b * B;
    constexpr int EU = 8;
    for (int p0 = tid; p0 < tot_k; p0 += EU * kMidThreads) {
        int o[EU], a[EU], c2[EU];
#pragma unroll
        for (int x = 0; x < EU; ++x) o[x] = p0 + x * kMidThreads < tot_k ? (int)olist[p0 + x * kMidThreads] : 0;
#pragma unroll
        for (int x = 0; x < EU; ++x) {
            if (r.item) {
                const bool role = o[x] >= B;
                const int t = role ? o[x] - B : o[x];
                a[x] = bu[t];
                c2[x] = (int)((uint32_t)(role ? bi[t] : bj[t]) | ((uint32_t)role << 31));
                o[x] = t;
            } else {
                a[x] = bi[o[x]];
                c2[x] = bj[o[x]];
            }
        }
#pragma unroll
        for (int x = 0; x < EU; ++x) {
            const int p = p0 + x * kMidThreads;
            if (p < tot_k) { occ[p] = make_int2(a[x], c2[x]); occt[p] = o[x]; }
        }
    }
    MID_STAMP(7);
    if (r.c == G - 1)
        for (int q = all.x + tid; q < 3 * B; q += kMidThreads) task[q] = make_int4(-1, 0, 0, 0);
}

// Batch-major touch maps (touch[b][row / 32], bit row % 32: the same n_rows * 64 bytes as the row-major maps of the other planners,
// which leave them zero between calls): every word becomes the PARITY of its 32 rows at the start of batch b -- the rows' counters
// and the exclusive XOR over the batches in front -- and the counters advance by the rows' touches.  A batch's parities are then one
// compact table (1.3 KB of items, 8.7 KB of users at the MovieLens-10M shape) instead of a counter and a bitmap word per lookup in
// two tables of megabytes: the resolve step's ~120,000 lookups per batch of 8192 were sector-sized gathers past the L2 (round 6: 55
// of its 93 us per 64 batches).
__global__ __launch_bounds__(256) void mid_prefix_kernel(int n_users, int n_items, int wu, int wi, int n_batches, int32_t* __restrict__ ucnt,
                                                         int32_t* __restrict__ icnt, uint32_t* __restrict__ touch_u, uint32_t* __restrict__ touch_i) {
    // a wave per word of 32 rows, a lane per batch (64 at a time)
    const int g = blockIdx.x * (blockDim.x / TKR_WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & (TKR_WAVE - 1);
    if (g >= wu + wi) return;
    const bool item = g >= wu;
    const int w = item ? g - wu : g, W = item ? wi : wu, n = item ? n_items : n_users;
    int32_t* cnt = (item ? icnt : ucnt) + w * 32;
    uint32_t* T = (item ? touch_i : touch_u) + w;
    const int rows = min(32, n - w * 32);
    const int c0 = lane < rows ? cnt[lane] : 0;
    uint32_t run = (uint32_t)__ballot(c0 & 1);             // bit r = parity of row r (rows <= 32: the low word)
    int add = 0;                                           // lane r < 32: touches of row r in this call
    for (int b0 = 0; b0 < n_batches; b0 += TKR_WAVE) {
        const int b = b0 + lane;
        const uint32_t t = b < n_batches ? T[(size_t)b * W] : 0u;
        uint32_t x = t;                                    // inclusive XOR scan over the lanes
#pragma unroll
        for (int d = 1; d < TKR_WAVE; d <<= 1) {
            const uint32_t up = __shfl_up(x, d);
            if (lane >= d) x ^= up;
        }
        if (b < n_batches) T[(size_t)b * W] = run ^ x ^ t;           // exclusive: the batches in front
        run ^= __shfl(x, TKR_WAVE - 1);
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int c = __popcll(__ballot((t >> r) & 1u));
            if (lane == r) add += c;
        }
    }
    if (lane < rows && add) cnt[lane] = c0 + add;
}

__global__ void mid_zero_kernel(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b, size_t nb) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < na) a[g] = 0;
    else if (g < na + nb) b[g - na] = 0;
}

template <bool TR>
__device__ __forceinline__ int mid_parity(const int32_t* __restrict__ cnt, const uint32_t* __restrict__ touch, int wt, int row, int b) {
    if constexpr (TR) return (int)((touch[(size_t)b * wt + (row >> 5)] >> (row & 31)) & 1u);
    else return parity_of(cnt, touch, row, b);
}

__device__ __forceinline__ int mid_pack_t16(int a, int b) { return (int)(((uint32_t)a & 0xffffu) | (((uint32_t)b & 0xffffu) << 16)); }

template <bool TR>
__global__ __launch_bounds__(kMidThreads, 8) void mid_resolve_kernel(int wu, int wi, int G, int B, int team, int lmax, int rec_stride, const int4* __restrict__ agg,
                                                                   int4* __restrict__ task_all, int2* __restrict__ occ_all,
                                                                   const int32_t* __restrict__ occt_all, const int32_t* __restrict__ out_u,
                                                                   const int32_t* __restrict__ ucnt, const int32_t* __restrict__ icnt,
                                                                   const uint32_t* __restrict__ touch_u, const uint32_t* __restrict__ touch_i,
                                                                   int32_t* __restrict__ tpar_all, int32_t* __restrict__ rec_all,
                                                                   int4* __restrict__ hdr_all) {
    __shared__ int scr[2 * kMidWaves];
    __shared__ int4 scr4[4];
    __shared__ int4 heavy[kMidThreads];                               // the heavy tasks of one round of kMidThreads tasks
    const int tid = threadIdx.x, b = blockIdx.y, c = blockIdx.x;
    int4 pre, all;
    mid_sums(agg + (size_t)b * G, G, c, scr4, pre, all);
    const int4 mine = agg[(size_t)b * G + c];
    int4* task = task_all + (size_t)b * 3 * B;
    int2* occ = occ_all + (size_t)b * 3 * B;
    const int32_t* occt = occt_all + (size_t)b * 3 * B;
    // parities of the partner rows of this range's occurrences (+ the per-triplet parities)
    // (four occurrences per thread at a time: every lookup is a gather with nothing to do behind it but wait)
    const int p_end = pre.y + mine.y;
    for (int p0 = pre.y + tid; p0 < p_end; p0 += 4 * kMidThreads) {
        int2 o[4];
        int pa[4], pb[4], tt[4], pu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + q * kMidThreads;
            o[q] = p < p_end ? occ[p] : make_int2(0, 0);
            tt[q] = (tpar_all && p < B) ? occt[p] : 0;                 // (p < B implies p < p_end: user occurrences come first)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + q * kMidThreads;
            pa[q] = pb[q] = pu[q] = 0;
            if (p < p_end) {
                if (p < B) {                                           // user occurrence: (i, j)
                    pa[q] = mid_parity<TR>(icnt, touch_i, wi, o[q].x, b);
                    pb[q] = mid_parity<TR>(icnt, touch_i, wi, o[q].y, b);
                    if (tpar_all) pu[q] = mid_parity<TR>(ucnt, touch_u, wu, out_u[(size_t)b * B + tt[q]], b);
                } else {                                               // item occurrence: (u, other | role << 31)
                    pa[q] = mid_parity<TR>(ucnt, touch_u, wu, o[q].x, b);
                    pb[q] = mid_parity<TR>(icnt, touch_i, wi, o[q].y & 0x3fffffff, b);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + q * kMidThreads;
            if (p < p_end) {
                if (tpar_all && p < B) tpar_all[(size_t)b * B + tt[q]] = pu[q] | (pa[q] << 1) | (pb[q] << 2);
                occ[p] = make_int2(o[q].x | (pa[q] << 30), o[q].y | (pb[q] << 30));
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    const int tot_l = all.z, tot_h = all.w;
    const int nlb = (tot_l + team - 1) / team;                        // light tasks per workgroup = team (plan_parts.h light_per_block)
    int32_t* rec = rec_all + (size_t)b * rec_stride * 16;
    int base_l = pre.z, n_heavy = 0;
    const int lane = tid & (TKR_WAVE - 1), wave = tid >> 6;
    for (int base = 0; base < mine.x; base += kMidThreads) {
        const int s = pre.x + base + tid;
        const bool valid = base + tid < mine.x;
        int4 t = valid ? task[s] : make_int4(-1, 0, 0, 0);
        if (valid) {
            const int row = t.x & 0x7fffffff;
            t.w = (t.x < 0) ? mid_parity<TR>(icnt, touch_i, wi, row, b) : mid_parity<TR>(ucnt, touch_u, wu, row, b);
            task[s] = t;
        }
        const int l = valid && t.z <= lmax, h = valid && t.z > lmax;
        int ex_l, ex_h, n_l, n_h;
        mid_scan2(l, h, scr, ex_l, ex_h, n_l, n_h);
        if (l) {
            const int li = base_l + ex_l;
            int4* r = reinterpret_cast<int4*>(rec + (size_t)li * 16);       // (li / lpb) * team + li % lpb with lpb == team
            int2 o[4];
            int tt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = (q < t.z) ? occ[t.y + q] : make_int2(0, 0);
                tt[q] = (q < t.z) ? occt[t.y + q] : 0;
            }
            r[0] = make_int4(t.x, t.w | (1 << 8), t.z, t.y);
            r[1] = make_int4(o[0].x, o[0].y, o[1].x, o[1].y);
            r[2] = make_int4(o[2].x, o[2].y, o[3].x, o[3].y);
            r[3] = make_int4(t.z, mid_pack_t16(tt[0], tt[1]), mid_pack_t16(tt[2], tt[3]), 0);
        } else if (h) {
            heavy[ex_h] = t;                                           // its records: below, a lane per wave record
        }
        __syncthreads();
        for (int e = wave * (TKR_WAVE / 16); e < n_h; e += kMidWaves * (TKR_WAVE / 16)) {       // four heavy tasks per wave: team <= 16
            const int mine_e = e + (lane >> 4), w = lane & 15;
            if (mine_e < n_h && w < team) {
                const int4 th = heavy[mine_e];
                const int hi = pre.w + n_heavy + mine_e;
                int4* r = reinterpret_cast<int4*>(rec + ((size_t)(nlb + hi) * team + w) * 16);
                const int n_mine = (th.z > w) ? (th.z - w + team - 1) / team : 0;
                int2 o[4];
                int tt[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q] = (q < n_mine) ? occ[th.y + w + q * team] : make_int2(0, 0);
                    tt[q] = (q < n_mine) ? occt[th.y + w + q * team] : 0;
                }
                r[0] = make_int4(th.x, th.w | (team << 8) | (w << 16), n_mine, th.y + w);
                r[1] = make_int4(o[0].x, o[0].y, o[1].x, o[1].y);
                r[2] = make_int4(o[2].x, o[2].y, o[3].x, o[3].y);
                r[3] = make_int4(th.z, mid_pack_t16(tt[0], tt[1]), mid_pack_t16(tt[2], tt[3]), 0);
            }
        }
        base_l += n_l;
        n_heavy += n_h;
    }
    if (c == G - 1) {
        if (tid == 0) hdr_all[b] = make_int4(nlb + tot_h, nlb, tot_h, tot_l + tot_h);
        for (int s = tot_l + tid; s < nlb * team; s += kMidThreads) {       // idle wave slots of the last light workgroup
            int4* r = reinterpret_cast<int4*>(rec + (size_t)s * 16);
            r[0] = make_int4(-1, 0, 0, 0);
            r[1] = r[2] = r[3] = make_int4(0, 0, 0, 0);
        }
    }
}

}  // namespace tkr

#ifdef TKR_MID_PROF
extern "C" int tkr_debug_mid_prof(unsigned long long* out /*[32] host*/) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tkr::mid_prof), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -100;
}
#endif
extern "C" int tkr_plan_team(int32_t batch_size);
extern "C" int tkr_plan_max_blocks(int32_t batch_size);
extern "C" __attribute__((visibility("hidden"))) int tkr_plan_commit(int32_t n_users, int32_t n_items, int32_t* ucnt, int32_t* icnt, uint32_t* touch_u,
                                                                     uint32_t* touch_i, void* stream);       // csrc/sampler.hip: K1c

// (helpers between translation units, not entry points)
extern "C" __attribute__((visibility("hidden"))) int tkr_plan_mid_ok(int32_t n_users, int32_t n_items, int32_t B) {
    static const int from = [] { const char* e = getenv("TKR_PLAN_MID_FROM"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1025; }();
    static const int upto = [] { const char* e = getenv("TKR_PLAN_MID_UPTO"); const int v = e ? atoi(e) : 0; return v > 0 ? v : tkr::kMidWideMaxB; }();
    if (B < from || B > upto || B > tkr::kMidWideMaxB) return 0;
    // every range's workgroup walks all of the batch's draws: worth it while a range keeps a fair share of them -- 480,189 users are 59
    // ranges, fine at batch 8192 (Netflix shape: 311 -> 330 M triplets/s on the line), a loss at 2048 (208 -> 183)
    const tkr::MidGeom g = tkr::mid_geom(n_users, n_items, B);
    const int G = g.gu + g.gi, worth = B / 128 > 16 ? B / 128 : 16;
    return G <= tkr::kMidMaxRanges && G <= worth;
}
extern "C" __attribute__((visibility("hidden"))) int64_t tkr_plan_mid_workspace_bytes(int32_t batch_size, int32_t n_batches) {
    if (batch_size <= 1024 || batch_size > tkr::kMidWideMaxB || n_batches <= 0) return 0;
    const int64_t sums = (int64_t)n_batches * tkr::kMidMaxRanges * (int64_t)sizeof(int4);
    return tkr::mid_wide(batch_size) ? sums + (int64_t)n_batches * 3 * batch_size * 4 : sums;         // the wide form's occurrence lists
}

extern "C" __attribute__((visibility("hidden"))) int tkr_sample_plan_mid(
    const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr, const int32_t* pos_cols, const int32_t* cols_sorted, int32_t n_users,
    int32_t n_items, uint64_t seed, uint64_t first_triplet, const int64_t* ctl, int32_t n_batches, int32_t B, int32_t* ucnt, int32_t* icnt,
    uint32_t* touch_u, uint32_t* touch_i, int32_t* out_u, int32_t* out_i, int32_t* out_j, int32_t* task, int32_t* occ, int32_t* rec,
    int32_t* hdr, int32_t* occt, int32_t* tpar, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace tkr;
    if (!tkr_plan_mid_ok(n_users, n_items, B)) return TKR_EUNSUPPORTED;
    if (!workspace || workspace_bytes < tkr_plan_mid_workspace_bytes(B, n_batches)) return TKR_EINVAL;
    const MidGeom g = mid_geom(n_users, n_items, B);
    const int G = g.gu + g.gi;
    hipStream_t s = (hipStream_t)stream;
    int4* agg = reinterpret_cast<int4*>(workspace);
    const size_t nB = (size_t)B * n_batches;
    const bool wide = mid_wide(B);
    uint32_t* olist = reinterpret_cast<uint32_t*>(agg + (size_t)n_batches * kMidMaxRanges);
    const size_t lds_build = wide ? mid_build_wide_lds(B) : mid_build_lds(B), lds_count = (size_t)kMidRows * 4;
    static bool attr_set[64] = {};                 // per device: the attribute belongs to the device's code object
    int dev = 0;
    TKR_CHECK(hipGetDevice(&dev));
    if (!(dev >= 0 && dev < 64 && attr_set[dev])) {
        TKR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_build_lds(kMidMaxB)));      // (+ the kernel's static LDS: 160 KB would be refused)
        TKR_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mid_build_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_build_wide_lds(kMidWideMaxB)));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(mid_draw_kernel, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, s, tr_users, (uint32_t)n_tr, row_ptr, pos_cols,
                       cols_sorted, (uint32_t)n_items, seed, first_triplet, ctl, B, nB, out_u, out_i, out_j);
    TKR_LAUNCH_CHECK();
    const int lmax = light_max(B), team = tkr_plan_team(B);
    hipLaunchKernelGGL(mid_count_kernel, dim3(G, n_batches), dim3(kMidThreads), lds_count, s, g, B, lmax, n_users, n_items, out_u, out_i,
                       out_j, agg);
    TKR_LAUNCH_CHECK();
    // batch-major touch maps when the call's batches fit the maps' n_rows * 16 words (all but the last few of 512 batches when n_rows is
    // not a multiple of 32); TKR_PLAN_MID_ROWMAJOR=1: the row-major maps and the lookups of the other planners
    static const bool rowmajor = getenv("TKR_PLAN_MID_ROWMAJOR") && getenv("TKR_PLAN_MID_ROWMAJOR")[0] == '1';
    const int wu = (n_users + 31) / 32, wi = (n_items + 31) / 32;
    const bool tr = !rowmajor && (size_t)n_batches * wu <= (size_t)n_users * kTouchWords && (size_t)n_batches * wi <= (size_t)n_items * kTouchWords;
    if (wide)
        hipLaunchKernelGGL(mid_build_wide_kernel, dim3(G, n_batches), dim3(kMidThreads), lds_build, s, g, B, n_users, n_items, out_u, out_i, out_j,
                           agg, reinterpret_cast<int4*>(task), reinterpret_cast<int2*>(occ), occt, touch_u, touch_i, tr ? wu : 0, tr ? wi : 0, olist);
    else
        hipLaunchKernelGGL(mid_build_kernel, dim3(G, n_batches), dim3(kMidThreads), lds_build, s, g, B, n_users, n_items, out_u, out_i, out_j,
                           agg, reinterpret_cast<int4*>(task), reinterpret_cast<int2*>(occ), occt, touch_u, touch_i, tr ? wu : 0, tr ? wi : 0);
    TKR_LAUNCH_CHECK();
    if (tr) {
        hipLaunchKernelGGL(mid_prefix_kernel, dim3((wu + wi + 3) / 4), dim3(256), 0, s, n_users, n_items, wu, wi, n_batches, ucnt, icnt,
                           touch_u, touch_i);
        hipLaunchKernelGGL(mid_resolve_kernel<true>, dim3(G, n_batches), dim3(kMidThreads), 0, s, wu, wi, G, B, team, lmax,
                           tkr_plan_max_blocks(B) * team, agg, reinterpret_cast<int4*>(task), reinterpret_cast<int2*>(occ), occt, out_u, ucnt,
                           icnt, touch_u, touch_i, tpar, rec, reinterpret_cast<int4*>(hdr));
        TKR_LAUNCH_CHECK();
        const size_t zu = (size_t)n_batches * wu, zi = (size_t)n_batches * wi;         // zero between calls, as every planner leaves them
        hipLaunchKernelGGL(mid_zero_kernel, dim3((unsigned)((zu + zi + 255) / 256)), dim3(256), 0, s, touch_u, zu, touch_i, zi);
        TKR_LAUNCH_CHECK();
        return TKR_OK;
    }
    hipLaunchKernelGGL(mid_resolve_kernel<false>, dim3(G, n_batches), dim3(kMidThreads), 0, s, 0, 0, G, B, team, lmax,
                       tkr_plan_max_blocks(B) * team, agg, reinterpret_cast<int4*>(task), reinterpret_cast<int2*>(occ), occt, out_u, ucnt, icnt,
                       touch_u, touch_i, tpar, rec, reinterpret_cast<int4*>(hdr));
    TKR_LAUNCH_CHECK();
    return tkr_plan_commit(n_users, n_items, ucnt, icnt, touch_u, touch_i, stream);
}
