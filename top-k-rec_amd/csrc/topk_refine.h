// K4, bound-and-refine, second form (round 6): interface between csrc/topk.hip (planning, bounds, tile image, fp32 fallback, merge)
// and csrc/topk_refine.hip (the tile kernel with packed candidate lists and the finish kernel that ranks a row once).
#pragma once
#include "tkr_common.h"

namespace tkr {

constexpr int kR2Users = 128;            // users per workgroup of the tile kernel (4 waves x 32)
constexpr int kR2Slots = 64;             // list slots per user: two private segments of 32 (one per half-lane)
constexpr int kR2SpansPerCU = 3;         // workgroups per CU the tile kernel is built for (LDS 48.3 KB, <= 168 registers)

struct Refine2Args {
    const float* U;
    const int32_t* uidx;
    int n_rows;
    const float* Vt;
    const float* bias;
    int n_cols, k;
    const uint32_t* mask;
    int mask_pitch, K;
    int grid_x, grid_y;                  // the plan's grid: pieces of the item table, or (user blocks, item ranges)
    int tiles_per_split;                 // plain grid only
    uint32_t* thr_shared;                // [n_rows] bounds the pieces of a block tell each other, or null
    const int4* items;                   // item table (block, t_begin, t_end, slot | stride << 16), or null: the plain grid
    const int32_t* nslots;               // [blocks] pieces per block (item table), or null: grid_y everywhere
    const int32_t* pbase;                // [blocks] first piece of a block in the dump (item table), or null: block * grid_y
    uint32_t* extra;                     // [0..2] bounds of topk_bounds_kernel, [4 + block] set when a list of the block overflowed
    const unsigned char* vimg;           // scaled fp16 image of the item tiles (topk_image_kernel)
    uint32_t* dump;                      // [pieces][128][64] packed entries
    float2* dhdr;                        // [pieces][128] {segment lengths, final threshold}
    int32_t* out_ids;
    float* out_scores;                   // or null
};

size_t refine2_dump_bytes(size_t n_pieces);                       // room for `dump` + `dhdr`
bool refine2_supports(int n_cols, int k);                          // 16-bit ids, k <= 128
int launch_refine2(const Refine2Args& a, hipStream_t stream);     // tile kernel + finish kernel; TKR_EUNSUPPORTED when !refine2_supports

}  // namespace tkr
