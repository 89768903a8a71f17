// Geometry of the fused local-window kernel (csrc/local_match.hip) and of the padded pooled planes it reads, shared
// with the per-frame prepare kernel (csrc/global_match.hip: manet_frame_prepare writes those planes and the tile table).
#pragma once
#include "manet_common.h"

namespace {

// F.interpolate(..., mode='bilinear', align_corners=True) coefficients (IntVOS.py:295):
// scale = (in-1)/(out-1), src = scale*dst, i0 = floor, i1 = i0 + (i0 < in-1), l1 = src - i0.
struct Bilin {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Bilin bilin_coeff(int dst, int in_size, int out_size)
{
    float scale = (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
    float src = scale * (float)dst;
    int a = (int)src;
    if (a > in_size - 1) a = in_size - 1;
    Bilin b;
    b.i0 = a;
    b.i1 = a + ((a < in_size - 1) ? 1 : 0);
    b.l1 = src - (float)a;
    b.l0 = 1.0f - b.l1;
    return b;
}
// smallest dst in [0, out_size] whose i0 is >= target (i0 is monotone in dst): an estimate from the inverse map,
// settled with the forward expression itself -- a couple of evaluations instead of a scan over the tile
__host__ __device__ __forceinline__ int bilin_first(int target, int in_size, int out_size)
{
    if (target <= 0) return 0;
    if (target > in_size - 1) return out_size;
    const float scale = (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;  // as bilin_coeff
    if (scale == 0.0f) return out_size;
    // i0(dst) of bilin_coeff is min((int)(scale * dst), in_size - 1); target <= in_size - 1 here, so the clamp never
    // decides the comparison
    int y = (int)((float)target * (1.0f / scale));
    y = y < 0 ? 0 : (y > out_size ? out_size : y);
    while (y > 0 && (int)(scale * (float)(y - 1)) >= target) --y;
    while (y < out_size && (int)(scale * (float)y) < target) ++y;
    return y;
}

__host__ __device__ constexpr int lf_cols(int d) { return d <= 6 ? 2 : 4; }                       // columns per thread
// d = 3, 4 and d >= 10: a thread owns one HALF of the window columns (d=4: dx 0..3 | 4..8, d=12: 0..11 | 12..24): half
// the running sums, twice the threads -- two waves per SIMD for the arithmetic (a workgroup per CU is all the grid
// offers, so threads are the only source of latency hiding) and twice the lanes for the per-pixel phase
__host__ __device__ constexpr int lf_dxs(int d) { return (d == 3 || d == 4 || d >= 10) ? 2 : 1; }
// window columns of the first half: a multiple of the column group, so the second half's vector reads stay aligned
__host__ __device__ constexpr int lf_ph(int d) { return lf_cols(d) == 4 ? 12 : ((d + 1) & ~1); }
__host__ __device__ constexpr int lf_pa(int d)  // running sums per column per thread
{
    return lf_dxs(d) == 1 ? 2 * d + 1 : ((2 * d + 1 - lf_ph(d)) > lf_ph(d) ? (2 * d + 1 - lf_ph(d)) : lf_ph(d));
}
__host__ __device__ constexpr int lf_nt(int d) { return d <= 2 ? 256 : (d <= 6 ? 512 : (d <= 9 ? 256 : 512)); }  // threads
__host__ __device__ constexpr int lf_slots(int d) { return lf_nt(d) / ((16 / lf_cols(d)) * lf_dxs(d)); }  // (row, dy) slots
__host__ __device__ constexpr int lf_nd(int d) { return d <= 10 ? 2 * d + 1 : (d == 11 ? 12 : 5); }  // dy per workgroup
__host__ __device__ constexpr int lf_ndg(int d) { return (2 * d + 1 + lf_nd(d) - 1) / lf_nd(d); }
__host__ __device__ constexpr int lf_sy(int d) { return lf_slots(d) / lf_nd(d) > 12 ? 12 : lf_slots(d) / lf_nd(d); }
constexpr int LF_SX = 16;  // columns of S
__host__ __device__ constexpr int lf_cw(int d)  // halo row stride (every thread reads whole vectors: room for the over-read)
{
    // COLS = 4 halves: 40 columns are needed; 48 = 16 (mod 64 banks) makes the four (row, dy) slots of a b128 lane
    // group land on disjoint banks (r2 PMC at stride 40: SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE; a
    // window read cost 10 LDS cycles instead of 4)
    return lf_cols(d) == 4 && lf_dxs(d) == 2 ? 48 : ((LF_SX + 2 * d + 3) & ~3);
}
__host__ __device__ constexpr int lf_yr(int d) { return lf_sy(d) + lf_nd(d) - 1; }                // halo rows
__host__ __device__ constexpr int lf_yplane(int d) { return lf_yr(d) * lf_cw(d); }
__host__ __device__ constexpr int lf_xplane(int d) { return lf_sy(d) * LF_SX; }
// channels per LDS stage: two stages <= 128 KiB (the phase-2 volume needs as much at the wide windows anyway)
__host__ __device__ constexpr int lf_cc(int d)
{
    int cc = 131072 / (8 * (lf_yplane(d) + lf_xplane(d)));
    return cc < 2 ? 2 : (cc > 25 ? 25 : cc);
}
// floats of one stage image, padded to whole 1 KiB LDS-DMA pieces
__host__ __device__ constexpr int lf_stage_floats(int d)
{
    return (lf_cc(d) * (lf_yplane(d) + lf_xplane(d)) / 4 + 63) / 64 * 256;
}
__host__ __device__ constexpr int lf_lab_rows(int d) { return 2 * (lf_sy(d) - 1) + 4 + 2 * (lf_nd(d) - 1); }
__host__ __device__ constexpr int lf_lab_cols(int d) { return 2 * (LF_SX - 1) + 4 + 4 * d; }
// phase 2 volume: [dy][cell of S][dx], dx contiguous, cell stride an ODD number of float4 -- a pixel's four taps are
// ds_read_b128 of 4 window columns each, and the 16 cells of a row of S land on 64 distinct banks
__host__ __device__ constexpr int lf_vs(int d) { return 4 * (((2 * d + 1 + 3) / 4) | 1); }
// pixels of a tile, upper bound -- rounded up to a multiple of the 64 LDS banks: the (id, pixel) slot a lane's atomic min goes to
// is id * lf_npix + pixel, a wave's lanes hold 64 CONSECUTIVE pixels, so whatever ids they carry they hit 64 distinct banks
// (r1-r5: 884 = 52 mod 64 -- lanes with different ids collided)
__host__ __device__ constexpr int lf_npix(int d) { return ((2 * (lf_sy(d) - 1) + 4) * (2 * (LF_SX - 1) + 4) + 63) / 64 * 64; }
constexpr int LF_NIP = 8;  // object ids per pass of the per-pixel phase
// LDS of the per-pixel phase: the volume of `rows` window rows (padded to whole 1 KiB LDS-DMA pieces), the label bytes around the tile
// for those rows, the per-(id, pixel) minima of `m2_rows` ids + the "no id" row, the bilinear tables, the label-change masks
__host__ __device__ constexpr int lf_lab_rows_of(int d, int rows) { return 2 * (lf_sy(d) - 1) + 4 + 2 * (rows - 1); }
__host__ __device__ constexpr size_t lf_vpad_bytes(int d, int rows) { return ((size_t)rows * lf_sy(d) * LF_SX * lf_vs(d) * 4 + 1023) / 1024 * 1024; }
__host__ __device__ constexpr size_t lf_lds_phase2_bytes(int d, int rows, int m2_rows)
{
    return lf_vpad_bytes(d, rows) + (((size_t)lf_lab_rows_of(d, rows) * lf_lab_cols(d) + 15) & ~(size_t)15) +
           (size_t)lf_npix(d) * (m2_rows + 1) * 4 + (size_t)(2 * (lf_sy(d) - 1) + 4 + 2 * (LF_SX - 1) + 4) * 16 +
           (size_t)lf_lab_rows_of(d, rows) * 16;
}
// The per-pixel phase ON A STORED VOLUME (the kernel's LF_VOL_IN mode, r6) can deal the window rows of an image to lf_nsub(d) workgroups of
// lf_ndv(d) rows each (two or more then fit a CU).  Measured at 480p, d = 12 (docs/history/r06_experiments.md): 5 rows per workgroup
// (one wave of 240 workgroups) 21.1 us; 2 rows (720 workgroups, two per CU) 23.8; 1 row (1 200, four per CU) 26-30 -- every
// workgroup pays the label fetch, the tables, the minima's initialisation and the closing atomics again, and those, not the volume
// stream, are what a workgroup waits for.  Shipped: no split (lf_ndv = lf_nd).
#ifndef MANET_LF_NDV12
#define MANET_LF_NDV12 5  // (A/B builds: window rows per workgroup of the stored-volume kernel at d = 12)
#endif
#ifndef MANET_LF_NTV
#define MANET_LF_NTV 0    // (A/B builds: threads per workgroup of the stored-volume kernel at d >= 10; 0 = the fused kernel's)
#endif
__host__ __device__ constexpr int lf_ndv(int d) { return d == 12 ? MANET_LF_NDV12 : lf_nd(d); }
__host__ __device__ constexpr int lf_ntv(int d) { return (MANET_LF_NTV > 0 && d >= 10) ? MANET_LF_NTV : lf_nt(d); }
__host__ __device__ constexpr int lf_nsub(int d) { return (lf_nd(d) + lf_ndv(d) - 1) / lf_ndv(d); }
__host__ __device__ constexpr size_t lf_lds_vol_bytes(int d, int n_ids)
{
    return lf_lds_phase2_bytes(d, lf_ndv(d), n_ids <= LF_NIP ? n_ids : LF_NIP);
}
__host__ __device__ constexpr size_t lf_lds_bytes(int d)
{
    size_t stage = 2 * (size_t)lf_stage_floats(d) * 4;
    size_t vol = lf_lds_phase2_bytes(d, lf_nd(d), LF_NIP);
    return stage > vol ? stage : vol;
}
// padded pooled plane [HPAD][WS]: image pixel (py, px) at (d + py, d + px)
struct PoolPad {
    int hp, wp, HPAD, WS;
    long plane;  // HPAD * WS
};
static PoolPad lf_pool_pad(int h, int w, int d)
{
    PoolPad G;
    G.hp = h / 2;
    G.wp = w / 2;
    G.HPAD = G.hp + lf_sy(d) + lf_ndg(d) * lf_nd(d) - 1;  // last tile row + halo rows of the last dy group
    G.WS = (G.wp + lf_cw(d) + 3) & ~3;                     // last tile column + halo columns
    G.plane = (long)G.HPAD * G.WS;
    return G;
}


}  // namespace
