// Global nearest-neighbour matching for MI355X (gfx950).
//
// Replaces networks/IntVOS.py:160-210 (nearest_neighbor_features_per_object) and its helpers
// (:23-40 pairwise distances, :62-97 masked min, :100-109 pixel selection, :113-157 chunk loop)
// plus, as a fused epilogue, :611-612 (normalise) and :615-622 (min-aggregation with the stored
// per-frame global map).
//
// Design (see DESIGN.md for the numbers):
//   * The N x M distance matrix is never materialised.  d(n,m) = (|q_n|^2 + |k_m|^2) - 2 q_n.k_m
//     is produced tile by tile on the matrix cores and reduced to a running minimum in registers.
//   * The contraction runs on v_mfma_f32_32x32x2_f32 with the operands SWAPPED (A = bank rows,
//     B = query pixels): an accumulator register then holds one query column and 16 bank rows, so
//     the reduction over the bank is lane-local (one v_min per element) and only the two 32-lane
//     halves have to be combined at the end (one cross-lane shuffle per object).
//   * bank rows are SORTED BY OBJECT ID in a pre-pass (counting sort; rows whose label is not an
//     object id are dropped, which is what the reference's pixel selection / 1e20 mask amounts
//     to).  Every 64-row bank tile then belongs to one object: no per-element label compare, and
//     work shrinks with the number of labelled pixels.
//   * the pre-pass writes the bank in the exact LDS image the MFMA loop wants ([16-byte unit][row]
//     [4 floats | 8 bf16], conflict-free ds_read_b128) so a tile is staged by linear, fully coalesced
//     16-byte loads, one tile-step ahead through registers, double buffered, one barrier per tile.
//   * three arithmetic modes: fp32 MFMA (exact), bf16 MFMA on rounded embeddings, and split-bf16
//     (hi+lo, three MFMAs: fp32-class accuracy); top-k (k_nn 2..8) as a variant of the fp32 kernel.
//   * the 64 queries x C operand of a wave lives in registers for the whole kernel.
//   * grid = (256-query tiles) x (S bank splits), split index tied to blockIdx % 8 so that the
//     workgroups of one XCD stream the same bank range through that XCD's L2.  Splits combine by
//     atomicMin on order-preserving integer keys; a last tiny kernel decodes, normalises and
//     min-merges with the stored map.
//
// Numerics of the fp32 path: the MFMA is a k-ascending fmaf chain from 0, d = fmaf(-2, mm, xs+ys);
// min is exact -- so the result is bit-identical to oracle/manet_oracle.c.
#include "manet_common.h"
#include <type_traits>
#include "local_geom.h"

namespace {

constexpr int QT = 256;       // queries per workgroup (4 waves x 2 blocks of 32)
constexpr int QB = 32;        // queries per packed block (one MFMA N dimension)
constexpr int BT = 64;        // bank rows per tile (2 MFMA M blocks)
constexpr int META_INTS = 256;
constexpr int META_T = 0;          // [0]        number of bank tiles actually used
constexpr int META_SEG = 1;        // [1..65]    first tile of object o (entry n_ids = T)
constexpr int META_CNT = 130;      // [130..193] rows per object
constexpr int META_KMAX = 200;     // [200]      max |k|^2 over the bank's rows, float bits (MANET_COMPUTE_BF16_REFINE)
// MANET_COMPUTE_BF16_REFINE: a pre-pass over every REFINE_SUB-th bank tile gives an upper bound of the minimum; the full
// bf16 pass then keeps, per (query, object), the bank rows that could beat it; up to REFINE_CAP of them are re-evaluated in
// the reference's fp32 arithmetic
// REFINE_SUB by bank size (refine_sub): the pre-pass costs 1 / REFINE_SUB of the filter pass's matrix work; a sparser sample
// leaves ~ln more candidates per pair for the cheap re-rank -- the larger the bank, the sparser the sample that pays.
constexpr int META_NAN = 66;       // [66..129]  object o's bank rows contain a NaN (MANET_COMPUTE_BF16_REFINE)
int refine_sub(long T_max)
{
    const int forced = manet_tune_get(MANET_TUNE_REFINE_SUB, 0);  // (experiments)
    if (forced > 0) return forced;
    return T_max <= 2500 ? 8 : 16;  // 480p x 5 frames: 2 007 tiles; 720p x 10 frames: 9 006
}
#ifndef MANET_REFINE_XCHG_MASK
#define MANET_REFINE_XCHG_MASK 3
#endif
constexpr int REFINE_XCHG_MASK = MANET_REFINE_XCHG_MASK;  // threshold exchange every (mask + 1) steps
#ifndef MANET_REFINE_CAP
#define MANET_REFINE_CAP 128
#endif
#ifndef MANET_REFINE_LDS_LIST
#define MANET_REFINE_LDS_LIST 2048
#endif
constexpr int REFINE_CAP = MANET_REFINE_CAP;  // capacity of a candidate bucket (one per 32-query block), in rows per (query, object) pair ON AVERAGE
constexpr int REFINE_LDS_LIST = MANET_REFINE_LDS_LIST;  // candidate entries a filter workgroup collects in LDS before it appends them in bulk
// A 32-query x 32-row block with more qualifying distances than a sub-list holds is listed as ONE "dense" entry {block's
// first pair, 0x80000000 | first bank slot of the pass}: the re-rank evaluates all of its 1 024 distances exactly.  At most
// REFINE_DENSE_CAP of them per 32-query bucket (counted in bcnt's second half); one more marks the bucket incomplete and
// the rescue pass (the exact fp32 kernel on the 256-query tile) takes over.  A dense entry is one 32 x 32 x C tile on the fp32
// matrix pipe for ONE wave (1.4 us at C = 100, operands from global memory / LDS per entry): 1 024 of them in every bucket of a
// 480p frame are ~1.1 ms of the chip, a quarter of the fp32 kernel -- beyond that the fp32 kernel's operand reuse wins.
#ifndef MANET_REFINE_DENSE_CAP
#define MANET_REFINE_DENSE_CAP 1024
#endif
constexpr int REFINE_DENSE_CAP = MANET_REFINE_DENSE_CAP;
constexpr unsigned REFINE_DENSE_BIT = 0x80000000u;
// ... and a block is listed whole as soon as more than REFINE_DENSE_MIN of its 1 024 distances qualify: a listed row costs the
// re-rank a 400-byte gather of its bank row (C = 100) per (query, row) pair, a dense entry 12.8 KB of rows per 1 024 pairs and
// 1.4 us of one wave's matrix pipe (more when the wave has nothing else to hide the row loads behind).  Measured on one box,
// ms per step on video-like / smooth embeddings at cfg2 size for (REFINE_DENSE_MIN, REFINE_DENSE_LANES): (128, off) 0.798 /
// 0.963, (64, 24) 0.799 / 0.866, (64, 16) 0.803 / 0.856, (32, 16) 0.806 / 0.858, (32, 8) 0.813 / 0.854.
#ifndef MANET_REFINE_DENSE_MIN
#define MANET_REFINE_DENSE_MIN 64
#endif
constexpr int REFINE_DENSE_MIN = MANET_REFINE_DENSE_MIN;
#ifndef MANET_REFINE_DENSE_LANES
#define MANET_REFINE_DENSE_LANES 16
#endif
constexpr int REFINE_DENSE_LANES = MANET_REFINE_DENSE_LANES;  // (64: off; smooth embeddings at cfg2 size: filter 0.66 -> 0.56 ms)
#ifndef MANET_REFINE_RZ
#define MANET_REFINE_RZ 4
#endif
constexpr int REFINE_RZ = MANET_REFINE_RZ;  // workgroups of the re-rank launch per bucket: each takes every 4th entry (the longest bucket is the launch's time)
constexpr int ONE_ROUND_OK = 1 << 29;   // block_map flag (fp32 pipe kernel): the device may swap the host's splits for ONE round of long ones
constexpr int RESCUE_LISTED = 1 << 30;  // block_map flag of the rescue launch: deal the workgroups to the LISTED tiles
// block_map word: bits 0-7 tuning, bits 8-20 small_S (<= 4096: 13 bits) -- or, in the rescue launch, bits 8-28 its split count --
// bit 29 ONE_ROUND_OK, bit 30 RESCUE_LISTED.  (the rescue field was masked with 22 bits and reached bit 29.)
constexpr int BLOCK_MAP_SMALL_S_MASK = 0x1fff, BLOCK_MAP_RESCUE_MASK = 0x1fffff;
static_assert(((BLOCK_MAP_RESCUE_MASK << 8) & (ONE_ROUND_OK | RESCUE_LISTED)) == 0 && ((BLOCK_MAP_SMALL_S_MASK << 8) & (ONE_ROUND_OK | RESCUE_LISTED)) == 0 &&
              4096 <= BLOCK_MAP_SMALL_S_MASK, "block_map fields overlap");

// Packed operand image of one row block: `units` 16-byte units per row, stored [unit][row][16 B].
//   f32    unit u = 2g+h holds k = 8g + 2j + h, j = 0..3 (4 floats)      -> v_mfma_f32_32x32x2_f32
//          followed by the rows' squared norms (fp32)
//   bf16   unit u = 2s+h holds k = 16s + 8h + e, e = 0..7 (8 bf16)        -> v_mfma_f32_32x32x16_bf16
//          no norm block: the norms ride in the spare k slots C..C+5 of the image itself (below)
//   bf16x3 the bf16 image of hi = bf16(x), then the image of lo = bf16(x - hi)
//
// bf16 images: d(n,m) = |q|^2 + |k|^2 - 2 q.k comes out of the MFMA chain itself, accumulator
// starting at 0, no per-element epilogue arithmetic:
//   query rows hold  -2*bf16(q)  (exact: a power of two)          bank rows hold  bf16(k)
//   query k = C+0..2 : 1, 1, 1                                      bank k = C+0..2 : |k|^2 as hi+mid+lo
//   query k = C+3..5 : |q|^2 as hi+mid+lo (three bf16 = 24 bits)    bank k = C+3..5 : 1, 1, 1
// (three bf16 pieces carry an fp32 value exactly; the products with 1 are exact, the sum is fp32.)
constexpr int BF16_SPECIAL = 6;  // spare k slots needed behind the C channels
__host__ __device__ constexpr size_t bank_tile_bytes_u(int units, bool norms)
{
    return ((size_t)units * BT * 16 + (norms ? BT * 4 : 0) + 1023) / 1024 * 1024;  // whole 1 KiB LDS-DMA pieces
}
__host__ __device__ constexpr size_t query_block_bytes_u(int units, bool norms)
{
    return (size_t)units * QB * 16 + (norms ? QB * 4 : 0);
}
__host__ __device__ constexpr size_t bank_tile_bytes(int NG) { return bank_tile_bytes_u(2 * NG, true); }
__host__ __device__ constexpr size_t query_block_bytes(int NG) { return query_block_bytes_u(2 * NG, true); }

// number of v_mfma_f32_32x32x2 k-steps (2 k each) the f32 kernel is instantiated for
int pick_ks(int C)
{
    if (C <= 32) return 16;
    if (C <= 100) return 50;   // the reference's embedding width: 50 MFMAs, not 52
    if (C <= 104) return 52;
    return 64;
}
// number of v_mfma_f32_32x32x16_bf16 k-steps (16 k each) the bf16 kernels are instantiated for:
// C channels + BF16_SPECIAL norm slots
int pick_ksb(int C)
{
    if (C + BF16_SPECIAL <= 32) return 2;
    if (C + BF16_SPECIAL <= 112) return 7;   // C = 100: 7 MFMAs per 32x32 block
    return 9;
}
constexpr int QT_BF16 = 512;  // queries per workgroup of the bf16 kernels (8 waves x 64)

struct Geom {
    int compute;
    int steps;   // MFMA k-steps of the instantiation
    int units;   // 16-byte units per row of the packed image
    int kpad;    // k extent covered by the image
    int qt;      // queries per workgroup
    size_t tile_bytes, qblk_bytes;
};

Geom geom_of(int C, int compute)
{
    Geom G;
    if (compute == MANET_COMPUTE_BF16_REFINE) compute = MANET_COMPUTE_BF16;  // same operand images as plain bf16
    G.compute = compute;
    if (compute == MANET_COMPUTE_F32) {
        G.steps = pick_ks(C);
        int NG = (G.steps + 3) / 4;
        G.units = 2 * NG;
        G.kpad = 8 * NG;
        G.qt = QT;
    } else {
        G.steps = pick_ksb(C);
        G.units = 2 * G.steps * (compute == MANET_COMPUTE_BF16X3 ? 2 : 1);
        G.kpad = 16 * G.steps;
        // 8 waves: the bf16 MFMA eats a bank tile 14x faster, so share it between more queries
        G.qt = QT_BF16;
    }
    const bool norms = (compute == MANET_COMPUTE_F32);
    G.tile_bytes = bank_tile_bytes_u(G.units, norms);
    G.qblk_bytes = query_block_bytes_u(G.units, norms);
    return G;
}

struct BankLayout {
    Geom G;
    size_t tile_bytes;
    long T_max;  // upper bound on tiles: every object wastes < 1 tile
    long nblocks;  // pre-pass blocks of RPB rows
    size_t off_meta, off_hist, off_src, off_pack, total;
    // MANET_COMPUTE_BF16_REFINE only: the sorted rows once more in fp32, row-major [T_max * 64][C] + their |k|^2 (what the
    // exact re-rank reads), and the sub-sampled bank of the pre-pass (its own meta block + every REFINE_SUB-th tile)
    long T_sub_max;
    size_t off_rows, off_norms, off_sub_meta, off_sub_pack;
    Geom G32;  // ... and its fp32 operand image (the rescue pass of incomplete candidate buckets: the exact fp32 kernel)
    size_t off_pack32;
};

BankLayout bank_layout(int64_t M0, int C, int n_ids, int compute)
{
    BankLayout L;
    L.G = geom_of(C, compute);
    L.tile_bytes = L.G.tile_bytes;
    L.T_max = (long)((M0 + BT - 1) / BT) + n_ids;
    L.nblocks = (long)((M0 + 255) / 256);
    L.off_meta = 0;
    L.off_hist = manet_align_up(META_INTS * sizeof(int), 256);
    L.off_src = manet_align_up(L.off_hist + (size_t)(L.nblocks > 0 ? L.nblocks : 1) * n_ids * sizeof(int), 256);
    L.off_pack = manet_align_up(L.off_src + (size_t)L.T_max * BT * sizeof(int), 1024);
    L.total = manet_align_up(L.off_pack + (size_t)L.T_max * L.tile_bytes, 1024);
    L.T_sub_max = 0;
    L.off_rows = L.off_norms = L.off_sub_meta = L.off_sub_pack = L.off_pack32 = 0;
    L.G32 = L.G;
    if (compute == MANET_COMPUTE_BF16_REFINE) {
        L.T_sub_max = L.T_max / refine_sub(L.T_max) + n_ids + 1;
        L.off_rows = L.total;
        L.off_norms = manet_align_up(L.off_rows + (size_t)L.T_max * BT * C * sizeof(float), 256);
        L.off_sub_meta = manet_align_up(L.off_norms + (size_t)L.T_max * BT * sizeof(float), 256);
        L.off_sub_pack = manet_align_up(L.off_sub_meta + META_INTS * sizeof(int), 1024);
        L.G32 = geom_of(C, MANET_COMPUTE_F32);
        L.off_pack32 = manet_align_up(L.off_sub_pack + (size_t)L.T_sub_max * L.tile_bytes, 1024);
        L.total = manet_align_up(L.off_pack32 + (size_t)L.T_max * L.G32.tile_bytes, 1024);
    }
    return L;
}

struct MatchLayout {
    Geom G;
    long N_pad;
    int nQT;
    size_t qblk_bytes, off_q, off_keys, off_topk, total;
    // MANET_COMPUTE_BF16_REFINE: per (object, query) threshold and exact-distance key, the flat candidate list
    // {pair, bank slot} with its capacity, and two counters (candidates appended, list overflowed)
    size_t off_thr, off_slack, off_keys2, off_list, off_stats, off_bcnt, off_q32;
    long list_cap, bucket_cap;  // the candidate list = N_pad / 32 buckets (one per 32-query block) of bucket_cap entries
};

constexpr int TOPK_SPLITS = 16;  // the top-k path trades a little tail balance for a bounded workspace

MatchLayout match_layout(int64_t N, int C, int n_ids, int compute, int k_nn = 1, bool arg = false)
{
    MatchLayout L;
    L.G = geom_of(C, compute);
    L.nQT = (int)((N + L.G.qt - 1) / L.G.qt);
    L.N_pad = (long)L.nQT * L.G.qt;
    L.qblk_bytes = L.G.qblk_bytes;
    L.off_q = 0;
    L.off_keys = manet_align_up((size_t)(L.N_pad / QB) * L.qblk_bytes, 256);
    L.off_topk = manet_align_up(L.off_keys + (size_t)n_ids * L.N_pad * sizeof(unsigned), 1024);
    L.total = L.off_topk;
    if (k_nn > 1)
        L.total = manet_align_up(L.off_topk + (size_t)TOPK_SPLITS * n_ids * L.N_pad * MANET_MAX_KNN * sizeof(float), 1024);
    if (arg)  // 64-bit (distance key, bank slot) pairs of the arg-min form live where the top-k lists would
        L.total = manet_align_up(L.off_topk + (size_t)n_ids * L.N_pad * sizeof(unsigned long long), 1024);
    L.off_thr = L.off_slack = L.off_keys2 = L.off_list = L.off_stats = L.off_bcnt = L.off_q32 = 0;
    L.list_cap = L.bucket_cap = 0;
    if (compute == MANET_COMPUTE_BF16_REFINE) {
        const size_t pairs = (size_t)n_ids * L.N_pad;
        L.list_cap = (long)pairs * REFINE_CAP;
        L.off_thr = L.off_topk;
        L.off_slack = manet_align_up(L.off_thr + pairs * sizeof(float), 256);
        L.off_keys2 = manet_align_up(L.off_slack + pairs * sizeof(float), 256);
        L.off_list = manet_align_up(L.off_keys2 + pairs * sizeof(unsigned), 256);
        L.bucket_cap = (long)n_ids * QB * REFINE_CAP;
        L.off_stats = manet_align_up(L.off_list + (size_t)L.list_cap * sizeof(uint2), 256);
        L.off_bcnt = L.off_stats + 256;
        // (bcnt: [N_pad / 32] entries appended | bit 31 incomplete, then [N_pad / 32] dense entries appended)
        // then {number of 256-query tiles to rescue, their ids} (written by the re-rank launch, read by the rescue launch)
        L.off_q32 = manet_align_up(L.off_bcnt + ((size_t)2 * (L.N_pad / QB) + 1 + (size_t)(L.N_pad / QT)) * sizeof(unsigned), 1024);  // fp32 query image (rescue)
        L.total = manet_align_up(L.off_q32 + (size_t)(L.N_pad / QB) * geom_of(C, MANET_COMPUTE_F32).qblk_bytes, 1024);
    }
    return L;
}

// Number of bank splits S (grid = query tiles x S).  Every split re-reads the query operand
// (S x 4C.N bytes of fabric traffic, the bank itself streams through each XCD's L2 once), so S
// should be as small as the tail allows: take the smallest multiple of 8 whose last round of
// workgroups over the `slots` resident workgroup slots is >= 97 % full, with >= 4 tiles per split.
int pick_splits(int nQT, long T_max, int slots)
{
    // Workgroups are dispatched as slots free up; what a split count costs is its last, partial round.  With two
    // workgroups per CU (slots = 512) a last round that fills at most half the slots leaves its workgroups alone on
    // their CUs, where they run about twice as fast: it costs half a round (r2 sweep at cfg2, S = 40 / 48 / 56 / 64 / 80:
    // 4.704 / 4.672 / 4.836 / 4.778 / 4.714 ms -- 48 leaves 0.47 of a round, 40 leaves 0.89).  Every workgroup also
    // pays a fixed prologue (query operand, first tile), a share that grows with S.
    long cap = T_max / 4 / 8 * 8;
    if (cap < 8) cap = 8;
    if (cap > 256) cap = 256;
    const bool two_per_cu = slots >= 512;
    int best = 8;
    double best_eff = 0.0;
    for (int S = 8; S <= cap; S += 8) {
        const double rounds = (double)nQT * S / slots;
        const double whole = (double)(long)rounds, frac = rounds - whole;
        const double last = frac <= 0.0 ? 0.0 : ((two_per_cu && frac <= 0.5) ? 0.5 : 1.0);
        const double eff = rounds / (whole + last) / (1.0 + 0.0003 * S);
        if (eff > best_eff) {
            best_eff = eff;
            best = S;
        }
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// Bank pre-pass = a stable counting sort of the rows by object id, without global atomics
// (deterministic packing order):
//   label_hist_kernel      per block of 256 rows: rows per object            -> hist[block][o]
//   label_scan_kernel      one wave per object: exclusive prefix over blocks -> base[block][o], cnt[o]
//   label_segments_kernel  tile range of every object (rows padded to whole 64-row tiles)
//   label_scatter_kernel   slot of row i = seg_start[o]*64 + base[block][o] + rank inside the block
//   pack_rows_kernel       gather + transpose the rows into the MFMA operand image
// A row counts for object o iff label == o (IntVOS.py:137); other labels (-1 = unlabelled) are
// dropped, which is what _selected_pixel (:100-109) / the 1e20 mask (:81-83) amount to for a minimum.
constexpr int RPB = 256;  // rows per pre-pass block

__global__ __launch_bounds__(RPB) void label_hist_kernel(const int *__restrict__ labels, long M0, int n_ids,
                                                         int *__restrict__ hist)
{
    __shared__ int h[MANET_MAX_IDS];
    if (threadIdx.x < MANET_MAX_IDS) h[threadIdx.x] = 0;
    __syncthreads();
    long i = (long)blockIdx.x * RPB + threadIdx.x;
    if (i < M0) {
        int lab = labels[i];
        if (lab >= 0 && lab < n_ids) atomicAdd(&h[lab], 1);  // LDS atomic
    }
    __syncthreads();
    if (threadIdx.x < n_ids) hist[(long)blockIdx.x * n_ids + threadIdx.x] = h[threadIdx.x];
}

// grid = n_ids blocks of one wave: exclusive prefix of hist[:, o] over the blocks, in place
__global__ __launch_bounds__(64) void label_scan_kernel(int *__restrict__ hist, int nblocks, int n_ids,
                                                        int *__restrict__ meta)
{
    const int o = blockIdx.x, lane = threadIdx.x;
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 64) {
        int b = b0 + lane;
        int v = (b < nblocks) ? hist[(long)b * n_ids + o] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (b < nblocks) hist[(long)b * n_ids + o] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
    if (lane == 0) meta[META_CNT + o] = carry;
}

__global__ void label_segments_kernel(int n_ids, int *meta)
{
    if (threadIdx.x == 0) {
        int t = 0;
        for (int o = 0; o < n_ids; ++o) {
            meta[META_SEG + o] = t;
            t += (meta[META_CNT + o] + BT - 1) / BT;
        }
        meta[META_SEG + n_ids] = t;
        meta[META_T] = t;
    }
}

// slot -> source row map (slots not hit stay -1 = padding row)
__global__ __launch_bounds__(RPB) void label_scatter_kernel(const int *__restrict__ labels, long M0, int n_ids,
                                                            const int *__restrict__ base,
                                                            const int *__restrict__ meta,
                                                            int *__restrict__ src_of)
{
    __shared__ int wcnt[RPB / 64][MANET_MAX_IDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < (RPB / 64) * MANET_MAX_IDS; j += RPB) (&wcnt[0][0])[j] = 0;
    __syncthreads();
    const long i = (long)blockIdx.x * RPB + threadIdx.x;
    const int lab = (i < M0) ? labels[i] : -1;
    const bool valid = (lab >= 0 && lab < n_ids);
    bool active = valid;
    int rank = 0;
    while (true) {  // ranks inside the wave from ballots, one pass per object present in the wave
        unsigned long long pending = __ballot(active);
        if (!pending) break;
        int leader = __ffsll((long long)pending) - 1;
        int L = __shfl(lab, leader);
        bool mine = active && (lab == L);
        unsigned long long mm = __ballot(mine);
        if (mine) {
            rank = __popcll(mm & ((1ull << lane) - 1ull));
            active = false;
        }
        if (lane == leader) wcnt[wave][L] = __popcll(mm);
    }
    __syncthreads();
    if (valid) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += wcnt[w][lab];
        int slot = meta[META_SEG + lab] * BT + base[(long)blockIdx.x * n_ids + lab] + before + rank;
        src_of[slot] = (int)i;
    }
}

// fp32 -> bf16, round to nearest even (NaN stays quiet NaN); same bits as the oracle's bf16_round
__device__ __forceinline__ unsigned f2bf(float x)
{
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf2f(unsigned b) { return __uint_as_float(b << 16); }

// an fp32 value as three bf16 pieces, p0 + p1 + p2 == x exactly (3 x 8 significand bits)
__device__ __forceinline__ void split3_bf16(float x, unsigned (&p)[3])
{
    p[0] = f2bf(x);
    float r = x - bf2f(p[0]);
    p[1] = f2bf(r);
    r = r - bf2f(p[1]);
    p[2] = f2bf(r);
}

// embedding element types the pack kernels read (the producer's storage):
// float, or bf16 as raw 16-bit words
__device__ __forceinline__ float emb_load(const float *p, long i) { return p[i]; }
__device__ __forceinline__ float emb_load(const unsigned short *p, long i) { return bf2f(p[i]); }

// One 16-byte unit of a row's operand image (see Geom): `row` = the row's kpad staged values.
__device__ __forceinline__ f32x4 image_unit_f32(const float *row, int u)
{
    const float *p = row + 8 * (u >> 1) + (u & 1);
    return f32x4{p[0], p[2], p[4], p[6]};
}
template <bool IS_QUERY>
__device__ __forceinline__ uint4 image_unit_bf16(const float *row, int u, int hi_units, int C, float norm)
{
    const float scale = IS_QUERY ? -2.0f : 1.0f;  // the query operand is -2q (exact in bf16)
    const bool lo = u >= hi_units;
    const int uu = lo ? u - hi_units : u;
    const int k0 = 16 * (uu >> 1) + 8 * (uu & 1);
    unsigned e8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = row[k0 + e];
        unsigned b = f2bf(x);
        if (lo) b = f2bf(x - bf2f(b));
        e8[e] = f2bf(scale * bf2f(b));
    }
    if (k0 + 8 > C && k0 < C + BF16_SPECIAL) {  // this unit holds norm slots (see Geom)
        unsigned piece[3];
        split3_bf16(norm, piece);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = k0 + e - C;  // 0..2: bank norm / query ones, 3..5: bank ones / query norm
            if (j >= 0 && j < BF16_SPECIAL) {
                const bool norm_slot = IS_QUERY ? (j >= 3) : (j < 3);
                // (a select chain, not piece[j % 3]: a run-time index keeps the three words in a private-memory stack object --
                // 36 bytes of scratch in frame_prepare_kernel<unsigned short, 32> through r5, tests/test_kernel_resources.py)
                const int j3 = j % 3;
                const unsigned pj = j3 == 0 ? piece[0] : (j3 == 1 ? piece[1] : piece[2]);
                e8[e] = lo ? 0u : (norm_slot ? pj : 0x3f80u);
            }
        }
    }
    return make_uint4(e8[0] | (e8[1] << 16), e8[2] | (e8[3] << 16), e8[4] | (e8[5] << 16), e8[6] | (e8[7] << 16));
}

// bank (ROWS = 64) and query (ROWS = 32) pack: rows -> MFMA operand image (see Geom).
// Rows are staged through LDS so that both the global reads (along k for row-major sources, along
// rows for C-major sources) and the 16-byte image writes are coalesced.  |row|^2 is the k-ascending
// fmaf chain of the oracle (IntVOS.py:32,35) -- over the bf16-rounded values in MANET_COMPUTE_BF16
// (the path then IS the reference formula on rounded embeddings), over the fp32 values otherwise.
// f32 images carry the norms in a trailing block; bf16 images carry them in the spare k slots (Geom).
// ROWS = rows staged per workgroup (a whole number of image blocks); IMG = rows per image block (64: bank tile,
// 32: query block).  keys != nullptr (query):
// the rows' match keys are reset to "no candidate" here, which saves the fill launch of the per-frame sequence.
template <int ROWS, int IMG, typename SRC>
__global__ __launch_bounds__(256) void pack_rows_kernel(const SRC *__restrict__ src, long s_row,
                                                        long s_c, const int *__restrict__ src_of,
                                                        const int *__restrict__ meta, long n_rows,
                                                        int C, int compute, int units, int kpad,
                                                        char *__restrict__ dst, long tile_bytes,
                                                        float pad_norm, unsigned *__restrict__ keys, long N_pad,
                                                        int n_ids)
{
    constexpr bool IS_QUERY = (IMG == QB);
    const long tile = blockIdx.x;
    if (meta && tile >= meta[META_T]) return;
    if (keys)
        for (int i = threadIdx.x; i < ROWS * n_ids; i += 256) keys[(size_t)(i / ROWS) * N_pad + tile * ROWS + (i % ROWS)] = 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) char pack_smem[];
    const int KP = kpad + 1;  // odd row stride: column reads are conflict-free
    float *rows = (float *)pack_smem;                  // [ROWS][KP]
    int *s_src = (int *)(rows + (long)ROWS * KP);      // [ROWS]
    float *s_norm = (float *)(s_src + ROWS);           // [ROWS]
    const int tid = threadIdx.x;
    if (tid < ROWS) {
        long slot = tile * ROWS + tid;
        s_src[tid] = src_of ? src_of[slot] : (slot < n_rows ? (int)slot : -1);
    }
    __syncthreads();
    if (s_c == 1) {  // row-major source: lanes along k
        for (int idx = tid; idx < ROWS * C; idx += 256) {
            int r = idx / C, k = idx - r * C;
            int sr = s_src[r];
            rows[r * KP + k] = (sr >= 0) ? emb_load(src, (long)sr * s_row + k) : 0.0f;
        }
    } else {  // C-major (or generic) source: lanes along rows
        for (int idx = tid; idx < ROWS * C; idx += 256) {
            int k = idx / ROWS, r = idx - k * ROWS;
            int sr = s_src[r];
            rows[r * KP + k] = (sr >= 0) ? emb_load(src, (long)sr * s_row + (long)k * s_c) : 0.0f;
        }
    }
    for (int idx = tid; idx < ROWS * (kpad - C); idx += 256) {
        int r = idx / (kpad - C), k = C + idx - r * (kpad - C);
        rows[r * KP + k] = 0.0f;
    }
    __syncthreads();
    if (tid < ROWS) {
        float n = pad_norm;
        if (s_src[tid] >= 0) {
            n = 0.0f;
            const float *row = rows + tid * KP;
            if (compute == MANET_COMPUTE_BF16) {
                for (int k = 0; k < C; ++k) {
                    float x = bf2f(f2bf(row[k]));
                    n = fmaf(x, x, n);
                }
            } else {
                for (int k = 0; k < C; ++k) n = fmaf(row[k], row[k], n);
            }
        }
        s_norm[tid] = n;
    }
    __syncthreads();
    // image block of staged row r: block (tile * ROWS/IMG + r / IMG), row r % IMG inside it
    char *out0 = dst + tile * (ROWS / IMG) * tile_bytes;
    if (compute == MANET_COMPUTE_F32) {
        for (int item = tid; item < units * ROWS; item += 256) {
            int r = item % ROWS, u = item / ROWS;
            *(f32x4 *)(out0 + (r / IMG) * tile_bytes + ((long)u * IMG + r % IMG) * 16) = image_unit_f32(rows + r * KP, u);
        }
        if (tid < ROWS) *(float *)(out0 + (tid / IMG) * tile_bytes + (long)units * IMG * 16 + (tid % IMG) * 4) = s_norm[tid];
    } else {
        const int hi_units = (compute == MANET_COMPUTE_BF16X3) ? units / 2 : units;
        for (int item = tid; item < units * ROWS; item += 256) {
            int r = item % ROWS, u = item / ROWS;
            *(uint4 *)(out0 + (r / IMG) * tile_bytes + ((long)u * IMG + r % IMG) * 16) =
                image_unit_bf16<IS_QUERY>(rows + r * KP, u, hi_units, C, s_norm[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-frame prepare (the producer side of the path): ONE read of a frame's C-major embedding
// writes BOTH per-frame operands of the propagation step --
//   * the query operand image of the global match (what pack_rows_kernel<32,32> writes), and
//   * the 2x2-average-pooled plane of the local match, padded with the reference's 1e20 (IntVOS.py:287), plus the
//     fused local kernel's tile table (what lf_pool_pad_kernel writes; r2 read both full-resolution frames again for
//     it, every frame: the stage moved 3x its algorithmic bytes) --
// so a propagated frame reads its embedding from HBM once, and the previous frame's not at all (its plane was made when
// it was the current frame).  Workgroup = one full-resolution row pair x XC columns, all channels, staged in LDS as
// [pixel][k] (odd stride: the row-wise and the k-wise accesses are both conflict-free); grid.z = frame of the batch.
// Blocks behind the data blocks fill the plane's top / bottom border rows, zero the image's padding rows, write the
// tile table and serve one optional caller fill (the local match's `out` pre-set, IntVOS.py:429-430's 1.0).
struct FramePrep {
    const void *emb;
    long s_f, s_y, s_x, s_c;
    int h, w, C, compute, units, kpad;
    char *ws;
    long ws_stride, qblk_bytes, off_plane, off_tab;
    int d, hp, wp, HPAD, WS;
    long PS;
    int TY, TX, nty, ntx;
    long N, N_pad;
    unsigned *fill_ptr;
    long fill_words;
    unsigned fill_value;
    int n_data, nxc;
    // manet_embed_finish: the embedding layer's epilogue in front of the staging -- y = relu(x * scale[c] + shift[c]), rounded to
    // the embedding's storage type, written to emb_out [frame][C][h][w] -- `emb` then is the 1x1 convolution's raw fp32 output
    const float *scale, *shift;
    void *emb_out;
    int emb_out_bf16, relu;
    int vec2;   // s_x == 1, even w / strides, aligned base: two pixels per load
    int rcopy;  // fp32 source + MANET_COMPUTE_BF16: LDS also holds a bf16-rounded copy
    int abl;  // -DMANET_ABLATION builds, timing experiments: 1 no loads, 2 no plane stores, 4 no image stores, 8 no norm chain, 16 no data blocks, 32 no aux blocks
};
#ifdef MANET_ABLATION
#define MANET_FP_ABL(bit_) (A.abl & (bit_))
#else
#define MANET_FP_ABL(bit_) false
#endif
// VEC2 (r6): the two staging forms are separate instantiations -- as a run-time branch (r3-r5) both lived in every kernel and the
// 2-byte instantiation ran out of scalar registers (22 SGPR spills and a 36-byte private segment; tests/test_kernel_resources.py)
template <typename SRC, int XC, bool VEC2>
__global__ __launch_bounds__(256) void frame_prepare_kernel(const FramePrep A)
{
    constexpr int PIX = 2 * XC;
    extern __shared__ __attribute__((aligned(16))) char pack_smem[];
    const int tid = threadIdx.x;
    const SRC *src = (const SRC *)A.emb + (long)blockIdx.z * A.s_f;
    char *ws = A.ws + (long)blockIdx.z * A.ws_stride;
    float *plane = (float *)(ws + A.off_plane);
    const int C = A.C, kpad = A.kpad, units = A.units;
    if (MANET_FP_ABL(16) && (int)blockIdx.x < A.n_data) return;
    if (MANET_FP_ABL(32) && (int)blockIdx.x >= A.n_data) return;
    if ((int)blockIdx.x >= A.n_data) {  // ---- auxiliary blocks
        const long gid = (long)(blockIdx.x - A.n_data) * 256 + tid, gstride = (long)(gridDim.x - A.n_data) * 256;
        if (A.d >= 0) {
            const int rows_b = A.HPAD - A.hp, WS4 = A.WS / 4;
            const long items = (long)C * rows_b * WS4;
            const f32x4 pad = {MANET_WRONG_LABEL_PADDING_DISTANCE, MANET_WRONG_LABEL_PADDING_DISTANCE,
                               MANET_WRONG_LABEL_PADDING_DISTANCE, MANET_WRONG_LABEL_PADDING_DISTANCE};
            for (long i = gid; i < items; i += gstride) {
                const int c = (int)(i / ((long)rows_b * WS4));
                const int rem = (int)(i - (long)c * rows_b * WS4);
                const int rb = rem / WS4, q = rem - rb * WS4;
                const int r = rb < A.d ? rb : A.hp + rb;  // rows [0, d) and [d + hp, HPAD)
                *(f32x4 *)(plane + (long)c * A.PS + (long)r * A.WS + 4 * q) = pad;
            }
            {  // ... and the left / right border columns of the data rows [d, d + hp): columns [0, d) and [d + wp, WS)
                const int nb = A.WS - A.wp;  // border floats per row
                const long items2 = (long)C * A.hp * nb;
                for (long i = gid; i < items2; i += gstride) {
                    const int c = (int)(i / ((long)A.hp * nb));
                    const int rem = (int)(i - (long)c * A.hp * nb);
                    const int r = rem / nb, j = rem - r * nb;
                    plane[(long)c * A.PS + (long)(A.d + r) * A.WS + (j < A.d ? j : A.wp + j)] = MANET_WRONG_LABEL_PADDING_DISTANCE;
                }
            }
            int *tab = (int *)(ws + A.off_tab);
            for (long i = gid; i <= A.nty + 1 + A.ntx; i += gstride)
                tab[i] = i <= A.nty ? bilin_first((int)i * A.TY, A.hp, A.h) : bilin_first((int)(i - A.nty - 1) * A.TX, A.wp, A.w);
        }
        {  // rows N .. N_pad of the image: zero operands (their results are never read)
            const long tail = A.N_pad - A.N;
            for (long i = gid; i < tail * units; i += gstride) {
                const long n = A.N + i % tail;
                const int u = (int)(i / tail);
                *(uint4 *)(ws + (n >> 5) * A.qblk_bytes + ((long)u * QB + (n & 31)) * 16) = make_uint4(0, 0, 0, 0);
            }
            if (A.compute == MANET_COMPUTE_F32)
                for (long i = gid; i < tail; i += gstride) {
                    const long n = A.N + i;
                    *(float *)(ws + (n >> 5) * A.qblk_bytes + (long)units * QB * 16 + (n & 31) * 4) = 0.0f;
                }
        }
        if (blockIdx.z == 0)
            for (long i = gid; i < A.fill_words; i += gstride) A.fill_ptr[i] = A.fill_value;
        return;
    }
    // ---- data blocks
    // LDS: rows [PIX][KP] = the embedding as stored (what the pooled plane, the f32 and the split-bf16 image are made from);
    // rq = the values the bf16 image and its |q|^2 are made from: bf16-rounded.  2-byte sources ARE rounded already (rq = rows);
    // fp32 sources with plain-bf16 arithmetic get a second, rounded copy (A.rcopy) so that neither the norm chain nor the image
    // assembly rounds per use (r3: 5 us of norm chain and 7 us of image assembly in a 20 us launch).
    const int KP = kpad + 1;
    float *rows = (float *)pack_smem;           // [PIX][KP]
    float *rq = A.rcopy ? rows + (long)PIX * KP : rows;
    const int rp = blockIdx.x / A.nxc, cx = blockIdx.x - rp * A.nxc;
    const int x0 = cx * XC, y0 = 2 * rp;
    // rq holds bf16-exact values (a 2-byte source, the rounded copy, or the embedding epilogue's 2-byte output)
    const bool bf16_exact = (A.compute == MANET_COMPUTE_BF16) && (A.rcopy || sizeof(SRC) == 2 || (A.scale && A.emb_out_bf16));
    // The launch is LATENCY-bound, not bandwidth-bound (1.6 workgroups per CU, 27 MB per frame; ablations in DESIGN 3.3): every
    // load of the workgroup is issued before the first one is waited for -- one memory round trip per workgroup.
    if constexpr (VEC2) {  // two horizontally adjacent pixels per lane (8-byte / 4-byte loads): half the load instructions
        constexpr int NP2 = PIX / 2, NKQ = 256 / NP2;
        const int pp = tid % NP2, kq = tid / NP2;
        const int p = 2 * pp, y = y0 + p / XC, x = x0 + p % XC;  // (XC is even: both pixels in one row; w is even)
        const bool in = (y < A.h && x < A.w);
        const SRC *sp = src + (long)(y < A.h ? y : A.h - 1) * A.s_y + (long)(x < A.w ? x : A.w - 2);
        long sc_ = A.s_c;
        if (MANET_FP_ABL(1)) { sp = src; sc_ = 0; }  // (timing ablation: every load hits one cached line)
        float *r0 = rows + p * KP, *q0 = rq + p * KP;
        auto stage2 = [&](auto kb_tag) __attribute__((always_inline)) {
            constexpr int KB = decltype(kb_tag)::value;
            for (int k0 = kq; k0 < C; k0 += NKQ * KB) {
                float va[KB], vb[KB];
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int k = k0 + j * NKQ;
                    const SRC *a = sp + (long)(k < C ? k : C - 1) * sc_;
                    if (sizeof(SRC) == 4) {
                        const float2 t = *(const float2 *)a;
                        va[j] = t.x; vb[j] = t.y;
                    } else {
                        const unsigned t = *(const unsigned *)a;
                        va[j] = bf2f(t & 0xffffu); vb[j] = bf2f(t >> 16);
                    }
                }
                if (sizeof(SRC) == 4 && A.scale) {  // the embedding layer's epilogue (block-uniform)
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        const int k = k0 + j * NKQ, kc = k < C ? k : C - 1;
                        const float sc = A.scale[kc], sh = A.shift[kc];
                        float a = fmaf(va[j], sc, sh), b = fmaf(vb[j], sc, sh);
                        if (A.relu) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
                        const long eo = (((long)blockIdx.z * C + kc) * A.h + (y < A.h ? y : A.h - 1)) * A.w + (x < A.w ? x : A.w - 2);
                        if (A.emb_out_bf16) {
                            const unsigned ba = f2bf(a), bb = f2bf(b);
                            a = bf2f(ba); b = bf2f(bb);  // the operands are made from the embedding AS STORED
                            if (in && k < C) *(unsigned *)((unsigned short *)A.emb_out + eo) = ba | (bb << 16);
                        } else if (in && k < C) {
                            *(float2 *)((float *)A.emb_out + eo) = float2{a, b};
                        }
                        va[j] = a; vb[j] = b;
                    }
                }
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int k = k0 + j * NKQ;
                    if (k < C) {
                        const float a = in ? va[j] : 0.0f, b = in ? vb[j] : 0.0f;
                        r0[k] = a; r0[KP + k] = b;
                        if (A.rcopy) { q0[k] = bf2f(f2bf(a)); q0[KP + k] = bf2f(f2bf(b)); }
                    }
                }
            }
        };
        if (C <= 13 * NKQ) stage2(std::integral_constant<int, 13>{});
        else stage2(std::integral_constant<int, 16>{});
    } else {  // generic strides: one pixel per lane, lanes along x
        constexpr int NKQ = 256 / PIX;
        const int p = tid % PIX, kq = tid / PIX;
        const int y = y0 + p / XC, x = x0 + p % XC;
        const bool in = (y < A.h && x < A.w);
        const SRC *sp = src + (long)(y < A.h ? y : A.h - 1) * A.s_y + (long)(x < A.w ? x : A.w - 1) * A.s_x;
        long sc_ = A.s_c;
        if (MANET_FP_ABL(1)) { sp = src; sc_ = 0; }
        float *rp_ = rows + p * KP, *qp_ = rq + p * KP;
        auto stage = [&](auto kb_tag) __attribute__((always_inline)) {
            constexpr int KB = decltype(kb_tag)::value;
            for (int k0 = kq; k0 < C; k0 += NKQ * KB) {
                float v[KB];
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int k = k0 + j * NKQ;
                    v[j] = emb_load(sp, (long)(k < C ? k : C - 1) * sc_);
                }
                if (sizeof(SRC) == 4 && A.scale) {  // the embedding layer's epilogue (block-uniform)
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        const int k = k0 + j * NKQ, kc = k < C ? k : C - 1;
                        float a = fmaf(v[j], A.scale[kc], A.shift[kc]);
                        if (A.relu) a = fmaxf(a, 0.0f);
                        const long eo = (((long)blockIdx.z * C + kc) * A.h + (y < A.h ? y : A.h - 1)) * A.w + (x < A.w ? x : A.w - 1);
                        if (A.emb_out_bf16) {
                            const unsigned ba = f2bf(a);
                            a = bf2f(ba);
                            if (in && k < C) ((unsigned short *)A.emb_out)[eo] = (unsigned short)ba;
                        } else if (in && k < C) {
                            ((float *)A.emb_out)[eo] = a;
                        }
                        v[j] = a;
                    }
                }
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int k = k0 + j * NKQ;
                    if (k < C) {
                        const float a = in ? v[j] : 0.0f;
                        rp_[k] = a;
                        if (A.rcopy) qp_[k] = bf2f(f2bf(a));
                    }
                }
            }
        };
        // (the whole channel range in one batch: C <= 100 -> 25 x NKQ, C <= 128 -> 32 x NKQ channels in flight per thread)
        if (C <= 25 * NKQ) stage(std::integral_constant<int, 25>{});
        else stage(std::integral_constant<int, 32>{});
    }
    for (int idx = tid; idx < PIX * (kpad - C); idx += 256) {
        const int p = idx / (kpad - C), k = C + idx - p * (kpad - C);
        rows[p * KP + k] = 0.0f;
        if (A.rcopy) rq[p * KP + k] = 0.0f;
    }
    __syncthreads();
    // ---- roles (no further barrier): wave 0 walks the |q|^2 chain -- 100 dependent fmaf, 1.5 us -- and then writes what needs
    // it (the f32 image's norm block / the bf16 image's units with norm slots) and the plane's border columns; waves 1..3 write
    // the pooled plane and the image units that do not depend on the norm while that chain runs.
    const bool f32img = (A.compute == MANET_COMPUTE_F32);
    const int hi_units = (A.compute == MANET_COMPUTE_BF16X3) ? units / 2 : units;
    // bf16 images: unit u carries norm slots iff its k range reaches past C (both halves of the split image are handled alike)
    auto unit_is_special = [&](int u) { const int uu = u >= hi_units ? u - hi_units : u; return 16 * (uu >> 1) + 8 * (uu & 1) + 8 > C; };
    auto image_addr = [&](long n, int u) { return ws + (n >> 5) * A.qblk_bytes + ((long)u * QB + (n & 31)) * 16; };
    // the bf16 image unit from bf16-EXACT staged values: -2 x is exact, its upper 16 bits are the bf16 (no rounding, no NaN fix-up:
    // identical bits to f2bf(-2 * bf2f(f2bf(x))))
    auto unit_bf16_exact = [&](const float *row, int u) {
        const int k0 = 16 * (u >> 1) + 8 * (u & 1);
        unsigned e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = __float_as_uint(-2.0f * row[k0 + e]) >> 16;
        return make_uint4(e8[0] | (e8[1] << 16), e8[2] | (e8[3] << 16), e8[4] | (e8[5] << 16), e8[6] | (e8[7] << 16));
    };
    // (f32 image: the last units / 6 units of every pixel are wave 0's as well -- its share of the store work behind the chain)
    const int u_split = f32img ? units - units / 6 : units;
    if (tid < PIX) {  // ---- wave 0 (PIX = 64 lanes; XC = 64 builds: two waves)
        const int p = tid;
        const int y = y0 + p / XC, x = x0 + p % XC;
        const bool in = (y < A.h && x < A.w);
        const long n = (long)y * A.w + x;
        float nrm = 0.0f;
        if (!MANET_FP_ABL(8)) {  // |q|^2: the k-ascending fmaf chain of the oracle, as pack_rows_kernel (reads batched ahead)
            const float *row = (A.compute == MANET_COMPUTE_BF16 ? rq : rows) + p * KP;
            const bool rnd = (A.compute == MANET_COMPUTE_BF16) && !bf16_exact;
            int k = 0;
            for (; k + 10 <= C; k += 10) {
                float v[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) v[j] = row[k + j];
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const float xv = rnd ? bf2f(f2bf(v[j])) : v[j];
                    nrm = fmaf(xv, xv, nrm);
                }
            }
            for (; k < C; ++k) {
                const float xv = rnd ? bf2f(f2bf(row[k])) : row[k];
                nrm = fmaf(xv, xv, nrm);
            }
        }
        if (in && !MANET_FP_ABL(4)) {
            if (f32img) {
                *(float *)(ws + (n >> 5) * A.qblk_bytes + (long)units * QB * 16 + (n & 31) * 4) = nrm;
                for (int u = u_split; u < units; ++u) *(f32x4 *)image_addr(n, u) = image_unit_f32(rows + p * KP, u);
            } else {
                for (int u = 0; u < units; ++u)
                    if (unit_is_special(u)) *(uint4 *)image_addr(n, u) = image_unit_bf16<true>(rows + p * KP, u, hi_units, C, nrm);
            }
        }
    }
    if (tid >= PIX) {  // ---- the other waves
        const int t = tid - PIX, NT = 256 - PIX;
        // pooled plane row d + rp (IntVOS.py:282-284: window summed row-major, times 1/4)
        if (A.d >= 0 && rp < A.hp && !MANET_FP_ABL(2)) {
            float *prow = plane + (long)(A.d + rp) * A.WS + A.d + x0 / 2;
            for (int idx = t; idx < (XC / 2) * C; idx += NT) {
                const int c = idx / (XC / 2), px = idx - c * (XC / 2);
                if (x0 / 2 + px < A.wp) {
                    const float *q = rows + (2 * px) * KP + c;
                    prow[(long)c * A.PS + px] = (((q[0] + q[KP]) + q[XC * KP]) + q[(XC + 1) * KP]) * 0.25f;
                }
            }
        }
        // operand image: pixel (y, x) is query row n = y w + x -> block n / 32, row n % 32
        if (!MANET_FP_ABL(4)) {
            for (int item = t; item < u_split * PIX; item += NT) {
                const int p = item % PIX, u = item / PIX;
                const int y = y0 + p / XC, x = x0 + p % XC;
                if (y >= A.h || x >= A.w) continue;
                const long n = (long)y * A.w + x;
                if (f32img) *(f32x4 *)image_addr(n, u) = image_unit_f32(rows + p * KP, u);
                else if (unit_is_special(u)) continue;  // (wave 0, behind the norm chain)
                else if (bf16_exact) *(uint4 *)image_addr(n, u) = unit_bf16_exact(rq + p * KP, u);
                else *(uint4 *)image_addr(n, u) = image_unit_bf16<true>(rows + p * KP, u, hi_units, C, 0.0f);
            }
        }
    }
}

// Workspace initialisation as a plain kernel.  (hipMemsetAsync nodes were observed to replay with the
// wrong fill value from the second replay of a captured HIP graph on ROCm 7.2; a kernel node has no
// such problem, and callers may capture a frame's launch sequence.)
__global__ void fill32_kernel(unsigned *__restrict__ p, unsigned value, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = value;
}
void fill32(void *p, unsigned value, size_t words, hipStream_t st)
{
    if (!words) return;
    unsigned blocks = (unsigned)((words + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill32_kernel, dim3(blocks), dim3(256), 0, st, (unsigned *)p, value, (long)words);
}

// order-preserving float -> uint key (so atomicMin on keys == min on floats, negatives included:
// d may be slightly negative from rounding and must not be clamped)
// NaN -> key 0: it wins every atomicMin and float_of(0) is a NaN again, so a NaN distance propagates
// to the output like torch.min does (IntVOS.py:84).
__device__ __forceinline__ unsigned key_of(float f)
{
    unsigned u = __float_as_uint(f);
    if (f != f) return 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of(unsigned k)
{
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// NaN-propagating minimum (v_minimum3_f32 on gfx950): torch.min semantics, and -- unlike fminf on an MFMA
// result -- needs no canonicalising v_max in front of it.
__device__ __forceinline__ float min3p(float m, float a, float b)
{
    return __builtin_elementwise_minimum(__builtin_elementwise_minimum(m, a), b);
}

// block -> (query tile, bank split, tile range [t0, t1)); false = nothing to do.
// Big banks: S splits chosen on the host (pick_splits), split index tied to the XCD (block b runs on XCD b % 8 --
// observed, used for speed only): the blocks of one XCD that are resident together stream the same bank range
// through that XCD's L2.
// Small banks (T < S tiles: a scribble-sized memory, the reference's normal case, test.py:170-176): the host cannot know
// T without a sync (it only knows the upper bound M0 / 64), and S splits of at most one tile each would make every
// workgroup load its 256-query operand for a single tile.  The kernel re-decides: S' = clamp(small_S, 1, T) splits with
// small_S = resident workgroup slots / query tiles -- one full round of workgroups, whole tiles each -- and a flat
// block -> (tile, split) map (such a bank fits every XCD's L2 anyway).  The minimum is order-independent: same bits.
// `block_map`: bits 0-7 = tuning (0: XCD-aware, 1: tile fastest, 2: split fastest, 4..7: XCD-aware with 2..5 splits
// fastest), bits 8.. = small_S.
#ifndef MANET_FILTER_TAIL_CUTS
#define MANET_FILTER_TAIL_CUTS 4
#endif
constexpr int FILTER_TAIL_CUTS = MANET_FILTER_TAIL_CUTS;  // (1: bm == 3's map)
__device__ __forceinline__ bool split_of_block(int b, int nQT, int S, int T, int block_map, int &qt, int &s, int &t0,
                                               int &t1)
{
    const int bm = block_map & 0xff, small_S = (block_map >> 8) & BLOCK_MAP_SMALL_S_MASK;
    bool one_round = T < S && small_S > 0;
    if (!one_round && (block_map & ONE_ROUND_OK) && small_S > 0 && small_S < S) {
        // r5.  The host sizes S from the bank's UPPER-bound tile count (it cannot know how many rows are labelled without a
        // sync) for many rounds of short splits; a workgroup's fixed cost (query operand, pipeline fill, closing atomics) is
        // worth ~0.6 tile-times after the overlap of two workgroups per CU.  With the REAL tile count: cost of the host's
        // choice = its rounds over the 512 slots (a last round at most half full counts half, as in pick_splits) x (tiles per
        // split + 0.6) against ONE round of small_S = floor(512 / query tiles) long splits.  The reference driver's first-round
        // bank (rough_ROI: ~17 000 rows = 268 tiles at 480p) ran 48 splits of 5.6 tiles: 673 us; one round of 5 x 54: 633.
        // Banks above ~600 tiles keep the host's choice (a tie there, and the XCD-aware map).  Same bits either way.
        const float rounds = (float)nQT * (float)S / 512.0f;
        const float whole = floorf(rounds), frac = rounds - whole;
        const float last = frac <= 0.0f ? 0.0f : (frac <= 0.5f ? 0.5f : 1.0f);
        const float cost_host = (whole + last) * ((float)T / (float)S + 0.6f);
        const float cost_one = ceilf((float)T / (float)small_S) + 0.6f;
        one_round = cost_one < 0.97f * cost_host;
    }
    if (one_round) {
        const int S2 = small_S < T ? small_S : T;
        qt = b % nQT;
        s = b / nQT;
        if (s >= S2) return false;
        t0 = (int)((long)s * T / S2);
        t1 = (int)((long)(s + 1) * T / S2);
        return t0 < t1;
    }
    if (bm == 0 || bm == 3) {
        const int xcd = b & 7;
        const int idx = b >> 3;
        qt = idx % nQT;
        s = xcd + 8 * (idx / nQT);
        // bm == 3 (the FILTER pass of MANET_COMPUTE_BF16_REFINE; S is a multiple of 8): XCD x walks the CONTIGUOUS splits
        // x S/8 .. (x+1) S/8 - 1 one after the other instead of x, x + 8, ...: at any time the eight resident splits are
        // spread over the whole bank, i.e. over all objects, so most of an object's splits start after earlier ones of the
        // same object have published their tightened thresholds
        if (bm == 3 && (S & 7) == 0) s = xcd * (S >> 3) + idx / nQT;
    } else if (bm >= 4 && bm <= 7 && (S & 7) == 0) {
        // XCD-aware with PB splits fastest: the workgroups of an XCD that are resident together cover 64 / PB query tiles x
        // PB of the XCD's splits -- a query operand is loaded by PB workgroups at about the same time (one L2 miss, PB - 1
        // hits) at the price of PB splits streaming through the L2 side by side
        const int PB = bm - 2, S8 = S >> 3;
        const int xcd = b & 7, idx = b >> 3;
        const int g = idx / (nQT * PB), r = idx - g * (nQT * PB);
        const int pbg = (S8 - g * PB) < PB ? (S8 - g * PB) : PB;
        if (pbg <= 0) return false;
        qt = r / pbg;
        s = xcd + 8 * (g * PB + (r - qt * pbg));
        if (qt >= nQT) return false;
    } else if (bm == 1) {
        qt = b % nQT;
        s = b / nQT;
    } else {
        s = b % S;
        qt = b / S;
    }
    t0 = (int)((long)s * T / S);
    t1 = (int)((long)(s + 1) * T / S);
    return t0 < t1;
}

// bm == 3's map with a TAPERED tail (the FILTER pass only): each XCD's LAST split is cut into FILTER_TAIL_CUTS pieces.  The
// listing behind the filter's tests is a few per cent of all wave cycles but sits in a few workgroups -- on video-like data a
// (query tile, split) that holds the tile's image region of a bank frame lists thousands of rows and runs twice as long as its
// neighbours (measured: listing 3.9 % of the wave cycles, up to 45 % of one wave's) -- and whatever such a workgroup adds in
// the launch's LAST round is the launch's tail: shorter last workgroups, a shorter tail.
__device__ __forceinline__ bool tapered_split_of_block(int b, int nQT, int S, int T, int &qt, int &s, int &t0, int &t1)
{
    const int xcd = b & 7, idx = b >> 3, S8 = S >> 3;
    const int full = nQT * (S8 - 1);
    int cut = 0, cuts = 1;
    if (idx < full) {
        qt = idx % nQT;
        s = xcd * S8 + idx / nQT;
    } else {
        const int r = idx - full;
        cut = r / nQT;
        cuts = FILTER_TAIL_CUTS;
        if (cut >= FILTER_TAIL_CUTS) return false;
        qt = r - cut * nQT;
        s = xcd * S8 + S8 - 1;
    }
    const int a = (int)((long)s * T / S), e = (int)((long)(s + 1) * T / S);
    t0 = a + (int)((long)cut * (e - a) / cuts);
    t1 = a + (int)((long)(cut + 1) * (e - a) / cuts);
    return t0 < t1;
}


// ---------------------------------------------------------------------------------------------
// main kernel, fp32: one workgroup = 256 queries x one bank split
// sorted insert of d into the ascending list m[0..K-1] (drops the largest)
template <int K>
__device__ __forceinline__ void topk_insert(float (&m)[K], float d)
{
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float lo = fminf(m[j], d);
        d = fmaxf(m[j], d);
        m[j] = lo;
    }
}

// KNN = 1: masked minimum (IntVOS.py:84-85), splits meet through atomicMin on `keys`.
// KNN = 8: the MANET_MAX_KNN smallest distances per (query, object) for the top-k path
//          (IntVOS.py:87-94); every split writes its sorted list to `topk` [S][n_ids][N_pad][8].
// ARG (with KNN = 1): also track WHICH bank row attains the minimum (training: the gradient of torch.min flows to
//          that row only, IntVOS.py:84); splits meet through a 64-bit atomicMin on (distance key << 32 | bank slot)
//          in `keys64` -- equal distances resolve to the smallest slot, i.e. the first row in the sorted bank.
// (KS = 64 -- C in 105..128 -- with the top-k lists or the arg-min slots does not fit 256 VGPRs: those two forms take one
// workgroup per CU instead of spilling)
// NTH (with ARG, r5 -- the training path of k_nearest_neighbors > 1, IntVOS.py:87-94): the minimum is taken over the (distance
//          key, bank slot) pairs STRICTLY ABOVE a per-(object, query) bound -- `keys` then points at the previous pass's 64-bit
//          pairs -- so pass j of k yields the j-th nearest row and its slot, exactly, ties ordered by slot.
template <int KS, int KNN, bool ARG = false, bool NTH = false>
__global__ __launch_bounds__(256, ((KS == 64 && (KNN > 1 || ARG)) || NTH) ? 1 : 2) void global_match_f32_kernel(const char *__restrict__ qpack,
                                                                  const char *__restrict__ bpack,
                                                                  const int *__restrict__ meta,
                                                                  int n_ids, int nQT, int S,
                                                                  long N_pad,
                                                                  unsigned *__restrict__ keys,
                                                                  float *__restrict__ topk, int block_map)
{
    static_assert(!ARG || KNN == 1, "arg-min tracking is the k = 1 path");
    static_assert(!NTH || ARG, "the bounded form is a variant of the arg-min form");
    unsigned long long *keys64 = (unsigned long long *)topk;  // ARG: the top-k region holds the 64-bit pairs
    const unsigned long long *bound64 = (const unsigned long long *)keys;  // NTH: the previous pass's pairs
    constexpr int NG = (KS + 3) / 4;
    constexpr size_t TILE_BYTES = bank_tile_bytes(NG);
    constexpr size_t QBLK_BYTES = query_block_bytes(NG);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x TILE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware mapping: block b runs on XCD b % 8 (observed, used for speed only).  All blocks
    // of one XCD that are resident together share a bank split -> the split streams through L2.
    int qt, s, t0, t1;
    const int T = meta[META_T];
    if (!split_of_block(blockIdx.x, nQT, S, T, block_map, qt, s, t0, t1)) return;

    // Tile staging through registers (issue the global loads a whole tile-step early, write them to
    // LDS after the next barrier).  global_load_lds would save the VGPR round trip, but hipcc cannot
    // tell the DMA's LDS destination from the ds_reads of the other buffer and drains vmcnt(0) in
    // front of the first ds_read of every tile, which serialises the prefetch.
    constexpr int NV = (int)(TILE_BYTES / 16);    // 16-byte vectors per tile
    constexpr int NLD = (NV + 255) / 256;         // per thread
    u32x4 R[NLD];
#define MANET_GLOAD(R_, t_)                                                                \
    {                                                                                      \
        const u32x4 *g_ = (const u32x4 *)(bpack + (size_t)(t_) * TILE_BYTES);              \
        _Pragma("unroll") for (int i_ = 0; i_ < NLD; ++i_)                                 \
        {                                                                                  \
            const int idx_ = i_ * NTHR + tid;                                              \
            R_[i_] = g_[idx_ < NV ? idx_ : NV - 1]; /* clamped: always a valid address */  \
        }                                                                                  \
    }
#define MANET_LSTORE(R_, slot_)                                                            \
    {                                                                                      \
        u32x4 *l_ = (u32x4 *)(smem + (size_t)(slot_) * TILE_BYTES);                        \
        _Pragma("unroll") for (int i_ = 0; i_ < NLD; ++i_)                                 \
            if (i_ * NTHR + tid < NV) l_[i_ * NTHR + tid] = R_[i_];                        \
    }
    constexpr int NTHR = 256;
    MANET_GLOAD(R, t0);

    // this wave's 2 x 32 queries, resident in registers for the whole kernel (B operand:
    // lane holds q[j = lane&31][k = 8g + 2jj + (lane>>5)])
    f32x4 q0[NG], q1[NG];
    float xs0, xs1;
    {
        const char *qb0 = qpack + (size_t)(qt * (QT / QB) + wave * 2) * QBLK_BYTES;
        const char *qb1 = qb0 + QBLK_BYTES;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            q0[g] = *(const f32x4 *)(qb0 + ((size_t)(g * 2 + h) * QB + l31) * 16);
            q1[g] = *(const f32x4 *)(qb1 + ((size_t)(g * 2 + h) * QB + l31) * 16);
        }
        xs0 = *(const float *)(qb0 + (size_t)NG * 2 * QB * 16 + l31 * 4);
        xs1 = *(const float *)(qb1 + (size_t)NG * 2 * QB * 16 + l31 * 4);
    }
    const long qbase = (long)qt * QT + wave * 64 + l31;

    int o = 0;
    while (meta[META_SEG + o + 1] <= t0) ++o;  // object owning tile t0
    int seg_end = meta[META_SEG + o + 1];
    float m0[KNN], m1[KNN];
    int a0 = -1, a1 = -1;  // ARG: bank slot (without the lane's +4h) of the running minimum, -1 = none yet
    unsigned long long B0 = ~0ull, B1 = ~0ull, lo0 = 0, lo1 = 0;  // NTH: running minimum pairs, and the bounds they must exceed
    auto reset = [&]() {
#pragma unroll
        for (int j = 0; j < KNN; ++j) m0[j] = m1[j] = (KNN == 1) ? MANET_WRONG_LABEL_PADDING_DISTANCE : INFINITY;
        a0 = a1 = -1;
        B0 = B1 = ~0ull;
    };
    auto load_bounds = [&](int obj) {
        if (NTH) {
            lo0 = bound64[(size_t)obj * N_pad + qbase];
            lo1 = bound64[(size_t)obj * N_pad + qbase + 32];
        }
    };
    reset();
    load_bounds(o);

    auto flush = [&](int obj) {
        if (ARG) {
            unsigned long long k0 = ((unsigned long long)key_of(m0[0]) << 32) | (unsigned)(a0 < 0 ? -1 : a0 + 4 * h);
            unsigned long long k1 = ((unsigned long long)key_of(m1[0]) << 32) | (unsigned)(a1 < 0 ? -1 : a1 + 4 * h);
            if (NTH) {
                k0 = B0;
                k1 = B1;
            }
            const unsigned long long o0 = __shfl_xor(k0, 32), o1 = __shfl_xor(k1, 32);
            k0 = k0 < o0 ? k0 : o0;
            k1 = k1 < o1 ? k1 : o1;
            if (h == 0) {
                atomicMin(keys64 + (size_t)obj * N_pad + qbase, k0);
                atomicMin(keys64 + (size_t)obj * N_pad + qbase + 32, k1);
            }
        } else if (KNN == 1) {
            float a = min3p(m0[0], m0[0], __shfl_xor(m0[0], 32));
            float c = min3p(m1[0], m1[0], __shfl_xor(m1[0], 32));
            if (h == 0) {
                atomicMin(keys + (size_t)obj * N_pad + qbase, key_of(a));
                atomicMin(keys + (size_t)obj * N_pad + qbase + 32, key_of(c));
            }
        } else {
            // merge the other half-wave's list into ours, then the lower half writes the split's list
            float o0[KNN], o1[KNN];
#pragma unroll
            for (int j = 0; j < KNN; ++j) {
                o0[j] = __shfl_xor(m0[j], 32);
                o1[j] = __shfl_xor(m1[j], 32);
            }
#pragma unroll
            for (int j = 0; j < KNN; ++j) {
                topk_insert<KNN>(m0, o0[j]);
                topk_insert<KNN>(m1, o1[j]);
            }
            if (h == 0) {
                float *p0 = topk + (((size_t)s * n_ids + obj) * N_pad + qbase) * KNN;
                float *p1 = p0 + (size_t)32 * KNN;
#pragma unroll
                for (int j = 0; j < KNN; ++j) {
                    p0[j] = m0[j];
                    p1[j] = m1[j];
                }
            }
        }
    };

    MANET_LSTORE(R, 0);
    if (t0 + 1 < t1) MANET_GLOAD(R, t0 + 1);
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        __syncthreads();  // tile t is visible in buffer buf; buffer buf^1 (tile t-1) is free
        if (t + 1 < t1) {
            MANET_LSTORE(R, buf ^ 1);                 // tile t+1, loaded during the previous step
            if (t + 2 < t1) MANET_GLOAD(R, t + 2);    // in flight during this step's MFMAs
        }
        if (t >= seg_end) {  // wave-uniform: crossed into the next object's rows
            flush(o);
            reset();
            do { ++o; seg_end = meta[META_SEG + o + 1]; } while (t >= seg_end);
            load_bounds(o);
        }
        const char *tb = smem + (size_t)buf * TILE_BYTES;
        const f32x4 *A = (const f32x4 *)tb;
        f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 a0 = A[(g * 2 + h) * BT + l31];
            f32x4 a1 = A[(g * 2 + h) * BT + 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (g * 4 + j < KS) {  // compile-time: k-steps beyond C are all-zero, skip them
                    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], q0[g][j], c00, 0, 0, 0);
                    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], q1[g][j], c01, 0, 0, 0);
                    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], q0[g][j], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], q1[g][j], c11, 0, 0, 0);
                }
            }
        }
        // epilogue: register r of block rb holds bank row rb*32 + (r&3) + 8*(r>>2) + 4*h
        const float *ysl = (const float *)(tb + (size_t)NG * 2 * BT * 16);
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            f32x4 y0 = *(const f32x4 *)(ysl + 8 * tq + 4 * h);
            f32x4 y1 = *(const f32x4 *)(ysl + 32 + 8 * tq + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * tq + i;
                const float d00 = fmaf(-2.0f, c00[r], xs0 + y0[i]);  // IntVOS.py:39
                const float d01 = fmaf(-2.0f, c01[r], xs1 + y0[i]);
                const float d10 = fmaf(-2.0f, c10[r], xs0 + y1[i]);
                const float d11 = fmaf(-2.0f, c11[r], xs1 + y1[i]);
                if (NTH) {  // the smallest (distance key, slot) pair above the bound; a NaN (key 0) never qualifies
                    const unsigned slot = (unsigned)(t * BT + (r & 3) + 8 * (r >> 2) + 4 * h);
                    const unsigned long long k00 = ((unsigned long long)key_of(d00) << 32) | slot;
                    const unsigned long long k10 = ((unsigned long long)key_of(d10) << 32) | (slot + 32u);
                    const unsigned long long k01 = ((unsigned long long)key_of(d01) << 32) | slot;
                    const unsigned long long k11 = ((unsigned long long)key_of(d11) << 32) | (slot + 32u);
                    if (k00 > lo0 && k00 < B0) B0 = k00;
                    if (k10 > lo0 && k10 < B0) B0 = k10;
                    if (k01 > lo1 && k01 < B1) B1 = k01;
                    if (k11 > lo1 && k11 < B1) B1 = k11;
                } else if (ARG) {  // strict <: the first row in bank order wins a tie; a NaN never wins
                    const int slot = t * BT + (r & 3) + 8 * (r >> 2);
                    if (d00 < m0[0]) { m0[0] = d00; a0 = slot; }
                    if (d10 < m0[0]) { m0[0] = d10; a0 = slot + 32; }
                    if (d01 < m1[0]) { m1[0] = d01; a1 = slot; }
                    if (d11 < m1[0]) { m1[0] = d11; a1 = slot + 32; }
                } else if (KNN == 1) {  // NaN-propagating, like torch.min (IntVOS.py:84)
                    m0[0] = min3p(m0[0], d00, d10);
                    m1[0] = min3p(m1[0], d01, d11);
                } else {
                    topk_insert<KNN>(m0, d00);
                    topk_insert<KNN>(m0, d10);
                    topk_insert<KNN>(m1, d01);
                    topk_insert<KNN>(m1, d11);
                }
            }
        }
    }
    flush(o);
}

// ---------------------------------------------------------------------------------------------
// Half-tile pipelined form of the fp32 kernel (k = 1, the shipped headline path).  Same decomposition, operand
// images and arithmetic as global_match_f32_kernel<KS, 1> -- every accumulator is the same k-ascending fmaf chain,
// d = fmaf(-2, mm, xs + ys): bit-identical results -- but the tile is computed as two phases of 100 MFMAs:
//   phase A  rows  0-31 (c00, c01)   while the VALU reduces rows 32-63 of the PREVIOUS tile (c10, c11 still hold
//                                     them; their |k|^2 were copied to 16 registers before the buffer was recycled)
//   phase B  rows 32-63 (c10, c11)   while the VALU reduces rows 0-31 of this tile
// r2 PMC of the un-pipelined kernel: matrix pipe 90 % busy -- the two co-resident workgroups of a CU run in
// lockstep, so their ~1.3 k-cycle epilogues (add, fma, min per element) coincide and the pipe idles for exactly
// that share of a 12.8 k-cycle tile.  Here a wave's MFMA stream never stops for an epilogue.
// Staging by asm LDS-DMA (see lds_dma16): it frees the 28 staging VGPRs the 16 carried |k|^2 need, and removes
// the ds_write pass; one raw s_barrier per tile.
// RESCUE (MANET_COMPUTE_BF16_REFINE): the same kernel as the exact fall-back of the bf16 filter -- a workgroup returns at
// once unless one of its query tile's eight 32-query blocks has an incomplete candidate bucket (bit 31 or a count past
// the capacity); the minima meet the re-rank's by atomicMin on the same keys (the same fp32 chains: the same bits).
template <int KS, bool RESCUE = false>
__global__ __launch_bounds__(256, 2) void global_match_f32_pipe_kernel(const char *__restrict__ qpack,
                                                                       const char *__restrict__ bpack,
                                                                       const int *__restrict__ meta, int n_ids,
                                                                       int nQT, int S, long N_pad,
                                                                       unsigned *__restrict__ keys, int block_map,
                                                                       const unsigned *__restrict__ bcnt = nullptr,
                                                                       long bucket_cap = 0)
{
    constexpr int NG = (KS + 3) / 4;
    constexpr size_t TILE_BYTES = bank_tile_bytes(NG);  // whole KiB
    constexpr size_t QBLK_BYTES = query_block_bytes(NG);
    constexpr int PIECES = (int)(TILE_BYTES / 1024);
    constexpr int NW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x TILE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    int qt, s, t0, t1;
    const int T = meta[META_T];
    if (RESCUE && (block_map & RESCUE_LISTED)) {
        // the tiles to rescue were listed by the re-rank launch (rs = {count, tile ids}, behind bcnt's two halves): the
        // launch's workgroups are dealt to THOSE tiles, with as many bank splits each as the grid allows -- a frame with a
        // handful of incomplete tiles spreads them over the whole chip instead of leaving each to 16 workgroups (r4: 1.5 ms
        // for 4 of 102 tiles at cfg2 size, the time one workgroup needs for a 16th of the bank)
        const unsigned *rs = bcnt + 2 * (N_pad >> 5);
        const int nr = (int)rs[0];
        if (nr <= 0) return;
        // splits per tile: whole rounds of the chip's 512 workgroup slots (a workgroup's fixed cost -- its 106 KB query
        // operand, the pipeline fill, the closing atomics -- is worth ~6 tiles of matrix work: few long workgroups beat many
        // short ones; 4 tiles of 102 at cfg2 size: 390 us with 404 splits of 5 tiles, 3.2 rounds)
        int Sd = (block_map >> 8) & BLOCK_MAP_RESCUE_MASK;  // (experiments: MANET_TUNE_RESCUE_SPLITS)
        if (Sd > 0) {
            const int most_grid = (int)gridDim.x / nr;
            Sd = Sd > most_grid ? most_grid : Sd;
            Sd = Sd < 1 ? 1 : Sd;
        } else {
            const int most_grid = (int)gridDim.x / nr, most_t = T / 8 > 1 ? T / 8 : 1;
            const int most = most_grid < most_t ? most_grid : most_t;
            long best = -1;
            for (int r = 1; r <= 4; ++r) {
                int c = (512 * r) / nr;
                c = c < 1 ? 1 : (c > most ? most : c);
                const long rounds = ((long)nr * c + 511) / 512;
                const long cost = rounds * ((long)(T + c - 1) / c + 6);
                if (best < 0 || cost < best) { best = cost; Sd = c; }
            }
        }
        if ((int)blockIdx.x >= nr * Sd) return;
        s = (int)blockIdx.x / nr;
        qt = (int)rs[1 + ((int)blockIdx.x - s * nr)];
        t0 = (int)((long)s * T / Sd);
        t1 = (int)((long)(s + 1) * T / Sd);
        if (t0 >= t1) return;
    } else {
        if (!split_of_block(blockIdx.x, nQT, S, T, block_map, qt, s, t0, t1)) return;
        if (RESCUE) {
            bool need = false;
            for (int i = 0; i < QT / QB; ++i) {
                const unsigned raw = bcnt[(long)qt * (QT / QB) + i];
                need = need || (raw >> 31) || (long)raw > bucket_cap;
            }
            if (!need) return;  // (wave-uniform: every lane read the same eight counters)
        }
    }

    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    auto stage_dma = [&](int t, int slot) __attribute__((always_inline)) {
        const char *g = bpack + (size_t)t * TILE_BYTES + (size_t)lane * 16;
        const unsigned l = smem_base + (unsigned)slot * (unsigned)TILE_BYTES;
#pragma unroll
        for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
            const int pc = wave + i * NW;  // wave-uniform
            if (pc < PIECES) lds_dma16(g + (size_t)pc * 1024, l + (unsigned)pc * 1024u);
        }
    };
    stage_dma(t0, 0);

    f32x4 q0[NG], q1[NG];
    float xs0, xs1;
    {
        const char *qb0 = qpack + (size_t)(qt * (QT / QB) + wave * 2) * QBLK_BYTES;
        const char *qb1 = qb0 + QBLK_BYTES;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            q0[g] = *(const f32x4 *)(qb0 + ((size_t)(g * 2 + h) * QB + l31) * 16);
            q1[g] = *(const f32x4 *)(qb1 + ((size_t)(g * 2 + h) * QB + l31) * 16);
        }
        xs0 = *(const float *)(qb0 + (size_t)NG * 2 * QB * 16 + l31 * 4);
        xs1 = *(const float *)(qb1 + (size_t)NG * 2 * QB * 16 + l31 * 4);
    }
    const long qbase = (long)qt * QT + wave * 64 + l31;
    // the compiler waits for the query operand HERE (its counted vmcnt waits must not sink into the tile loop,
    // where they would drain the LDS-DMA it does not know about)
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(q0[g]), "+v"(q1[g]));
    asm volatile("" : "+v"(xs0), "+v"(xs1));

    int o = 0;
    while (meta[META_SEG + o + 1] <= t0) ++o;  // object owning tile t0
    int seg_end = meta[META_SEG + o + 1];
    float m0a, m0b, m1a, m1b;  // two running minima per query block: short dependency chains
    m0a = m0b = m1a = m1b = MANET_WRONG_LABEL_PADDING_DISTANCE;
    auto flush = [&](int obj) {
        const float v0 = min3p(m0a, m0b, m0b), v1 = min3p(m1a, m1b, m1b);
        const float a = min3p(v0, v0, __shfl_xor(v0, 32));
        const float c = min3p(v1, v1, __shfl_xor(v1, 32));
        if (h == 0) {
            atomicMin(keys + (size_t)obj * N_pad + qbase, key_of(a));
            atomicMin(keys + (size_t)obj * N_pad + qbase + 32, key_of(c));
        }
    };

    // pending = rows 32-63 of the previous tile: still in c10 / c11, their |k|^2 in yp[]; "nothing pending" is
    // expressed by values that cannot win (c = 0, |k|^2 = 1e20 -> d = 1e20)
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    float yp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) yp[r] = MANET_WRONG_LABEL_PADDING_DISTANCE;
    // the reference's d = (xs + ys) - 2 mm with its two roundings (IntVOS.py:39), then the running minimum
#define MANET_REDUCE(ca_, cb_, y_, r_)                                                             \
    {                                                                                              \
        const float da_ = fmaf(-2.0f, ca_[r_], xs0 + (y_));                                        \
        const float db_ = fmaf(-2.0f, cb_[r_], xs1 + (y_));                                        \
        if ((r_) & 1) {                                                                            \
            m0b = min3p(m0b, da_, da_);                                                            \
            m1b = min3p(m1b, db_, db_);                                                            \
        } else {                                                                                   \
            m0a = min3p(m0a, da_, da_);                                                            \
            m1a = min3p(m1a, db_, db_);                                                            \
        }                                                                                          \
    }
    auto settle = [&]() __attribute__((always_inline)) {  // reduce what is pending, serially (object boundary, end)
#pragma unroll
        for (int r = 0; r < 16; ++r) MANET_REDUCE(c10, c11, yp[r], r);
#pragma unroll
        for (int r = 0; r < 16; ++r) yp[r] = MANET_WRONG_LABEL_PADDING_DISTANCE;
        c10 = f32x16{0};
        c11 = f32x16{0};
    };

    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        // this wave's pieces of tile t have landed (issued one tile ago); the barrier publishes the tile and tells
        // us that every wave is done with the other buffer (tile t-1: fragments consumed, |k|^2 copied to yp)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < t1) stage_dma(t + 1, buf ^ 1);  // in flight during this tile's 200 MFMAs
        if (t >= seg_end) {  // wave-uniform: crossed into the next object's rows
            settle();
            flush(o);
            m0a = m0b = m1a = m1b = MANET_WRONG_LABEL_PADDING_DISTANCE;
            do { ++o; seg_end = meta[META_SEG + o + 1]; } while (t >= seg_end);
        }
        const char *tb = smem + (size_t)buf * TILE_BYTES;
        const f32x4 *A = (const f32x4 *)tb;
        const float *ysl = (const float *)(tb + (size_t)NG * 2 * BT * 16);
        // ---- phase A: rows 0-31 of tile t on the matrix pipe, rows 32-63 of tile t-1 on the VALU
        c00 = f32x16{0};
        c01 = f32x16{0};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 a0 = A[(g * 2 + h) * BT + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (g * 4 + j < KS) {  // compile-time: k-steps beyond C are all-zero, skip them
                    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], q0[g][j], c00, 0, 0, 0);
                    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], q1[g][j], c01, 0, 0, 0);
                }
            }
            // the 16 pending registers spread over the NG fragment groups
#pragma unroll
            for (int r = (g * 16) / NG; r < ((g + 1) * 16) / NG; ++r) MANET_REDUCE(c10, c11, yp[r], r);
        }
        // |k|^2 of rows 32-63 of THIS tile, for the next tile's phase A (register r = row 32 + (r&3) + 8(r>>2) + 4h)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 y1 = *(const f32x4 *)(ysl + 32 + 8 * tq + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) yp[4 * tq + i] = y1[i];
        }
        // ---- phase B: rows 32-63 of tile t on the matrix pipe, rows 0-31 of tile t on the VALU
        c10 = f32x16{0};
        c11 = f32x16{0};
        f32x4 y0v[4];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) y0v[tq] = *(const f32x4 *)(ysl + 8 * tq + 4 * h);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 a1 = A[(g * 2 + h) * BT + 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (g * 4 + j < KS) {
                    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], q0[g][j], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], q1[g][j], c11, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = (g * 16) / NG; r < ((g + 1) * 16) / NG; ++r) MANET_REDUCE(c00, c01, y0v[r >> 2][r & 3], r);
        }
    }
    settle();
#undef MANET_REDUCE
    flush(o);
}

// ---------------------------------------------------------------------------------------------
// main kernel, bf16 operands (MANET_COMPUTE_BF16 / _BF16X3): one workgroup = 512 queries x one bank
// split, 8 waves.  Same decomposition as the f32 kernel (swapped operands, object-pure 64-row tiles,
// lane-local running min, atomicMin across splits); the contraction is v_mfma_f32_32x32x16_bf16 with fp32
// accumulation:
//   X3 = false: one MFMA per 16 k on embeddings rounded to bf16 (7 instead of 50 MFMAs per block at C=100)
//   X3 = true : hi*hi + hi*lo + lo*hi with x = hi + lo, hi = bf16(x), lo = bf16(x - hi): the dropped
//               lo*lo term is <= 2^-16 relative, i.e. fp32-class distances at 3/16 of the f32 MFMA cost.
// The operand images carry -2q and both squared norms (see Geom), so an accumulator element IS
// d(n, m) = |q|^2 + |k|^2 - 2 q.k when the k loop ends: the epilogue is ONE v_minimum3_f32 per two
// elements (r1: add + fma + min per element = 6x the VALU work, 0.32 of 0.68 ms).
// Staging: TPS tiles per step, double buffered in LDS; DMA = true: asm LDS-DMA (no VGPR round trip, no
// ds_write pass, the loads of step s+1 fly during the whole of step s); DMA = false: r1's register staging.
template <int KSB, bool X3, int TPS, bool DMA>
__global__ __launch_bounds__(512, 1) void global_match_bf16_kernel(const char *__restrict__ qpack,
                                                                   const char *__restrict__ bpack,
                                                                   const int *__restrict__ meta, int n_ids,
                                                                   int nQT, int S, long N_pad,
                                                                   unsigned *__restrict__ keys, int block_map,
                                                                   int young_prio)
{
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    constexpr int NW = 8;
    constexpr int UNITS = 2 * KSB * (X3 ? 2 : 1);
    constexpr int LO = 2 * KSB;  // first unit of the lo image
    constexpr size_t TILE_BYTES = bank_tile_bytes_u(UNITS, false);  // = UNITS KiB
    constexpr size_t QBLK_BYTES = query_block_bytes_u(UNITS, false);
    constexpr size_t STEP_BYTES = TILE_BYTES * TPS;
    constexpr int QTB = NW * 64;
    static_assert(TILE_BYTES == (size_t)UNITS * 1024, "a bf16 tile is a whole number of 1 KiB DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x STEP_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    int qt, s, t0, t1;
    const int T = meta[META_T];
    if (!split_of_block(blockIdx.x, nQT, S, T, block_map, qt, s, t0, t1)) return;

    // ---- staging of one STEP = TPS consecutive tiles ------------------------------------------------
    constexpr int PIECES = UNITS * TPS;  // 1 KiB pieces per step
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    auto stage_dma = [&](int t, int slot) __attribute__((always_inline)) {
        const int np = ((t1 - t) < TPS ? (t1 - t) : TPS) * UNITS;  // the split's last step may be short
        const char *g = bpack + (size_t)t * TILE_BYTES + (size_t)lane * 16;
        const unsigned l = smem_base + (unsigned)slot * (unsigned)STEP_BYTES;
#pragma unroll
        for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
            const int pc = wave + i * NW;  // wave-uniform
            if (pc < np) lds_dma16(g + (size_t)pc * 1024, l + (unsigned)pc * 1024u);
        }
    };
    constexpr int NV = (int)(TILE_BYTES / 16) * TPS;
    constexpr int NLD = DMA ? 1 : (NV + NW * 64 - 1) / (NW * 64);
    constexpr int NTHR = NW * 64;
    u32x4 Ra[NLD];
    auto gload = [&](int t) __attribute__((always_inline)) {
        const u32x4 *g_ = (const u32x4 *)(bpack + (size_t)t * TILE_BYTES);
        const int lim_ = ((t1 - t) < TPS ? (t1 - t) : TPS) * (int)(TILE_BYTES / 16);
#pragma unroll
        for (int i_ = 0; i_ < NLD; ++i_) {
            const int idx_ = i_ * NTHR + tid;
            Ra[i_] = g_[idx_ < lim_ ? idx_ : lim_ - 1];  // clamped into the split's own range
        }
    };
    auto lstore = [&](int slot) __attribute__((always_inline)) {
        u32x4 *l_ = (u32x4 *)(smem + (size_t)slot * STEP_BYTES);
#pragma unroll
        for (int i_ = 0; i_ < NLD; ++i_)
            if (i_ * NTHR + tid < NV) l_[i_ * NTHR + tid] = Ra[i_];
    };
    if (DMA) stage_dma(t0, 0);
    else gload(t0);

    // B operand: lane holds (-2q | norm slots)[j = lane&31][k = 16s + 8*(lane>>5) + 0..7] for its two blocks
    u32x4 q0[KSB], q1[KSB], q0l[X3 ? KSB : 1], q1l[X3 ? KSB : 1];
    {
        const char *qb0 = qpack + (size_t)(qt * (QTB / QB) + wave * 2) * QBLK_BYTES;
        const char *qb1 = qb0 + QBLK_BYTES;
#pragma unroll
        for (int k = 0; k < KSB; ++k) {
            q0[k] = *(const u32x4 *)(qb0 + ((size_t)(k * 2 + h) * QB + l31) * 16);
            q1[k] = *(const u32x4 *)(qb1 + ((size_t)(k * 2 + h) * QB + l31) * 16);
            if (X3) {
                q0l[k] = *(const u32x4 *)(qb0 + ((size_t)(LO + k * 2 + h) * QB + l31) * 16);
                q1l[k] = *(const u32x4 *)(qb1 + ((size_t)(LO + k * 2 + h) * QB + l31) * 16);
            }
        }
    }
    const long qbase = (long)qt * QTB + wave * 64 + l31;
    // Make the compiler wait for the query operand HERE.  Otherwise its counted vmcnt waits for these
    // loads sink into the tile loop (down to vmcnt(0)), where they would also drain the asm LDS-DMA of the
    // next step that the compiler does not know about.
#pragma unroll
    for (int k = 0; k < KSB; ++k) {
        asm volatile("" : "+v"(q0[k]), "+v"(q1[k]));
        if (X3) asm volatile("" : "+v"(q0l[k]), "+v"(q1l[k]));
    }

    int o = 0;
    while (meta[META_SEG + o + 1] <= t0) ++o;
    int seg_end = meta[META_SEG + o + 1];
    // two running minima per query block (even / odd accumulator registers): short dependency chains
    float m0a, m0b, m1a, m1b;
    m0a = m0b = m1a = m1b = MANET_WRONG_LABEL_PADDING_DISTANCE;
    auto flush = [&](int obj) {
        float a = min3p(m0a, m0b, __shfl_xor(min3p(m0a, m0b, m0b), 32));
        float c = min3p(m1a, m1b, __shfl_xor(min3p(m1a, m1b, m1b), 32));
        if (h == 0) {
            atomicMin(keys + (size_t)obj * N_pad + qbase, key_of(a));
            atomicMin(keys + (size_t)obj * N_pad + qbase + 32, key_of(c));
        }
    };
#define MANET_BF(x) __builtin_bit_cast(bf16x8_t, x)
#define MANET_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a_), MANET_BF(b_), c_, 0, 0, 0)
    // one 64-row tile: 4 x KSB (x3) MFMAs from zero accumulators, then the running minimum
    auto tile = [&](const char *tb) __attribute__((always_inline)) {
        const u32x4 *A = (const u32x4 *)tb;
        f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
        if (!X3) {
            // all A fragments of the tile first (2 x KSB ds_read_b128 in flight), then the MFMAs back to
            // back behind counted lgkmcnt waits: the matrix pipe never waits for an LDS round trip
            u32x4 a0[KSB], a1[KSB];
#pragma unroll
            for (int k = 0; k < KSB; ++k) {
                a0[k] = A[(k * 2 + h) * BT + l31];
                a1[k] = A[(k * 2 + h) * BT + 32 + l31];
            }
#pragma unroll
            for (int k = 0; k < KSB; ++k) {
                MANET_MFMA(a0[k], q0[k], c00);
                MANET_MFMA(a1[k], q0[k], c10);
                MANET_MFMA(a0[k], q1[k], c01);
                MANET_MFMA(a1[k], q1[k], c11);
            }
        } else {
#pragma unroll
            for (int k = 0; k < KSB; ++k) {
                u32x4 a0 = A[(k * 2 + h) * BT + l31];
                u32x4 a1 = A[(k * 2 + h) * BT + 32 + l31];
                u32x4 a0l = A[(LO + k * 2 + h) * BT + l31];
                u32x4 a1l = A[(LO + k * 2 + h) * BT + 32 + l31];
                MANET_MFMA(a0, q0[k], c00);
                MANET_MFMA(a1, q0[k], c10);
                MANET_MFMA(a0, q1[k], c01);
                MANET_MFMA(a1, q1[k], c11);
                MANET_MFMA(a0, q0l[k], c00);
                MANET_MFMA(a1, q0l[k], c10);
                MANET_MFMA(a0, q1l[k], c01);
                MANET_MFMA(a1, q1l[k], c11);
                MANET_MFMA(a0l, q0[k], c00);
                MANET_MFMA(a1l, q0[k], c10);
                MANET_MFMA(a0l, q1[k], c01);
                MANET_MFMA(a1l, q1[k], c11);
            }
        }
        // accumulator register r of block rb = bank row rb*32 + (r&3) + 8*(r>>2) + 4*h; all of them belong
        // to the same object, so the reduction over rows is register-wise
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            m0a = min3p(m0a, c00[r], c10[r]);
            m0b = min3p(m0b, c00[r + 1], c10[r + 1]);
            m1a = min3p(m1a, c01[r], c11[r]);
            m1b = min3p(m1b, c01[r + 1], c11[r + 1]);
        }
    };
    auto next_object = [&](int t) __attribute__((always_inline)) {
        if (t >= seg_end) {  // wave-uniform: tile t starts another object's rows
            flush(o);
            m0a = m0b = m1a = m1b = MANET_WRONG_LABEL_PADDING_DISTANCE;
            do { ++o; seg_end = meta[META_SEG + o + 1]; } while (t >= seg_end);
        }
    };
    // static priority for the younger half of the workgroup (waves 4-7 lose every VALU arbitration against
    // their SIMD partner otherwise; MI355X_MICROARCH.md "two waves per SIMD", item 4)
    if (young_prio && wave >= 4) __builtin_amdgcn_s_setprio(1);

    if (!DMA) {
        lstore(0);
        if (t0 + TPS < t1) gload(t0 + TPS);
    }
    int buf = 0;
    for (int t = t0; t < t1; t += TPS, buf ^= 1) {
        if (DMA) {
            // this wave's pieces of the step have landed; the barrier then publishes everybody's pieces
            // and tells us that every wave is done reading the other buffer (the previous step)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + TPS < t1) stage_dma(t + TPS, buf ^ 1);  // in flight during the whole step
        } else {
            __syncthreads();  // step's tiles visible in `buf`; the other buffer is free
            if (t + TPS < t1) {
                lstore(buf ^ 1);
                if (t + 2 * TPS < t1) gload(t + 2 * TPS);
            }
        }
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            if (t + u < t1) {
                next_object(t + u);
                tile(smem + (size_t)buf * STEP_BYTES + (size_t)u * TILE_BYTES);
            }
        }
    }
#undef MANET_MFMA
#undef MANET_BF
    flush(o);
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined form of the plain-bf16 kernel (the shipped one for MANET_COMPUTE_BF16).
// Same decomposition and arithmetic as global_match_bf16_kernel<KSB, false, 2, true>; what changes is WHEN
// things are issued, so that the matrix pipe never waits for LDS (r2 PMC of the kernel above: pipe 67 %
// busy, waves parked 37 % of their cycles on s_waitcnt -- the A-fragment reads right behind every barrier):
//   * a step = 2 tiles (A, B); fragment registers F[k] are bound to k-step k: as soon as the four MFMAs of
//     (tile, k) have been issued, F[k] is refilled with the NEXT tile's k-step -- every ds_read_b128 is in
//     flight for a whole tile (~7 x 128 matrix cycles) before its MFMA needs it;
//   * the step's barrier sits BETWEEN tile A and tile B.  By then tile B's fragments are already in
//     registers, so the barrier (a) publishes the next step's buffer, whose first fragments tile B's k-steps
//     prefetch, and (b) frees the current buffer for the LDS-DMA of the step after next.  No fragment read
//     ever follows a barrier directly; two LDS buffers suffice.
// ABL: timing ablations only (results are garbage): 1 = no DMA / no vmcnt wait, 2 = no barrier, 4 = epilogue
// reduced to one min3, 8 = no fragment refills.  (A 3-deep LDS ring with counted vmcnt(N) waits -- the DMA two
// steps ahead -- was measured slower, 0.529 vs 0.498 ms, and removed: DMA latency is not the stall.)
template <int KSB, int ABL>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_waves_per_eu(2, 2)))
void global_match_bf16_pipe_kernel(const char *__restrict__ qpack, const char *__restrict__ bpack,
                                   const int *__restrict__ meta, int n_ids, int nQT, int S, long N_pad,
                                   unsigned *__restrict__ keys, int block_map, int young_prio)
{
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    constexpr int NW = 8, TPS = 2;
    constexpr int UNITS = 2 * KSB;
    constexpr size_t TILE_BYTES = bank_tile_bytes_u(UNITS, false);  // = UNITS KiB
    constexpr size_t QBLK_BYTES = query_block_bytes_u(UNITS, false);
    constexpr size_t STEP_BYTES = TILE_BYTES * TPS;
    constexpr int QTB = NW * 64;
    constexpr int NBUF = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF x STEP_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    int qt, s, t0, t1;
    const int T = meta[META_T];
    if (!split_of_block(blockIdx.x, nQT, S, T, block_map, qt, s, t0, t1)) return;

    constexpr int PIECES = UNITS * TPS;  // 1 KiB pieces per step
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    auto stage_dma = [&](int t, int slot, bool prologue = false) __attribute__((always_inline)) {
        if ((ABL & 1) && !prologue) return;  // ablation: only the first steps are staged (real data in LDS)
        const int np = ((t1 - t) < TPS ? (t1 - t) : TPS) * UNITS;  // the split's last step may be short
        const char *g = bpack + (size_t)t * TILE_BYTES + (size_t)lane * 16;
        const unsigned l = smem_base + (unsigned)slot * (unsigned)STEP_BYTES;
#pragma unroll
        for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
            const int pc = wave + i * NW;  // wave-uniform
            if (pc < np) lds_dma16(g + (size_t)pc * 1024, l + (unsigned)pc * 1024u);
        }
    };
    stage_dma(t0, 0, true);
    if (t0 + TPS < t1) stage_dma(t0 + TPS, 1, true);

    u32x4 q0[KSB], q1[KSB];
    {
        const char *qb0 = qpack + (size_t)(qt * (QTB / QB) + wave * 2) * QBLK_BYTES;
        const char *qb1 = qb0 + QBLK_BYTES;
#pragma unroll
        for (int k = 0; k < KSB; ++k) {
            q0[k] = *(const u32x4 *)(qb0 + ((size_t)(k * 2 + h) * QB + l31) * 16);
            q1[k] = *(const u32x4 *)(qb1 + ((size_t)(k * 2 + h) * QB + l31) * 16);
        }
    }
    const long qbase = (long)qt * QTB + wave * 64 + l31;
#pragma unroll
    for (int k = 0; k < KSB; ++k) asm volatile("" : "+v"(q0[k]), "+v"(q1[k]));  // compiler waits for q here

    int o = 0;
    while (meta[META_SEG + o + 1] <= t0) ++o;
    int seg_end = meta[META_SEG + o + 1];
    float m0a, m0b, m1a, m1b;
    m0a = m0b = m1a = m1b = MANET_WRONG_LABEL_PADDING_DISTANCE;
    auto flush = [&](int obj) {
        float a = min3p(m0a, m0b, __shfl_xor(min3p(m0a, m0b, m0b), 32));
        float c = min3p(m1a, m1b, __shfl_xor(min3p(m1a, m1b, m1b), 32));
        if (h == 0) {
            atomicMin(keys + (size_t)obj * N_pad + qbase, key_of(a));
            atomicMin(keys + (size_t)obj * N_pad + qbase + 32, key_of(c));
        }
    };
    auto next_object = [&](int t) __attribute__((always_inline)) {
        if (t >= seg_end) {  // wave-uniform: tile t starts another object's rows
            flush(o);
            m0a = m0b = m1a = m1b = MANET_WRONG_LABEL_PADDING_DISTANCE;
            do { ++o; seg_end = meta[META_SEG + o + 1]; } while (t >= seg_end);
        }
    };
    if (young_prio && wave >= 4) __builtin_amdgcn_s_setprio(1);

    // this lane's fragment offset inside a tile image: unit (2k + h), rows l31 and 32 + l31
    const unsigned frag_off = (unsigned)((h * BT + l31) * 16);
    u32x4 F0[KSB], F1[KSB];
#define MANET_BF(x) __builtin_bit_cast(bf16x8_t, x)
#define MANET_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a_), MANET_BF(b_), c_, 0, 0, 0)
#define MANET_LOADF(k_, tile_base_)                                                                \
    {                                                                                              \
        const char *f_ = (tile_base_) + frag_off + (size_t)(k_) * (2 * BT * 16);                   \
        F0[k_] = *(const u32x4 *)f_;                                                               \
        F1[k_] = *(const u32x4 *)(f_ + 32 * 16);                                                   \
    }
    // one tile: MFMAs of k-step k from F[k], then F[k] <- k-step k of the tile at `next_base`
#define MANET_TILE(next_base_)                                                                     \
    {                                                                                              \
        f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};                                         \
        _Pragma("unroll") for (int k = 0; k < KSB; ++k)                                            \
        {                                                                                          \
            MANET_MFMA(F0[k], q0[k], c00);                                                         \
            MANET_MFMA(F1[k], q0[k], c10);                                                         \
            MANET_MFMA(F0[k], q1[k], c01);                                                         \
            MANET_MFMA(F1[k], q1[k], c11);                                                         \
            if (!(ABL & 8)) MANET_LOADF(k, next_base_);                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); /* 4 MFMA  */                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); /* 2 DS read: the refill, right behind them */ \
        }                                                                                          \
        _Pragma("unroll") for (int r = 0; r < ((ABL & 4) ? 2 : 16); r += 2)                        \
        {                                                                                          \
            m0a = min3p(m0a, c00[r], c10[r]);                                                      \
            m0b = min3p(m0b, c00[r + 1], c10[r + 1]);                                              \
            m1a = min3p(m1a, c01[r], c11[r]);                                                      \
            m1b = min3p(m1b, c01[r + 1], c11[r + 1]);                                              \
        }                                                                                          \
    }

    // prologue: steps 0 and 1 in flight; publish step 0, fetch tile A's fragments
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < KSB; ++k) MANET_LOADF(k, smem);

    int buf = 0;
    for (int t = t0; t < t1; t += TPS, buf ^= 1) {
        const char *cur = smem + (size_t)buf * STEP_BYTES;
        const char *nxt = smem + (size_t)(buf ^ 1) * STEP_BYTES;
        // ---- tile A (its fragments are in F; refill F with tile B of the same buffer)
        next_object(t);
        MANET_TILE(cur + TILE_BYTES);
        // ---- mid-step: everything of this buffer is in registers now.  This wave's pieces of the next step have
        // landed (issued one step ago); the barrier publishes the next step's buffer and frees this one for the
        // step after next.
        if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of `cur` have returned
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NBUF * TPS < t1) stage_dma(t + NBUF * TPS, buf);
        // ---- tile B (refill F with tile A of the next step's buffer; a stale read if there is none)
        if (t + 1 < t1) {
            next_object(t + 1);
            MANET_TILE(nxt);
        }
    }
#undef MANET_TILE
#undef MANET_LOADF
#undef MANET_MFMA
#undef MANET_BF
    flush(o);
}

// ---------------------------------------------------------------------------------------------
// "Wide" form of the pipelined plain-bf16 kernel: 4 waves per workgroup, each wave owns 128 queries (four
// 32-query blocks, 112 VGPRs of -2q operand) and walks the bank in PASSES of 32 rows: 4 x KSB MFMAs per pass,
// four independent accumulators, one A fragment per k-step (7 ds_read_b128 per 28 MFMAs -- half the LDS
// fragment traffic of the 64 x 64 wave tile).  Same 512 queries per workgroup and the same packed images as
// the other bf16 kernels, but TWO workgroups per CU (256 threads, <= 256 VGPRs): their barriers are
// independent, and the prologue / flush of one overlaps the matrix work of the other.
//   step = 2 tiles = 4 passes; fragments F[k] bound to k, refilled with the next pass's k-step right behind
//   the MFMAs that consumed them; the step's barrier sits before the LAST pass (its fragments are already in
//   registers): it publishes the next step's buffer and frees the current one for the LDS-DMA of step + 2.
// FILTER (MANET_COMPUTE_BF16_REFINE, second pass): instead of reducing to a minimum, every bank row whose bf16 distance is
// within the query's threshold thr[object][query] is appended to the flat candidate list {pair, bank slot} (stats[0] = its
// fill count; an entry that does not fit raises stats[1]).  A pass's 16 distances per lane are first reduced to their minimum --
// the same eight v_minimum3 the plain kernel spends -- and only a wave in which some lane's minimum passes its threshold
// takes the slow path that looks at the individual rows.
template <int KSB, int ABL, bool FILTER = false>
__global__ __launch_bounds__(256, 2) void global_match_bf16_wide_kernel(const char *__restrict__ qpack,
                                                                        const char *__restrict__ bpack,
                                                                        const int *__restrict__ meta, int n_ids,
                                                                        int nQT, int S, long N_pad,
                                                                        unsigned *__restrict__ keys, int block_map,
                                                                        int young_prio, unsigned *__restrict__ thr,
                                                                        const float *__restrict__ slack,
                                                                        unsigned long long *__restrict__ stats,
                                                                        uint2 *__restrict__ list, long bucket_cap,
                                                                        unsigned *__restrict__ bcnt)
{
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    constexpr int NW = 4, TPS = 2, NQB = 4;  // waves, tiles per step, query blocks per wave
    constexpr int UNITS = 2 * KSB;
    constexpr size_t TILE_BYTES = bank_tile_bytes_u(UNITS, false);  // = UNITS KiB
    constexpr size_t QBLK_BYTES = query_block_bytes_u(UNITS, false);
    constexpr size_t STEP_BYTES = TILE_BYTES * TPS;
    constexpr int QTB = QT_BF16;
    static_assert(NW * NQB * QB == QTB, "4 waves x 4 blocks x 32 queries");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x STEP_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    int qt, s, t0, t1;
    const int T = meta[META_T];
    if (FILTER && (block_map & 0xff) == 8) {
        if (!tapered_split_of_block(blockIdx.x, nQT, S, T, qt, s, t0, t1)) return;
    } else if (!split_of_block(blockIdx.x, nQT, S, T, block_map, qt, s, t0, t1))
        return;

    constexpr int PIECES = UNITS * TPS;  // 1 KiB pieces per step
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    auto stage_dma = [&](int t, int slot, bool prologue = false) __attribute__((always_inline)) {
        if ((ABL & 1) && !prologue) return;
        const int np = ((t1 - t) < TPS ? (t1 - t) : TPS) * UNITS;  // the split's last step may be short
        const unsigned l = smem_base + (unsigned)slot * (unsigned)STEP_BYTES;
        if (FILTER) {  // (at the register limit: a uniform base + this lane's 32-bit offset, no 64-bit address pair)
            const char *g = bpack + (size_t)t * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
                const int pc = wave + i * NW;  // wave-uniform
                if (pc < np) lds_dma16_s(g + (size_t)pc * 1024, (unsigned)lane * 16u, l + (unsigned)pc * 1024u);
            }
            return;
        }
        const char *g = bpack + (size_t)t * TILE_BYTES + (size_t)lane * 16;
#pragma unroll
        for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
            const int pc = wave + i * NW;  // wave-uniform
            if (pc < np) lds_dma16(g + (size_t)pc * 1024, l + (unsigned)pc * 1024u);
        }
    };
    stage_dma(t0, 0, true);
    if (t0 + TPS < t1) stage_dma(t0 + TPS, 1, true);

    u32x4 q[NQB][KSB];
    {
        const char *qb = qpack + (size_t)(qt * (QTB / QB) + wave * NQB) * QBLK_BYTES;
#pragma unroll
        for (int j = 0; j < NQB; ++j)
#pragma unroll
            for (int k = 0; k < KSB; ++k)
                q[j][k] = *(const u32x4 *)(qb + (size_t)j * QBLK_BYTES + ((size_t)(k * 2 + h) * QB + l31) * 16);
    }
    const unsigned qbase = (unsigned)(qt * QTB + wave * (NQB * QB) + l31);  // (N_pad < 2^31: check_common)
#pragma unroll
    for (int j = 0; j < NQB; ++j)
#pragma unroll
        for (int k = 0; k < KSB; ++k) asm volatile("" : "+v"(q[j][k]));  // compiler waits for q here

    int o = 0;
    while (meta[META_SEG + o + 1] <= t0) ++o;
    int seg_end = meta[META_SEG + o + 1];
    float ma[NQB], mb[NQB];  // two running minima per query block (even / odd accumulator registers)
    // FILTER: this lane's four queries' thresholds for the current object.  tq starts at the pair's current global
    // threshold -- the pre-pass's bound + slack, already tightened by the workgroups that finished before this one -- and
    // tightens to (smallest bf16 distance this lane meets) + slack: any row's bf16 distance bounds the minimum the same way
    // the pre-pass's does, and the slack 2.1 E(U) only shrinks with U.  Leaving an object, the lane publishes its threshold
    // (atomicMin on the key): the later rounds of workgroups filter against nearly the final minimum, which is what keeps
    // the candidate lists short.
    float tq[NQB], sq[NQB];
#pragma unroll
    for (int j = 0; j < NQB; ++j) ma[j] = mb[j] = MANET_WRONG_LABEL_PADDING_DISTANCE;
    // FILTER: the lowest key this lane KNOWS to be published for its four (query, object) pairs -- what it read or wrote
    // last -- lives in LDS (touched at exchanges only).  A lane publishes only below it: every lane every time was 13 M
    // atomics per launch, most of them no-ops at the L2 -- the filter pass's main overhead over the plain kernel.
    unsigned *pks = (unsigned *)(smem + 2 * STEP_BYTES + (size_t)REFINE_LDS_LIST * 8) + wave * 256 + lane;
    auto load_thr = [&](int obj) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NQB; ++j) {
            const unsigned kpub = FILTER ? (thr + (size_t)obj * N_pad)[qbase + 32u * j] : 0u;
            tq[j] = FILTER ? float_of(kpub) : 0.0f;
            sq[j] = FILTER ? (slack + (size_t)obj * N_pad)[qbase + 32u * j] : 0.0f;
            if (FILTER) pks[64 * j] = kpub;
        }
    };
    if (FILTER) load_thr(o);
    auto flush = [&](int obj) {
        if (FILTER) {
#pragma unroll
            for (int j = 0; j < NQB; ++j) {
                const float a = fminf(tq[j], __shfl_xor(tq[j], 32));
                const unsigned ka = key_of(a);
                if (h == 0 && a == a && ka < pks[64 * j]) {
                    atomicMin(&(thr + (size_t)obj * N_pad)[qbase + 32u * j], ka);
                    pks[64 * j] = ka;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NQB; ++j) {
            const float v = min3p(ma[j], mb[j], mb[j]);
            const float a = min3p(v, v, __shfl_xor(v, 32));
            if (h == 0) atomicMin(&(keys + (size_t)obj * N_pad)[qbase + 32u * j], key_of(a));
        }
    };
    (void)young_prio;
    // FILTER: minimum of a pass's 16 distances of one query block, and the slow path that appends the qualifying rows
    // (accumulator register r of lane (l31, h) is bank row (r & 3) + 8 (r >> 2) + 4 h of the pass)
    auto min16 = [&](const f32x16 &c) __attribute__((always_inline)) {
        float p = min3p(c[0], c[1], c[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) p = min3p(p, c[r], c[r + 1]);
        return min3p(p, c[15], c[15]);
    };
    // A qualifying row goes to the WAVE's own LDS list first -- its fill count lives in a scalar register, a hit costs a
    // ballot, a population count and a ds_write, no atomic (a returning global atomic per row stalled the wave for a memory
    // round trip: 2x the kernel's time; a returning LDS atomic: +45 %) -- and reaches the global list in bulk when the wave
    // is done, or earlier when a sub-list fills up (flush_sub).
    // The wave's list is four sub-lists, one per query block: the re-rank kernel's waves then read 32 neighbouring queries
    // (for a fixed channel one 128-byte line of the C-major embedding) instead of the wave's 128.
    constexpr int WL = REFINE_LDS_LIST / NW / NQB;  // entries per wave and query block
    uint2 *fl = (uint2 *)(smem + 2 * STEP_BYTES) + wave * (WL * NQB);
    int wl_n[NQB] = {0, 0, 0, 0};  // (wave-uniform)
    int wl_total = 0;              // (statistics)
    // sub-list j of this wave -> its bucket of the global list (block 16 qt + 4 wave + j): one returning atomic reserves the
    // places.  Called when the sub-list cannot take a block's hits (so nothing is dropped here: on spatially smooth
    // embeddings whole regions qualify while the thresholds are still loose) and at the wave's end.
    auto flush_sub = [&](int j) {
        const int cnt = wl_n[j];
        if (cnt == 0) return;  // (wave-uniform)
        // (the bucket's address arithmetic stays HERE, in scalar registers: hipcc otherwise hoists four per-lane 64-bit list
        // addresses + four LDS addresses out of the step loop, 15 VGPRs it can only spill at this kernel's register budget --
        // r3: scratch_load + s_waitcnt vmcnt(0) in the flush path, which drained the LDS-DMA prefetch on every flush)
        int b32 = qt * (QTB / QB) + wave * NQB + j, sub_off = j * WL;
        asm volatile("" : "+s"(b32), "+s"(sub_off));
        const long b = (long)b32;
        unsigned base = 0u;
        if (!(ABL & 64))
        if (lane == 0) base = atomicAdd(&bcnt[b], (unsigned)cnt) & 0x7fffffffu;  // (bit 31 = the bucket's "incomplete" mark)
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        uint2 *dst = list + b * bucket_cap + base;  // (wave-uniform)
        const long room = bucket_cap - (long)base;
        for (int i = lane; i < cnt; i += 64) {
            const uint2 e = fl[sub_off + i];  // (this wave's own ds_writes: ordered behind them in the LDS queue)
            if ((long)i < room)
                dst[i] = make_uint2((unsigned)((size_t)(e.x >> 16) * N_pad + (size_t)qt * QTB + (e.x & 0xffffu)), e.y);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the sub-list was read before it is refilled
        wl_n[j] = 0;
        wl_total += cnt;
    };
    auto emit_regs = [&](const f32x16 &c, float t, int j, int row0) __attribute__((always_inline)) {
        if (ABL & 128) {  // (timing: the tests without the listing)
            wl_n[j] += (int)(__popcll(__ballot(c[0] <= t)) & 1);
            return;
        }
        const unsigned lq = (unsigned)(wave * (NQB * QB) + l31 + 32 * j);  // query inside the workgroup's 512
        // (four registers at a time first: a block usually has its one or two hits in one group -- 8 tests instead of 16)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            // (the group's minimum is computed here, behind the branch: carrying the four group minima out of the pass's
            // own min16 -- 10 operations instead of 8 in front of the branch -- measured the same)
            const float gm = fminf(min3p(c[4 * g4], c[4 * g4 + 1], c[4 * g4 + 2]), c[4 * g4 + 3]);
            if (!__ballot(gm <= t)) continue;  // wave-uniform
#pragma unroll
            for (int r = 4 * g4; r < 4 * g4 + 4; ++r) {
                const bool hit = c[r] <= t;
                const unsigned long long m = __ballot(hit);
                if (m) {  // wave-uniform: most registers hold no qualifying row for any lane
                    const int slot = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int idx = wl_n[j] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (hit && idx < WL) fl[j * WL + idx] = make_uint2(((unsigned)o << 16) | lq, (unsigned)slot);
                    wl_n[j] += __popcll(m);
                }
            }
        }
    };
    // The hits of one 32 x 32 block go to sub-list j optimistically (the fast path is the loop above and one scalar compare).
    // If they did not fit behind what the sub-list held, nothing is lost: the sub-list is rolled back to its old fill,
    // flushed to its bucket, and the block is listed again into the empty sub-list.  A block with more hits than a whole
    // sub-list (> 128 of its 1 024 distances inside the threshold: embeddings the bf16 pass cannot tell apart) is listed as
    // ONE dense entry (r4; REFINE_DENSE_CAP per bucket, beyond that the bucket is marked incomplete and the rescue pass --
    // the exact fp32 kernel -- takes that query tile; through r4's first captures every such block went to the rescue pass:
    // 99 % of the tiles of a spatially smooth clip).
    auto emit = [&](const f32x16 &c, float t, int j, int row0, unsigned long long hm) __attribute__((always_inline)) {
        const int before = wl_n[j];
        // hm = the lanes (query, half of the pass's rows) whose 16 distances hold a qualifying one: more than
        // REFINE_DENSE_LANES of the 64 and the block goes dense without looking at single registers (video-like data:
        // neighbouring queries find their rows in the same pass -- the listing of such blocks was the filter pass's tail)
        const bool many = REFINE_DENSE_LANES < 64 && __popcll(hm) > REFINE_DENSE_LANES;
        if (!many) emit_regs(c, t, j, row0);
        if (many || wl_n[j] > WL || wl_n[j] - before > REFINE_DENSE_MIN) {  // (wave-uniform, rare)
            const int total = many ? __popcll(hm) : wl_n[j] - before;
            wl_n[j] = before;
            if (before >= WL || !(many || total > REFINE_DENSE_MIN)) flush_sub(j);  // (room for the dense entry / for the block's rows)
            if (many || total > REFINE_DENSE_MIN) {
                // a DENSE block (embeddings the bf16 pass cannot tell apart over this neighbourhood): one entry for its 1 024
                // distances, into the sub-list that was just emptied; the bucket's dense count lives behind the fill counts
                int b32 = qt * (QTB / QB) + wave * NQB + j, nb32 = (int)(N_pad >> 5);
                asm volatile("" : "+s"(b32), "+s"(nb32));
                if (lane == 0) {
                    if (atomicAdd(&bcnt[(long)nb32 + b32], 1u) >= (unsigned)REFINE_DENSE_CAP)
                        atomicOr(&bcnt[b32], 0x80000000u);  // (bit 31: incomplete -- the rescue pass takes this query tile)
                    fl[j * WL + wl_n[j]] = make_uint2(((unsigned)o << 16) | (unsigned)(wave * (NQB * QB) + 32 * j), REFINE_DENSE_BIT | (unsigned)row0);
                }
                wl_n[j] += 1;
                wl_total += total;  // (statistics: qualifying rows SEEN)
            } else
                emit_regs(c, t, j, row0);
        }
    };

    // this lane's fragment offset inside a tile image: unit (2k + h), row rb * 32 + l31
    const unsigned frag_off = (unsigned)((h * BT + l31) * 16);
    u32x4 F[KSB];
#define MANET_BF(x) __builtin_bit_cast(bf16x8_t, x)
#define MANET_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a_), MANET_BF(b_), c_, 0, 0, 0)
#define MANET_LOADF(k_, pass_base_) F[k_] = *(const u32x4 *)((pass_base_) + frag_off + (size_t)(k_) * (2 * BT * 16));
    // one pass = 32 bank rows x 128 queries: MFMAs of k-step k from F[k], then F[k] <- k-step k of the pass
    // at `next_base` (a tile's second row block is 32 * 16 bytes behind its first inside every unit)
#define MANET_PASS(next_base_, row0_)                                                              \
    {                                                                                              \
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};                                             \
        _Pragma("unroll") for (int k = 0; k < KSB; ++k)                                            \
        {                                                                                          \
            MANET_MFMA(F[k], q[0][k], c0);                                                         \
            MANET_MFMA(F[k], q[1][k], c1);                                                         \
            MANET_MFMA(F[k], q[2][k], c2);                                                         \
            MANET_MFMA(F[k], q[3][k], c3);                                                         \
            if (!(ABL & 8)) MANET_LOADF(k, next_base_);                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); /* 4 MFMA */                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* the refill right behind them */  \
        }                                                                                          \
        {                                                                                          \
        _Pragma("unroll") for (int r = 0; r < ((ABL & 4) ? 2 : 16); r += 4)                        \
        {                                                                                          \
            ma[0] = min3p(ma[0], c0[r], c0[r + 2]);                                                \
            mb[0] = min3p(mb[0], c0[r + 1], c0[r + 3]);                                            \
            ma[1] = min3p(ma[1], c1[r], c1[r + 2]);                                                \
            mb[1] = min3p(mb[1], c1[r + 1], c1[r + 3]);                                            \
            ma[2] = min3p(ma[2], c2[r], c2[r + 2]);                                                \
            mb[2] = min3p(mb[2], c2[r + 1], c2[r + 3]);                                            \
            ma[3] = min3p(ma[3], c3[r], c3[r + 2]);                                                \
            mb[3] = min3p(mb[3], c3[r + 1], c3[r + 3]);                                            \
        }                                                                                          \
        }                                                                                          \
    }

    // FILTER form of a pass: the threshold test of two query blocks runs UNDER the MFMAs of the other two -- the pass is two
    // half-passes (blocks 0,1 then blocks 2,3, KSB k-steps each), and the test of a half's accumulators sits in the basic
    // block of the NEXT half's MFMAs, in front of its (rarely taken) branch to the slow path.  (First form: the test of all
    // four blocks behind the pass's last MFMA, then the branch -- hipcc does not move MFMAs across a branch, so the ~50 VALU
    // instructions of the test ran exposed in every pass: the filter cost +13 % over the plain kernel.)  Accumulators: two
    // accumulate while two are tested, 64 registers as before.  pc2 / pc3 = the pending second half (blocks 2,3 of the
    // previous pass, rows pend_row0..), tested under the next pass's first half; an object change or the end of the split
    // tests it on the spot.
#define MANET_TEST2(ca_, cb_, ja_, jb_, row0_)                                                     \
    {                                                                                              \
        const float pa = min16(ca_), pb = min16(cb_);                                              \
        tq[ja_] = fminf(tq[ja_], pa + sq[ja_]);                                                    \
        tq[jb_] = fminf(tq[jb_], pb + sq[jb_]);                                                    \
        const bool ha = pa <= tq[ja_], hb = pb <= tq[jb_];                                         \
        if (!(ABL & 16) && __ballot(ha | hb)) { /* wave-uniform */                                 \
            const unsigned long long tk0_ = (ABL & 256) ? __builtin_readcyclecounter() : 0ull;     \
            const unsigned long long ma_ = __ballot(ha), mb_ = __ballot(hb);                       \
            if (ma_) emit(ca_, tq[ja_], ja_, (row0_), ma_);                                        \
            if (mb_) emit(cb_, tq[jb_], jb_, (row0_), mb_);                                        \
            if (ABL & 256) {                                                                       \
                dbg_cycles += __builtin_readcyclecounter() - tk0_;                                 \
                dbg_events += 1;                                                                   \
            }                                                                                      \
        }                                                                                          \
    }
#define MANET_PASS_F(next_base_, row0_)                                                            \
    {                                                                                              \
        f32x16 c0 = {0}, c1 = {0};                                                                 \
        _Pragma("unroll") for (int k = 0; k < KSB; ++k)                                            \
        {                                                                                          \
            MANET_MFMA(F[k], q[0][k], c0);                                                         \
            MANET_MFMA(F[k], q[1][k], c1);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); /* 2 MFMA */                        \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); /* a slice of the pending test */   \
        }                                                                                          \
        MANET_TEST2(pc2, pc3, 2, 3, pend_row0);                                                    \
        pc2 = (f32x16){0};                                                                         \
        pc3 = (f32x16){0};                                                                         \
        _Pragma("unroll") for (int k = 0; k < KSB; ++k)                                            \
        {                                                                                          \
            MANET_MFMA(F[k], q[2][k], pc2);                                                        \
            MANET_MFMA(F[k], q[3][k], pc3);                                                        \
            MANET_LOADF(k, next_base_);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* the refill right behind them */  \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                     \
        }                                                                                          \
        MANET_TEST2(c0, c1, 0, 1, (row0_));                                                        \
        pend_row0 = (row0_);                                                                       \
    }
    unsigned long long dbg_cycles = 0ull, dbg_events = 0ull;  // (ABL & 256: cycles this wave spent behind the tests, and how often)
    const unsigned long long dbg_t0 = (ABL & 256) ? __builtin_readcyclecounter() : 0ull;
    unsigned xk[NQB] = {0u, 0u, 0u, 0u};  // FILTER: threshold keys asked for in the previous exchange step, of object xo
    int xo = -1;
    f32x16 pc2, pc3;  // FILTER: the pending half (nothing pending: distances no threshold admits)
    int pend_row0 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) pc2[r] = pc3[r] = MANET_WRONG_LABEL_PADDING_DISTANCE;
    auto test_pending = [&]() __attribute__((always_inline)) {
        if (FILTER) {
            MANET_TEST2(pc2, pc3, 2, 3, pend_row0);
#pragma unroll
            for (int r = 0; r < 16; ++r) pc2[r] = pc3[r] = MANET_WRONG_LABEL_PADDING_DISTANCE;
        }
    };
    auto next_object = [&](int t) __attribute__((always_inline)) {
        if (t >= seg_end) {  // wave-uniform: tile t starts another object's rows
            test_pending();  // (FILTER: the previous pass's second half still belongs to the old object)
            flush(o);
#pragma unroll
            for (int j = 0; j < NQB; ++j) ma[j] = mb[j] = MANET_WRONG_LABEL_PADDING_DISTANCE;
            do { ++o; seg_end = meta[META_SEG + o + 1]; } while (t >= seg_end);
            if (FILTER) load_thr(o);
        }
    };

    // prologue: steps 0 and 1 in flight; publish step 0, fetch the first pass's fragments
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < KSB; ++k) MANET_LOADF(k, smem);

    int buf = 0;
    for (int t = t0; t < t1; t += TPS, buf ^= 1) {
        const char *cur = smem + (size_t)buf * STEP_BYTES;
        const char *nxt = smem + (size_t)(buf ^ 1) * STEP_BYTES;
        const bool has_b = (t + 1 < t1);
        // ---- tile A: rows 0-31 (refill: A rows 32-63), rows 32-63 (refill: B rows 0-31)
        next_object(t);
        if (FILTER) {
            MANET_PASS_F(cur + 32 * 16, t * BT);
            MANET_PASS_F(cur + TILE_BYTES, t * BT + 32);
        } else {
            MANET_PASS(cur + 32 * 16, t * BT);
            MANET_PASS(cur + TILE_BYTES, t * BT + 32);
        }
        // ---- tile B rows 0-31 (refill: B rows 32-63).  After this pass every fragment of `cur` is in
        // registers.
        if (has_b) {
            next_object(t + 1);
            if (FILTER) {
                MANET_PASS_F(cur + TILE_BYTES + 32 * 16, (t + 1) * BT);
            } else {
                MANET_PASS(cur + TILE_BYTES + 32 * 16, (t + 1) * BT);
            }
        }
        // ---- this wave's pieces of the next step have landed (issued one step ago) and its reads of `cur`
        // have returned; the barrier publishes the next buffer and frees `cur` for step + 2
        if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (FILTER) {
            // threshold exchange with the workgroups running beside this one on the same queries and object: every fourth
            // step the wave publishes its thresholds (non-returning atomics) and ASKS for the published ones; the answer is
            // taken one step later, right here behind the step's vmcnt(0) -- the loads never stall the wave (first form:
            // load and use in one place, a memory round trip exposed per exchange).
            if (xo == o) {
#pragma unroll
                for (int j = 0; j < NQB; ++j) {
                    tq[j] = fminf(tq[j], float_of(xk[j]));
                    if (h == 0 && xk[j] < pks[64 * j]) pks[64 * j] = xk[j];
                }
            }
            xo = -1;
            if (!(ABL & 32) && (((t - t0) / TPS) & REFINE_XCHG_MASK) == REFINE_XCHG_MASK) {
                flush(o);
                xo = o;
#pragma unroll
                for (int j = 0; j < NQB; ++j) xk[j] = (thr + (size_t)o * N_pad)[qbase + 32u * j];
            }
        }
        if (t + 2 * TPS < t1) stage_dma(t + 2 * TPS, buf);
        // ---- tile B rows 32-63 (refill: first pass of the next step; a stale read if there is none)
        if (has_b) {
            if (FILTER) {
                MANET_PASS_F(nxt, (t + 1) * BT + 32);
            } else {
                MANET_PASS(nxt, (t + 1) * BT + 32);
            }
        }
    }
    test_pending();
#undef MANET_PASS
#undef MANET_PASS_F
#undef MANET_TEST2
#undef MANET_LOADF
#undef MANET_MFMA
#undef MANET_BF
    flush(o);
    if (FILTER) {
        // the wave's remaining candidates -> the global list, which is one BUCKET per 32-query block (flush_sub).  The
        // re-rank kernel runs one workgroup per bucket with the block's 32 query vectors in LDS.  A bucket that outgrows its
        // capacity keeps counting (bcnt > bucket_cap marks it) and the rescue pass takes that query tile.
#pragma unroll
        for (int j = 0; j < NQB; ++j) flush_sub(j);
        if (lane == 0 && wl_total) atomicAdd(&stats[0], (unsigned long long)wl_total);  // (statistics only)
        if ((ABL & 256) && lane == 0) {  // timing experiments: stats[4..7] = listing cycles, listing events, wave cycles, waves
            atomicAdd(&stats[4], dbg_cycles);
            atomicAdd(&stats[5], dbg_events);
            atomicAdd(&stats[6], __builtin_readcyclecounter() - dbg_t0);
            atomicAdd(&stats[7], 1ull);
            atomicMax(&stats[8], dbg_cycles);
            atomicMax(&stats[9], __builtin_readcyclecounter() - dbg_t0);
        }
    }
}

// MANET_COMPUTE_BF16_REFINE: fp32-exact minima at bf16 cost (VERDICT r2 "next" #4).
//   IntVOS.py:81-85 is a MINIMUM over the bank, so any filter that keeps the true arg-min row may discard the rest:
//   1. pre-pass  : the plain bf16 kernel over every REFINE_SUB-th bank tile -> U(n,o), a bf16 distance of SOME row, i.e. an
//                  upper bound (to within the rounding bound E) of the minimum;
//   2. threshold : thr(n,o) = U + 2 E(n,o), E from |q_n|, max |k| and U (below): every row m with fp32 distance
//                  <= the fp32 minimum has bf16 distance <= thr;
//   3. filter    : the bf16 wide kernel over the WHOLE bank (FILTER form) appends the rows with bf16 distance <= thr to the
//                  pair's candidate list (REFINE_CAP slots);
//   4. re-rank   : the candidates are re-evaluated in the reference's fp32 arithmetic -- the oracle's fmaf chains, from
//                  the fp32 copy of the sorted bank -- and reduced; a pair whose list overflowed scans its object's rows
//                  instead.  The result is the fp32 kernel's, bit for bit.
// The bound: with u = 2^-9 (bf16 round-to-nearest), q~ = bf16(q), k~ = bf16(k), e = (q~ - q) - (k~ - k), |e| <= u (|q| + |k|):
//   | |q~ - k~|^2 - |q - k|^2 | = | 2 (q - k).e + |e|^2 | <= 2 u s sqrt(d) + u^2 s^2,   s = |q| + max |k|,  d = |q - k|^2
// plus the fp32 accumulation error of either evaluation (a = 8 (C + 8) 2^-24 s^2, generous).  For the two rows that matter
// (the pre-pass's best row and the true arg-min) d <= U + E0 + a with the crude E0 = (2u + u^2) s^2 + a, hence
//   E = 2 u s sqrt(max(U, 0) + E0 + a) + u^2 s^2 + a   bounds both, and thr = U + 2 E (x 1.05 against the bound's own rounding).

// the sorted bank once more as fp32 rows + |k|^2 (k-ascending fmaf chain over the fp32 values, as the f32 images carry
// it) + the bank's max |k|^2; grid = tiles, 256 threads
template <typename SRC>
__global__ __launch_bounds__(256) void bank_rows_f32_kernel(const SRC *__restrict__ src, long s_row, long s_c,
                                                            const int *__restrict__ src_of, int *__restrict__ meta, int C,
                                                            float *__restrict__ rows, float *__restrict__ norms)
{
    const long tile = blockIdx.x;
    if (tile >= meta[META_T]) return;
    for (int idx = threadIdx.x; idx < BT * C; idx += 256) {
        const int r = s_c == 1 ? idx / C : idx % BT, k = s_c == 1 ? idx % C : idx / BT;  // lanes along the source's fast axis
        const int sr = src_of[tile * BT + r];
        rows[(tile * BT + r) * C + k] = sr >= 0 ? emb_load(src, (long)sr * s_row + (long)k * s_c) : 0.0f;
    }
    if (threadIdx.x < BT) {
        const long slot = tile * BT + threadIdx.x;
        float n = MANET_WRONG_LABEL_PADDING_DISTANCE;
        const int sr = src_of[slot];
        if (sr >= 0) {
            n = 0.0f;
            for (int k = 0; k < C; ++k) {
                const float x = emb_load(src, (long)sr * s_row + (long)k * s_c);
                n = fmaf(x, x, n);
            }
            if (n == n) {
                atomicMax((unsigned *)&meta[META_KMAX], __float_as_uint(n));  // n >= 0: the bit pattern orders like the value
            } else {  // a NaN row: its object's minimum is NaN for every query (what MANET_COMPUTE_F32's min3p gives); keep it out
                int o = 0;  // of max |k|^2, which would void every OTHER object's threshold
                while (meta[META_SEG + o + 1] <= tile) ++o;
                meta[META_NAN + o] = 1;
            }
        }
        norms[slot] = n;
    }
}

// sub-sampled bank of the pre-pass: every REFINE_SUB-th tile of every object (at least one per non-empty object)
__global__ void sub_segments_kernel(int n_ids, const int *__restrict__ meta, int *__restrict__ sub_meta, int REFINE_SUB)
{
    if (threadIdx.x == 0) {
        int t = 0;
        for (int o = 0; o < n_ids; ++o) {
            sub_meta[META_SEG + o] = t;
            t += (meta[META_SEG + o + 1] - meta[META_SEG + o] + REFINE_SUB - 1) / REFINE_SUB;
        }
        sub_meta[META_SEG + n_ids] = t;
        sub_meta[META_T] = t;
    }
}
__global__ __launch_bounds__(256) void sub_copy_kernel(int n_ids, const int *__restrict__ meta,
                                                       const int *__restrict__ sub_meta, const char *__restrict__ bpack,
                                                       char *__restrict__ spack, long tile_bytes, int REFINE_SUB)
{
    const int j = blockIdx.x;
    if (j >= sub_meta[META_T]) return;
    int o = 0;
    while (sub_meta[META_SEG + o + 1] <= j) ++o;
    const long src_tile = meta[META_SEG + o] + (long)(j - sub_meta[META_SEG + o]) * REFINE_SUB;
    const uint4 *a = (const uint4 *)(bpack + src_tile * tile_bytes);
    uint4 *b = (uint4 *)(spack + (long)j * tile_bytes);
    for (long i = threadIdx.x; i < tile_bytes / 16; i += 256) b[i] = a[i];
}

// |q~_n|^2 out of the bf16 query image: the three bf16 pieces in k slots C+3 .. C+5 (see Geom)
__device__ __forceinline__ float query_norm_from_image(const char *qimg, long qblk_bytes, long n, int C)
{
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = C + 3 + i, u = 2 * (k >> 4) + ((k >> 3) & 1), e = k & 7;
        const unsigned short b = *(const unsigned short *)(qimg + (n >> 5) * qblk_bytes + ((long)u * QB + (n & 31)) * 16 + e * 2);
        s += bf2f(b);  // hi + mid + lo: exact in fp32
    }
    return s;
}

__global__ void refine_threshold_kernel(const unsigned *__restrict__ keys, const char *__restrict__ qimg, long qblk_bytes,
                                        int C, const int *__restrict__ meta, long N, long N_pad, int n_ids,
                                        unsigned *__restrict__ thr, float *__restrict__ slack, unsigned *__restrict__ keys2,
                                        unsigned long long *__restrict__ stats, unsigned *__restrict__ bcnt)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) stats[0] = stats[1] = 0ull;
    if (i >= 4 && i < 10) stats[i] = 0ull;  // (ablation builds: cycle counters of the filter pass)
    if (i <= 2 * (N_pad / QB)) bcnt[i] = 0u;  // (entries appended, dense entries appended, number of tiles to rescue)
    if (i >= (long)n_ids * N_pad) return;
    const long n = i % N_pad;
    const int o = (int)(i / N_pad);
    unsigned k2 = 0xffffffffu;  // "no row": the exact distances meet here by atomicMin
    const unsigned k = keys[i];
    float t = -INFINITY, sl = 0.0f;  // no row of this object anywhere: no candidates, the result is the padding distance
    if (k != 0xffffffffu && n < N) {
        const float U = float_of(k);
        const float u = 0.001953125f;  // 2^-9
        const float qn2 = query_norm_from_image(qimg, qblk_bytes, n, C);
        // NaN embeddings: a NaN in the query row, or in any row of this object, makes every distance of the pair's
        // minimum NaN in MANET_COMPUTE_F32 (min3p propagates it): the pair's exact key is NaN's (0, it wins every atomicMin) --
        // no threshold is meaningful for it, and none is needed
        if (qn2 != qn2 || meta[META_NAN + o]) k2 = 0u;
        const float qn = sqrtf(fmaxf(qn2, 0.0f)) * 1.004f;  // |q| <= |q~| / (1 - u)
        const float kn = sqrtf(__uint_as_float((unsigned)meta[META_KMAX])) * 1.0001f;
        const float s = qn + kn, s2 = s * s;
        const float a = 8.0f * (float)(C + 8) * 5.9604645e-8f * s2;
        const float E0 = (2.0f * u + u * u) * s2 + a;
        const float E = 2.0f * u * s * sqrtf(fmaxf(U, 0.0f) + E0 + a) + u * u * s2 + a;
        sl = 2.1f * E;
        t = U + sl;  // (NaN stays NaN: no candidates; such a pair's result was settled above)
    }
    keys2[i] = k2;
    thr[i] = key_of(t);  // as an order-preserving key: the filter pass's workgroups tighten it by atomicMin
    slack[i] = sl;       // ... to (smallest bf16 distance they met) + slack: E only shrinks with U
}

// MANET_EPI_REFINE_EXACT: no filter at all -- every 32-query block is marked incomplete, so the re-rank launch only writes the
// fp32 query images and the rescue pass (the exact fp32 kernel) takes every tile: the fp32 path's cost + ~0.03 ms, for callers
// that KNOW (ops.PreparedBank: from the previous frame's rescue share) that the bf16 pass cannot tell these embeddings apart
__global__ void refine_force_kernel(long N_pad, int n_ids, unsigned *__restrict__ keys2, unsigned long long *__restrict__ stats,
                                    unsigned *__restrict__ bcnt)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        stats[0] = 0ull;
        stats[1] = 1ull;
    }
    if (i < N_pad / QB) bcnt[i] = 0x80000000u;
    if (i == 0) bcnt[2 * (N_pad / QB)] = 0u;  // (the re-rank launch lists the tiles to rescue: all of them)
    if (i < (long)n_ids * N_pad) keys2[i] = 0xffffffffu;
}

#ifdef MANET_ABLATION
// timing experiments (manet_tune_set(MANET_TUNE_ABLATION, 256)): what the filter pass's waves spent behind their threshold tests
__global__ void refine_debug_kernel(const unsigned long long *stats)
{
    printf("filter pass: %llu waves, %.0f cycles each (longest %llu); listing path: %llu events, %.0f cycles each, %.1f %% of the wave "
           "cycles (most in one wave: %llu cycles)\n", stats[7], (double)stats[6] / (double)(stats[7] ? stats[7] : 1), stats[9], stats[5],
           (double)stats[4] / (double)(stats[5] ? stats[5] : 1), 100.0 * (double)stats[4] / (double)(stats[6] ? stats[6] : 1), stats[8]);
}
#endif

// how many 256-query tiles of the last filter pass went through the rescue pass -> out2 = {rescued, tiles} (device memory: the
// caller copies it out asynchronously and reads it a frame later)
__global__ void refine_rescued_kernel(const unsigned *__restrict__ bcnt, long tiles, long bucket_cap, int *__restrict__ out2)
{
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (long t = threadIdx.x; t < tiles; t += blockDim.x) {
        bool need = false;
        for (int i = 0; i < QT / QB; ++i) {
            const unsigned raw = bcnt[t * (QT / QB) + i];
            need = need || (raw >> 31) || (long)raw > bucket_cap;
        }
        mine += need ? 1 : 0;
    }
    atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = cnt;
        out2[1] = (int)tiles;
    }
}

// exact re-rank: one workgroup per bucket of the candidate list (= 32 neighbouring queries, every object), one thread per
// candidate {pair, bank slot}: the reference's fp32 distance (the oracle's fmaf chains, IntVOS.py:32-39) of that (query,
// bank row), reduced per pair by atomicMin on the order-preserving key; the usual finish kernel decodes the keys.
// The block's 32 query vectors are read ONCE, coalesced, into LDS as [channel][query] (same query: broadcast, different
// queries: different banks) -- with a flat list in emission order a wave's candidates spread over a filter workgroup's 512
// queries, so each of the 100 query loads touched ~14 cache lines: measured, the query gather cost as much as the bank-row
// gather (48 us each of the kernel's 87 at cfg3 shape; tools/ history in DESIGN.md 3.2b).  A lane's bank row is read 32
// channels per round, all eight 16-byte loads issued together: one 128-byte line's worth, crossing L2 -> L1 once.
template <typename SRC>
__global__ __launch_bounds__(256) void refine_rerank_kernel(const SRC *__restrict__ q, long q_sn, long q_sc,
                                                            const float *__restrict__ rows, const float *__restrict__ norms,
                                                            const uint2 *__restrict__ list, unsigned *__restrict__ bcnt,
                                                            long bucket_cap, long N, long N_pad, int C,
                                                            unsigned *__restrict__ keys2, unsigned long long *__restrict__ stats,
                                                            char *__restrict__ q32, int units32, long qblk_bytes32)
{
    extern __shared__ __attribute__((aligned(16))) char rr_smem[];
    float *qs = (float *)rr_smem;  // [C][QB]
    const long b = blockIdx.x;
    const unsigned raw = bcnt[b], have = raw & 0x7fffffffu;  // (bit 31: a block's hits were not listed, see emit)
    const int cnt = (long)have < bucket_cap ? (int)have : (int)bucket_cap;
    const int tid = threadIdx.x;
    if (((long)have > bucket_cap || (raw >> 31)) && tid == 0) stats[1] = 1ull;  // (statistics; the rescue pass looks at bcnt itself)
    // the rescue pass (the exact fp32 kernel) takes every 256-query tile that holds an incomplete bucket: the blocks of such
    // a tile also write their fp32 operand image here, where the block's queries are in LDS anyway (a separate pack launch
    // cost 5 us per healthy frame)
    bool rescue = false;
    {
        const long t0 = b / (QT / QB) * (QT / QB);
        for (int i = 0; i < QT / QB; ++i) {
            const unsigned r2 = bcnt[t0 + i];
            rescue = rescue || (r2 >> 31) || (long)r2 > bucket_cap;
        }
    }
    // this workgroup's share of the bucket: entries z, z + zn, ... -- zn of the launch's REFINE_RZ workgroups per bucket take
    // part: one per 512 entries, a dense entry (a 32 x 32 x C matrix tile for one wave) counted as 64 listed rows.  (All four on
    // every bucket: 4 x the query-block loads and a quarter of the work each -- +8 us of the re-rank on video-like data, where a
    // bucket holds ~320 rows; one: -45 us on smooth data, where the longest bucket is the launch's time.)
    const int z = blockIdx.y;
    int zn = (int)(((long)cnt + 64l * (long)bcnt[N_pad / QB + b] + 511) / 512);
    zn = zn < 1 ? 1 : (zn > REFINE_RZ ? REFINE_RZ : zn);
    if (rescue && tid == 0 && z == 0 && b % (QT / QB) == 0) {  // the tile's first block lists it for the rescue launch
        unsigned *rs = bcnt + 2 * (N_pad / QB);
        rs[1 + atomicAdd(&rs[0], 1u)] = (unsigned)(b / (QT / QB));
    }
    if (z > 0 && (rescue || z >= zn)) return;  // (a rescued tile's candidates are moot; its image is written by z == 0)
    if (cnt == 0 && !rescue) return;
    __shared__ int dense_n;
    __shared__ int dense_at[REFINE_DENSE_CAP];
    if (tid == 0) dense_n = 0;
    for (int idx = tid; idx < QB * C; idx += 256) {
        long n = b * QB + (idx & (QB - 1));
        n = n < N ? n : N - 1;  // (padding queries have no candidates)
        qs[idx] = emb_load(q + n * q_sn, (long)(idx / QB) * q_sc);
    }
    __syncthreads();
    if (rescue) {  // (block-uniform; z == 0) the block's image as pack_rows_kernel<32,32> writes it for MANET_COMPUTE_F32
        char *out0 = q32 + b * qblk_bytes32;
        for (int item = tid; item < units32 * QB; item += 256) {
            const int r = item & (QB - 1), u = item / QB;
            const int k0 = 8 * (u >> 1) + (u & 1);  // image_unit_f32: k0, k0 + 2, k0 + 4, k0 + 6
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (k0 + 2 * e < C) ? qs[(k0 + 2 * e) * QB + r] : 0.0f;
            *(f32x4 *)(out0 + ((long)u * QB + r) * 16) = v;
        }
        if (tid < QB) {
            float nrm = 0.0f;
            for (int k = 0; k < C; ++k) nrm = fmaf(qs[k * QB + tid], qs[k * QB + tid], nrm);
            *(float *)(out0 + (long)units32 * QB * 16 + tid * 4) = nrm;
        }
        return;  // (the rescue pass evaluates every pair of this tile: its candidates are moot)
    }
    for (int i = z + zn * tid; i < cnt; i += zn * 256) {
        const uint2 e = list[b * bucket_cap + i];
        if (e.y & REFINE_DENSE_BIT) {  // a dense block: the whole workgroup evaluates it below
            if (!rescue) {             // (a tile that is rescued anyway gets every distance from the fp32 kernel)
                const int at = atomicAdd(&dense_n, 1);
                if (at < REFINE_DENSE_CAP) dense_at[at] = i;  // (more than the cap: the filter pass marked the bucket incomplete)
            }
            continue;
        }
        const float *x = qs + (int)((e.x % (unsigned long)N_pad) & (QB - 1));
        const float *kr = rows + (long)e.y * C;
        float xs = 0.0f, mm = 0.0f;
        int k = 0;
        for (; (C & 3) == 0 && k + 32 <= C; k += 32) {
            f32x4 y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = *(const f32x4 *)(kr + k + 4 * j);  // (rows are 16-byte aligned: C % 4 == 0)
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float xv = x[(k + j) * QB];
                xs = fmaf(xv, xv, xs);
                mm = fmaf(xv, y[j >> 2][j & 3], mm);
            }
        }
        for (; k < C; ++k) {
            const float xv = x[k * QB];
            xs = fmaf(xv, xv, xs);
            mm = fmaf(xv, kr[k], mm);
        }
        atomicMin(keys2 + e.x, key_of(fmaf(-2.0f, mm, xs + norms[e.y])));  // IntVOS.py:39
    }
    __syncthreads();
    // dense entries: 32 bank rows x this block's 32 queries each -- exactly one 32 x 32 tile of the fp32 matrix pipe: a WAVE
    // per entry, A = the 32 bank rows (lane = row, read from the fp32 copy of the bank), B = the block's queries from LDS,
    // ceil(C / 2) v_mfma_f32_32x32x2_f32 in k order -- the fp32 kernel's own chain for these pairs, hence its bits.  (First
    // form: VALU, a thread per (query, 4 rows): 1.4 us per entry per workgroup, a quarter of the CU's rate.)
    const int nd = dense_n < REFINE_DENSE_CAP ? dense_n : REFINE_DENSE_CAP;
    if (nd > 0) {
        const int lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
        const float *x = qs + l31;
        float xs = 0.0f;
        for (int k = 0; k < C; ++k) xs = fmaf(x[k * QB], x[k * QB], xs);
        for (int d = w; d < nd; d += 4) {
            const uint2 e = list[b * bucket_cap + dense_at[d]];
            const long slot0 = (long)(e.y & ~REFINE_DENSE_BIT);
            const float *kr = rows + (slot0 + l31) * C;
            f32x16 acc = {0};
            if ((C & 3) == 0) {
                // batches of five 16-byte loads = ten k-steps, two batches in flight: the loads of batch i + 1 are issued before
                // the MFMAs of batch i (a lone wave has nothing else to hide a row load behind: the first form -- load, wait,
                // ten MFMAs, five times per entry -- spent most of an entry waiting)
                f32x4 ya[5], yb[5];
                auto load5 = [&](f32x4 (&y)[5], int k0) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) y[j] = (k0 + 4 * j < C) ? *(const f32x4 *)(kr + k0 + 4 * j) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                };
                auto mfma5 = [&](const f32x4 (&y)[5], int k0) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        if (k0 + 4 * j < C) {  // (wave-uniform)
                            const int k = k0 + 4 * j + hh;  // this half's k of the step: even lanes' half 2s, the other 2s + 1
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hh ? y[j][1] : y[j][0], x[k * QB], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hh ? y[j][3] : y[j][2], x[(k + 2) * QB], acc, 0, 0, 0);
                        }
                    }
                };
                load5(ya, 0);
                for (int k0 = 0; k0 < C; k0 += 40) {
                    if (k0 + 20 < C) load5(yb, k0 + 20);
                    mfma5(ya, k0);
                    if (k0 + 40 < C) load5(ya, k0 + 40);
                    if (k0 + 20 < C) mfma5(yb, k0 + 20);
                }
            } else {
                for (int k0 = 0; k0 < C; k0 += 2) {
                    const int k = k0 + hh;
                    const bool ok = k < C;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? kr[k] : 0.0f, ok ? x[k * QB] : 0.0f, acc, 0, 0, 0);
                }
            }
            // C/D layout: column = lane & 31 (query), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (bank row of the pass)
            unsigned best = 0xffffffffu;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const f32x4 nv = *(const f32x4 *)(norms + slot0 + 8 * tq + 4 * hh);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned kd = key_of(fmaf(-2.0f, acc[4 * tq + i], xs + nv[i]));  // IntVOS.py:39
                    best = kd < best ? kd : best;
                }
            }
            const unsigned other = (unsigned)__shfl_xor((int)best, 32);
            best = other < best ? other : best;
            if (hh == 0) atomicMin(keys2 + e.x + l31, best);  // (e.x = the pair of the block's first query)
        }
    }
}


// decode + (sigmoid-0.5)*2 (IntVOS.py:611-612) + min-merge with the stored map (IntVOS.py:620-622)
// MANET_EPI_KEYS_ARMED: the keys are put back to "no candidate" as they are read (every key, padding rows included), so
// the next armed call on the same match workspace needs no fill launch
__global__ void global_finish_kernel(unsigned *__restrict__ keys, long N, long N_pad, int n_ids,
                                     int flags, float *__restrict__ out, float *__restrict__ mem)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((flags & MANET_EPI_KEYS_ARMED) && i >= N * n_ids && i < N_pad * n_ids) {
        const long j = i - N * n_ids;  // the padding rows' keys: n = N + j / n_ids
        keys[(size_t)(j % n_ids) * N_pad + N + j / n_ids] = 0xffffffffu;
    }
    if (i >= N * n_ids) return;
    long n = i / n_ids;
    int o = (int)(i - n * n_ids);
    unsigned k = keys[(size_t)o * N_pad + n];
    if (flags & MANET_EPI_KEYS_ARMED) keys[(size_t)o * N_pad + n] = 0xffffffffu;
    // an object with no bank row keeps the initial key: padding distance (IntVOS.py:81-83)
    float g = (k == 0xffffffffu) ? MANET_WRONG_LABEL_PADDING_DISTANCE : float_of(k);
    if (flags & MANET_EPI_NORMALIZE) g = manet_normalize_dist(g);
    if (mem) {
        float mv = mem[i];
        g = (g <= mv) ? g : mv;
        mem[i] = g;
    }
    out[i] = g;
}

// top-k epilogue (IntVOS.py:87-94): merge the splits' lists, keep the k smallest; entries >= 1e20
// are the reference's masked rows (here: tile padding rows / missing rows): replaced by
// pad = max(valid distances, and 0 if any entry is invalid) -- `dists * valid_mask` zeroes them
// before the max -- then the mean.  Then the same normalise / merge as the k=1 path.
__global__ void global_finish_topk_kernel(const float *__restrict__ topk, int S, int k_nn, long N, long N_pad,
                                          int n_ids, int flags, float *__restrict__ out,
                                          float *__restrict__ mem)
{
    constexpr int K = MANET_MAX_KNN;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * n_ids) return;
    long n = i / n_ids;
    int o = (int)(i - n * n_ids);
    float best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = INFINITY;
    for (int sp = 0; sp < S; ++sp) {
        const float *p = topk + (((size_t)sp * n_ids + o) * N_pad + n) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) topk_insert<K>(best, p[j]);
    }
    float pad = -INFINITY, sum = 0.0f;
    bool any_invalid = false;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k_nn) {
            if (best[j] < MANET_WRONG_LABEL_PADDING_DISTANCE) pad = fmaxf(pad, best[j]);
            else any_invalid = true;
        }
    if (any_invalid) pad = fmaxf(pad, 0.0f);
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < k_nn) sum += (best[j] < MANET_WRONG_LABEL_PADDING_DISTANCE) ? best[j] : pad;
    float g = sum / (float)k_nn;
    if (flags & MANET_EPI_NORMALIZE) g = manet_normalize_dist(g);
    if (mem) {
        float mv = mem[i];
        g = (g <= mv) ? g : mv;
        mem[i] = g;
    }
    out[i] = g;
}

__global__ void normalize_merge_kernel(float *__restrict__ x, float *__restrict__ mem, long n,
                                       int normalize)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = x[i];
    if (normalize) g = manet_normalize_dist(g);
    if (mem) {
        float mv = mem[i];
        g = (g <= mv) ? g : mv;
        mem[i] = g;
    }
    x[i] = g;
}

// bank pack (64-row tiles) and query pack (32-row blocks; keys != nullptr: also resets the rows' match keys)
// for either embedding storage type
int launch_bank_pack(const void *src, int emb_dtype, long s_row, long s_c, const int *src_of, const int *meta, long n_rows,
                     int C, const Geom &G, char *dst, long ntiles, hipStream_t st)
{
    if (ntiles <= 0) return MANET_OK;
    const size_t lds = (size_t)BT * (G.kpad + 1) * sizeof(float) + 2 * BT * sizeof(int);
    if (emb_dtype == MANET_EMB_F32)
        hipLaunchKernelGGL((pack_rows_kernel<BT, BT, float>), dim3((unsigned)ntiles), dim3(256), lds, st, (const float *)src,
                           s_row, s_c, src_of, meta, n_rows, C, G.compute, G.units, G.kpad, dst, (long)G.tile_bytes,
                           MANET_WRONG_LABEL_PADDING_DISTANCE, (unsigned *)nullptr, 0L, 0);
    else if (emb_dtype == MANET_EMB_BF16)
        hipLaunchKernelGGL((pack_rows_kernel<BT, BT, unsigned short>), dim3((unsigned)ntiles), dim3(256), lds, st,
                           (const unsigned short *)src, s_row, s_c, src_of, meta, n_rows, C, G.compute, G.units, G.kpad,
                           dst, (long)G.tile_bytes, MANET_WRONG_LABEL_PADDING_DISTANCE, (unsigned *)nullptr, 0L, 0);
    else
        return manet_set_error(MANET_E_INVALID, "embedding dtype %d (MANET_EMB_F32 / MANET_EMB_BF16)", emb_dtype);
    return MANET_OK;
}

int launch_query_pack(const void *src, int emb_dtype, long s_row, long s_c, long N, long N_pad, int C, const Geom &G,
                      char *dst, unsigned *keys, int n_ids, hipStream_t st)
{
    constexpr int SR = QB;  // staged rows per workgroup (two blocks per workgroup measured slower: 22 vs 16 us at 480p)
    const size_t lds = (size_t)SR * (G.kpad + 1) * sizeof(float) + 2 * SR * sizeof(int);
    const unsigned blocks = (unsigned)(N_pad / SR);
    if (emb_dtype == MANET_EMB_F32)
        hipLaunchKernelGGL((pack_rows_kernel<SR, QB, float>), dim3(blocks), dim3(256), lds, st, (const float *)src, s_row, s_c,
                           (const int *)nullptr, (const int *)nullptr, N, C, G.compute, G.units, G.kpad, dst,
                           (long)G.qblk_bytes, 0.0f, keys, N_pad, n_ids);
    else if (emb_dtype == MANET_EMB_BF16)
        hipLaunchKernelGGL((pack_rows_kernel<SR, QB, unsigned short>), dim3(blocks), dim3(256), lds, st,
                           (const unsigned short *)src, s_row, s_c, (const int *)nullptr, (const int *)nullptr, N, C,
                           G.compute, G.units, G.kpad, dst, (long)G.qblk_bytes, 0.0f, keys, N_pad, n_ids);
    else
        return manet_set_error(MANET_E_INVALID, "embedding dtype %d (MANET_EMB_F32 / MANET_EMB_BF16)", emb_dtype);
    return MANET_OK;
}

// arg-min form: decode (distance key, bank slot) -> raw distance + source row of the caller's bank (-1: no row)
__global__ void global_finish_arg_kernel(const unsigned long long *__restrict__ keys64, const int *__restrict__ src_of,
                                         const int *__restrict__ meta, long N, long N_pad, int n_ids,
                                         float *__restrict__ out, int *__restrict__ arg)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * n_ids) return;
    long n = i / n_ids;
    int o = (int)(i - n * n_ids);
    const unsigned long long k = keys64[(size_t)o * N_pad + n];
    const unsigned hi = (unsigned)(k >> 32), slot = (unsigned)k;
    out[i] = (k == ~0ull) ? MANET_WRONG_LABEL_PADDING_DISTANCE : float_of(hi);
    arg[i] = (slot != 0xffffffffu && (long)slot < (long)meta[META_T] * BT) ? src_of[slot] : -1;
}

// Backward of the k = 1 global match w.r.t. both embeddings (reference: autograd through IntVOS.py:32-39 and the
// torch.min of :84): with m* = arg[n][o] and g = grad_out[n][o],
//   d/dq_n += 2 g (q_n - k_m*),   d/dk_m* += 2 g (k_m* - q_n)       (d = |q|^2 + |k|^2 - 2 q.k)
// thread = (query n, channel c), lanes along n; grad_bank is accumulated with atomicAdd (zeroed by the caller side
// of this entry point).  Element strides for every tensor.
__global__ void global_match_backward_kernel(const float *__restrict__ q, long q_sn, long q_sc,
                                             const float *__restrict__ k, long k_sm, long k_sc,
                                             const int *__restrict__ arg, const float *__restrict__ gout, long N,
                                             int C, int n_ids, float *__restrict__ gq, long gq_sn, long gq_sc,
                                             float *__restrict__ gk, long gk_sm, long gk_sc)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const long n = i % N;
    const int c = (int)(i / N);
    const float qv = q[n * q_sn + (long)c * q_sc];
    float acc = 0.0f;
    for (int o = 0; o < n_ids; ++o) {
        const int m = arg[n * n_ids + o];
        const float g = gout[n * n_ids + o];
        if (m < 0 || g == 0.0f) continue;
        const float t = 2.0f * g * (qv - k[(long)m * k_sm + (long)c * k_sc]);
        acc += t;
        if (gk) atomicAdd(gk + (long)m * gk_sm + (long)c * gk_sc, -t);  // (NULL: the bank needs no gradient)
    }
    if (gq) gq[n * gq_sn + (long)c * gq_sc] = acc;
}

__global__ void zero_strided_kernel(float *__restrict__ p, long n0, long n1, long s0, long s1)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0 * n1) return;
    p[(i % n0) * s0 + (i / n0) * s1] = 0.0f;
}

int check_common(int64_t N, int64_t M0, int C, int n_ids, int k_nn, int compute)
{
    if (N <= 0 || M0 < 0) return manet_set_error(MANET_E_INVALID, "N=%lld M0=%lld", (long long)N, (long long)M0);
    if (M0 >= (1LL << 31) - 64 * (MANET_MAX_IDS + 1) || N >= (1LL << 31) - QT)
        return manet_set_error(MANET_E_INVALID, "N or M0 too large for 32-bit row indices");
    if (C <= 0 || C > MANET_MAX_C) return manet_set_error(MANET_E_INVALID, "C=%d (supported 1..%d)", C, MANET_MAX_C);
    if (n_ids <= 0 || n_ids > MANET_MAX_IDS)
        return manet_set_error(MANET_E_INVALID, "n_ids=%d (supported 1..%d)", n_ids, MANET_MAX_IDS);
    if (k_nn < 1 || k_nn > MANET_MAX_KNN)
        return manet_set_error(MANET_E_INVALID, "k_nn=%d (supported 1..%d)", k_nn, MANET_MAX_KNN);
    if (k_nn > 1 && compute != MANET_COMPUTE_F32)
        return manet_set_error(MANET_E_INVALID, "k_nn > 1 needs MANET_COMPUTE_F32");
    if (compute != MANET_COMPUTE_F32 && compute != MANET_COMPUTE_BF16 && compute != MANET_COMPUTE_BF16X3 &&
        compute != MANET_COMPUTE_BF16_REFINE)
        return manet_set_error(MANET_E_INVALID, "compute=%d (MANET_COMPUTE_F32 / _BF16 / _BF16X3 / _BF16_REFINE)", compute);
    if (compute == MANET_COMPUTE_BF16_REFINE && C + BF16_SPECIAL > 112)
        return manet_set_error(MANET_E_INVALID, "MANET_COMPUTE_BF16_REFINE supports C <= 106 (the wide bf16 kernel)");
    return MANET_OK;
}

// the kernels' `block_map` argument: mapping bits + small_S (see split_of_block).
// Mapping (r3b): XCD-aware with PB splits fastest, PB = how many of this bank's splits fit an XCD's L2 side by side
// (<= 3; 2.4 MB of the 4 MB: the query operands in flight want the rest).  The workgroups of an XCD that are resident
// together then cover 64 / PB query tiles x PB splits instead of 64 x 1, a query operand is fetched over the fabric once
// per PB workgroups, and the splits still stream through the L2 once.  Measured (FETCH_SIZE x 2, kernel time unchanged
// within 0.3 %): cfg2 fp32 692 -> 420 MB per launch (PB = 2; PB = 3: 538), cfg3 bf16 277 -> 142 MB (PB = 3; 4: 168),
// cfg5 bf16 1.26 GB at PB = 1, which its 1.6 MB splits keep (PB = 3: 1.64 GB).
thread_local double tl_bank_bytes_hint = 0.0;  // packed bytes of the bank about to be matched (set by the entry points)
int block_map_arg(int nQT, int slots, int S = 0)
{
    int small_S = slots / (nQT > 0 ? nQT : 1);
    if (small_S < 1) small_S = 1;
    if (small_S > 4096) small_S = 4096;
    int bm = manet_tune_get(MANET_TUNE_BLOCK_MAP, -1);
    if (bm < 0) {
        bm = 0;
        if (S >= 16 && (S & 7) == 0 && tl_bank_bytes_hint > 0.0) {
            int pb = (int)(2.4e6 / (tl_bank_bytes_hint / S));
            pb = pb > 3 ? 3 : pb;
            if (pb > S / 8) pb = S / 8;
            if (pb >= 2) bm = pb + 2;
        }
    }
    return (bm & 0xff) | (small_S << 8);
}

template <int KS, int KNN, bool ARG = false, bool NTH = false>
void launch_main_f32(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S,
                     long N_pad, unsigned *keys, float *topk, hipStream_t st)
{
    size_t lds = 2 * bank_tile_bytes((KS + 3) / 4);
    // per call (cheap, host side): the attribute is per device and the library keeps no state
    (void)hipFuncSetAttribute((const void *)global_match_f32_kernel<KS, KNN, ARG, NTH>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    manet_profile_record(st, true);
    hipLaunchKernelGGL((global_match_f32_kernel<KS, KNN, ARG, NTH>), dim3((unsigned)(nQT * S)), dim3(256), lds, st, qpack,
                       bpack, meta, n_ids, nQT, S, N_pad, keys, topk, block_map_arg(nQT, 512, S));
    manet_profile_record(st, false);
}

// the exact fp32 kernel as MANET_COMPUTE_BF16_REFINE's rescue pass (workgroups of complete query tiles return at once)
template <int KS>
void launch_rescue_f32_pipe(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S, long N_pad,
                            unsigned *keys, const unsigned *bcnt, long bucket_cap, bool listed, hipStream_t st)
{
    size_t lds = 2 * bank_tile_bytes((KS + 3) / 4);
    (void)hipFuncSetAttribute((const void *)global_match_f32_pipe_kernel<KS, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    // listed: the workgroups are dealt to the tiles the re-rank launch listed (few, usually none); else (every tile is
    // matched: MANET_EPI_REFINE_EXACT) the fp32 kernel's own block map
    hipLaunchKernelGGL((global_match_f32_pipe_kernel<KS, true>), dim3((unsigned)(nQT * S)), dim3(256), lds, st, qpack, bpack, meta,
                       n_ids, nQT, S, N_pad, keys,
                       listed ? (RESCUE_LISTED | ((manet_tune_get(MANET_TUNE_RESCUE_SPLITS, 0) & BLOCK_MAP_RESCUE_MASK) << 8)) : block_map_arg(nQT, 512, S),
                       bcnt, bucket_cap);
}

template <int KS>
void launch_main_f32_pipe(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S, long N_pad,
                          unsigned *keys, hipStream_t st)
{
    size_t lds = 2 * bank_tile_bytes((KS + 3) / 4);
    (void)hipFuncSetAttribute((const void *)global_match_f32_pipe_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    manet_profile_record(st, true);
    const int one_round = manet_tune_get(MANET_TUNE_ONE_ROUND, 0) == 1 ? 0 : ONE_ROUND_OK;  // (1: the host's splits always, A/B timing)
    hipLaunchKernelGGL((global_match_f32_pipe_kernel<KS>), dim3((unsigned)(nQT * S)), dim3(256), lds, st, qpack, bpack, meta,
                       n_ids, nQT, S, N_pad, keys, block_map_arg(nQT, 512, S) | one_round, (const unsigned *)nullptr, 0L);
    manet_profile_record(st, false);
}

template <int KSB, bool X3, int TPS, bool DMA>
void launch_main_bf16_v(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S, long N_pad,
                        unsigned *keys, int young_prio, hipStream_t st)
{
    size_t lds = 2 * TPS * bank_tile_bytes_u(2 * KSB * (X3 ? 2 : 1), false);
    (void)hipFuncSetAttribute((const void *)global_match_bf16_kernel<KSB, X3, TPS, DMA>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    manet_profile_record(st, true);
    hipLaunchKernelGGL((global_match_bf16_kernel<KSB, X3, TPS, DMA>), dim3((unsigned)(nQT * S)), dim3(512), lds, st,
                       qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, block_map_arg(nQT, 256, S),
                       young_prio);
    manet_profile_record(st, false);
}

// Shipped: plain bf16 -> global_match_bf16_wide_kernel (global_match_bf16_pipe_kernel for C > 106);
// split-bf16 -> global_match_bf16_kernel<KSB, true, 1, true> (1 tile per step, asm LDS-DMA staging).
// MANET_TUNE_BF16_VARIANT (experiments; bit field, none of it changes a workspace layout):
//   bits 0-1  global_match_bf16_kernel: tiles per step 0 = default, 1 = one, 2 = four (two for split-bf16)
//   bit 2     global_match_bf16_kernel: register staging instead of LDS-DMA
//   bit 3     static s_setprio 1 for waves 4-7 (8-wave kernels)
//   bit 4     plain bf16 on global_match_bf16_kernel (the un-pipelined loop)
//   bit 6     plain bf16 on global_match_bf16_pipe_kernel (8 waves, 64 x 64 wave tile)
// MANET_TUNE_ABLATION (key 3) selects the timing-ablation instantiations of the two pipelined kernels.
template <int KSB, bool X3>
void launch_main_bf16(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S, long N_pad,
                      unsigned *keys, hipStream_t st, int prof_channel = 0)
{
#ifdef MANET_ABLATION
    const int v = manet_tune_get(MANET_TUNE_BF16_VARIANT, 0);
#else
    const int v = 0;  // (the default build carries the shipped kernels only: make EXTRA=-DMANET_ABLATION for the rest)
#endif
    const int flat = (v >> 4) & 1;
    int prio = (v >> 3) & 1;
    if constexpr (!X3) {
        if (!flat) {  // plain-bf16 kernels: software-pipelined fragments, barrier in front of the step's last pass
            // the 8-wave 64x64-wave-tile form: KSB = 9 (C > 106), whose 144 operand VGPRs do not fit the wide form; tuning
            constexpr bool NARROW_ONLY = (KSB == 9);
            const bool narrow = NARROW_ONLY || ((v >> 6) & 1);
            const void *fn = nullptr;
#ifdef MANET_ABLATION
            const int abl = KSB == 7 ? manet_tune_get(MANET_TUNE_ABLATION, 0) : 0;  // timing experiments only
            if (narrow) {
#define MANET_PK(AB_) \
    if (abl == AB_) fn = (const void *)global_match_bf16_pipe_kernel<KSB, (KSB == 7 ? AB_ : 0)>;
                MANET_PK(0) MANET_PK(1) MANET_PK(2) MANET_PK(4) MANET_PK(8) MANET_PK(15)
#undef MANET_PK
                if (!fn) fn = (const void *)global_match_bf16_pipe_kernel<KSB, 0>;
            } else if constexpr (!NARROW_ONLY) {
#define MANET_WK(AB_) \
    if (abl == AB_) fn = (const void *)global_match_bf16_wide_kernel<KSB, (KSB == 7 ? AB_ : 0)>;
                MANET_WK(0) MANET_WK(1) MANET_WK(2) MANET_WK(4) MANET_WK(8) MANET_WK(15)
#undef MANET_WK
                if (!fn) fn = (const void *)global_match_bf16_wide_kernel<KSB, 0>;
            }
#else
            if constexpr (NARROW_ONLY) fn = (const void *)global_match_bf16_pipe_kernel<KSB, 0>;
            else fn = (const void *)global_match_bf16_wide_kernel<KSB, 0>;
#endif
            const unsigned threads = narrow ? 512 : 256;
            const size_t lds = (size_t)2 * 2 * bank_tile_bytes_u(2 * KSB, false);
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int bm = block_map_arg(nQT, narrow ? 256 : 512, S);
            unsigned *no_thr = nullptr;
            const float *no_slack = nullptr;
            unsigned long long *no_stats = nullptr;
            uint2 *no_list = nullptr;
            unsigned *no_bcnt = nullptr;
            long no_cap = 0;
            // (the narrow kernel takes the first ten arguments; the wide one also the FILTER form's five, unused here)
            void *args[] = {(void *)&qpack, (void *)&bpack, (void *)&meta, (void *)&n_ids, (void *)&nQT, (void *)&S,
                            (void *)&N_pad, (void *)&keys, (void *)&bm, (void *)&prio, (void *)&no_thr, (void *)&no_slack,
                            (void *)&no_stats, (void *)&no_list, (void *)&no_cap, (void *)&no_bcnt};
            manet_profile_record(st, true, prof_channel);
            (void)hipLaunchKernel(fn, dim3((unsigned)(nQT * S)), dim3(threads), args, lds, st);
            manet_profile_record(st, false, prof_channel);
            return;
        }
    }
#ifdef MANET_ABLATION
    const int tps = v & 3, reg = (v >> 2) & 1;
#define MANET_BV(TPS_)                                                                                         \
    if (reg) launch_main_bf16_v<KSB, X3, TPS_, false>(qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, prio, st); \
    else launch_main_bf16_v<KSB, X3, TPS_, true>(qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, prio, st);
    if (tps == 1) { MANET_BV(1) }
    else if (tps == 2) { MANET_BV((X3 ? 2 : 4)) }
    else { MANET_BV((X3 ? 1 : 2)) }
#undef MANET_BV
#else
    // split-bf16: global_match_bf16_kernel<KSB, true, 1, true> (1 tile per step, asm LDS-DMA staging)
    if constexpr (X3) launch_main_bf16_v<KSB, true, 1, true>(qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, prio, st);
#endif
}

// MANET_COMPUTE_BF16_REFINE on a prepared bank (see the kernels' header comment).  `qimg` = the query's bf16 operand
// image (made by the caller of this function), `qraw` = the same query as stored (fp32 / bf16, element strides).
int run_refine(const char *qimg, const void *qraw, int q_dtype, long q_sn, long q_sc, const char *bws, const BankLayout &BL,
               char *mws, const MatchLayout &ML, long N, int C, int n_ids, float *out, float *mem, int flags, hipStream_t st)
{
    const int *meta = (const int *)(bws + BL.off_meta), *sub_meta = (const int *)(bws + BL.off_sub_meta);
    unsigned *keys = (unsigned *)(mws + ML.off_keys), *keys2 = (unsigned *)(mws + ML.off_keys2);
    unsigned *thr = (unsigned *)(mws + ML.off_thr);
    float *slack = (float *)(mws + ML.off_slack);
    uint2 *list = (uint2 *)(mws + ML.off_list);
    unsigned long long *stats = (unsigned long long *)(mws + ML.off_stats);
    unsigned *bcnt = (unsigned *)(mws + ML.off_bcnt);
    const bool force_exact = (flags & MANET_EPI_REFINE_EXACT) != 0;
    flags &= ~MANET_EPI_REFINE_EXACT;
    const long pairs = (long)n_ids * ML.N_pad;
    if (force_exact) {
        hipLaunchKernelGGL(refine_force_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, ML.N_pad, n_ids, keys2, stats,
                           bcnt);
    } else {
    // 1. pre-pass over the sub-sampled bank -> keys = U
    const int S1 = pick_splits(ML.nQT, BL.T_sub_max, 512);
    tl_bank_bytes_hint = (double)BL.T_sub_max * (double)BL.tile_bytes;  // (block_map_arg: the pre-pass's bank)
    if (ML.G.steps == 2) launch_main_bf16<2, false>(qimg, bws + BL.off_sub_pack, sub_meta, n_ids, ML.nQT, S1, ML.N_pad, keys, st, 3);
    else launch_main_bf16<7, false>(qimg, bws + BL.off_sub_pack, sub_meta, n_ids, ML.nQT, S1, ML.N_pad, keys, st, 3);
    // 2. thresholds; exact keys and counters reset
    hipLaunchKernelGGL(refine_threshold_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, (const unsigned *)keys,
                       qimg, (long)ML.qblk_bytes, C, meta, N, ML.N_pad, n_ids, thr, slack, keys2, stats, bcnt);
    // 3. filter pass over the whole bank -> candidate list
    {
        const int S = pick_splits(ML.nQT, BL.T_max, 512);
        const size_t lds = (size_t)2 * 2 * bank_tile_bytes_u(2 * ML.G.steps, false) + (size_t)REFINE_LDS_LIST * 8 + 4 * 256 * 4;  // + the published keys
        const void *fn = ML.G.steps == 2 ? (const void *)global_match_bf16_wide_kernel<2, 0, true>
                                         : (const void *)global_match_bf16_wide_kernel<7, 0, true>;
#ifdef MANET_ABLATION
        if (ML.G.steps != 2) {  // timing experiments only (results are wrong): 16 no slow path, 32 no threshold exchange
            const int fabl = manet_tune_get(MANET_TUNE_ABLATION, 0);
            if (fabl == 16) fn = (const void *)global_match_bf16_wide_kernel<7, 16, true>;
            if (fabl == 32) fn = (const void *)global_match_bf16_wide_kernel<7, 32, true>;
            if (fabl == 48) fn = (const void *)global_match_bf16_wide_kernel<7, 48, true>;
            if (fabl == 64) fn = (const void *)global_match_bf16_wide_kernel<7, 64, true>;    // no returning atomic in flush_sub
            if (fabl == 128) fn = (const void *)global_match_bf16_wide_kernel<7, 128, true>;  // no listing behind the tests
            if (fabl == 192) fn = (const void *)global_match_bf16_wide_kernel<7, 192, true>;
            if (fabl == 256) fn = (const void *)global_match_bf16_wide_kernel<7, 256, true>;  // cycle counts of the listing path -> stats[4..9]
        }
#endif
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const char *bpack = bws + BL.off_pack;
        int nQT = ML.nQT, Sv = S, bm = block_map_arg(ML.nQT, 512), prio = 0;
        // (see split_of_block) the spread map.  (PB splits fastest on top of it, as the plain kernels have it: fabric fetch
        // of this pass 301 -> 176 MB at cfg3 shape, but concurrent splits of one object share thresholds later --
        // 4.3 -> 5.05 candidate rows per pair -- and the step got 1 % slower: not taken.)
        unsigned fgrid = (unsigned)(nQT * S);
        if ((bm & 0xff) == 0) {
            bm |= 3;
            if (FILTER_TAIL_CUTS > 1 && (S & 7) == 0) {  // (tapered tail: see split_of_block)
                bm = (bm & ~0xff) | 8;
                fgrid = (unsigned)(nQT * (S + 8 * (FILTER_TAIL_CUTS - 1)));
            }
        }
        long N_pad = ML.N_pad, cap = ML.bucket_cap;
        unsigned *thr_c = thr;
        const float *slack_c = slack;
        void *args[] = {(void *)&qimg, (void *)&bpack, (void *)&meta, (void *)&n_ids, (void *)&nQT, (void *)&Sv,
                        (void *)&N_pad, (void *)&keys, (void *)&bm, (void *)&prio, (void *)&thr_c, (void *)&slack_c, (void *)&stats,
                        (void *)&list, (void *)&cap, (void *)&bcnt};
        manet_profile_record(st, true, 0);
        (void)hipLaunchKernel(fn, dim3(fgrid), dim3(256), args, lds, st);
        manet_profile_record(st, false, 0);
#ifdef MANET_ABLATION
        if (manet_tune_get(MANET_TUNE_ABLATION, 0) == 256) hipLaunchKernelGGL(refine_debug_kernel, dim3(1), dim3(1), 0, st, stats);
#endif
    }
    }  // (!force_exact)
    // 4. exact re-rank of the candidates
    const float *rows = (const float *)(bws + BL.off_rows), *norms = (const float *)(bws + BL.off_norms);
    const dim3 rgrid((unsigned)(ML.N_pad / QB), REFINE_RZ);
    const size_t rlds = (size_t)QB * C * sizeof(float);
    const Geom G32 = BL.G32;
    char *q32 = mws + ML.off_q32;
    if (q_dtype == MANET_EMB_F32)
        hipLaunchKernelGGL(refine_rerank_kernel<float>, rgrid, dim3(256), rlds, st, (const float *)qraw, q_sn, q_sc, rows, norms,
                           (const uint2 *)list, bcnt, ML.bucket_cap, N, ML.N_pad, C, keys2, stats, q32, G32.units,
                           (long)G32.qblk_bytes);
    else
        hipLaunchKernelGGL(refine_rerank_kernel<unsigned short>, rgrid, dim3(256), rlds, st, (const unsigned short *)qraw, q_sn, q_sc,
                           rows, norms, (const uint2 *)list, bcnt, ML.bucket_cap, N, ML.N_pad, C, keys2, stats, q32,
                           G32.units, (long)G32.qblk_bytes);
    // 5. rescue: the 256-query tiles that hold a 32-query block whose bucket is incomplete (a block's hits were not listed,
    //    or the bucket overflowed) go through the exact fp32 kernel against the whole bank -- their fp32 operand image was
    //    written by the re-rank launch above; the workgroups of complete tiles return at once; the minima meet the
    //    re-rank's by atomicMin on keys2
    {
        const int nQT32 = (int)(ML.N_pad / QT);
        // (at most 16 splits: a healthy frame pays for the dispatch of nQT32 x S32 workgroups that return at once -- 7.7 us
        // at 48 splits -- and a full rescue loses ~10 % to the coarser last round)
        int S32 = pick_splits(nQT32, BL.T_max, 512);
        if (!force_exact) S32 = S32 > 16 ? 16 : S32;  // (forced: every tile is matched -- the fp32 kernel's own split count)
        if (force_exact) manet_profile_record(st, true, 0);
        tl_bank_bytes_hint = (double)BL.T_max * (double)G32.tile_bytes;
        const char *bpack32 = bws + BL.off_pack32;
        switch (G32.steps) {
        case 16: launch_rescue_f32_pipe<16>(q32, bpack32, meta, n_ids, nQT32, S32, ML.N_pad, keys2, bcnt, ML.bucket_cap, !force_exact, st); break;
        case 50: launch_rescue_f32_pipe<50>(q32, bpack32, meta, n_ids, nQT32, S32, ML.N_pad, keys2, bcnt, ML.bucket_cap, !force_exact, st); break;
        case 52: launch_rescue_f32_pipe<52>(q32, bpack32, meta, n_ids, nQT32, S32, ML.N_pad, keys2, bcnt, ML.bucket_cap, !force_exact, st); break;
        default: launch_rescue_f32_pipe<64>(q32, bpack32, meta, n_ids, nQT32, S32, ML.N_pad, keys2, bcnt, ML.bucket_cap, !force_exact, st); break;
        }
        if (force_exact) manet_profile_record(st, false, 0);
    }
    const long total = N * n_ids;
    hipLaunchKernelGGL(global_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, keys2, N, ML.N_pad, n_ids,
                       flags & ~MANET_EPI_KEYS_ARMED, out, mem);
    return manet_check_launch("manet_global_match (bf16 filter + fp32 re-rank)");
}

}  // namespace

ManetFrameLayout manet_frame_layout(int h, int w, int C, int compute, int max_distance)
{
    ManetFrameLayout F;
    const MatchLayout ML = match_layout((int64_t)h * w, C, 1, compute);
    F.N_pad = ML.N_pad;
    F.off_image = 0;
    F.image_bytes = (size_t)(ML.N_pad / QB) * ML.qblk_bytes;
    F.off_plane = manet_align_up(F.image_bytes, 1024);
    F.hp = h / 2;
    F.wp = w / 2;
    F.HPAD = F.WS = F.TY = F.TX = F.nty = F.ntx = 0;
    F.PS = 0;
    F.plane_bytes = F.tab_bytes = 0;
    if (max_distance >= 0) {
        const PoolPad G = lf_pool_pad(h, w, max_distance);
        F.HPAD = G.HPAD;
        F.WS = G.WS;
        F.PS = G.plane;
        F.TY = lf_sy(max_distance) - 1;
        F.TX = LF_SX - 1;
        F.nty = (F.hp + F.TY - 1) / F.TY;
        F.ntx = (F.wp + F.TX - 1) / F.TX;
        F.plane_bytes = (size_t)G.plane * C * sizeof(float);
        F.tab_bytes = (size_t)(F.nty + F.ntx + 2) * sizeof(int);
    }
    F.off_tab = manet_align_up(F.off_plane + F.plane_bytes, 256);
    F.total = manet_align_up(F.off_tab + F.tab_bytes, 1024);
    return F;
}

extern "C" {

int manet_frame_workspace_bytes(int h, int w, int C, int compute, int max_distance, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    if (h <= 0 || w <= 0) return manet_set_error(MANET_E_INVALID, "h=%d w=%d", h, w);
    int rc = check_common((int64_t)h * w, 0, C, 1, 1, compute);
    if (rc) return rc;
    if (max_distance > MANET_MAX_LOCAL_DISTANCE || (max_distance >= 0 && (h < 2 || w < 2)))
        return manet_set_error(MANET_E_INVALID, "max_distance=%d (supported -1 = no pooled plane, 0..%d; h, w >= 2)",
                               max_distance, MANET_MAX_LOCAL_DISTANCE);
    *bytes = manet_frame_layout(h, w, C, compute, max_distance).total;
    return MANET_OK;
}

static int frame_prepare_impl(const void *emb, int emb_dtype, int64_t s_f, int64_t s_y, int64_t s_x, int64_t s_c, int n_frames,
                              int h, int w, int C, int compute, int max_distance, void *frames_ws, size_t frame_ws_stride,
                              void *fill_ptr, int64_t fill_words, uint32_t fill_value, const float *scale, const float *shift,
                              int relu, void *emb_out, int emb_out_dtype, manet_stream_t stream);

int manet_frame_prepare(const void *emb, int emb_dtype, int64_t s_f, int64_t s_y, int64_t s_x, int64_t s_c, int n_frames,
                        int h, int w, int C, int compute, int max_distance, void *frames_ws, size_t frame_ws_stride,
                        void *fill_ptr, int64_t fill_words, uint32_t fill_value, manet_stream_t stream)
{
    return frame_prepare_impl(emb, emb_dtype, s_f, s_y, s_x, s_c, n_frames, h, w, C, compute, max_distance, frames_ws,
                              frame_ws_stride, fill_ptr, fill_words, fill_value, nullptr, nullptr, 0, nullptr, MANET_EMB_F32, stream);
}

int manet_embed_finish(const float *conv_out, int64_t s_f, int64_t s_y, int64_t s_x, int64_t s_c, const float *scale,
                       const float *shift, int relu, void *emb_out, int emb_out_dtype, int n_frames, int h, int w, int C,
                       int compute, int max_distance, void *frames_ws, size_t frame_ws_stride, manet_stream_t stream)
{
    if (!scale || !shift || !emb_out) return manet_set_error(MANET_E_INVALID, "null pointer");
    if (emb_out_dtype != MANET_EMB_F32 && emb_out_dtype != MANET_EMB_BF16)
        return manet_set_error(MANET_E_INVALID, "embedding dtype %d (MANET_EMB_F32 / MANET_EMB_BF16)", emb_out_dtype);
    if (((size_t)emb_out & 7) != 0) return manet_set_error(MANET_E_INVALID, "emb_out must be 8-byte aligned");
    return frame_prepare_impl(conv_out, MANET_EMB_F32, s_f, s_y, s_x, s_c, n_frames, h, w, C, compute, max_distance, frames_ws,
                              frame_ws_stride, nullptr, 0, 0u, scale, shift, relu, emb_out, emb_out_dtype, stream);
}

static int frame_prepare_impl(const void *emb, int emb_dtype, int64_t s_f, int64_t s_y, int64_t s_x, int64_t s_c, int n_frames,
                              int h, int w, int C, int compute, int max_distance, void *frames_ws, size_t frame_ws_stride,
                              void *fill_ptr, int64_t fill_words, uint32_t fill_value, const float *scale, const float *shift,
                              int relu, void *emb_out, int emb_out_dtype, manet_stream_t stream)
{
    size_t need = 0;
    int rc = manet_frame_workspace_bytes(h, w, C, compute, max_distance, &need);
    if (rc) return rc;
    if (!emb || !frames_ws || n_frames <= 0 || n_frames > 65535)
        return manet_set_error(MANET_E_INVALID, "null pointer or n_frames=%d", n_frames);
    if (frame_ws_stride < need || (frame_ws_stride & 1023))
        return manet_set_error(MANET_E_WORKSPACE, "frame workspace stride %zu < %zu bytes (or not a multiple of 1024)",
                               frame_ws_stride, need);
    if (fill_words < 0 || (fill_words > 0 && !fill_ptr)) return manet_set_error(MANET_E_INVALID, "bad fill request");
    const ManetFrameLayout F = manet_frame_layout(h, w, C, compute, max_distance);
    const Geom G = geom_of(C, compute);
#ifdef MANET_ABLATION
    const int XC = manet_tune_get(MANET_TUNE_FRAME_XC, 0) == 1 ? 64 : 32;  // (tuning: 64-column workgroups)
#else
    const int XC = 32;
#endif
    FramePrep A;
    A.emb = emb;
    A.s_f = (long)s_f; A.s_y = (long)s_y; A.s_x = (long)s_x; A.s_c = (long)s_c;
    A.h = h; A.w = w; A.C = C; A.compute = G.compute; A.units = G.units; A.kpad = G.kpad;  // (G.compute: _BF16_REFINE packs as _BF16)
    A.ws = (char *)frames_ws; A.ws_stride = (long)frame_ws_stride; A.qblk_bytes = (long)G.qblk_bytes;
    A.off_plane = (long)F.off_plane; A.off_tab = (long)F.off_tab;
    A.d = max_distance; A.hp = F.hp; A.wp = F.wp; A.HPAD = F.HPAD; A.WS = F.WS; A.PS = F.PS;
    A.TY = F.TY; A.TX = F.TX; A.nty = F.nty; A.ntx = F.ntx;
    A.N = (long)h * w; A.N_pad = F.N_pad;
    A.fill_ptr = (unsigned *)fill_ptr; A.fill_words = (long)fill_words; A.fill_value = fill_value;
    A.nxc = (w + XC - 1) / XC;
#ifdef MANET_ABLATION
    A.abl = manet_tune_get(MANET_TUNE_ABLATION, 0);
#else
    A.abl = 0;
#endif
    A.n_data = ((h + 1) / 2) * A.nxc;
    long aux_items = (long)(F.N_pad - A.N) * G.units + fill_words + 64;
    if (max_distance >= 0) aux_items += (long)C * (F.HPAD - F.hp) * (F.WS / 4) + (long)C * F.hp * (F.WS - F.wp) / 4;
    long aux = (aux_items + 1023) / 1024;
    if (aux < 1) aux = 1;
    if (aux > 256) aux = 256;
    const size_t esz = emb_dtype == MANET_EMB_F32 ? 4 : 2;
    A.vec2 = (s_x == 1 && (w & 1) == 0 && (s_y & 1) == 0 && (s_c & 1) == 0 && (n_frames == 1 || (s_f & 1) == 0) &&
              ((size_t)emb % (2 * esz)) == 0) ? 1 : 0;
    A.scale = scale; A.shift = shift; A.relu = relu; A.emb_out = emb_out; A.emb_out_bf16 = (scale && emb_out_dtype == MANET_EMB_BF16) ? 1 : 0;
    // (the epilogue form stores 2-pixel pairs: emb_out rows must pair up as the source's do)
    if (scale && (w & 1)) A.vec2 = 0;
    A.rcopy = (emb_dtype == MANET_EMB_F32 && G.compute == MANET_COMPUTE_BF16 && !A.emb_out_bf16) ? 1 : 0;
    const size_t lds = (size_t)(A.rcopy ? 2 : 1) * 2 * XC * (G.kpad + 1) * sizeof(float) + 2 * XC * sizeof(float);
    const dim3 grid((unsigned)(A.n_data + aux), 1, (unsigned)n_frames);
    hipStream_t st = (hipStream_t)stream;
    if (emb_dtype != MANET_EMB_F32 && emb_dtype != MANET_EMB_BF16)
        return manet_set_error(MANET_E_INVALID, "embedding dtype %d (MANET_EMB_F32 / MANET_EMB_BF16)", emb_dtype);
    const void *fn = emb_dtype == MANET_EMB_F32
                         ? (A.vec2 ? (const void *)frame_prepare_kernel<float, 32, true> : (const void *)frame_prepare_kernel<float, 32, false>)
                         : (A.vec2 ? (const void *)frame_prepare_kernel<unsigned short, 32, true>
                                   : (const void *)frame_prepare_kernel<unsigned short, 32, false>);
#ifdef MANET_ABLATION
    if (XC == 64)
        fn = emb_dtype == MANET_EMB_F32
                 ? (A.vec2 ? (const void *)frame_prepare_kernel<float, 64, true> : (const void *)frame_prepare_kernel<float, 64, false>)
                 : (A.vec2 ? (const void *)frame_prepare_kernel<unsigned short, 64, true>
                           : (const void *)frame_prepare_kernel<unsigned short, 64, false>);
#endif
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void *args[] = {(void *)&A};
    manet_profile_record(st, true, 2);
    (void)hipLaunchKernel(fn, grid, dim3(256), args, lds, st);
    manet_profile_record(st, false, 2);
    return manet_check_launch("manet_frame_prepare");
}

int manet_bank_workspace_bytes(int64_t M0, int C, int n_ids, int compute, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_common(1, M0, C, n_ids, 1, compute);
    if (rc) return rc;
    *bytes = bank_layout(M0, C, n_ids, compute).total;
    return MANET_OK;
}

int manet_match_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, int k_nn, int compute,
                                size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_common(N, M0, C, n_ids, k_nn, compute);
    if (rc) return rc;
    *bytes = match_layout(N, C, n_ids, compute, k_nn).total;
    return MANET_OK;
}

int manet_global_match_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, int k_nn, int compute,
                                       size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_common(N, M0, C, n_ids, k_nn, compute);
    if (rc) return rc;
    *bytes = bank_layout(M0, C, n_ids, compute).total + match_layout(N, C, n_ids, compute, k_nn).total;
    return MANET_OK;
}

int manet_bank_prepare(const float *bank, int64_t b_stride_m, int64_t b_stride_c, const int32_t *labels,
                       int64_t M0, int C, int n_ids, int compute, void *bank_ws, size_t bank_ws_bytes,
                       manet_stream_t stream)
{
    return manet_bank_prepare_ex(bank, MANET_EMB_F32, b_stride_m, b_stride_c, labels, M0, C, n_ids, compute, bank_ws,
                                 bank_ws_bytes, stream);
}

int manet_bank_prepare_ex(const void *bank, int emb_dtype, int64_t b_stride_m, int64_t b_stride_c, const int32_t *labels,
                          int64_t M0, int C, int n_ids, int compute, void *bank_ws, size_t bank_ws_bytes,
                          manet_stream_t stream)
{
    int rc = check_common(1, M0, C, n_ids, 1, compute);
    if (rc) return rc;
    if ((M0 > 0 && (!bank || !labels)) || !bank_ws) return manet_set_error(MANET_E_INVALID, "null pointer");
    BankLayout L = bank_layout(M0, C, n_ids, compute);
    if (bank_ws_bytes < L.total)
        return manet_set_error(MANET_E_WORKSPACE, "bank workspace %zu < %zu bytes", bank_ws_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)bank_ws;
    int *meta = (int *)(ws + L.off_meta);
    int *hist = (int *)(ws + L.off_hist);
    int *src_of = (int *)(ws + L.off_src);
    fill32(meta, 0u, META_INTS, st);
    fill32(src_of, 0xffffffffu, (size_t)L.T_max * BT, st);
    if (M0 > 0) {
        hipLaunchKernelGGL(label_hist_kernel, dim3((unsigned)L.nblocks), dim3(RPB), 0, st, labels, (long)M0, n_ids, hist);
        hipLaunchKernelGGL(label_scan_kernel, dim3((unsigned)n_ids), dim3(64), 0, st, hist, (int)L.nblocks, n_ids, meta);
    }
    hipLaunchKernelGGL(label_segments_kernel, dim3(1), dim3(64), 0, st, n_ids, meta);
    if (M0 > 0)
        hipLaunchKernelGGL(label_scatter_kernel, dim3((unsigned)L.nblocks), dim3(RPB), 0, st, labels, (long)M0, n_ids,
                           (const int *)hist, (const int *)meta, src_of);
    rc = launch_bank_pack(bank, emb_dtype, (long)b_stride_m, (long)b_stride_c, (const int *)src_of, (const int *)meta, (long)M0,
                          C, L.G, ws + L.off_pack, L.T_max, st);
    if (rc) return rc;
    if (compute == MANET_COMPUTE_BF16_REFINE) {
        // the fp32 operand image of the same sorted bank: what the rescue pass (the exact fp32 kernel on the query blocks
        // whose candidate buckets are incomplete) multiplies against
        rc = launch_bank_pack(bank, emb_dtype, (long)b_stride_m, (long)b_stride_c, (const int *)src_of, (const int *)meta,
                              (long)M0, C, L.G32, ws + L.off_pack32, L.T_max, st);
        if (rc) return rc;
        // the fp32 copy of the sorted rows the exact re-rank reads, and the sub-sampled bank of the pre-pass
        float *rows = (float *)(ws + L.off_rows), *norms = (float *)(ws + L.off_norms);
        int *sub_meta = (int *)(ws + L.off_sub_meta);
        if (L.T_max > 0) {
            if (emb_dtype == MANET_EMB_F32)
                hipLaunchKernelGGL(bank_rows_f32_kernel<float>, dim3((unsigned)L.T_max), dim3(256), 0, st, (const float *)bank,
                                   (long)b_stride_m, (long)b_stride_c, (const int *)src_of, meta, C, rows, norms);
            else
                hipLaunchKernelGGL(bank_rows_f32_kernel<unsigned short>, dim3((unsigned)L.T_max), dim3(256), 0, st,
                                   (const unsigned short *)bank, (long)b_stride_m, (long)b_stride_c, (const int *)src_of, meta, C,
                                   rows, norms);
        }
        fill32(sub_meta, 0u, META_INTS, st);
        const int sub = refine_sub(L.T_max);
        hipLaunchKernelGGL(sub_segments_kernel, dim3(1), dim3(64), 0, st, n_ids, (const int *)meta, sub_meta, sub);
        hipLaunchKernelGGL(sub_copy_kernel, dim3((unsigned)L.T_sub_max), dim3(256), 0, st, n_ids, (const int *)meta,
                           (const int *)sub_meta, (const char *)(ws + L.off_pack), ws + L.off_sub_pack, (long)L.tile_bytes, sub);
    }
    return manet_check_launch("manet_bank_prepare");
}

int manet_query_pack_bytes(int64_t N, int C, int compute, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_common(N, 0, C, 1, 1, compute);
    if (rc) return rc;
    MatchLayout ML = match_layout(N, C, 1, compute);
    *bytes = (size_t)(ML.N_pad / QB) * ML.qblk_bytes;
    return MANET_OK;
}

int manet_query_pack(const void *query, int emb_dtype, int64_t q_stride_n, int64_t q_stride_c, int64_t N, int C,
                     int compute, void *packed, size_t packed_bytes, manet_stream_t stream)
{
    int rc = check_common(N, 0, C, 1, 1, compute);
    if (rc) return rc;
    if (!query || !packed) return manet_set_error(MANET_E_INVALID, "null pointer");
    MatchLayout ML = match_layout(N, C, 1, compute);
    const size_t need = (size_t)(ML.N_pad / QB) * ML.qblk_bytes;
    if (packed_bytes < need) return manet_set_error(MANET_E_WORKSPACE, "packed query buffer %zu < %zu bytes", packed_bytes, need);
    rc = launch_query_pack(query, emb_dtype, (long)q_stride_n, (long)q_stride_c, (long)N, ML.N_pad, C, ML.G, (char *)packed,
                           nullptr, 0, (hipStream_t)stream);
    if (rc) return rc;
    return manet_check_launch("manet_query_pack");
}

int manet_global_match_prepared(const float *query, int64_t q_stride_n, int64_t q_stride_c,
                                const void *bank_ws, int64_t N, int64_t M0, int C, int n_ids, int k_nn,
                                int compute, float *out, float *mem_inout, int epilogue_flags,
                                void *match_ws, size_t match_ws_bytes, manet_stream_t stream)
{
    return manet_global_match_prepared_ex(query, MANET_EMB_F32, q_stride_n, q_stride_c, bank_ws, N, M0, C, n_ids, k_nn,
                                          compute, out, mem_inout, epilogue_flags, match_ws, match_ws_bytes, stream);
}

int manet_global_match_prepared_ex(const void *query, int emb_dtype, int64_t q_stride_n, int64_t q_stride_c,
                                   const void *bank_ws, int64_t N, int64_t M0, int C, int n_ids, int k_nn,
                                   int compute, float *out, float *mem_inout, int epilogue_flags,
                                   void *match_ws, size_t match_ws_bytes, manet_stream_t stream)
{
    int rc = check_common(N, M0, C, n_ids, k_nn, compute);
    if (rc) return rc;
    if (!query || !bank_ws || !out || !match_ws) return manet_set_error(MANET_E_INVALID, "null pointer");
    BankLayout BL = bank_layout(M0, C, n_ids, compute);
    MatchLayout ML = match_layout(N, C, n_ids, compute, k_nn);
    if (match_ws_bytes < ML.total)
        return manet_set_error(MANET_E_WORKSPACE, "match workspace %zu < %zu bytes", match_ws_bytes, ML.total);
    hipStream_t st = (hipStream_t)stream;
    const char *bws = (const char *)bank_ws;
    char *mws = (char *)match_ws;
    const int *meta = (const int *)(bws + BL.off_meta);
    unsigned *keys = (unsigned *)(mws + ML.off_keys);
    // MANET_EMB_PACKED: `query` already is the operand image manet_query_pack / manet_frame_prepare wrote (same N, C,
    // compute)
    const char *qpack = (const char *)query;
    const bool armed = (epilogue_flags & MANET_EPI_KEYS_ARMED) && k_nn == 1;
    if (!armed) epilogue_flags &= ~MANET_EPI_KEYS_ARMED;
    if (emb_dtype != MANET_EMB_PACKED) {
        rc = launch_query_pack(query, emb_dtype, (long)q_stride_n, (long)q_stride_c, (long)N, ML.N_pad, C, ML.G,
                               mws + ML.off_q, keys, n_ids, st);  // also resets the keys
        if (rc) return rc;
        qpack = mws + ML.off_q;
    } else if (!armed) {
        fill32(keys, 0xffffffffu, (size_t)n_ids * ML.N_pad, st);
    }
    if (compute == MANET_COMPUTE_BF16_REFINE) {
        if (emb_dtype == MANET_EMB_PACKED)
            return manet_set_error(MANET_E_INVALID, "MANET_COMPUTE_BF16_REFINE re-ranks in fp32 from the query as stored: pass "
                                                    "it to manet_global_match_refine next to its packed image");
        return run_refine(qpack, query, emb_dtype, (long)q_stride_n, (long)q_stride_c, bws, BL, mws, ML, (long)N, C, n_ids, out,
                          mem_inout, epilogue_flags, st);
    }
    // resident workgroup slots: f32 and plain bf16 (wide kernel) = 2 x 256-thread workgroups per CU,
    // split-bf16 (and the tuning-only narrow/flat bf16 forms) = 1 x 512-thread workgroup per CU
#ifdef MANET_ABLATION
    const int bv = manet_tune_get(MANET_TUNE_BF16_VARIANT, 0);
#else
    const int bv = 0;
#endif
    const bool two_per_cu = compute == MANET_COMPUTE_F32 ||
                            (compute == MANET_COMPUTE_BF16 && !(bv & (16 | 64)) && ML.G.steps != 9);
    int S = pick_splits(ML.nQT, BL.T_max, two_per_cu ? 512 : 256);
    tl_bank_bytes_hint = (double)BL.T_max * (double)BL.tile_bytes;  // (block_map_arg)
    {
        int forced = manet_tune_get(MANET_TUNE_SPLITS, 0);  // tuning only
        if (forced > 0) S = (forced + 7) / 8 * 8;
    }
    if (k_nn > 1) S = TOPK_SPLITS;
    const char *bpack = bws + BL.off_pack;
    float *topk = (float *)(mws + ML.off_topk);
    if (k_nn > 1) {
        // splits that own no tile never write their lists: start from "no candidate"
        size_t words = (size_t)TOPK_SPLITS * n_ids * ML.N_pad * MANET_MAX_KNN;
        fill32(topk, 0x7f7f7f7fu, words, st);  // 3.39e38 >= 1e20: invalid
    }
    if (compute != MANET_COMPUTE_F32) {
        const bool x3 = (compute == MANET_COMPUTE_BF16X3);
#define MANET_GB_CASE(K_)                                                                              \
    case K_:                                                                                           \
        if (x3) launch_main_bf16<K_, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st);  \
        else launch_main_bf16<K_, false>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st);    \
        break;
        switch (ML.G.steps) {
            MANET_GB_CASE(2) MANET_GB_CASE(7)
        default:
            if (x3) launch_main_bf16<9, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st);
            else launch_main_bf16<9, false>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st);
            break;
        }
#undef MANET_GB_CASE
    } else {
#ifdef MANET_ABLATION
        const bool pipe = !manet_tune_get(MANET_TUNE_F32_UNPIPED, 0);  // tuning: 1 = the un-pipelined k = 1 kernel
#define MANET_GM_UNPIPED(KS_) launch_main_f32<KS_, 1>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, topk, st)
#else
        const bool pipe = true;
#define MANET_GM_UNPIPED(KS_) (void)0
#endif
#define MANET_GM_CASE(KS_)                                                                                    \
    case KS_:                                                                                                 \
        if (k_nn == 1 && pipe) launch_main_f32_pipe<KS_>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st); \
        else if (k_nn == 1) MANET_GM_UNPIPED(KS_);                                                            \
        else launch_main_f32<KS_, MANET_MAX_KNN>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, topk, st); \
        break;
    switch (pick_ks(C)) {
        MANET_GM_CASE(16) MANET_GM_CASE(50) MANET_GM_CASE(52)
    default:
        if (k_nn == 1 && pipe) launch_main_f32_pipe<64>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st);
        else if (k_nn == 1) MANET_GM_UNPIPED(64);
        else launch_main_f32<64, MANET_MAX_KNN>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, topk, st);
        break;
    }
#undef MANET_GM_CASE
#undef MANET_GM_UNPIPED
    }
    long total = (long)(armed ? ML.N_pad : N) * n_ids;
    if (k_nn == 1)
        hipLaunchKernelGGL(global_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           keys, (long)N, ML.N_pad, n_ids, epilogue_flags, out, mem_inout);
    else
        hipLaunchKernelGGL(global_finish_topk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const float *)topk, S, k_nn, (long)N, ML.N_pad, n_ids, epilogue_flags, out, mem_inout);
    return manet_check_launch("manet_global_match_prepared");
}

int manet_global_match(const float *query, int64_t q_stride_n, int64_t q_stride_c, const float *bank,
                       int64_t b_stride_m, int64_t b_stride_c, const int32_t *labels, int64_t N, int64_t M0,
                       int C, int n_ids, int k_nn, int compute, float *out, float *mem_inout,
                       int epilogue_flags, void *workspace, size_t workspace_bytes, manet_stream_t stream)
{
    return manet_global_match_ex(query, MANET_EMB_F32, q_stride_n, q_stride_c, bank, MANET_EMB_F32, b_stride_m, b_stride_c,
                                 labels, N, M0, C, n_ids, k_nn, compute, out, mem_inout, epilogue_flags, workspace,
                                 workspace_bytes, stream);
}

int manet_global_match_ex(const void *query, int q_dtype, int64_t q_stride_n, int64_t q_stride_c, const void *bank,
                          int b_dtype, int64_t b_stride_m, int64_t b_stride_c, const int32_t *labels, int64_t N,
                          int64_t M0, int C, int n_ids, int k_nn, int compute, float *out, float *mem_inout,
                          int epilogue_flags, void *workspace, size_t workspace_bytes, manet_stream_t stream)
{
    int rc = check_common(N, M0, C, n_ids, k_nn, compute);
    if (rc) return rc;
    if (!workspace) return manet_set_error(MANET_E_INVALID, "workspace == NULL");
    size_t bbytes = bank_layout(M0, C, n_ids, compute).total;
    size_t mbytes = match_layout(N, C, n_ids, compute, k_nn).total;
    if (workspace_bytes < bbytes + mbytes)
        return manet_set_error(MANET_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, bbytes + mbytes);
    char *ws = (char *)workspace;
    rc = manet_bank_prepare_ex(bank, b_dtype, b_stride_m, b_stride_c, labels, M0, C, n_ids, compute, ws, bbytes, stream);
    if (rc) return rc;
    return manet_global_match_prepared_ex(query, q_dtype, q_stride_n, q_stride_c, ws, N, M0, C, n_ids, k_nn, compute, out,
                                          mem_inout, epilogue_flags, ws + bbytes, mbytes, stream);
}

int manet_global_match_arg_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_common(N, M0, C, n_ids, 1, MANET_COMPUTE_F32);
    if (rc) return rc;
    *bytes = bank_layout(M0, C, n_ids, MANET_COMPUTE_F32).total + match_layout(N, C, n_ids, MANET_COMPUTE_F32, 1, true).total;
    return MANET_OK;
}

int manet_global_match_arg_f32(const float *query, int64_t q_stride_n, int64_t q_stride_c, const float *bank,
                               int64_t b_stride_m, int64_t b_stride_c, const int32_t *labels, int64_t N, int64_t M0,
                               int C, int n_ids, float *out, int32_t *arg_out, void *workspace, size_t workspace_bytes,
                               manet_stream_t stream)
{
    const int compute = MANET_COMPUTE_F32;
    int rc = check_common(N, M0, C, n_ids, 1, compute);
    if (rc) return rc;
    if (!query || !out || !arg_out || !workspace) return manet_set_error(MANET_E_INVALID, "null pointer");
    BankLayout BL = bank_layout(M0, C, n_ids, compute);
    MatchLayout ML = match_layout(N, C, n_ids, compute, 1, true);
    if (workspace_bytes < BL.total + ML.total)
        return manet_set_error(MANET_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, BL.total + ML.total);
    char *bws = (char *)workspace, *mws = bws + BL.total;
    rc = manet_bank_prepare(bank, b_stride_m, b_stride_c, labels, M0, C, n_ids, compute, bws, BL.total, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int *meta = (const int *)(bws + BL.off_meta);
    unsigned long long *keys64 = (unsigned long long *)(mws + ML.off_topk);
    fill32(keys64, 0xffffffffu, (size_t)2 * n_ids * ML.N_pad, st);
    rc = launch_query_pack(query, MANET_EMB_F32, (long)q_stride_n, (long)q_stride_c, (long)N, ML.N_pad, C, ML.G,
                           mws + ML.off_q, nullptr, 0, st);
    if (rc) return rc;
    int S = pick_splits(ML.nQT, BL.T_max, 512);
    tl_bank_bytes_hint = (double)BL.T_max * (double)BL.tile_bytes;  // (block_map_arg)
    const char *qpack = mws + ML.off_q, *bpack = bws + BL.off_pack;
    unsigned *keys = (unsigned *)(mws + ML.off_keys);
    switch (pick_ks(C)) {
    case 16: launch_main_f32<16, 1, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, (float *)keys64, st); break;
    case 50: launch_main_f32<50, 1, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, (float *)keys64, st); break;
    case 52: launch_main_f32<52, 1, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, (float *)keys64, st); break;
    default: launch_main_f32<64, 1, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, (float *)keys64, st); break;
    }
    long total = (long)N * n_ids;
    hipLaunchKernelGGL(global_finish_arg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const unsigned long long *)keys64, (const int *)(bws + BL.off_src), meta, (long)N, ML.N_pad, n_ids,
                       out, arg_out);
    return manet_check_launch("manet_global_match_arg_f32");
}

/* k_nn nearest rows per (query, object), with their row numbers: the forward of the TRAINING path of k_nearest_neighbors > 1
 * (IntVOS.py:87-94; manet_hip.h).  k_nn passes of the arg-min kernel, pass j bounded from below by pass j - 1's (distance key,
 * slot) pair: exact, ascending, ties ordered by bank slot.  out / arg_out: [k_nn][N][n_ids]; beyond an object's row count the
 * distance is the padding value 1e20 and the row -1. */
int manet_global_match_topk_arg_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_common(N, M0, C, n_ids, 1, MANET_COMPUTE_F32);
    if (rc) return rc;
    MatchLayout ML = match_layout(N, C, n_ids, MANET_COMPUTE_F32, 1, true);
    *bytes = bank_layout(M0, C, n_ids, MANET_COMPUTE_F32).total + ML.total +
             manet_align_up((size_t)n_ids * ML.N_pad * sizeof(unsigned long long), 1024);
    return MANET_OK;
}

int manet_global_match_topk_arg_f32(const float *query, int64_t q_stride_n, int64_t q_stride_c, const float *bank,
                                    int64_t b_stride_m, int64_t b_stride_c, const int32_t *labels, int64_t N, int64_t M0,
                                    int C, int n_ids, int k_nn, float *out, int32_t *arg_out, void *workspace,
                                    size_t workspace_bytes, manet_stream_t stream)
{
    const int compute = MANET_COMPUTE_F32;
    int rc = check_common(N, M0, C, n_ids, k_nn, compute);
    if (rc) return rc;
    if (!query || !out || !arg_out || !workspace) return manet_set_error(MANET_E_INVALID, "null pointer");
    BankLayout BL = bank_layout(M0, C, n_ids, compute);
    MatchLayout ML = match_layout(N, C, n_ids, compute, 1, true);
    const size_t kbytes = manet_align_up((size_t)n_ids * ML.N_pad * sizeof(unsigned long long), 1024);
    if (workspace_bytes < BL.total + ML.total + kbytes)
        return manet_set_error(MANET_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, BL.total + ML.total + kbytes);
    char *bws = (char *)workspace, *mws = bws + BL.total;
    rc = manet_bank_prepare(bank, b_stride_m, b_stride_c, labels, M0, C, n_ids, compute, bws, BL.total, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int *meta = (const int *)(bws + BL.off_meta);
    unsigned long long *kbuf[2] = {(unsigned long long *)(mws + ML.off_topk), (unsigned long long *)(mws + ML.total)};
    rc = launch_query_pack(query, MANET_EMB_F32, (long)q_stride_n, (long)q_stride_c, (long)N, ML.N_pad, C, ML.G,
                           mws + ML.off_q, nullptr, 0, st);
    if (rc) return rc;
    int S = pick_splits(ML.nQT, BL.T_max, 512);
    tl_bank_bytes_hint = (double)BL.T_max * (double)BL.tile_bytes;  // (block_map_arg)
    const char *qpack = mws + ML.off_q, *bpack = bws + BL.off_pack;
    const long total = (long)N * n_ids;
    for (int j = 0; j < k_nn; ++j) {
        unsigned long long *cur = kbuf[j & 1], *prev = kbuf[(j & 1) ^ 1];
        fill32(cur, 0xffffffffu, (size_t)2 * n_ids * ML.N_pad, st);
        if (j == 0) fill32(prev, 0u, (size_t)2 * n_ids * ML.N_pad, st);  // bound 0: every real pair qualifies
        switch (pick_ks(C)) {
        case 16: launch_main_f32<16, 1, true, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, (unsigned *)prev, (float *)cur, st); break;
        case 50: launch_main_f32<50, 1, true, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, (unsigned *)prev, (float *)cur, st); break;
        case 52: launch_main_f32<52, 1, true, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, (unsigned *)prev, (float *)cur, st); break;
        default: launch_main_f32<64, 1, true, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, (unsigned *)prev, (float *)cur, st); break;
        }
        hipLaunchKernelGGL(global_finish_arg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const unsigned long long *)cur, (const int *)(bws + BL.off_src), meta, (long)N, ML.N_pad, n_ids,
                           out + (size_t)j * total, arg_out + (size_t)j * total);
    }
    return manet_check_launch("manet_global_match_topk_arg_f32");
}

int manet_global_match_backward_f32(const float *query, int64_t q_stride_n, int64_t q_stride_c, const float *bank,
                                    int64_t b_stride_m, int64_t b_stride_c, const int32_t *arg, const float *grad_out,
                                    int64_t N, int64_t M0, int C, int n_ids, float *grad_query, int64_t gq_stride_n,
                                    int64_t gq_stride_c, float *grad_bank, int64_t gb_stride_m, int64_t gb_stride_c,
                                    manet_stream_t stream)
{
    int rc = check_common(N, M0, C, n_ids, 1, MANET_COMPUTE_F32);
    if (rc) return rc;
    if (!query || !arg || !grad_out || (!grad_query && !grad_bank) || (M0 > 0 && !bank))
        return manet_set_error(MANET_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (M0 > 0 && grad_bank) {
        long tb = (long)M0 * C;
        hipLaunchKernelGGL(zero_strided_kernel, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, st, grad_bank, (long)M0,
                           (long)C, (long)gb_stride_m, (long)gb_stride_c);
    }
    long tq = (long)N * C;
    hipLaunchKernelGGL(global_match_backward_kernel, dim3((unsigned)((tq + 255) / 256)), dim3(256), 0, st, query,
                       (long)q_stride_n, (long)q_stride_c, bank, (long)b_stride_m, (long)b_stride_c, arg, grad_out, (long)N, C,
                       n_ids, grad_query, (long)gq_stride_n, (long)gq_stride_c, grad_bank, (long)gb_stride_m,
                       (long)gb_stride_c);
    return manet_check_launch("manet_global_match_backward_f32");
}

int manet_global_match_refine(const void *query, int emb_dtype, int64_t q_stride_n, int64_t q_stride_c,
                              const void *query_image, const void *bank_ws, int64_t N, int64_t M0, int C, int n_ids,
                              float *out, float *mem_inout, int epilogue_flags, void *match_ws, size_t match_ws_bytes,
                              manet_stream_t stream)
{
    const int compute = MANET_COMPUTE_BF16_REFINE;
    int rc = check_common(N, M0, C, n_ids, 1, compute);
    if (rc) return rc;
    if (!query || !bank_ws || !out || !match_ws) return manet_set_error(MANET_E_INVALID, "null pointer");
    if (emb_dtype != MANET_EMB_F32 && emb_dtype != MANET_EMB_BF16)
        return manet_set_error(MANET_E_INVALID, "embedding dtype %d (MANET_EMB_F32 / MANET_EMB_BF16)", emb_dtype);
    if (!query_image)  // no image at hand: the general entry point packs one
        return manet_global_match_prepared_ex(query, emb_dtype, q_stride_n, q_stride_c, bank_ws, N, M0, C, n_ids, 1, compute, out,
                                              mem_inout, epilogue_flags, match_ws, match_ws_bytes, stream);
    BankLayout BL = bank_layout(M0, C, n_ids, compute);
    MatchLayout ML = match_layout(N, C, n_ids, compute, 1);
    if (match_ws_bytes < ML.total)
        return manet_set_error(MANET_E_WORKSPACE, "match workspace %zu < %zu bytes", match_ws_bytes, ML.total);
    hipStream_t st = (hipStream_t)stream;
    fill32((char *)match_ws + ML.off_keys, 0xffffffffu, (size_t)n_ids * ML.N_pad, st);
    return run_refine((const char *)query_image, query, emb_dtype, (long)q_stride_n, (long)q_stride_c, (const char *)bank_ws, BL,
                      (char *)match_ws, ML, (long)N, C, n_ids, out, mem_inout, epilogue_flags & ~MANET_EPI_KEYS_ARMED, st);
}

int manet_global_match_refine_stats(const void *match_ws, int64_t N, int C, int n_ids, int64_t *candidates,
                                    int64_t *list_overflowed)
{
    if (!match_ws) return manet_set_error(MANET_E_INVALID, "null pointer");
    int rc = check_common(N, 0, C, n_ids, 1, MANET_COMPUTE_BF16_REFINE);
    if (rc) return rc;
    MatchLayout ML = match_layout(N, C, n_ids, MANET_COMPUTE_BF16_REFINE, 1);
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, (const char *)match_ws + ML.off_stats, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)  // (blocks)
        return manet_set_error(MANET_E_LAUNCH, "reading the statistics failed");
    if (candidates) *candidates = (int64_t)h[0];
    if (list_overflowed) *list_overflowed = (int64_t)h[1];
    return MANET_OK;
}

int manet_global_match_refine_rescued_async(const void *match_ws, int64_t N, int C, int n_ids, int32_t *out2_device,
                                            manet_stream_t stream)
{
    if (!match_ws || !out2_device) return manet_set_error(MANET_E_INVALID, "null pointer");
    int rc = check_common(N, 0, C, n_ids, 1, MANET_COMPUTE_BF16_REFINE);
    if (rc) return rc;
    MatchLayout ML = match_layout(N, C, n_ids, MANET_COMPUTE_BF16_REFINE, 1);
    hipLaunchKernelGGL(refine_rescued_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned *)((const char *)match_ws + ML.off_bcnt), (long)(ML.N_pad / QT), (long)ML.bucket_cap, out2_device);
    return manet_check_launch("manet_global_match_refine_rescued_async");
}

int manet_global_match_refine_stats2(const void *match_ws, int64_t N, int C, int n_ids, int64_t *stats4)
{
    if (!match_ws || !stats4) return manet_set_error(MANET_E_INVALID, "null pointer");
    int rc = check_common(N, 0, C, n_ids, 1, MANET_COMPUTE_BF16_REFINE);
    if (rc) return rc;
    MatchLayout ML = match_layout(N, C, n_ids, MANET_COMPUTE_BF16_REFINE, 1);
    unsigned long long h[2] = {0, 0};
    const long nb = ML.N_pad / QB;
    unsigned *bc = (unsigned *)malloc((size_t)nb * sizeof(unsigned));
    if (!bc) return manet_set_error(MANET_E_INVALID, "out of host memory");
    if (hipMemcpy(h, (const char *)match_ws + ML.off_stats, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(bc, (const char *)match_ws + ML.off_bcnt, (size_t)nb * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) {
        free(bc);
        return manet_set_error(MANET_E_LAUNCH, "reading the statistics failed");
    }
    // a 256-query tile goes through the rescue pass (the exact fp32 kernel) iff one of its eight 32-query blocks has an
    // incomplete bucket -- the same test global_match_f32_pipe_kernel<KS, RESCUE> makes
    long rescued = 0;
    const long tiles = ML.N_pad / QT;
    for (long t = 0; t < tiles; ++t) {
        bool need = false;
        for (int i = 0; i < QT / QB; ++i) {
            const unsigned raw = bc[t * (QT / QB) + i];
            need = need || (raw >> 31) || (long)raw > ML.bucket_cap;
        }
        rescued += need ? 1 : 0;
    }
    free(bc);
    stats4[0] = (int64_t)h[0];
    stats4[1] = (int64_t)h[1];
    stats4[2] = (int64_t)rescued;
    stats4[3] = (int64_t)tiles;
    return MANET_OK;
}

int manet_normalize_merge_f32(float *x, float *mem_inout, int64_t n, int normalize, manet_stream_t stream)
{
    if (n < 0 || (n > 0 && !x)) return manet_set_error(MANET_E_INVALID, "bad arguments");
    if (n == 0) return MANET_OK;
    hipLaunchKernelGGL(normalize_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, mem_inout, (long)n, normalize);
    return manet_check_launch("manet_normalize_merge_f32");
}

}  // extern "C"
