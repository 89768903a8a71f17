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

namespace {

constexpr int QT = 256;       // queries per workgroup (4 waves x 2 blocks of 32)
constexpr int QB = 32;        // queries per packed block (one MFMA N dimension)
constexpr int BT = 64;        // bank rows per tile (2 MFMA M blocks)
constexpr int META_INTS = 256;
constexpr int META_T = 0;          // [0]        number of bank tiles actually used
constexpr int META_SEG = 1;        // [1..65]    first tile of object o (entry n_ids = T)
constexpr int META_CNT = 130;      // [130..193] rows per object

// Packed operand image of one row block: `units` 16-byte units per row, stored [unit][row][16 B],
// then the rows' squared norms (fp32).
//   f32    unit u = 2g+h holds k = 8g + 2j + h, j = 0..3 (4 floats)      -> v_mfma_f32_32x32x2_f32
//   bf16   unit u = 2s+h holds k = 16s + 8h + e, e = 0..7 (8 bf16)        -> v_mfma_f32_32x32x16_bf16
//   bf16x3 the bf16 image of hi = bf16(x), then the image of lo = bf16(x - hi)
__host__ __device__ constexpr size_t bank_tile_bytes_u(int units)
{
    return ((size_t)units * BT * 16 + BT * 4 + 1023) / 1024 * 1024;  // whole 1 KiB global_load_lds pieces
}
__host__ __device__ constexpr size_t query_block_bytes_u(int units) { return (size_t)units * QB * 16 + QB * 4; }
__host__ __device__ constexpr size_t bank_tile_bytes(int NG) { return bank_tile_bytes_u(2 * NG); }
__host__ __device__ constexpr size_t query_block_bytes(int NG) { return query_block_bytes_u(2 * NG); }

// number of v_mfma_f32_32x32x2 k-steps (2 k each) the f32 kernel is instantiated for
int pick_ks(int C)
{
    if (C <= 32) return 16;
    if (C <= 100) return 50;   // the reference's embedding width: 50 MFMAs, not 52
    if (C <= 104) return 52;
    return 64;
}
// number of v_mfma_f32_32x32x16_bf16 k-steps (16 k each) the bf16 kernels are instantiated for
int pick_ksb(int C)
{
    if (C <= 32) return 2;
    if (C <= 112) return 7;
    return 8;
}

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
        G.qt = (manet_tune_get(MANET_TUNE_BF16_VARIANT, 0) == 1) ? 256 : 512;
    }
    G.tile_bytes = bank_tile_bytes_u(G.units);
    G.qblk_bytes = query_block_bytes_u(G.units);
    return G;
}

struct BankLayout {
    Geom G;
    size_t tile_bytes;
    long T_max;  // upper bound on tiles: every object wastes < 1 tile
    long nblocks;  // pre-pass blocks of RPB rows
    size_t off_meta, off_hist, off_src, off_pack, total;
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
    return L;
}

struct MatchLayout {
    Geom G;
    long N_pad;
    int nQT;
    size_t qblk_bytes, off_q, off_keys, off_topk, total;
};

constexpr int TOPK_SPLITS = 16;  // the top-k path trades a little tail balance for a bounded workspace

MatchLayout match_layout(int64_t N, int C, int n_ids, int compute, int k_nn = 1)
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
    return L;
}

// Number of bank splits S (grid = query tiles x S).  Every split re-reads the query operand
// (S x 4C.N bytes of fabric traffic, the bank itself streams through each XCD's L2 once), so S
// should be as small as the tail allows: take the smallest multiple of 8 whose last round of
// workgroups over the `slots` resident workgroup slots is >= 97 % full, with >= 4 tiles per split.
int pick_splits(int nQT, long T_max, int slots)
{
    long cap = T_max / 4 / 8 * 8;
    if (cap < 8) cap = 8;
    if (cap > 256) cap = 256;
    int best = 8;
    double best_eff = 0.0;
    for (int S = 8; S <= cap; S += 8) {
        double rounds = (double)nQT * S / slots;
        double full = (double)(long)rounds;
        if (full < rounds) full += 1.0;
        double eff = rounds / full;
        if (eff >= 0.97) return S;
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

// bank (ROWS = 64) and query (ROWS = 32) pack: rows -> MFMA operand image (see Geom), then |row|^2.
// Rows are staged through LDS so that both the global reads (along k for row-major sources, along
// rows for C-major sources) and the 16-byte image writes are coalesced.  |row|^2 is the k-ascending
// fmaf chain of the oracle (IntVOS.py:32,35) -- over the bf16-rounded values in MANET_COMPUTE_BF16
// (the path then IS the reference formula on rounded embeddings), over the fp32 values otherwise.
template <int ROWS>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float *__restrict__ src, long s_row,
                                                        long s_c, const int *__restrict__ src_of,
                                                        const int *__restrict__ meta, long n_rows,
                                                        int C, int compute, int units, int kpad,
                                                        char *__restrict__ dst, long tile_bytes,
                                                        float pad_norm)
{
    const long tile = blockIdx.x;
    if (meta && tile >= meta[META_T]) return;
    extern __shared__ __attribute__((aligned(16))) char pack_smem[];
    const int KP = kpad + 1;  // odd row stride: column reads are conflict-free
    float *rows = (float *)pack_smem;                  // [ROWS][KP]
    int *s_src = (int *)(rows + (long)ROWS * KP);      // [ROWS]
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
            rows[r * KP + k] = (sr >= 0) ? src[(long)sr * s_row + k] : 0.0f;
        }
    } else {  // C-major (or generic) source: lanes along rows
        for (int idx = tid; idx < ROWS * C; idx += 256) {
            int k = idx / ROWS, r = idx - k * ROWS;
            int sr = s_src[r];
            rows[r * KP + k] = (sr >= 0) ? src[(long)sr * s_row + (long)k * s_c] : 0.0f;
        }
    }
    for (int idx = tid; idx < ROWS * (kpad - C); idx += 256) {
        int r = idx / (kpad - C), k = C + idx - r * (kpad - C);
        rows[r * KP + k] = 0.0f;
    }
    __syncthreads();
    char *out = dst + tile * tile_bytes;
    if (compute == MANET_COMPUTE_F32) {
        for (int item = tid; item < units * ROWS; item += 256) {
            int r = item % ROWS, u = item / ROWS;
            const float *row = rows + r * KP + 8 * (u >> 1) + (u & 1);
            f32x4 v = {row[0], row[2], row[4], row[6]};
            *(f32x4 *)(out + ((long)u * ROWS + r) * 16) = v;
        }
    } else {
        const int hi_units = (compute == MANET_COMPUTE_BF16X3) ? units / 2 : units;
        for (int item = tid; item < units * ROWS; item += 256) {
            int r = item % ROWS, u = item / ROWS;
            const bool lo = u >= hi_units;
            const int uu = lo ? u - hi_units : u;
            const float *row = rows + r * KP + 16 * (uu >> 1) + 8 * (uu & 1);
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x0 = row[2 * e], x1 = row[2 * e + 1];
                unsigned b0 = f2bf(x0), b1 = f2bf(x1);
                if (lo) {
                    b0 = f2bf(x0 - bf2f(b0));
                    b1 = f2bf(x1 - bf2f(b1));
                }
                w[e] = b0 | (b1 << 16);
            }
            *(uint4 *)(out + ((long)u * ROWS + r) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
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
        *(float *)(out + (long)units * ROWS * 16 + tid * 4) = n;
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
__device__ __forceinline__ unsigned key_of(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of(unsigned k)
{
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
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
template <int KS, int KNN>
__global__ __launch_bounds__(256, 2) void global_match_f32_kernel(const char *__restrict__ qpack,
                                                                  const char *__restrict__ bpack,
                                                                  const int *__restrict__ meta,
                                                                  int n_ids, int nQT, int S,
                                                                  long N_pad,
                                                                  unsigned *__restrict__ keys,
                                                                  float *__restrict__ topk, int block_map)
{
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
    const int b = blockIdx.x;
    int qt, s;
    if (block_map == 0) {
        const int xcd = b & 7;
        const int idx = b >> 3;
        qt = idx % nQT;
        s = xcd + 8 * (idx / nQT);
    } else if (block_map == 1) {  // tuning only: query tile fastest, no XCD awareness
        qt = b % nQT;
        s = b / nQT;
    } else {                      // tuning only: split fastest
        s = b % S;
        qt = b / S;
    }
    const int T = meta[META_T];
    const int t0 = (int)((long)s * T / S);
    const int t1 = (int)((long)(s + 1) * T / S);
    if (t0 >= t1) return;

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
    auto reset = [&]() {
#pragma unroll
        for (int j = 0; j < KNN; ++j) m0[j] = m1[j] = (KNN == 1) ? MANET_WRONG_LABEL_PADDING_DISTANCE : INFINITY;
    };
    reset();

    auto flush = [&](int obj) {
        if (KNN == 1) {
            float a = fminf(m0[0], __shfl_xor(m0[0], 32));
            float c = fminf(m1[0], __shfl_xor(m1[0], 32));
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
                if (KNN == 1) {
                    m0[0] = fminf(m0[0], fminf(d00, d10));
                    m1[0] = fminf(m1[0], fminf(d01, d11));
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
// main kernel, bf16 operands (MANET_COMPUTE_BF16 / _BF16X3): one workgroup = 512 queries x one bank
// split, 8 waves.  Same structure as the f32 kernel (swapped operands, object-pure 64-row tiles,
// lane-local running min, double-buffered global_load_lds staging, atomicMin across splits); the
// contraction is v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   X3 = false: one MFMA per 16 k on embeddings rounded to bf16 (7 instead of 50 MFMAs per block at C=100)
//   X3 = true : hi*hi + hi*lo + lo*hi with x = hi + lo, hi = bf16(x), lo = bf16(x - hi): the dropped
//               lo*lo term is <= 2^-16 relative, i.e. fp32-class distances at 3/16 of the f32 MFMA cost.
template <int KSB, bool X3, int NW, int TPS>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void global_match_bf16_kernel(const char *__restrict__ qpack,
                                                                const char *__restrict__ bpack,
                                                                const int *__restrict__ meta, int n_ids,
                                                                int nQT, int S, long N_pad,
                                                                unsigned *__restrict__ keys, int block_map)
{
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    constexpr int UNITS = 2 * KSB * (X3 ? 2 : 1);
    constexpr int LO = 2 * KSB;  // first unit of the lo image
    constexpr size_t TILE_BYTES = bank_tile_bytes_u(UNITS);
    constexpr size_t QBLK_BYTES = query_block_bytes_u(UNITS);
    constexpr int QTB = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x TPS x TILE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    const int b = blockIdx.x;
    int qt, s;
    if (block_map == 0) {
        const int xcd = b & 7;
        const int idx = b >> 3;
        qt = idx % nQT;
        s = xcd + 8 * (idx / nQT);
    } else {
        qt = b % nQT;
        s = b / nQT;
    }
    const int T = meta[META_T];
    const int t0 = (int)((long)s * T / S);
    const int t1 = (int)((long)(s + 1) * T / S);
    if (t0 >= t1) return;

    // register staging (see the f32 kernel) of one STEP = TPS consecutive tiles: one barrier per step
    constexpr int NV = (int)(TILE_BYTES / 16) * TPS;
    constexpr int NLD = (NV + NW * 64 - 1) / (NW * 64);
    constexpr int NTHR = NW * 64;
    constexpr size_t STEP_BYTES = TILE_BYTES * TPS;
    u32x4 Ra[NLD];
    // loads of a step's last partial tile group are clamped into the split's own range
#define MANET_GLOAD_STEP(R_, t_)                                                                    \
    {                                                                                               \
        const u32x4 *g_ = (const u32x4 *)(bpack + (size_t)(t_) * TILE_BYTES);                       \
        const int lim_ = ((t1 - (t_)) < TPS ? (t1 - (t_)) : TPS) * (int)(TILE_BYTES / 16);          \
        _Pragma("unroll") for (int i_ = 0; i_ < NLD; ++i_)                                          \
        {                                                                                           \
            const int idx_ = i_ * NTHR + tid;                                                       \
            R_[i_] = g_[idx_ < lim_ ? idx_ : lim_ - 1];                                             \
        }                                                                                           \
    }
#define MANET_LSTORE_STEP(R_, slot_)                                                                \
    {                                                                                               \
        u32x4 *l_ = (u32x4 *)(smem + (size_t)(slot_) * STEP_BYTES);                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < NLD; ++i_)                                          \
            if (i_ * NTHR + tid < NV) l_[i_ * NTHR + tid] = R_[i_];                                 \
    }
    MANET_GLOAD_STEP(Ra, t0);

    // B operand: lane holds q[j = lane&31][k = 16s + 8*(lane>>5) + 0..7] for its two query blocks
    uint4 q0[KSB], q1[KSB], q0l[X3 ? KSB : 1], q1l[X3 ? KSB : 1];
    float xs0, xs1;
    {
        const char *qb0 = qpack + (size_t)(qt * (QTB / QB) + wave * 2) * QBLK_BYTES;
        const char *qb1 = qb0 + QBLK_BYTES;
#pragma unroll
        for (int k = 0; k < KSB; ++k) {
            q0[k] = *(const uint4 *)(qb0 + ((size_t)(k * 2 + h) * QB + l31) * 16);
            q1[k] = *(const uint4 *)(qb1 + ((size_t)(k * 2 + h) * QB + l31) * 16);
            if (X3) {
                q0l[k] = *(const uint4 *)(qb0 + ((size_t)(LO + k * 2 + h) * QB + l31) * 16);
                q1l[k] = *(const uint4 *)(qb1 + ((size_t)(LO + k * 2 + h) * QB + l31) * 16);
            }
        }
        xs0 = *(const float *)(qb0 + (size_t)UNITS * QB * 16 + l31 * 4);
        xs1 = *(const float *)(qb1 + (size_t)UNITS * QB * 16 + l31 * 4);
    }
    const long qbase = (long)qt * QTB + wave * 64 + l31;

    int o = 0;
    while (meta[META_SEG + o + 1] <= t0) ++o;
    int seg_end = meta[META_SEG + o + 1];
    float m0 = MANET_WRONG_LABEL_PADDING_DISTANCE, m1 = MANET_WRONG_LABEL_PADDING_DISTANCE;
    auto flush = [&](int obj) {
        float a = fminf(m0, __shfl_xor(m0, 32));
        float c = fminf(m1, __shfl_xor(m1, 32));
        if (h == 0) {
            atomicMin(keys + (size_t)obj * N_pad + qbase, key_of(a));
            atomicMin(keys + (size_t)obj * N_pad + qbase + 32, key_of(c));
        }
    };
#define MANET_BF(x) __builtin_bit_cast(bf16x8_t, x)
    struct Acc {
        f32x16 c00, c01, c10, c11;
    };
    // the 4 x KSB (x3) MFMAs of one 64 x 64 tile block
    auto mfma_tile = [&](const char *tb, Acc &c) __attribute__((always_inline)) {
        const u32x4 *A = (const u32x4 *)tb;
        c.c00 = c.c01 = c.c10 = c.c11 = f32x16{0};
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
                c.c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0[k]), MANET_BF(q0[k]), c.c00, 0, 0, 0);
                c.c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0[k]), MANET_BF(q1[k]), c.c01, 0, 0, 0);
                c.c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1[k]), MANET_BF(q0[k]), c.c10, 0, 0, 0);
                c.c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1[k]), MANET_BF(q1[k]), c.c11, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < KSB; ++k) {
                u32x4 a0 = A[(k * 2 + h) * BT + l31];
                u32x4 a1 = A[(k * 2 + h) * BT + 32 + l31];
                u32x4 a0l = A[(LO + k * 2 + h) * BT + l31];
                u32x4 a1l = A[(LO + k * 2 + h) * BT + 32 + l31];
                c.c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0), MANET_BF(q0[k]), c.c00, 0, 0, 0);
                c.c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0), MANET_BF(q1[k]), c.c01, 0, 0, 0);
                c.c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1), MANET_BF(q0[k]), c.c10, 0, 0, 0);
                c.c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1), MANET_BF(q1[k]), c.c11, 0, 0, 0);
                c.c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0), MANET_BF(q0l[k]), c.c00, 0, 0, 0);
                c.c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0), MANET_BF(q1l[k]), c.c01, 0, 0, 0);
                c.c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1), MANET_BF(q0l[k]), c.c10, 0, 0, 0);
                c.c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1), MANET_BF(q1l[k]), c.c11, 0, 0, 0);
                c.c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0l), MANET_BF(q0[k]), c.c00, 0, 0, 0);
                c.c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a0l), MANET_BF(q1[k]), c.c01, 0, 0, 0);
                c.c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1l), MANET_BF(q0[k]), c.c10, 0, 0, 0);
                c.c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MANET_BF(a1l), MANET_BF(q1[k]), c.c11, 0, 0, 0);
            }
        }
    };
    // running min over the 64 x 64 distances of a finished tile block
    auto epilogue = [&](const char *tb, const Acc &c) __attribute__((always_inline)) {
        const float *ysl = (const float *)(tb + (size_t)UNITS * BT * 16);
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            f32x4 y0 = *(const f32x4 *)(ysl + 8 * tq + 4 * h);
            f32x4 y1 = *(const f32x4 *)(ysl + 32 + 8 * tq + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * tq + i;
                m0 = fminf(m0, fminf(fmaf(-2.0f, c.c00[r], xs0 + y0[i]), fmaf(-2.0f, c.c10[r], xs0 + y1[i])));
                m1 = fminf(m1, fminf(fmaf(-2.0f, c.c01[r], xs1 + y0[i]), fmaf(-2.0f, c.c11[r], xs1 + y1[i])));
            }
        }
    };
    auto next_object = [&](int t) __attribute__((always_inline)) {
        if (t >= seg_end) {  // wave-uniform: tile t starts another object's rows
            flush(o);
            m0 = m1 = MANET_WRONG_LABEL_PADDING_DISTANCE;
            do { ++o; seg_end = meta[META_SEG + o + 1]; } while (t >= seg_end);
        }
    };
    {
        // double-buffered LDS (2 x TPS tiles), register-staged prefetch one step ahead, ONE barrier per
        // step of TPS tiles.  With the bf16 MFMA a 64-row tile is only ~0.4 us of matrix work per wave, so
        // the fixed cost of a step (barrier skew, LDS write, first-fragment latency) is amortised over
        // TPS tiles.  (A ping-pong schedule of the two waves of a SIMD and a dual-accumulator software
        // pipeline were both measured: within noise of this simpler loop, see DESIGN.md.)
        Acc c;
        MANET_LSTORE_STEP(Ra, 0);
        if (t0 + TPS < t1) MANET_GLOAD_STEP(Ra, t0 + TPS);
        int buf = 0;
        for (int t = t0; t < t1; t += TPS, buf ^= 1) {
            __syncthreads();  // step's tiles visible in `buf`; the other buffer is free
            if (t + TPS < t1) {
                MANET_LSTORE_STEP(Ra, buf ^ 1);
                if (t + 2 * TPS < t1) MANET_GLOAD_STEP(Ra, t + 2 * TPS);
            }
#pragma unroll
            for (int u = 0; u < TPS; ++u) {
                if (t + u < t1) {
                    const char *tb = smem + (size_t)buf * STEP_BYTES + (size_t)u * TILE_BYTES;
                    next_object(t + u);
                    mfma_tile(tb, c);
                    epilogue(tb, c);
                }
            }
        }
    }
#undef MANET_GLOAD_STEP
#undef MANET_LSTORE_STEP
#undef MANET_BF
    flush(o);
}

// decode + (sigmoid-0.5)*2 (IntVOS.py:611-612) + min-merge with the stored map (IntVOS.py:620-622)
__global__ void global_finish_kernel(const unsigned *__restrict__ keys, long N, long N_pad, int n_ids,
                                     int flags, float *__restrict__ out, float *__restrict__ mem)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * n_ids) return;
    long n = i / n_ids;
    int o = (int)(i - n * n_ids);
    unsigned k = keys[(size_t)o * N_pad + n];
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
    if (compute != MANET_COMPUTE_F32 && compute != MANET_COMPUTE_BF16 && compute != MANET_COMPUTE_BF16X3)
        return manet_set_error(MANET_E_INVALID, "compute=%d (MANET_COMPUTE_F32 / _BF16 / _BF16X3)", compute);
    return MANET_OK;
}

template <int KS, int KNN>
void launch_main_f32(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S,
                     long N_pad, unsigned *keys, float *topk, hipStream_t st)
{
    size_t lds = 2 * bank_tile_bytes((KS + 3) / 4);
    // per call (cheap, host side): the attribute is per device and the library keeps no state
    (void)hipFuncSetAttribute((const void *)global_match_f32_kernel<KS, KNN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    manet_profile_record(st, true);
    hipLaunchKernelGGL((global_match_f32_kernel<KS, KNN>), dim3((unsigned)(nQT * S)), dim3(256), lds, st, qpack,
                       bpack, meta, n_ids, nQT, S, N_pad, keys, topk, manet_tune_get(MANET_TUNE_BLOCK_MAP, 0));
    manet_profile_record(st, false);
}

template <int KSB, bool X3, int NW, int TPS>
void launch_main_bf16_v(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S, long N_pad,
                        unsigned *keys, hipStream_t st)
{
    size_t lds = 2 * TPS * bank_tile_bytes_u(2 * KSB * (X3 ? 2 : 1));
    (void)hipFuncSetAttribute((const void *)global_match_bf16_kernel<KSB, X3, NW, TPS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    manet_profile_record(st, true);
    hipLaunchKernelGGL((global_match_bf16_kernel<KSB, X3, NW, TPS>), dim3((unsigned)(nQT * S)), dim3(NW * 64), lds, st,
                       qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, manet_tune_get(MANET_TUNE_BLOCK_MAP, 0));
    manet_profile_record(st, false);
}

// workgroup shape of the bf16 kernels: 0 (default) = 8 waves, 2 tiles per barrier for plain bf16;
// tuning: 1 = 4 waves x 2 workgroups per CU, 2 = 8 waves, 1 tile per barrier, 3 = 8 waves, 4 tiles
int bf16_variant() { return manet_tune_get(MANET_TUNE_BF16_VARIANT, 0); }

template <int KSB, bool X3>
void launch_main_bf16(const char *qpack, const char *bpack, const int *meta, int n_ids, int nQT, int S, long N_pad,
                      unsigned *keys, hipStream_t st)
{
    switch (bf16_variant()) {
    case 1: launch_main_bf16_v<KSB, X3, 4, X3 ? 1 : 2>(qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, st); break;
    case 2: launch_main_bf16_v<KSB, X3, 8, 1>(qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, st); break;
    case 3: launch_main_bf16_v<KSB, X3, 8, X3 ? 2 : 4>(qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, st); break;
    default: launch_main_bf16_v<KSB, X3, 8, X3 ? 1 : 2>(qpack, bpack, meta, n_ids, nQT, S, N_pad, keys, st); break;
    }
}

}  // namespace

extern "C" {

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
    {
        size_t lds = (size_t)BT * (L.G.kpad + 1) * sizeof(float) + BT * sizeof(int);
        hipLaunchKernelGGL(pack_rows_kernel<BT>, dim3((unsigned)L.T_max), dim3(256), lds, st, bank, (long)b_stride_m,
                           (long)b_stride_c, (const int *)src_of, (const int *)meta, (long)M0, C, compute, L.G.units,
                           L.G.kpad, ws + L.off_pack, (long)L.tile_bytes, MANET_WRONG_LABEL_PADDING_DISTANCE);
    }
    return manet_check_launch("manet_bank_prepare");
}

int manet_global_match_prepared(const float *query, int64_t q_stride_n, int64_t q_stride_c,
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
    fill32(keys, 0xffffffffu, (size_t)n_ids * ML.N_pad, st);
    {
        size_t lds = (size_t)QB * (ML.G.kpad + 1) * sizeof(float) + QB * sizeof(int);
        hipLaunchKernelGGL(pack_rows_kernel<QB>, dim3((unsigned)(ML.N_pad / QB)), dim3(256), lds, st, query,
                           (long)q_stride_n, (long)q_stride_c, (const int *)nullptr, (const int *)nullptr, (long)N,
                           C, compute, ML.G.units, ML.G.kpad, mws + ML.off_q, (long)ML.qblk_bytes, 0.0f);
    }
    // resident workgroup slots: f32 = 2 x 256-thread workgroups per CU, bf16 = 1 x 512-thread workgroup per CU
    int S = pick_splits(ML.nQT, BL.T_max, (compute == MANET_COMPUTE_F32 || ML.G.qt == 256) ? 512 : 256);
    {
        int forced = manet_tune_get(MANET_TUNE_SPLITS, 0);  // tuning only
        if (forced > 0) S = (forced + 7) / 8 * 8;
    }
    if (k_nn > 1) S = TOPK_SPLITS;
    const char *qpack = mws + ML.off_q;
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
            if (x3) launch_main_bf16<8, true>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st);
            else launch_main_bf16<8, false>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, st);
            break;
        }
#undef MANET_GB_CASE
    } else {
#define MANET_GM_CASE(KS_)                                                                                    \
    case KS_:                                                                                                 \
        if (k_nn == 1) launch_main_f32<KS_, 1>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, topk, st); \
        else launch_main_f32<KS_, MANET_MAX_KNN>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, topk, st); \
        break;
    switch (pick_ks(C)) {
        MANET_GM_CASE(16) MANET_GM_CASE(50) MANET_GM_CASE(52)
    default:
        if (k_nn == 1) launch_main_f32<64, 1>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, topk, st);
        else launch_main_f32<64, MANET_MAX_KNN>(qpack, bpack, meta, n_ids, ML.nQT, S, ML.N_pad, keys, topk, st);
        break;
    }
#undef MANET_GM_CASE
    }
    long total = (long)N * n_ids;
    if (k_nn == 1)
        hipLaunchKernelGGL(global_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const unsigned *)keys, (long)N, ML.N_pad, n_ids, epilogue_flags, out, mem_inout);
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
    int rc = check_common(N, M0, C, n_ids, k_nn, compute);
    if (rc) return rc;
    if (!workspace) return manet_set_error(MANET_E_INVALID, "workspace == NULL");
    size_t bbytes = bank_layout(M0, C, n_ids, compute).total;
    size_t mbytes = match_layout(N, C, n_ids, compute, k_nn).total;
    if (workspace_bytes < bbytes + mbytes)
        return manet_set_error(MANET_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, bbytes + mbytes);
    char *ws = (char *)workspace;
    rc = manet_bank_prepare(bank, b_stride_m, b_stride_c, labels, M0, C, n_ids, compute, ws, bbytes, stream);
    if (rc) return rc;
    return manet_global_match_prepared(query, q_stride_n, q_stride_c, ws, N, M0, C, n_ids, k_nn, compute, out,
                                       mem_inout, epilogue_flags, ws + bbytes, mbytes, stream);
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
