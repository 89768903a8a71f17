// Local (2d+1)^2 window matching against the previous frame, MI355X (gfx950).
//
// Replaces networks/IntVOS.py:266-315 (local_pairwise_distances2) and :345-434
// (local_previous_frame_nearest_neighbor_features_per_object; USE_CORRELATION_COST=False,
// MODEL_UNFOLD=True -- the live configuration, IntVOS.py:15-16).
//
// HBM-bound at small windows (d=4: ~7 flop/B), VALU-bound at the reference's default d=12 (SURVEY.md
// 8d): the point is to read each embedding once, coalesced along x out of the caller's C-major
// planes, and never to build the reference's [1,C,h',w',(2d+1)^2] unfold tensor (1.6 GB at d=12).
//   pool2x2_kernel      both frames -> [C][h'][w'] planes (2x2 mean; floor sizes)
//   local_dist_kernel<D> workgroup = rows x 16 columns of the (pooled) grid; the previous frame's halo
//                       tile goes through LDS in channel stages, a thread slides a 2-column window over
//                       one window row: 2 x (2d+1) running sums over C of (x - y)^2, direct form as the
//                       reference (no |x|^2+|y|^2-2xy cancellation); out-of-image neighbours are the
//                       reference's 1e20 padding -> inf -> 1.0 after normalisation
//   local_min_kernel    64 full-resolution pixels x 8 waves: bilinear (align_corners) sample of the
//                       pooled, normalised volume + stride-2 label gather + masked min per object;
//                       waves split the window rows, partial minima meet in LDS
//   local_upsample_kernel  only for the stand-alone local_pairwise_distances2 API
#include "manet_common.h"
#include "local_geom.h"
#include <type_traits>

namespace {

// (window width P = 2d+1 <= 2*MANET_MAX_LOCAL_DISTANCE+1 = 25)

// IntVOS.py:282-284  F.avg_pool2d(x, (2,2), (2,2)): window summed row-major, times 1/4 (exact)
__global__ void pool2x2_kernel(const float *__restrict__ a, long a_sy, long a_sx, long a_sc,
                               const float *__restrict__ b, long b_sy, long b_sx, long b_sc, int C, int hp,
                               int wp, float *__restrict__ ap, float *__restrict__ bp)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long plane = (long)hp * wp;
    if (i >= plane * C) return;
    int c = (int)(i / plane);
    int rem = (int)(i - (long)c * plane);
    int py = rem / wp, px = rem - py * wp;
    const float *p = a + (2L * py) * a_sy + (2L * px) * a_sx + (long)c * a_sc;
    const float *q = b + (2L * py) * b_sy + (2L * px) * b_sx + (long)c * b_sc;
    ap[i] = (((p[0] + p[a_sx]) + p[a_sy]) + p[a_sy + a_sx]) * 0.25f;
    bp[i] = (((q[0] + q[b_sx]) + q[b_sy]) + q[b_sy + b_sx]) * 0.25f;
}

// IntVOS.py:288-294 (pooled grid, normalised) and :300-313 (full grid, raw).
// x = current/query frame, y = previous frame, both [H][W][C] with element strides.
// pooled_out != 0: out[l][H][W] = (sigmoid(dist)-0.5)*2 ; else out[H][W][P*P] = dist.
//
// Workgroup = RY grid rows x 16 columns, all P*P offsets.  Thread = (row ry, window row dy, pair of
// columns): 2 x P running sums in registers.  Per chunk of LD_CC channels the y halo tile
// [(RY+2d) rows][16+2d cols] (out-of-image = the reference's 1e20 padding, IntVOS.py:287, so
// (x-1e20)^2 = inf needs no bounds logic in the inner loop) and the x tile are staged in LDS with
// row-contiguous global reads; a thread then slides its 2+2d wide window over the P offsets out
// of registers: (2+2d)/2 ds_read_b64 feed 2*P sub+fma pairs.  Channels are accumulated in ascending
// order with fmaf -- the same arithmetic as the oracle.
constexpr int LD_TX = 16;  // columns per workgroup
// tile geometry as a function of the window radius d (all compile-time in the kernel):
__host__ __device__ constexpr int ld_ry(int d)  // grid rows per workgroup: 32 (row, dy) slots
{
    return (32 / (2 * d + 1)) < 1 ? 1 : ((32 / (2 * d + 1)) > 8 ? 8 : (32 / (2 * d + 1)));
}
__host__ __device__ constexpr int ld_cw(int d) { return (LD_TX + 2 * d + 3) & ~3; }  // halo row stride, 16 B rows
__host__ __device__ constexpr int ld_yplane(int d) { return (ld_ry(d) + 2 * d) * ld_cw(d); }
__host__ __device__ constexpr int ld_nj(int d) { return (ld_yplane(d) + 255) / 256; }  // halo elements per thread
// channels per LDS stage: as many as two stages fit in 64 KiB and ~80 staging registers allow -- a
// stage costs one global round trip + one barrier, so few, fat stages (4 at d=4, 13 at d=12 for C=100)
__host__ __device__ constexpr int ld_cc(int d)
{
    int by_lds = 65536 / (8 * (ld_yplane(d) + ld_ry(d) * LD_TX));
    int by_regs = 80 / (ld_nj(d) + 1);
    int cc = by_lds < by_regs ? by_lds : by_regs;
    return cc > 25 ? 25 : (cc < 4 ? 4 : cc);
}
template <int D>  // window radius: compile-time so the 2 x P accumulators unroll without predicates
__global__ __launch_bounds__(256, 2) void local_dist_kernel(const float *__restrict__ x, long x_sy, long x_sx,
                                                            long x_sc, const float *__restrict__ y, long y_sy,
                                                            long y_sx, long y_sc, int H, int W, int C,
                                                            int pooled_out, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int d = D;
    constexpr int P = 2 * d + 1;
    constexpr int MAXP = P;  // arrays are exactly P wide
    constexpr int RY = ld_ry(D), LD_CC = ld_cc(D), LD_NJ = ld_nj(D);
    constexpr int RYH = RY + 2 * d;  // halo rows
    constexpr int CW = ld_cw(D);
    constexpr int yplane = RYH * CW, xplane = RY * LD_TX;
    const int buf_floats = LD_CC * (yplane + xplane);  // one stage: ys [LD_CC][RYH][CW] then xs [LD_CC][RY][16]
    float *smem = (float *)smem_raw;
    const int tid = threadIdx.x;
    const int px0 = blockIdx.x * LD_TX, py0 = blockIdx.y * RY;

    // staging map, fixed for the whole kernel: this thread's LD_NJ elements of the halo plane and its
    // one element of the x plane (global offsets without the channel term; -1 = outside the image)
    long yoff[LD_NJ];
    int ylds[LD_NJ];
#pragma unroll
    for (int j = 0; j < LD_NJ; ++j) {
        int e = tid + 256 * j;
        ylds[j] = (e < yplane) ? e : -1;
        int r = e / CW, col = e - r * CW;
        int yy = py0 - d + r, xx = px0 - d + col;
        yoff[j] = (e < yplane && yy >= 0 && yy < H && xx >= 0 && xx < W) ? ((long)yy * y_sy + (long)xx * y_sx) : -1;
    }
    long xoff = -1;
    const bool has_x = tid < xplane;
    if (has_x) {
        int r = tid / LD_TX, col = tid - r * LD_TX;
        if (py0 + r < H && px0 + col < W) xoff = (long)(py0 + r) * x_sy + (long)(px0 + col) * x_sx;
    }
    float ry_[LD_CC][LD_NJ], rx_[LD_CC];
    auto load_regs = [&](int c0) {  // all loads of a stage issued back to back
#pragma unroll
        for (int c = 0; c < LD_CC; ++c) {
            const bool cin = (c0 + c) < C;
#pragma unroll
            for (int j = 0; j < LD_NJ; ++j)
                ry_[c][j] = (cin && yoff[j] >= 0) ? y[yoff[j] + (long)(c0 + c) * y_sc] : (cin ? 1e20f : 0.0f);
            rx_[c] = (cin && xoff >= 0) ? x[xoff + (long)(c0 + c) * x_sc] : 0.0f;
        }
    };
    auto store_lds = [&](int buf) {
        float *ys = smem + (long)buf * buf_floats;
        float *xs = ys + LD_CC * yplane;
#pragma unroll
        for (int c = 0; c < LD_CC; ++c) {
#pragma unroll
            for (int j = 0; j < LD_NJ; ++j)
                if (ylds[j] >= 0) ys[c * yplane + ylds[j]] = ry_[c][j];
            if (has_x) xs[c * xplane + tid] = rx_[c];
        }
    };

    // thread -> (ry, dy, g): g = column pair
    const int g = tid & 7;
    const int dy = (tid >> 3) % P;
    const int ry = (tid >> 3) / P;
    const bool active = ry < RY;
    float acc0[MAXP], acc1[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) acc0[i] = acc1[i] = 0.0f;

    load_regs(0);
    store_lds(0);
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < C; c0 += LD_CC, buf ^= 1) {
        const bool more = (c0 + LD_CC) < C;
        if (more) load_regs(c0 + LD_CC);  // next stage's global loads fly under this stage's math
        if (active) {
            const float *ys = smem + (long)buf * buf_floats;
            const float *xs = ys + LD_CC * yplane;
#pragma unroll 2
            for (int c = 0; c < LD_CC; ++c) {  // channels beyond C were staged as x = y = 0 (adds 0) / 1e20 (already inf)
                const float *yrow = ys + c * yplane + (ry + dy) * CW + 2 * g;
                const float2 xv = *(const float2 *)(xs + c * xplane + ry * LD_TX + 2 * g);
                float win[MAXP + 1];
#pragma unroll
                for (int i = 0; i < (MAXP + 1) / 2; ++i) {
                    if (2 * i < P + 1) {
                        float2 t = *(const float2 *)(yrow + 2 * i);
                        win[2 * i] = t.x;
                        win[2 * i + 1] = t.y;
                    }
                }
#pragma unroll
                for (int dx = 0; dx < MAXP; ++dx) {
                    if (dx < P) {
                        float d0 = xv.x - win[dx];
                        float d1 = xv.y - win[dx + 1];
                        acc0[dx] = fmaf(d0, d0, acc0[dx]);
                        acc1[dx] = fmaf(d1, d1, acc1[dx]);
                    }
                }
            }
        }
        if (more) store_lds(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    const int py = py0 + ry, pxa = px0 + 2 * g;
    if (py >= H) return;
#pragma unroll
    for (int dx = 0; dx < MAXP; ++dx) {
        if (dx < P) {
            const int l = dy * P + dx;
            if (pooled_out) {
                float *o = out + ((long)l * H + py) * W + pxa;
                if (pxa < W) o[0] = manet_normalize_dist_local(acc0[dx]);
                if (pxa + 1 < W) o[1] = manet_normalize_dist_local(acc1[dx]);
            } else {
                if (pxa < W) out[((long)py * W + pxa) * (P * P) + l] = acc0[dx];
                if (pxa + 1 < W) out[((long)py * W + pxa + 1) * (P * P) + l] = acc1[dx];
            }
        }
    }
}

struct DistLaunch;
template <int D>
static void launch_dist_d(const DistLaunch &DL, hipStream_t st, const float *x, long x_sy, long x_sx, long x_sc,
                          const float *y, long y_sy, long y_sx, long y_sc, int H, int W, int C, int pooled_out,
                          float *out);
struct DistLaunch {
    int RY;
    dim3 grid;
    size_t lds;
};
static DistLaunch dist_launch(int H, int W, int d)
{
    DistLaunch L;
    const int P = 2 * d + 1;
    (void)P;
    int ry = ld_ry(d);  // 256 threads = 32 (row, dy) slots x 8 column pairs
    L.RY = ry;
    L.grid = dim3((unsigned)((W + LD_TX - 1) / LD_TX), (unsigned)((H + ry - 1) / ry));
    L.lds = 2 * (size_t)ld_cc(d) * ((size_t)ld_yplane(d) + (size_t)ry * LD_TX) * sizeof(float);  // 2 stages
    return L;
}
template <int D>
static void launch_dist_d(const DistLaunch &DL, hipStream_t st, const float *x, long x_sy, long x_sx, long x_sc,
                          const float *y, long y_sy, long y_sx, long y_sc, int H, int W, int C, int pooled_out,
                          float *out)
{
    hipLaunchKernelGGL(local_dist_kernel<D>, DL.grid, dim3(256), DL.lds, st, x, x_sy, x_sx, x_sc, y, y_sy, y_sx, y_sc,
                       H, W, C, pooled_out, out);
}
static void launch_dist(int d, hipStream_t st, const float *x, long x_sy, long x_sx, long x_sc, const float *y,
                        long y_sy, long y_sx, long y_sc, int H, int W, int C, int pooled_out, float *out)
{
    DistLaunch DL = dist_launch(H, W, d);
#define MANET_LD_CASE(D_) case D_: launch_dist_d<D_>(DL, st, x, x_sy, x_sx, x_sc, y, y_sy, y_sx, y_sc, H, W, C, pooled_out, out); break;
    switch (d) {
        MANET_LD_CASE(0) MANET_LD_CASE(1) MANET_LD_CASE(2) MANET_LD_CASE(3) MANET_LD_CASE(4) MANET_LD_CASE(5)
        MANET_LD_CASE(6) MANET_LD_CASE(7) MANET_LD_CASE(8) MANET_LD_CASE(9) MANET_LD_CASE(10) MANET_LD_CASE(11)
        MANET_LD_CASE(12)
    default: break;  // check_local() rejected it already
    }
#undef MANET_LD_CASE
}

__device__ __forceinline__ float bilin_sample(const float *__restrict__ pl, int wp, const Bilin &by,
                                              const Bilin &bx)
{
    return by.l0 * (bx.l0 * pl[by.i0 * wp + bx.i0] + bx.l1 * pl[by.i0 * wp + bx.i1]) +
           by.l1 * (bx.l0 * pl[by.i1 * wp + bx.i0] + bx.l1 * pl[by.i1 * wp + bx.i1]);
}

// IntVOS.py:398-408 (label unfold, stride 2, zero padding) + :428-432 (where / min).
// pooled != 0: dvol = [P*P][hp][wp] normalised pooled volume, sampled bilinearly at (y, x);
// pooled == 0: dvol = [h][w][P*P] full-resolution raw distances.
// Workgroup = 64 consecutive pixels x LM_WAVES waves; wave k takes window rows by = k, k+LM_WAVES, ..
// (lanes run along x, so tap and label loads are row segments); partial minima meet in LDS.
constexpr int NI = 8;        // object ids per pass
constexpr int LM_WAVES = 8;  // waves per workgroup
__global__ __launch_bounds__(64 * LM_WAVES) void local_min_kernel(const float *__restrict__ dvol, int pooled,
                                                                  const int *__restrict__ labels, int h, int w,
                                                                  int hp, int wp, int d, int n_ids,
                                                                  float *__restrict__ out)
{
    __shared__ float red[LM_WAVES][NI][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    const bool valid = i < (long)h * w;
    const int y = valid ? (int)(i / w) : 0, x = valid ? (int)(i - (long)y * w) : 0;
    const int P = 2 * d + 1;
    Bilin cy = {0, 0, 0.f, 0.f}, cx = {0, 0, 0.f, 0.f};
    if (pooled) {
        cy = bilin_coeff(y, hp, h);
        cx = bilin_coeff(x, wp, w);
    }
    const long plane = (long)hp * wp;
    for (int o0 = 0; o0 < n_ids; o0 += NI) {
        float m[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) m[k] = INFINITY;
        if (valid) {
            for (int by = wave; by < P; by += LM_WAVES) {
                const int yy = y + 2 * (by - d);
                const bool yin = (yy >= 0 && yy < h);
#pragma unroll 5
                for (int bx = 0; bx < P; ++bx) {
                    const int xx = x + 2 * (bx - d);
                    const int lab = (yin && xx >= 0 && xx < w) ? labels[(long)yy * w + xx] : 0;
                    const int l = by * P + bx;
                    const float v = pooled ? bilin_sample(dvol + (long)l * plane, wp, cy, cx) : dvol[i * (P * P) + l];
#pragma unroll
                    for (int k = 0; k < NI; ++k) m[k] = fminf(m[k], (lab == o0 + k) ? v : 1.0f);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) red[wave][k][lane] = m[k];
        __syncthreads();
        if (wave == 0 && valid) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                float r = red[0][k][lane];
#pragma unroll
                for (int ww = 1; ww < LM_WAVES; ++ww) r = fminf(r, red[ww][k][lane]);
                if (o0 + k < n_ids) out[i * n_ids + o0 + k] = r;
            }
        }
        __syncthreads();
    }
}

// IntVOS.py:295-296: the resized volume itself, [h][w][P*P]
__global__ void local_upsample_kernel(const float *__restrict__ dvol, int h, int w, int hp, int wp, int PP,
                                      float *__restrict__ out)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)h * w * PP) return;
    int l = (int)(i % PP);
    long pix = i / PP;
    int y = (int)(pix / w), x = (int)(pix - (long)y * w);
    Bilin cy = bilin_coeff(y, hp, h), cx = bilin_coeff(x, wp, w);
    out[i] = bilin_sample(dvol + (long)l * hp * wp, wp, cy, cx);
}

// ---------------------------------------------------------------------------------------------
// FUSED local match (the live configuration: downsample on): window distances -> normalise -> bilinear ->
// stride-2 label gather -> masked min in ONE kernel behind the pooling pass; the (2d+1)^2 volume never touches
// memory (r1: pooled volume written and re-read by a third launch, 47 us at d=4 / 125 us at d=12).
//
// Decomposition.  A full-resolution pixel (y, x) reads the pooled volume at rows {i0, i1}, columns {j0, j1}
// (bilinear, align_corners).  A workgroup owns the pixels whose (i0, j0) falls into a TY x 15 block of the
// pooled grid; they need the volume on a (TY+1) x 16 block S (a one-sided apron), for every window offset.
//   phase 1  distances on S: thread = (row of S, window row dy, group of COLS columns[, half of the window columns]);
//            running sums over C held as packed pairs of neighbouring window positions (v_pk_add_f32 / v_pk_fma_f32);
//            channels staged through LDS in double-buffered stages by asm LDS-DMA, one barrier per stage; a thread
//            slides its window over the offsets out of registers, the channel loop rotated by hand.  Per stored sum the
//            same ascending chain of fma(d, d, acc), d = x - y, as local_dist_kernel and the oracle.
//   phase 2  the normalised volume of S goes to LDS ([dy][cell][dx], dx contiguous), the previous frame's labels
//            around the tile too (one byte each, prefetched during phase 1), and every (pixel, window row) item walks
//            its row four columns at a time: b128 taps, the oracle's bilinear expression on two columns per packed op,
//            an LDS atomic min on the float bits into the (id, pixel) slot.  Bit-identical to r1's three-launch path
//            (local_dist_kernel -> local_min_kernel), which shares manet_normalize_dist_local.
// The pooling pass writes PADDED planes (lf_pool_pad_kernel): a border of the reference's padding value (1e20 for
// the previous frame, IntVOS.py:287) wide enough that no tile ever leaves the plane, rows a multiple of 16 bytes --
// staging is branch-free -- and the tile table (first pixel row / column of every tile).
// Wide windows (d >= 7) use COLS = 4: window reads are aligned ds_read_b128 and feed 4 x P fma pairs.
// For d >= 11 the window rows are dealt to NDG workgroups per tile (d=11: 2, d=12: 5); their partial minima meet
// by atomicMin on the float bits (all candidates lie in [0, 1]; `out` is pre-set to 1.0, the reference's
// "no match" value, IntVOS.py:429-430).
// DESIGN.md 3.4 has the measured phase timeline and the VALU-rate analysis.
// IntVOS.py:282-284 F.avg_pool2d(x, (2,2), (2,2)) of both frames into padded planes: window summed row-major,
// times 1/4 (exact) -- the arithmetic of pool2x2_kernel; border = 0 for the current frame (never used), 1e20 for
// the previous frame (IntVOS.py:287: (x - 1e20)^2 = inf -> 1.0 after normalisation, no bounds logic downstream)
__device__ __forceinline__ float lf_ld(const float *p, long i) { return p[i]; }
__device__ __forceinline__ float lf_ld(const unsigned short *p, long i) { return __uint_as_float((unsigned)p[i] << 16); }  // bf16
// out_init != nullptr (window rows dealt to several workgroups per tile): `out` is pre-set to 1.0 here, the value the
// partial minima are atomicMin'ed into -- one launch fewer than a separate fill.
template <typename SRC>  // float, or bf16 as raw 16-bit words (the producer's storage)
__global__ void lf_pool_pad_kernel(const SRC *__restrict__ a, long a_sy, long a_sx, long a_sc,
                                   const SRC *__restrict__ b, long b_sy, long b_sx, long b_sc, int C, int hp, int wp,
                                   int d, int HPAD, int WS, float *__restrict__ ap, float *__restrict__ bp,
                                   float *__restrict__ out_init, long n_out, int *__restrict__ tab, int TY, int TX,
                                   int nty, int ntx, int h, int w)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long plane = (long)HPAD * WS;
    if (out_init && i < n_out) out_init[i] = 1.0f;  // the launcher checked n_out <= plane * C
    // tile table of the fused kernel: tile row t owns the full-resolution rows [tab[t], tab[t+1]) (those whose upper
    // bilinear source row i0 lies in [t TY, (t+1) TY)), tile column t the columns [tab[nty+1+t], tab[nty+2+t]).
    // Computed here once instead of by every workgroup (four divisions and two scans on its critical path).
    if (i <= nty) tab[i] = bilin_first((int)i * TY, hp, h);
    else if (i <= nty + 1 + ntx) tab[i] = bilin_first((int)(i - nty - 1) * TX, wp, w);
    if (i >= plane * C) return;
    int c = (int)(i / plane);
    int rem = (int)(i - (long)c * plane);
    int r = rem / WS, col = rem - r * WS;
    int py = r - d, px = col - d;
    float va = 0.0f, vb = 1e20f;
    if (py >= 0 && py < hp && px >= 0 && px < wp) {
        const SRC *p = a + (2L * py) * a_sy + (2L * px) * a_sx + (long)c * a_sc;
        const SRC *q = b + (2L * py) * b_sy + (2L * px) * b_sx + (long)c * b_sc;
        va = (((lf_ld(p, 0) + lf_ld(p, a_sx)) + lf_ld(p, a_sy)) + lf_ld(p, a_sy + a_sx)) * 0.25f;
        vb = (((lf_ld(q, 0) + lf_ld(q, b_sx)) + lf_ld(q, b_sy)) + lf_ld(q, b_sy + b_sx)) * 0.25f;
    }
    ap[i] = va;
    bp[i] = vb;
}

// Phase timeline of the fused kernel (development aid, off by default): build with
// `make -C cvpr2020_manet_amd/csrc EXTRA=-DMANET_LF_TIMELINE`, run tools/local_timeline.py on the GPU box.
#ifdef MANET_LF_TIMELINE
__device__ unsigned long long lf_dbg[8192 * 8];
extern "C" int manet_dbg_read(unsigned long long *host, size_t n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(lf_dbg), n * 8); }
#define LF_T(k) if (threadIdx.x == 0) lf_dbg[blockIdx.x * 8 + (k)] = wall_clock64();
#else
#define LF_T(k)
#endif
// MODE (r6): the window distances depend on the two EMBEDDINGS only (IntVOS.py:266-296), the masked minimum (:398-432) on the
// previous frame's labels -- and a clip's embeddings never change between the interaction rounds of a session (test.py:137-154
// extracts them once per sequence).  LF_VOL_OUT: phase 1 alone, for a BATCH of frame pairs in one launch (blockIdx.y = pair): each
// workgroup's normalised volume image -- the exact LDS image phase 2 reads, [ND][SY * 16][VS] floats -- goes to memory.  LF_VOL_IN:
// phase 2 alone on such an image (one LDS-DMA stream instead of the channel stages): what the sequential chain runs from a
// clip's second round on.  LF_FUSED: both in one launch, the volume never in memory (r1-r5).  Same arithmetic, same bits.
constexpr int LF_FUSED = 0, LF_VOL_OUT = 1, LF_VOL_IN = 2;
constexpr int LF_BATCH = 32;  // frame pairs per LF_VOL_OUT launch (their pointers travel as a kernel argument)
struct LfBatch {
    const float *cur[LF_BATCH], *prev[LF_BATCH];
    float *vol[LF_BATCH];
};
// LF_VOL_IN: the tile table (first pixel row / column of every tile) computed on the HOST with the device's expression (fp32 IEEE
// operations, no contraction: the same integers) and handed over as a kernel argument -- a workgroup's label loads then depend on no
// memory round trip (under the volume stream the table read alone took 2.5 us, timeline in docs/history/r06_experiments.md)
constexpr int LF_TAB_MAX = 96;
struct LfTab {
    int n;  // entries (nty + 1 + ntx + 1), or 0: read the frame's table from memory
    int v[LF_TAB_MAX];
};
// wave-uniform counted wait (LDS-DMA pieces and asm-issued loads are invisible to the compiler's own counters)
__device__ __forceinline__ void lf_wait_vmcnt(int n)
{
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // (more pieces per wave than the table: wait for all)
    }
}
// floats of one workgroup's volume image in memory: the LDS image padded to whole 1 KiB LDS-DMA pieces
__host__ __device__ constexpr int lf_img_floats(int d) { return (lf_nd(d) * lf_sy(d) * LF_SX * lf_vs(d) + 255) / 256 * 256; }
template <int D, int MODE>
__global__ __launch_bounds__(lf_nt(D)) void local_fused_kernel(const float *__restrict__ curp_arg,
                                                               const float *__restrict__ prevp_arg, int WS, long PS,
                                                               const int *__restrict__ labels, int h, int w, int C,
                                                               int n_ids, float *__restrict__ out,
                                                               const int *__restrict__ tab, int abl_arg, int ntx, int nty,
                                                               int rw, int rh, float *__restrict__ vol_arg,
                                                               const typename std::conditional<MODE == LF_VOL_OUT, LfBatch,
                                                                   typename std::conditional<MODE == LF_VOL_IN, LfTab, int>::type>::type batch)
{
    const float *curp = curp_arg, *prevp = prevp_arg;
    float *vol = vol_arg;
    if constexpr (MODE == LF_VOL_OUT) {
        curp = batch.cur[blockIdx.y];
        prevp = batch.prev[blockIdx.y];
        vol = batch.vol[blockIdx.y];
    }
    // (the ablation switch is a compile-time 0 outside -DMANET_ABLATION builds: no run-time tests in the loops)
#ifdef MANET_ABLATION
    const int abl = abl_arg;
#else
    constexpr int abl = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int P = 2 * D + 1, NT = MODE == LF_VOL_IN ? lf_ntv(D) : lf_nt(D), ND = lf_nd(D), NDG = lf_ndg(D), SY = lf_sy(D), TY = SY - 1;
    constexpr int TX = LF_SX - 1, CW = lf_cw(D), YR = lf_yr(D), CC = lf_cc(D), COLS = lf_cols(D), NG = LF_SX / COLS;
    constexpr int DXS = lf_dxs(D), PA = lf_pa(D), LF_PH = lf_ph(D);
    constexpr int yplane = YR * CW, xplane = SY * LF_SX;
    constexpr int NVY = yplane / 4, NVX = xplane / 4;          // float4 per channel
    constexpr int NITEM = CC * (NVY + NVX);                    // float4 per stage
    constexpr int NPIECE = (NITEM + 63) / 64;                  // 1 KiB LDS-DMA pieces per stage
    constexpr int NWV = NT / 64, KD = (NPIECE + NWV - 1) / NWV;  // pieces per wave
    constexpr int buf_floats = lf_stage_floats(D);             // = NPIECE * 256
    float *smem = (float *)smem_raw;
    const int tid = threadIdx.x;
    LF_T(0)
    const int hp = h / 2, wp = w / 2;
    // XCD-aware block -> (tile column, tile row, window-row group): block L runs on XCD L % 8 (observed, used for speed
    // only), whose L2 serves the staging DMA.  Each XCD gets a compact rw x rh block of tiles (4 x 2 such regions cover the
    // grid) with all their window-row groups: its L2 then holds about a fifth of each pooled plane (region + halo) instead
    // of a full-height strip per tile column -- r3 PMC at 480p, d=12: the kernel's fabric fetch 33 -> 22 MB.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tix = (xcd & 3) * rw + idx % rw, tmp_ = idx / rw;
    // LF_VOL_IN: an image's window rows are dealt to NSUB workgroups of NDR rows (lf_ndv): two of them fit a CU
    constexpr int NDR = MODE == LF_VOL_IN ? lf_ndv(D) : ND, NSUB = MODE == LF_VOL_IN ? lf_nsub(D) : 1;
    const int tiy = (xcd >> 2) * rh + tmp_ % rh, tizs = tmp_ / rh;
    const int tiz = tizs / NSUB, sub = tizs - tiz * NSUB;
    if (tix >= ntx || tiy >= nty) return;  // (the regions' padding)
    const int a = tiy * TY, b0 = tix * TX;  // pooled origin of S
    const int dy0 = tiz * ND + sub * NDR;   // first window row of this workgroup
    // window rows this workgroup really owns (the last group of an image / the last image of a tile may hold fewer)
    const int nd_here = min(min(P - dy0, ND - sub * NDR), NDR);
    if (nd_here <= 0) return;
    float *vimg = MODE == LF_FUSED ? nullptr
                                   : vol + (long)((tiz * nty + tiy) * ntx + tix) * lf_img_floats(D) + (long)sub * NDR * (SY * LF_SX * lf_vs(D));

    // ---- phase 1: distances on S for window rows dy0 .. dy0+ND-1 ---------------------------------
    // Staging by LDS-DMA (lds_dma16: 64 lanes x 16 bytes land in 1 KiB of LDS, no VGPR hop, no ds_write): a stage is
    // NPIECE such pieces, dealt round-robin to the waves.  The stage image is [CC][YR][CW] previous-frame halo rows then
    // [CC][SY][16] current-frame rows; a lane's 16 bytes of piece p are float4 number 64 p + lane of that image.  Its
    // source address for stage 0 is computed once (below); every later stage is a uniform step of CC planes.
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem_raw);
    const float *src[KD];
    int sch[KD];  // the item's channel inside the stage (the last stage may run past C: clamped there)
#pragma unroll
    for (int k = 0; k < KD; ++k) {
        int i = (k * NWV + wave) * 64 + lane;
        i = i < NITEM ? i : NITEM - 1;  // the tail of the last piece lands in the padding of the stage image
        if (i < CC * NVY) {
            const int c = i / NVY, v = i - c * NVY;
            const int r = v / (CW / 4), q = v - r * (CW / 4);
            // padded coordinates: image row a - D + dy0 + r sits at plane row a + dy0 + r
            src[k] = prevp + (long)c * PS + (a + dy0 + r) * WS + b0 + 4 * q;
            sch[k] = c;
        } else {
            const int j = i - CC * NVY;
            const int c = j / NVX, v = j - c * NVX;
            const int r = v / 4, q = v - r * 4;
            src[k] = curp + (long)c * PS + (D + a + r) * WS + D + b0 + 4 * q;
            sch[k] = c;
        }
    }
    auto stage_dma = [&](int c0, int buf) __attribute__((always_inline)) {
        const bool tail = c0 + CC > C;  // uniform
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            if (k * NWV + wave < NPIECE) {  // wave-uniform
                const float *g_ = src[k] + (long)c0 * PS;
                if (tail) {
                    const int over = c0 + sch[k] - (C - 1);
                    if (over > 0) g_ -= (long)over * PS;  // channels past C re-read plane C-1; the arithmetic skips them
                }
                lds_dma16(g_, smem_base + (unsigned)buf * (unsigned)(buf_floats * 4) + (unsigned)(k * NWV + wave) * 1024u);
            }
        }
    };

    // lane -> (column group, slot): the window half is the thread index's TOP bit, so a wave's lanes read the same half
    // (with lf_cw's row stride that keeps every b128 lane group of the window reads conflict-free)
    const int g = tid % NG;
    constexpr bool XH_TOP = COLS == 4;  // (b64 reads of the COLS = 2 kernels: 32-lane groups, the interleaved order measured better)
    const int xh = XH_TOP ? tid / (NT / DXS) : (tid / NG) % DXS;  // which half of the window columns
    const int slot = XH_TOP ? (tid % (NT / DXS)) / NG : tid / (NG * DXS);
    const int dyi = slot % ND;
    const int ry = slot / ND;
    const bool active = ry < SY;
    const int dx_lo = xh * LF_PH;
    const int ndx = (xh == DXS - 1) ? P - dx_lo : LF_PH;  // window columns this thread really owns (<= PA)
    constexpr int WN = (COLS + PA - 1 + COLS - 1) / COLS * COLS;  // window floats read per channel (whole vectors)
    // Running sums, held as PAIRS of neighbouring window positions k = dx + j (k even first): one v_pk_add_f32 and one
    // v_pk_fma_f32 do two (x - y)^2 steps.  A wave issues a VALU instruction every ~5 cycles whatever its width
    // (tools/ubench/valu_rate.hip: v_fma_f32 5.3, v_pk_fma_f32 5.5 ticks per instruction per wave, flat from 1 to 4 waves
    // per SIMD), so at the 2 waves per SIMD this kernel runs at, packed math is twice the arithmetic per issue slot.
    // The packed operands must be even-aligned register pairs, which window pairs starting at even k are (they come out
    // of b64 / b128 LDS reads); column j's range k = j .. j+PA-1 is covered by the aligned pairs around it, so up to
    // one lane per end accumulates a neighbour that is never stored.  Every stored sum is the same ascending chain of
    // fma(d, d, acc), d = x - y, as local_dist_kernel's and the oracle's.
    constexpr int NP = (PA + 2) / 2;  // pairs per column
    f32x2 accp[COLS][NP];             // sum (j, dx) = accp[j][(dx + (j & 1)) >> 1][(dx + j) & 1]
#define LF_ACC(j_, dx_) accp[j_][((dx_) + ((j_) & 1)) >> 1][((dx_) + (j_)) & 1]
#pragma unroll
    for (int j = 0; j < COLS; ++j)
#pragma unroll
        for (int i = 0; i < NP; ++i) accp[j][i] = f32x2{0.0f, 0.0f};

    // One channel's operands: the thread's COLS current-frame values and its WN-wide slice of the previous frame's row
    struct Chan {
        f32x2 win[WN / 2];
        float xv[COLS];
    };
    auto compute = [&](int buf_, int nch) __attribute__((always_inline)) {
        if (active && !(abl & 1)) {
        const float *ys = smem + (long)buf_ * buf_floats + (ry + dyi) * CW + COLS * g + dx_lo;
        const float *xs = smem + (long)buf_ * buf_floats + CC * yplane + ry * LF_SX + COLS * g;
        auto fetch = [&](Chan &W, int c) __attribute__((always_inline)) {
            const float *yrow = ys + c * yplane;
            const float *xrow = xs + c * xplane;
            if constexpr (COLS == 2) {
                const f32x2 t = *(const f32x2 *)xrow;
                W.xv[0] = t[0];
                W.xv[1] = t[1];
#pragma unroll
                for (int i = 0; i < WN / 2; ++i) W.win[i] = *(const f32x2 *)(yrow + 2 * i);
            } else {
                const f32x4 t = *(const f32x4 *)xrow;
#pragma unroll
                for (int j = 0; j < 4; ++j) W.xv[j] = t[j];
#pragma unroll
                for (int i = 0; i < WN / 4; ++i) {
                    const f32x4 u = *(const f32x4 *)(yrow + 4 * i);
                    W.win[2 * i] = f32x2{u[0], u[1]};
                    W.win[2 * i + 1] = f32x2{u[2], u[3]};
                }
            }
        };
        auto fma_chan = [&](const Chan &U) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < COLS; ++j) {
                const f32x2 xx = {U.xv[j], U.xv[j]};
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int kp = (j >> 1) + i;  // window pair (2 kp, 2 kp + 1)
                    if (2 * kp <= j + PA - 1) {
                        const f32x2 dd = xx - U.win[kp];
                        accp[j][i] = __builtin_elementwise_fma(dd, dd, accp[j][i]);
                    }
                }
            }
        };
        if constexpr (COLS == 2 || true) {
            // the channel loop rotated by hand: channel c+1's LDS reads are issued before channel c's arithmetic
            Chan W0, W1;  // loop-carried: the loop stays rolled, or the compiler hoists every channel's reads and spills
            fetch(W0, 0);
            int c = 0;
#pragma unroll 1
            for (; c + 2 < nch; c += 2) {  // branch-free body (a conditional fetch costs a register shuffle per trip)
                fetch(W1, c + 1);
                fma_chan(W0);
                fetch(W0, c + 2);
                fma_chan(W1);
            }
            if (c + 1 < nch) {
                fetch(W1, c + 1);
                fma_chan(W0);
                fma_chan(W1);
            } else {
                fma_chan(W0);
            }
        } else {  // wide windows: the arithmetic of one channel covers the reads of the next in an unrolled pair
#pragma unroll 2
            for (int c = 0; c < nch; ++c) {
                Chan W;
                fetch(W, c);
                fma_chan(W);
            }
        }
    }
    };
    if constexpr (MODE != LF_VOL_IN) {
        stage_dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // LF_VOL_IN: order of the memory traffic.  (1) the label loads -- issued from asm so that the compiler's own counted waits (which
    // know nothing of the LDS-DMA pieces behind them) do not drain the volume stream when the labels are used; their addresses come
    // from the host's tile table, i.e. from the kernel argument: no dependent memory round trip; (2) the volume image of this (tile,
    // window-row group), all pieces; then counted waits: the labels (oldest) -> label bytes, change masks, tables, minima; the first
    // VR1 window rows -> their items; the rest -> the other items.  The launch is one wave of workgroups: without this every
    // workgroup waited 7 us for the whole 25.8 MB and then all of them computed (timeline in docs/history/r06_experiments.md).
    constexpr int VR1 = (2 * NDR) / 5 > 0 ? (2 * NDR) / 5 : 1;  // window rows whose items start first
    constexpr int IMG_PIECES = (int)(lf_vpad_bytes(D, NDR) / 1024);  // (whole 1 KiB pieces: the tail lands in the V region's padding)
    constexpr int IMG_P1 = NDR > VR1 ? (VR1 * SY * LF_SX * lf_vs(D) * 4 + 1023) / 1024 : IMG_PIECES;
    // this wave's pieces among the first x pieces of the image (dealt round-robin)
    auto my_pieces = [&](int x) __attribute__((always_inline)) { return x > wave ? (x - wave + NWV - 1) / NWV : 0; };
    LF_T(1)
    // full-resolution pixels of this tile: rows with i0(y) in [a, a+TY), columns with j0(x) in [b0, b0+TX) -- the
    // pooling pass left the ranges in `tab`
    int ya = 0, yb = 0, xa = 0, xb = 0;
    bool host_tab = false;
    if constexpr (MODE == LF_VOL_IN) host_tab = batch.n > 0;
    if constexpr (MODE == LF_VOL_IN) {
        if (host_tab) {
            ya = batch.v[tiy], yb = batch.v[tiy + 1];
            xa = batch.v[nty + 1 + tix], xb = batch.v[nty + 2 + tix];
        }
    }
    if (!host_tab) {
        ya = tab[tiy], yb = tab[tiy + 1];
        xa = tab[nty + 1 + tix], xb = tab[nty + 2 + tix];
    }
    const int ny = yb - ya, nx = xb - xa;
    if (ny > 2 * TY + 4 || nx > 2 * TX + 4) __builtin_trap();  // cannot happen (ratio (hp-1)/(h-1) < 1/2): fail loudly, never overrun L
    // the previous frame's labels around the tile: rows ya + 2(dy0 - D) .., columns xa - 2D ..; outside the image = 0
    // (zero padding, IntVOS.py:400).  Loaded NOW (behind the first two stages' loads, under the first stage's arithmetic), used after phase 1: they come
    // from HBM (nobody has touched them this frame) and would otherwise cost a full miss latency between the phases
    constexpr int KL = (lf_lab_rows_of(D, NDR) * lf_lab_cols(D) + NT - 1) / NT;
    const int lrows = ny + 2 * (NDR - 1), lcols = nx + 4 * D;
    const int ly0 = ya + 2 * (dy0 - D), lx0 = xa - 2 * D;
    int labr[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        const int e = tid + NT * k;
        const int r = e / lcols, c = e - r * lcols;
        const int yy = ly0 + r, xx = lx0 + c;
        const int yc = yy < 0 ? 0 : (yy < h ? yy : h - 1), xc = xx < 0 ? 0 : (xx < w ? xx : w - 1);
        int idx = yc * w + xc;
        asm volatile("" : "+v"(idx));  // the compiler must not turn the clamp back into a branch around the load:
        int v = 0;                     // unconditional loads issue back to back, a branchy one waits vmcnt(0) each time
        if constexpr (MODE == LF_FUSED) v = labels[idx];
        if constexpr (MODE == LF_VOL_IN) {
            const unsigned boff = 4u * (unsigned)idx;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(boff), "s"(labels) : "memory");
            labr[k] = v;  // (masked below, after the wait: the asm load's result must not be touched before it)
        } else {
            labr[k] = (yy == yc && xx == xc && !(abl & 8)) ? v : 0;
        }
    }
    if constexpr (MODE == LF_VOL_IN) {
        for (int pc = wave; pc < IMG_PIECES; pc += NWV) lds_dma16(vimg + (pc * 64 + lane) * 4, smem_base + (unsigned)pc * 1024u);
        lf_wait_vmcnt(my_pieces(IMG_PIECES));  // the label loads are older than this wave's pieces: they have landed
    }
    if constexpr (MODE == LF_VOL_IN) {
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            asm volatile("" : "+v"(labr[k]));  // (uses stay behind the wait)
            const int e = tid + NT * k;
            const int r = e / lcols, c = e - r * lcols;
            const int yy = ly0 + r, xx = lx0 + c;
            const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
            labr[k] = (inside && !(abl & 8)) ? labr[k] : 0;
        }
    }

    // stage s is in LDS buffer s & 1.  The DMA of stage s + 1 into the other buffer (free: its last readers passed the
    // barrier) is issued first and runs under the arithmetic of stage s; the wave waits for its own pieces (vmcnt --
    // the DMA is hidden from the compiler's counters, this is the only wait on it) and the barrier publishes them.
    if constexpr (MODE != LF_VOL_IN) {
        for (int c0 = 0, st_i = 0; c0 < ((abl & 16) ? 0 : C); c0 += CC, ++st_i) {
            if (c0 + CC < C) stage_dma(c0 + CC, (st_i & 1) ^ 1);
            compute(st_i & 1, (C - c0) < CC ? (C - c0) : CC);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    LF_T(2)
    constexpr int VS = lf_vs(D), NPS = lf_npix(D);
    float *V = smem;                                                 // [NDR][SY * 16][VS] (+ the tail of the last LDS-DMA piece)
    unsigned char *L = (unsigned char *)smem + lf_vpad_bytes(D, NDR);  // [lab_rows][lab_cols]; a byte >= the pass's ids = "no id"
    if (MODE != LF_VOL_IN && active) {
        float *vp0 = V + ((dyi * SY + ry) * LF_SX + COLS * g) * VS + dx_lo;
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
#pragma unroll
            for (int d4 = 0; d4 < PA; d4 += 4) {
                if (LF_PH % 4 == 0 && d4 + 3 < PA && d4 + 3 < ndx) {
                    f32x4 t;
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = (abl & 4) ? LF_ACC(j, d4 + i) : manet_normalize_dist_local(LF_ACC(j, d4 + i));
                    *(f32x4 *)(vp0 + j * VS + d4) = t;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (d4 + i < PA && d4 + i < ndx) vp0[j * VS + d4 + i] = manet_normalize_dist_local(LF_ACC(j, d4 + i));
                }
            }
        }
    }
    LF_T(3)
    if constexpr (MODE == LF_VOL_OUT) {
        // the image as phase 2 reads it, to memory in one linear pass (entries no thread wrote -- the cells' padding past the
        // window width, window rows past 2d+1 in the last group -- travel along and are never used)
        __syncthreads();
        constexpr int IMG4 = ND * SY * LF_SX * VS / 4;
        for (int i = tid; i < IMG4; i += NT) ((f32x4 *)vimg)[i] = ((const f32x4 *)V)[i];
        return;
    }
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        const int e = tid + NT * k;
        // one pass over the ids (n_ids <= LF_NIP): store the M2 row directly -- the id, or n_ids for "no id" (the row behind the ids')
        const int lmax = n_ids <= LF_NIP ? n_ids : 255;
        if (e < lrows * lcols) L[e] = (labr[k] >= 0 && labr[k] < (n_ids <= LF_NIP ? n_ids : MANET_MAX_IDS)) ? (unsigned char)labr[k] : (unsigned char)lmax;
    }
    LF_T(7)
    // per-(id, pixel) minima [LF_NIP + 1][NPS] (row LF_NIP collects the candidates whose label is not an id of this
    // pass), then the separable bilinear tables: tap offsets into V and the two weights, per pixel row / column
    unsigned *M2 = (unsigned *)(L + (((size_t)lf_lab_rows_of(D, NDR) * lf_lab_cols(D) + 15) & ~(size_t)15));
    const int m2_rows = n_ids <= LF_NIP ? n_ids : LF_NIP;  // + the "no id" row (the launcher sized the LDS for it)
    struct Tap {
        int o0, o1;
        float l0, l1;
    };
    Tap *RT = (Tap *)(M2 + (m2_rows + 1) * NPS), *CT = RT + (2 * TY + 4);
    // r6: label-change masks.  CM[2 r + parity] bit k = (L[r][2 k + parity] != L[r][2 k + 2 + parity]): a (pixel, window row)
    // item's 2d+1 labels are the entries pxx, pxx + 2, .. of one row -- all the SAME iff bits pxx / 2 .. pxx / 2 + 2d - 1 of
    // its parity's mask are zero.  Masks are piecewise constant (objects are blobs): most items then take the minimum of
    // their window row in registers and touch ONE (id, pixel) slot instead of reading 2d+1 label bytes and issuing 2d+1 LDS
    // atomics (the per-pixel phase was LDS-bound on exactly those: 25 + 25 of 78 LDS instructions per item at d = 12).
    unsigned long long *CM = (unsigned long long *)(CT + (2 * TX + 4));
    if constexpr (P > 1) {
        __syncthreads();  // L is complete
        for (int rp = wave; rp < 2 * lrows; rp += NWV) {
            const int r_ = rp >> 1, c0 = 2 * lane + (rp & 1);
            const bool ok = c0 + 2 < lcols;
            const unsigned char *lr = L + r_ * lcols + (ok ? c0 : 0);
            const bool diff = ok && lr[0] != lr[2];
            const unsigned long long m_ = __builtin_amdgcn_ballot_w64(diff);
            if (lane == 0) CM[rp] = m_;
        }
    }
    if (tid < ny) {
        const Bilin cy = bilin_coeff(ya + tid, hp, h);
        RT[tid] = Tap{(cy.i0 - a) * LF_SX * VS, (cy.i1 - a) * LF_SX * VS, cy.l0, cy.l1};
    } else if (tid >= 64 && tid < 64 + nx) {
        const Bilin cx = bilin_coeff(xa + tid - 64, wp, w);
        CT[tid - 64] = Tap{(cx.i0 - b0) * VS, (cx.i1 - b0) * VS, cx.l0, cx.l1};
    }
    // work item = (pixel, window row): every lane busy whatever the tile's pixel count.  The item walks its window
    // row four columns at a time (four b128 taps), and every candidate goes to its (id, pixel) slot by an LDS
    // atomic min on the float bits -- all candidates lie in [0, 1], start value 1.0 = "no match" (IntVOS.py:429-432:
    // where(label == id, dist, 1) then min) -- instead of a compare/select/min per id in registers.
    const int npix = ny * nx;
    const float inv_npix = 1.0f / (float)npix, inv_nx = 1.0f / (float)nx;  // exact quotients below: see DESIGN 3.4
    for (int o0 = 0; o0 < n_ids; o0 += LF_NIP) {
        const int nk = (n_ids - o0) < LF_NIP ? (n_ids - o0) : LF_NIP;
        for (int e = tid; e < nk * NPS; e += NT) M2[e] = 0x3f800000u;  // the rows that are read back
        // (LF_VOL_IN: this wave's pieces of the first VR1 window rows have landed; the barrier publishes everyone's)
        if constexpr (MODE == LF_VOL_IN) lf_wait_vmcnt(my_pieces(IMG_PIECES) - my_pieces(IMG_P1));
        __syncthreads();
        LF_T(4)
        // n_ids <= LF_NIP (one pass): the label byte, clamped to LF_NIP, IS the row of M2
        auto items = [&](auto single_pass, int item_lo, int item_hi) __attribute__((always_inline)) {
            constexpr bool SINGLE = decltype(single_pass)::value;
            for (int item = item_lo + tid; item < ((abl & 2) ? 0 : item_hi); item += NT) {
                const int by = (int)(((float)item + 0.5f) * inv_npix), pix = item - by * npix;
                const int py = (int)(((float)pix + 0.5f) * inv_nx), pxx = pix - py * nx;
                const Tap r = RT[py], c = CT[pxx];
                const float *vb = V + by * (SY * LF_SX * VS);
                const float *p00 = vb + r.o0 + c.o0, *p01 = vb + r.o0 + c.o1, *p10 = vb + r.o1 + c.o0, *p11 = vb + r.o1 + c.o1;
                const unsigned char *lrow = L + (py + 2 * by) * lcols + pxx;
                unsigned *mrow = M2 + pix;
                const f32x2 cl0 = {c.l0, c.l0}, cl1 = {c.l1, c.l1}, rl0 = {r.l0, r.l0}, rl1 = {r.l1, r.l1};
                if constexpr (P > 1) {
                    // every item of this wave sees ONE label along its window row (wave-uniform test: no divergence): the
                    // row's minimum in registers, one atomic.  Same candidates, same minimum: the same bits.
                    const unsigned long long cm = CM[2 * (py + 2 * by) + (pxx & 1)];
                    const bool mixed = ((cm >> (pxx >> 1)) & ((1ull << (P - 1)) - 1ull)) != 0ull;
                    if (__builtin_amdgcn_ballot_w64(mixed) == 0ull && !(abl & 32)) {
                        float m_ = INFINITY;
#pragma unroll
                        for (int q = 0; q < (P + 3) / 4; ++q) {
                            const f32x4 t00 = *(const f32x4 *)(p00 + 4 * q), t01 = *(const f32x4 *)(p01 + 4 * q);
                            const f32x4 t10 = *(const f32x4 *)(p10 + 4 * q), t11 = *(const f32x4 *)(p11 + 4 * q);
#pragma unroll
                            for (int i = 0; i < 4; i += 2) {
                                if (4 * q + i >= P) continue;
                                const f32x2 a00 = {t00[i], t00[i + 1]}, a01 = {t01[i], t01[i + 1]};
                                const f32x2 a10 = {t10[i], t10[i + 1]}, a11 = {t11[i], t11[i + 1]};
                                const f32x2 v2 = rl0 * (cl0 * a00 + cl1 * a01) + rl1 * (cl0 * a10 + cl1 * a11);
                                m_ = (4 * q + i + 1 < P) ? fminf(fminf(m_, v2[0]), v2[1]) : fminf(m_, v2[0]);
                            }
                        }
                        unsigned idx = (unsigned)lrow[0];
                        if (!SINGLE) {
                            idx -= (unsigned)o0;
                            idx = idx < (unsigned)LF_NIP ? idx : (unsigned)LF_NIP;
                        }
                        atomicMin(mrow + idx * NPS, __float_as_uint(m_));
                        continue;
                    }
                }
#pragma unroll
                for (int q = 0; q < (P + 3) / 4; ++q) {
                    const f32x4 t00 = *(const f32x4 *)(p00 + 4 * q), t01 = *(const f32x4 *)(p01 + 4 * q);
                    const f32x4 t10 = *(const f32x4 *)(p10 + 4 * q), t11 = *(const f32x4 *)(p11 + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        if (4 * q + i >= P) continue;
                        // two window columns per packed op; per element the expression of bilin_sample / the oracle:
                        // l0y (l0x v00 + l1x v01) + l1y (l0x v10 + l1x v11)
                        const f32x2 a00 = {t00[i], t00[i + 1]}, a01 = {t01[i], t01[i + 1]};
                        const f32x2 a10 = {t10[i], t10[i + 1]}, a11 = {t11[i], t11[i + 1]};
                        const f32x2 v2 = rl0 * (cl0 * a00 + cl1 * a01) + rl1 * (cl0 * a10 + cl1 * a11);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int bx = 4 * q + i + e;
                            if (bx < P) {
                                unsigned idx = (unsigned)lrow[2 * bx];
                                if (!SINGLE) {
                                    idx -= (unsigned)o0;
                                    idx = idx < (unsigned)LF_NIP ? idx : (unsigned)LF_NIP;
                                }
                                atomicMin(mrow + idx * NPS, __float_as_uint(v2[e]));
                            }
                        }
                    }
                }
            }
        };
        // (LF_VOL_IN: the first batch's rows, then -- once the second batch has landed -- the others)
        const int item_mid = (MODE == LF_VOL_IN && NDR > VR1) ? npix * (nd_here < VR1 ? nd_here : VR1) : npix * nd_here;
        if (n_ids <= LF_NIP) items(std::true_type{}, 0, item_mid);
        else items(std::false_type{}, 0, item_mid);
        if constexpr (MODE == LF_VOL_IN && NDR > VR1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (n_ids <= LF_NIP) items(std::true_type{}, item_mid, npix * nd_here);
            else items(std::false_type{}, item_mid, npix * nd_here);
        }
        __syncthreads();
        LF_T(5)
        for (int e = tid; e < npix * nk; e += NT) {
            const int pix = e / nk, k = e - pix * nk;
            const int py = pix / nx, pxx = pix - py * nx;
            float *o = out + ((long)(ya + py) * w + (xa + pxx)) * n_ids + o0 + k;
            if (NDG * NSUB == 1) *o = __uint_as_float(M2[k * NPS + pix]);
            else atomicMin((unsigned *)o, M2[k * NPS + pix]);  // several workgroups per tile: `out` was pre-set to 1.0
        }
        __syncthreads();
        LF_T(6)
    }
}

template <int D, int MODE = LF_FUSED>
static void launch_fused_d(hipStream_t st, const float *ap, const float *bp, const PoolPad &G, const int *labels, int h,
                           int w, int C, int n_ids, float *out, const int *tab, float *vol = nullptr,
                           const LfBatch *batch = nullptr, int n_pairs = 1)
{
    constexpr int TY = lf_sy(D) - 1, TX = LF_SX - 1;
    // i0 runs over 0..hp-1 (the last value only for the last row); tiles cover all of them
    const int ntx = (G.wp + TX - 1) / TX, nty = (G.hp + TY - 1) / TY;
    const int rw = (ntx + 3) / 4, rh = (nty + 1) / 2;  // tiles per XCD region (4 x 2 regions)
    dim3 grid((unsigned)(8 * rw * rh * lf_ndg(D) * (MODE == LF_VOL_IN ? lf_nsub(D) : 1)), (unsigned)(MODE == LF_VOL_OUT ? n_pairs : 1));
    const size_t lds = MODE == LF_VOL_IN ? lf_lds_vol_bytes(D, n_ids) : lf_lds_bytes(D);
    (void)hipFuncSetAttribute((const void *)local_fused_kernel<D, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(MODE == LF_VOL_IN ? lf_lds_vol_bytes(D, LF_NIP) : lds));
    if constexpr (MODE == LF_VOL_OUT)
        hipLaunchKernelGGL((local_fused_kernel<D, MODE>), grid, dim3(lf_nt(D)), lds, st, ap, bp, G.WS, G.plane, labels, h, w, C, n_ids,
                           out, tab, manet_tune_get(MANET_TUNE_ABLATION, 0), ntx, nty, rw, rh, vol, *batch);
    else if constexpr (MODE == LF_VOL_IN) {
        LfTab T;
        T.n = 0;
        if (nty + 1 + ntx + 1 <= LF_TAB_MAX) {  // the device's expression on the host: the same integers (lf_pool_pad_kernel / frame prepare)
            T.n = nty + 1 + ntx + 1;
            for (int i = 0; i <= nty; ++i) T.v[i] = bilin_first(i * TY, G.hp, h);
            for (int i = 0; i <= ntx; ++i) T.v[nty + 1 + i] = bilin_first(i * TX, G.wp, w);
        }
        hipLaunchKernelGGL((local_fused_kernel<D, MODE>), grid, dim3(lf_ntv(D)), lds, st, ap, bp, G.WS, G.plane, labels, h, w, C, n_ids,
                           out, tab, manet_tune_get(MANET_TUNE_ABLATION, 0), ntx, nty, rw, rh, vol, T);
    } else
        hipLaunchKernelGGL((local_fused_kernel<D, MODE>), grid, dim3(lf_nt(D)), lds, st, ap, bp, G.WS,
                           G.plane, labels, h, w, C, n_ids, out, tab, manet_tune_get(MANET_TUNE_ABLATION, 0), ntx, nty, rw, rh, vol, 0);
}
// workgroups (= volume images) of one frame pair
static long lf_images(int h, int w, int d)
{
    const int hp = h / 2, wp = w / 2, TY = lf_sy(d) - 1, TX = LF_SX - 1;
    return (long)((wp + TX - 1) / TX) * ((hp + TY - 1) / TY) * lf_ndg(d);
}

__global__ void fill_f32_kernel(float *__restrict__ p, float v, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// `pooled`: room for the two padded pooled frames (2 * C * lf_pool_pad().plane floats)
static void launch_fused(int d, hipStream_t st, const void *cur, long c_sy, long c_sx, long c_sc, const void *prev,
                         long p_sy, long p_sx, long p_sc, int emb_dtype, const int *labels, int h, int w, int C, int n_ids,
                         float *out, float *pooled, int *tab)
{
    const PoolPad G = lf_pool_pad(h, w, d);
    float *ap = pooled, *bp = pooled + G.plane * C;
    const int TY = lf_sy(d) - 1, TX = LF_SX - 1;
    const int nty = (G.hp + TY - 1) / TY, ntx = (G.wp + TX - 1) / TX;  // the fused kernel's grid
    {
        long n = G.plane * C;
        // partial minima of several workgroups per tile meet by atomicMin: `out` starts from the "no match" value,
        // set by the pooling pass when it has enough threads for it (always, for real shapes), else by a fill
        const long n_out = (long)h * w * n_ids;
        float *init = nullptr;
        if (lf_ndg(d) > 1) {
            if (n_out <= n) init = out;
            else {
                unsigned blocks = (unsigned)((n_out + 255) / 256);
                if (blocks > 1024) blocks = 1024;
                hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, st, out, 1.0f, n_out);
            }
        }
        if (emb_dtype == MANET_EMB_BF16)
            hipLaunchKernelGGL(lf_pool_pad_kernel<unsigned short>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               (const unsigned short *)cur, c_sy, c_sx, c_sc, (const unsigned short *)prev, p_sy, p_sx, p_sc, C,
                               G.hp, G.wp, d, G.HPAD, G.WS, ap, bp, init, n_out, tab, TY, TX, nty, ntx, h, w);
        else
            hipLaunchKernelGGL(lf_pool_pad_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               (const float *)cur, c_sy, c_sx, c_sc, (const float *)prev, p_sy, p_sx, p_sc, C, G.hp, G.wp, d,
                               G.HPAD, G.WS, ap, bp, init, n_out, tab, TY, TX, nty, ntx, h, w);
    }
#define MANET_LF_CASE(D_) case D_: launch_fused_d<D_>(st, ap, bp, G, labels, h, w, C, n_ids, out, tab); break;
    switch (d) {
        MANET_LF_CASE(0) MANET_LF_CASE(1) MANET_LF_CASE(2) MANET_LF_CASE(3) MANET_LF_CASE(4) MANET_LF_CASE(5)
        MANET_LF_CASE(6) MANET_LF_CASE(7) MANET_LF_CASE(8) MANET_LF_CASE(9) MANET_LF_CASE(10) MANET_LF_CASE(11)
        MANET_LF_CASE(12)
    default: break;
    }
#undef MANET_LF_CASE
}

struct LocalLayout {
    int hp, wp, PP;
    size_t off_ap, off_bp, off_vol, total;
};

LocalLayout local_layout(int h, int w, int C, int d, int downsample)
{
    LocalLayout L;
    L.PP = (2 * d + 1) * (2 * d + 1);
    L.hp = downsample ? h / 2 : h;
    L.wp = downsample ? w / 2 : w;
    size_t plane = (size_t)L.hp * L.wp;
    // pooled frames: the padded planes of the fused path (the unpadded ones of the other paths fit inside)
    size_t pooled = downsample ? (size_t)lf_pool_pad(h, w, d).plane * C * sizeof(float) : 0;
    L.off_ap = 0;
    L.off_bp = manet_align_up(pooled, 256);
    L.off_vol = L.off_bp + manet_align_up(pooled, 256);
    // the volume region doubles as the fused path's tile table (first pixel row / column of every tile)
    size_t vol = plane * L.PP * sizeof(float), tab = (size_t)(L.hp + L.wp + 4) * sizeof(int);
    L.total = manet_align_up(L.off_vol + (vol > tab ? vol : tab), 256);
    return L;
}

int check_local(int h, int w, int C, int d, int downsample)
{
    if (h <= 0 || w <= 0 || C <= 0) return manet_set_error(MANET_E_INVALID, "h=%d w=%d C=%d", h, w, C);
    if (d < 0 || d > MANET_MAX_LOCAL_DISTANCE)
        return manet_set_error(MANET_E_INVALID, "max_distance=%d (supported 0..%d)", d, MANET_MAX_LOCAL_DISTANCE);
    if (downsample && (h < 2 || w < 2)) return manet_set_error(MANET_E_INVALID, "downsample needs h,w >= 2");
    return MANET_OK;
}

// enqueue pooling (if any) + the distance volume into the workspace; returns the volume pointer
float *enqueue_volume(const float *cur, int64_t c_sy, int64_t c_sx, int64_t c_sc, const float *prev,
                      int64_t p_sy, int64_t p_sx, int64_t p_sc, int h, int w, int C, int d, int downsample,
                      char *ws, const LocalLayout &L, hipStream_t st)
{
    float *vol = (float *)(ws + L.off_vol);
    const int P = 2 * d + 1;
    if (downsample) {
        float *ap = (float *)(ws + L.off_ap), *bp = (float *)(ws + L.off_bp);
        long n = (long)C * L.hp * L.wp;
        hipLaunchKernelGGL(pool2x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cur, (long)c_sy,
                           (long)c_sx, (long)c_sc, prev, (long)p_sy, (long)p_sx, (long)p_sc, C, L.hp, L.wp, ap, bp);
        long plane = (long)L.hp * L.wp;
        launch_dist(d, st, (const float *)ap, (long)L.wp, 1L, plane, (const float *)bp, (long)L.wp, 1L, plane, L.hp,
                    L.wp, C, 1, vol);
    } else {
        launch_dist(d, st, cur, (long)c_sy, (long)c_sx, (long)c_sc, prev, (long)p_sy, (long)p_sx, (long)p_sc, h, w, C,
                    0, vol);
    }
    (void)P;
    return vol;
}

// ---------------------------------------------------------------------------------------------
// Training path: the local match with the winning window offset recorded, and its backward.
// The reference differentiates IntVOS.py:266-296 + :398-432 by autograd: avg_pool2d -> sum_c (x - y_off)^2 ->
// (sigmoid - 0.5) * 2 -> bilinear(align_corners) -> where(label mask, ., 1.0) -> min over the window.
// The gradient of the min flows to ONE window offset per (pixel, object) (none if the constant 1.0 wins).

// local_min_kernel with arg-min: arg[y][x][o] = window offset l = dy*P + dx whose masked value attains the
// minimum (strict <: the first offset wins a tie), -1 if no matching offset beats the constant 1.0
__global__ void local_min_arg_kernel(const float *__restrict__ dvol, const int *__restrict__ labels, int h, int w,
                                     int hp, int wp, int d, int n_ids, float *__restrict__ out, int *__restrict__ arg)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)h * w * n_ids) return;
    const int o = (int)(i % n_ids);
    const long pix = i / n_ids;
    const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
    const int P = 2 * d + 1;
    const Bilin cy = bilin_coeff(y, hp, h), cx = bilin_coeff(x, wp, w);
    const long plane = (long)hp * wp;
    float m = INFINITY;
    int am = -1;
    for (int by = 0; by < P; ++by) {
        const int yy = y + 2 * (by - d);
        const bool yin = (yy >= 0 && yy < h);
        for (int bx = 0; bx < P; ++bx) {
            const int xx = x + 2 * (bx - d);
            const int lab = (yin && xx >= 0 && xx < w) ? labels[(long)yy * w + xx] : 0;
            const int l = by * P + bx;
            const bool hit = (lab == o);
            const float v = hit ? bilin_sample(dvol + (long)l * plane, wp, cy, cx) : 1.0f;
            if (v < m) {
                m = v;
                am = hit ? l : -1;
            }
        }
    }
    out[i] = m;
    arg[i] = am;
}

// The same for MODEL_LOCAL_DOWNSAMPLE = False (IntVOS.py:299-313 + :398-432, r5): the volume is the RAW full-resolution
// distances [h][w][P*P] (no sigmoid, no bilinear), the labels are still gathered at stride 2 (the reference's unfold, :404)
// and the constant that competes with them is still 1.0 (:429) -- the reference's quirk, kept.
__global__ void local_min_arg_full_kernel(const float *__restrict__ vol, const int *__restrict__ labels, int h, int w, int d,
                                          int n_ids, float *__restrict__ out, int *__restrict__ arg)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)h * w * n_ids) return;
    const int o = (int)(i % n_ids);
    const long pix = i / n_ids;
    const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
    const int P = 2 * d + 1;
    const float *v0 = vol + pix * P * P;
    float m = INFINITY;
    int am = -1;
    for (int by = 0; by < P; ++by) {
        const int yy = y + 2 * (by - d);
        const bool yin = (yy >= 0 && yy < h);
        for (int bx = 0; bx < P; ++bx) {
            const int xx = x + 2 * (bx - d);
            const int lab = (yin && xx >= 0 && xx < w) ? labels[(long)yy * w + xx] : 0;
            const int l = by * P + bx;
            const bool hit = (lab == o);
            const float v = hit ? v0[l] : 1.0f;
            if (v < m) {
                m = v;
                am = hit ? l : -1;
            }
        }
    }
    out[i] = m;
    arg[i] = am;
}

// dV[l][pixel] += g for the winning offset of every (pixel, object) (several objects of a pixel may pick the same offset)
__global__ void local_bwd_scatter_full_kernel(const int *__restrict__ arg, const float *__restrict__ gout, long npix, int n_ids,
                                              float *__restrict__ dv)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * n_ids) return;
    const int l = arg[i];
    const float g = gout[i];
    if (l < 0 || g == 0.0f) return;
    atomicAdd(dv + (long)l * npix + i / n_ids, g);
}

// dVn[l][i][j] += g * bilinear weight, for the winning offset of every (pixel, object)  (backward of
// F.interpolate(..., 'bilinear', align_corners=True) restricted to the offsets the min selected)
__global__ void local_bwd_scatter_kernel(const int *__restrict__ arg, const float *__restrict__ gout, int h, int w,
                                         int hp, int wp, int n_ids, float *__restrict__ dvn)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)h * w * n_ids) return;
    const int l = arg[i];
    const float g = gout[i];
    if (l < 0 || g == 0.0f) return;
    const long pix = i / n_ids;
    const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
    const Bilin cy = bilin_coeff(y, hp, h), cx = bilin_coeff(x, wp, w);
    float *pl = dvn + (long)l * hp * wp;
    atomicAdd(pl + cy.i0 * wp + cx.i0, g * cy.l0 * cx.l0);
    atomicAdd(pl + cy.i0 * wp + cx.i1, g * cy.l0 * cx.l1);
    atomicAdd(pl + cy.i1 * wp + cx.i0, g * cy.l1 * cx.l0);
    atomicAdd(pl + cy.i1 * wp + cx.i1, g * cy.l1 * cx.l1);
}

// dV = dVn * d/dV[(sigmoid(V) - 0.5) * 2] = dVn * (1 - Vn^2) / 2   (in place; Vn = 1 where V = inf: gradient 0)
__global__ void local_bwd_dnorm_kernel(const float *__restrict__ vn, float *__restrict__ dvn, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = vn[i];
    dvn[i] = dvn[i] * (1.0f - v * v) * 0.5f;
}

// gradients w.r.t. the POOLED frames: V[l][p] = sum_c (x[c][p] - y[c][p + l])^2  =>
//   gx[c][p] =  sum_l 2 (x[c][p] - y[c][p + l]) dV[l][p]            (p + l inside the image)
//   gy[c][q] = -sum_l 2 (x[c][q - l] - y[c][q]) dV[l][q - l]        (q - l inside the image)
__global__ void local_bwd_dist_kernel(const float *__restrict__ xp, const float *__restrict__ yp,
                                      const float *__restrict__ dv, int C, int hp, int wp, int d,
                                      float *__restrict__ gxp, float *__restrict__ gyp)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long plane = (long)hp * wp;
    if (i >= plane * C) return;
    const int c = (int)(i / plane);
    const int rem = (int)(i - (long)c * plane);
    const int py = rem / wp, px = rem - py * wp;
    const int P = 2 * d + 1;
    const float *xc = xp + (long)c * plane, *yc = yp + (long)c * plane;
    const float xv = xc[rem], yv = yc[rem];
    float gx = 0.0f, gy = 0.0f;
    for (int dy = 0; dy < P; ++dy) {
        for (int dx = 0; dx < P; ++dx) {
            const long l = (long)(dy * P + dx) * plane;
            const int qy = py + dy - d, qx = px + dx - d;  // neighbour this pixel looked at
            if (qy >= 0 && qy < hp && qx >= 0 && qx < wp) gx += 2.0f * (xv - yc[qy * wp + qx]) * dv[l + rem];
            const int sy = py - (dy - d), sx = px - (dx - d);  // pixel that looked at this one
            if (sy >= 0 && sy < hp && sx >= 0 && sx < wp) gy -= 2.0f * (xc[sy * wp + sx] - yv) * dv[l + sy * wp + sx];
        }
    }
    gxp[i] = gx;
    gyp[i] = gy;
}

// backward of the 2x2 average pooling into the caller's (strided) gradient tensors
__global__ void local_bwd_unpool_kernel(const float *__restrict__ gxp, const float *__restrict__ gyp, int C, int h, int w,
                                        int hp, int wp, float *__restrict__ gcur, long c_sy, long c_sx, long c_sc,
                                        float *__restrict__ gprev, long p_sy, long p_sx, long p_sc)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long plane = (long)h * w;
    if (i >= plane * C) return;
    const int c = (int)(i / plane);
    const int rem = (int)(i - (long)c * plane);
    const int y = rem / w, x = rem - y * w;
    float a = 0.0f, b = 0.0f;
    if (y < 2 * hp && x < 2 * wp) {
        const long j = (long)c * hp * wp + (long)(y / 2) * wp + (x / 2);
        a = 0.25f * gxp[j];
        b = 0.25f * gyp[j];
    }
    gcur[(long)y * c_sy + (long)x * c_sx + (long)c * c_sc] = a;
    gprev[(long)y * p_sy + (long)x * p_sx + (long)c * p_sc] = b;
}

}  // namespace

extern "C" {

int manet_local_workspace_bytes(int h, int w, int C, int max_distance, int downsample, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_local(h, w, C, max_distance, downsample);
    if (rc) return rc;
    *bytes = local_layout(h, w, C, max_distance, downsample).total;
    return MANET_OK;
}

int manet_local_dist_f32(const float *cur, int64_t c_sy, int64_t c_sx, int64_t c_sc, const float *prev,
                         int64_t p_sy, int64_t p_sx, int64_t p_sc, int h, int w, int C, int max_distance,
                         int downsample, float *out, void *workspace, size_t workspace_bytes,
                         manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, downsample);
    if (rc) return rc;
    if (!cur || !prev || !out) return manet_set_error(MANET_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int d = max_distance;
    if (!downsample) {  // the volume is the result: write it straight into out
        launch_dist(d, st, cur, (long)c_sy, (long)c_sx, (long)c_sc, prev, (long)p_sy, (long)p_sx, (long)p_sc, h, w, C,
                    0, out);
        return manet_check_launch("manet_local_dist_f32");
    }
    LocalLayout L = local_layout(h, w, C, d, downsample);
    if (!workspace || workspace_bytes < L.total)
        return manet_set_error(MANET_E_WORKSPACE, "local workspace %zu < %zu bytes", workspace_bytes, L.total);
    float *vol = enqueue_volume(cur, c_sy, c_sx, c_sc, prev, p_sy, p_sx, p_sc, h, w, C, d, 1, (char *)workspace, L, st);
    long n = (long)h * w * L.PP;
    hipLaunchKernelGGL(local_upsample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const float *)vol, h, w, L.hp, L.wp, L.PP, out);
    return manet_check_launch("manet_local_dist_f32");
}

int manet_local_match_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc, const float *cur,
                          int64_t c_sy, int64_t c_sx, int64_t c_sc, const int32_t *prev_labels, int h, int w,
                          int C, int n_ids, int max_distance, int downsample, float *out, void *workspace,
                          size_t workspace_bytes, manet_stream_t stream)
{
    return manet_local_match_ex(prev, p_sy, p_sx, p_sc, cur, c_sy, c_sx, c_sc, MANET_EMB_F32, prev_labels, h, w, C, n_ids,
                                max_distance, downsample, out, workspace, workspace_bytes, stream);
}

int manet_local_match_ex(const void *prev_v, int64_t p_sy, int64_t p_sx, int64_t p_sc, const void *cur_v, int64_t c_sy,
                         int64_t c_sx, int64_t c_sc, int emb_dtype, const int32_t *prev_labels, int h, int w, int C,
                         int n_ids, int max_distance, int downsample, float *out, void *workspace, size_t workspace_bytes,
                         manet_stream_t stream)
{
    const float *prev = (const float *)prev_v, *cur = (const float *)cur_v;  // (typed below)
    int rc = check_local(h, w, C, max_distance, downsample);
    if (rc) return rc;
    if (emb_dtype != MANET_EMB_F32 && emb_dtype != MANET_EMB_BF16)
        return manet_set_error(MANET_E_INVALID, "embedding dtype %d (MANET_EMB_F32 / MANET_EMB_BF16)", emb_dtype);
    if (emb_dtype == MANET_EMB_BF16 && (!downsample || manet_tune_get(MANET_TUNE_LOCAL_UNFUSED, 0)))
        return manet_set_error(MANET_E_INVALID, "bf16 embeddings: the downsample configuration (fused path) only");
    if (n_ids <= 0 || n_ids > MANET_MAX_IDS)
        return manet_set_error(MANET_E_INVALID, "n_ids=%d (supported 1..%d)", n_ids, MANET_MAX_IDS);
    if (!cur || !prev || !prev_labels || !out) return manet_set_error(MANET_E_INVALID, "null pointer");
    LocalLayout L = local_layout(h, w, C, max_distance, downsample);
    if (!workspace || workspace_bytes < L.total)
        return manet_set_error(MANET_E_WORKSPACE, "local workspace %zu < %zu bytes", workspace_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    if (downsample && !manet_tune_get(MANET_TUNE_LOCAL_UNFUSED, 0)) {  // the live configuration: pooling pass + fused kernel
        manet_profile_record(st, true, 1);
        launch_fused(max_distance, st, cur_v, (long)c_sy, (long)c_sx, (long)c_sc, prev_v, (long)p_sy, (long)p_sx, (long)p_sc,
                     emb_dtype, prev_labels, h, w, C, n_ids, out, (float *)((char *)workspace + L.off_ap),
                     (int *)((char *)workspace + L.off_vol));
        manet_profile_record(st, false, 1);
        return manet_check_launch("manet_local_match_f32");
    }
    // IntVOS.py:370: local_pairwise_distances2(query_embedding, prev_frame_embedding)
    float *vol = enqueue_volume(cur, c_sy, c_sx, c_sc, prev, p_sy, p_sx, p_sc, h, w, C, max_distance, downsample,
                                (char *)workspace, L, st);
    long n = (long)h * w;
    hipLaunchKernelGGL(local_min_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * LM_WAVES), 0, st,
                       (const float *)vol, downsample ? 1 : 0, prev_labels, h, w, L.hp, L.wp, max_distance, n_ids, out);
    return manet_check_launch("manet_local_match_f32");
}

/* the fused kernel on two prepared frames (manet_frame_prepare): no pooling pass, no full-resolution read */
int manet_local_match_frames(const void *prev_frame_ws, const void *cur_frame_ws, const int32_t *prev_labels, int h, int w,
                             int C, int compute, int n_ids, int max_distance, float *out, int out_is_preset,
                             manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, 1);
    if (rc) return rc;
    if (n_ids <= 0 || n_ids > MANET_MAX_IDS)
        return manet_set_error(MANET_E_INVALID, "n_ids=%d (supported 1..%d)", n_ids, MANET_MAX_IDS);
    if (!prev_frame_ws || !cur_frame_ws || !prev_labels || !out) return manet_set_error(MANET_E_INVALID, "null pointer");
    const ManetFrameLayout F = manet_frame_layout(h, w, C, compute, max_distance);
    hipStream_t st = (hipStream_t)stream;
    PoolPad G;
    G.hp = F.hp; G.wp = F.wp; G.HPAD = F.HPAD; G.WS = F.WS; G.plane = F.PS;
    const float *ap = (const float *)((const char *)cur_frame_ws + F.off_plane);
    const float *bp = (const float *)((const char *)prev_frame_ws + F.off_plane);
    const int *tab = (const int *)((const char *)cur_frame_ws + F.off_tab);
    manet_profile_record(st, true, 1);
    if (lf_ndg(max_distance) > 1 && !out_is_preset) {  // partial minima of several workgroups per tile meet by atomicMin
        const long n_out = (long)h * w * n_ids;
        unsigned blocks = (unsigned)((n_out + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, st, out, 1.0f, n_out);
    }
#define MANET_LF_CASE(D_) case D_: launch_fused_d<D_>(st, ap, bp, G, prev_labels, h, w, C, n_ids, out, tab); break;
    switch (max_distance) {
        MANET_LF_CASE(0) MANET_LF_CASE(1) MANET_LF_CASE(2) MANET_LF_CASE(3) MANET_LF_CASE(4) MANET_LF_CASE(5)
        MANET_LF_CASE(6) MANET_LF_CASE(7) MANET_LF_CASE(8) MANET_LF_CASE(9) MANET_LF_CASE(10) MANET_LF_CASE(11)
        MANET_LF_CASE(12)
    default: break;
    }
#undef MANET_LF_CASE
    manet_profile_record(st, false, 1);
    return manet_check_launch("manet_local_match_frames");
}

/* r6: the label-independent half of the local match, kept per frame pair (see local_fused_kernel's MODE).
 * manet_local_volume_bytes: bytes of one frame pair's normalised window-distance volume, stored as the per-workgroup LDS images
 * of the per-pixel phase (1.5x the bare (2d+1)^2 x h/2 x w/2 floats at d=12: tile aprons, 16-byte cell padding). */
int manet_local_volume_bytes(int h, int w, int max_distance, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_local(h, w, 1, max_distance, 1);
    if (rc) return rc;
    // (+ 1 KiB: the per-pixel kernel fetches a sub-group's rows in whole 1 KiB pieces and may read past the last image's end)
    *bytes = (size_t)lf_images(h, w, max_distance) * lf_img_floats(max_distance) * sizeof(float) + 1024;
    return MANET_OK;
}

/* phase 1 (IntVOS.py:266-296) of n_pairs frame pairs, ceil(n_pairs / 32) launches: pair i = (prev_frame_ws[i], cur_frame_ws[i]),
 * two prepared frames (manet_frame_prepare) -> volumes[i] (manet_local_volume_bytes each, 16-byte aligned).  The three tables are
 * HOST arrays of device pointers. */
int manet_local_volume_frames(const void *const *prev_frame_ws, const void *const *cur_frame_ws, float *const *volumes, int n_pairs,
                              int h, int w, int C, int compute, int max_distance, manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, 1);
    if (rc) return rc;
    if (n_pairs < 0 || (n_pairs > 0 && (!prev_frame_ws || !cur_frame_ws || !volumes)))
        return manet_set_error(MANET_E_INVALID, "bad arguments");
    const ManetFrameLayout F = manet_frame_layout(h, w, C, compute, max_distance);
    hipStream_t st = (hipStream_t)stream;
    PoolPad G;
    G.hp = F.hp; G.wp = F.wp; G.HPAD = F.HPAD; G.WS = F.WS; G.plane = F.PS;
    for (int i0 = 0; i0 < n_pairs; i0 += LF_BATCH) {
        const int n = (n_pairs - i0) < LF_BATCH ? (n_pairs - i0) : LF_BATCH;
        LfBatch B;
        for (int i = 0; i < LF_BATCH; ++i) {
            const int j = i0 + (i < n ? i : 0);
            if (!prev_frame_ws[j] || !cur_frame_ws[j] || !volumes[j] || ((size_t)volumes[j] & 15))
                return manet_set_error(MANET_E_INVALID, "pair %d: null frame / volume pointer, or a volume not 16-byte aligned", j);
            B.cur[i] = (const float *)((const char *)cur_frame_ws[j] + F.off_plane);
            B.prev[i] = (const float *)((const char *)prev_frame_ws[j] + F.off_plane);
            B.vol[i] = volumes[j];
        }
        const int *tab = (const int *)((const char *)cur_frame_ws[i0] + F.off_tab);  // (the same table for every frame of a geometry)
#define MANET_LF_CASE(D_) case D_: launch_fused_d<D_, LF_VOL_OUT>(st, nullptr, nullptr, G, nullptr, h, w, C, 1, nullptr, tab, nullptr, &B, n); break;
        switch (max_distance) {
            MANET_LF_CASE(0) MANET_LF_CASE(1) MANET_LF_CASE(2) MANET_LF_CASE(3) MANET_LF_CASE(4) MANET_LF_CASE(5)
            MANET_LF_CASE(6) MANET_LF_CASE(7) MANET_LF_CASE(8) MANET_LF_CASE(9) MANET_LF_CASE(10) MANET_LF_CASE(11)
            MANET_LF_CASE(12)
        default: break;
        }
#undef MANET_LF_CASE
    }
    return manet_check_launch("manet_local_volume_frames");
}

/* phase 2 (IntVOS.py:398-432) on a stored volume: bilinear taps + stride-2 label gather + masked minimum -> out [h][w][n_ids]; the
 * same bits as manet_local_match_frames on the pair the volume was made from.  cur_frame_ws: the current frame's prepared
 * workspace (its tile table). */
int manet_local_match_volume(const float *volume, const void *cur_frame_ws, const int32_t *prev_labels, int h, int w, int C,
                             int compute, int n_ids, int max_distance, float *out, int out_is_preset, manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, 1);
    if (rc) return rc;
    if (n_ids <= 0 || n_ids > MANET_MAX_IDS)
        return manet_set_error(MANET_E_INVALID, "n_ids=%d (supported 1..%d)", n_ids, MANET_MAX_IDS);
    if (!volume || !cur_frame_ws || !prev_labels || !out || ((size_t)volume & 15))
        return manet_set_error(MANET_E_INVALID, "null pointer (or a volume not 16-byte aligned)");
    const ManetFrameLayout F = manet_frame_layout(h, w, C, compute, max_distance);
    hipStream_t st = (hipStream_t)stream;
    PoolPad G;
    G.hp = F.hp; G.wp = F.wp; G.HPAD = F.HPAD; G.WS = F.WS; G.plane = F.PS;
    const int *tab = (const int *)((const char *)cur_frame_ws + F.off_tab);
    manet_profile_record(st, true, 1);
    if (lf_ndg(max_distance) * lf_nsub(max_distance) > 1 && !out_is_preset) {  // partial minima of several workgroups per tile meet by atomicMin
        const long n_out = (long)h * w * n_ids;
        unsigned blocks = (unsigned)((n_out + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, st, out, 1.0f, n_out);
    }
#define MANET_LF_CASE(D_) case D_: launch_fused_d<D_, LF_VOL_IN>(st, nullptr, nullptr, G, prev_labels, h, w, C, n_ids, out, tab, (float *)volume); break;
    switch (max_distance) {
        MANET_LF_CASE(0) MANET_LF_CASE(1) MANET_LF_CASE(2) MANET_LF_CASE(3) MANET_LF_CASE(4) MANET_LF_CASE(5)
        MANET_LF_CASE(6) MANET_LF_CASE(7) MANET_LF_CASE(8) MANET_LF_CASE(9) MANET_LF_CASE(10) MANET_LF_CASE(11)
        MANET_LF_CASE(12)
    default: break;
    }
#undef MANET_LF_CASE
    manet_profile_record(st, false, 1);
    return manet_check_launch("manet_local_match_volume");
}

/* training path: downsample configuration only (the reference's live default, config.py:49) */
int manet_local_match_arg_workspace_bytes(int h, int w, int C, int max_distance, size_t *bytes)
{
    return manet_local_workspace_bytes(h, w, C, max_distance, 1, bytes);
}

int manet_local_match_arg_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc, const float *cur,
                              int64_t c_sy, int64_t c_sx, int64_t c_sc, const int32_t *prev_labels, int h, int w, int C,
                              int n_ids, int max_distance, float *out, int32_t *arg_out, float *vol_out, void *workspace,
                              size_t workspace_bytes, manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, 1);
    if (rc) return rc;
    if (n_ids <= 0 || n_ids > MANET_MAX_IDS)
        return manet_set_error(MANET_E_INVALID, "n_ids=%d (supported 1..%d)", n_ids, MANET_MAX_IDS);
    if (!cur || !prev || !prev_labels || !out || !arg_out || !vol_out) return manet_set_error(MANET_E_INVALID, "null pointer");
    LocalLayout L = local_layout(h, w, C, max_distance, 1);
    if (!workspace || workspace_bytes < L.total)
        return manet_set_error(MANET_E_WORKSPACE, "local workspace %zu < %zu bytes", workspace_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *ap = (float *)(ws + L.off_ap), *bp = (float *)(ws + L.off_bp);
    long n = (long)C * L.hp * L.wp;
    hipLaunchKernelGGL(pool2x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cur, (long)c_sy, (long)c_sx,
                       (long)c_sc, prev, (long)p_sy, (long)p_sx, (long)p_sc, C, L.hp, L.wp, ap, bp);
    long plane = (long)L.hp * L.wp;
    launch_dist(max_distance, st, (const float *)ap, (long)L.wp, 1L, plane, (const float *)bp, (long)L.wp, 1L, plane, L.hp,
                L.wp, C, 1, vol_out);  // normalised pooled volume [P*P][hp][wp]: kept for the backward
    long tot = (long)h * w * n_ids;
    hipLaunchKernelGGL(local_min_arg_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)vol_out,
                       prev_labels, h, w, L.hp, L.wp, max_distance, n_ids, out, arg_out);
    return manet_check_launch("manet_local_match_arg_f32");
}

int manet_local_match_backward_workspace_bytes(int h, int w, int C, int max_distance, size_t *bytes)
{
    if (!bytes) return manet_set_error(MANET_E_INVALID, "bytes == NULL");
    int rc = check_local(h, w, C, max_distance, 1);
    if (rc) return rc;
    size_t plane = (size_t)(h / 2) * (w / 2);
    size_t PP = (size_t)(2 * max_distance + 1) * (2 * max_distance + 1);
    *bytes = manet_align_up(4 * plane * C * sizeof(float), 256) + manet_align_up(plane * PP * sizeof(float), 256);
    return MANET_OK;
}

int manet_local_match_backward_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc, const float *cur,
                                   int64_t c_sy, int64_t c_sx, int64_t c_sc, const float *vol, const int32_t *arg,
                                   const float *grad_out, int h, int w, int C, int n_ids, int max_distance,
                                   float *grad_prev, int64_t gp_sy, int64_t gp_sx, int64_t gp_sc, float *grad_cur,
                                   int64_t gc_sy, int64_t gc_sx, int64_t gc_sc, void *workspace, size_t workspace_bytes,
                                   manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, 1);
    if (rc) return rc;
    if (!cur || !prev || !vol || !arg || !grad_out || !grad_prev || !grad_cur || !workspace)
        return manet_set_error(MANET_E_INVALID, "null pointer");
    size_t need = 0;
    (void)manet_local_match_backward_workspace_bytes(h, w, C, max_distance, &need);
    if (workspace_bytes < need) return manet_set_error(MANET_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
    const int hp = h / 2, wp = w / 2, P = 2 * max_distance + 1;
    const long plane = (long)hp * wp;
    hipStream_t st = (hipStream_t)stream;
    float *ap = (float *)workspace, *bp = ap + plane * C, *gxp = bp + plane * C, *gyp = gxp + plane * C;
    float *dvn = (float *)((char *)workspace + manet_align_up(4 * (size_t)plane * C * sizeof(float), 256));
    long n = plane * C;
    hipLaunchKernelGGL(pool2x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cur, (long)c_sy, (long)c_sx,
                       (long)c_sc, prev, (long)p_sy, (long)p_sx, (long)p_sc, C, hp, wp, ap, bp);
    long nv = plane * P * P;
    {
        unsigned blocks = (unsigned)((nv + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, st, dvn, 0.0f, nv);
    }
    long tot = (long)h * w * n_ids;
    hipLaunchKernelGGL(local_bwd_scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, arg, grad_out, h, w,
                       hp, wp, n_ids, dvn);
    hipLaunchKernelGGL(local_bwd_dnorm_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, vol, dvn, nv);
    hipLaunchKernelGGL(local_bwd_dist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float *)ap,
                       (const float *)bp, (const float *)dvn, C, hp, wp, max_distance, gxp, gyp);
    long nf = (long)h * w * C;
    hipLaunchKernelGGL(local_bwd_unpool_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, (const float *)gxp,
                       (const float *)gyp, C, h, w, hp, wp, grad_cur, (long)gc_sy, (long)gc_sx, (long)gc_sc, grad_prev,
                       (long)gp_sy, (long)gp_sx, (long)gp_sc);
    return manet_check_launch("manet_local_match_backward_f32");
}

/* MODEL_LOCAL_DOWNSAMPLE = False in training (r5; IntVOS.py:299-313, :398-432).  manet_local_match_full_arg_f32: the raw
 * full-resolution distance volume (kept in vol_out [h][w][(2d+1)^2] for nobody -- the backward recomputes differences from the
 * embeddings -- but it is the caller's memory: h*w*(2d+1)^2 floats), the masked minimum and the winning window offset.
 * manet_local_match_full_backward_f32: both embeddings as CONTIGUOUS [C][h][w] planes, gradients likewise; dv_ws
 * [(2d+1)^2][h*w] floats of scratch. */
int manet_local_match_full_arg_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc, const float *cur, int64_t c_sy,
                                   int64_t c_sx, int64_t c_sc, const int32_t *prev_labels, int h, int w, int C, int n_ids,
                                   int max_distance, float *out, int32_t *arg_out, float *vol_out, manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, 0);
    if (rc) return rc;
    if (n_ids <= 0 || n_ids > MANET_MAX_IDS)
        return manet_set_error(MANET_E_INVALID, "n_ids=%d (supported 1..%d)", n_ids, MANET_MAX_IDS);
    if (!cur || !prev || !prev_labels || !out || !arg_out || !vol_out) return manet_set_error(MANET_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    launch_dist(max_distance, st, cur, (long)c_sy, (long)c_sx, (long)c_sc, prev, (long)p_sy, (long)p_sx, (long)p_sc, h, w, C, 0,
                vol_out);
    long tot = (long)h * w * n_ids;
    hipLaunchKernelGGL(local_min_arg_full_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)vol_out,
                       prev_labels, h, w, max_distance, n_ids, out, arg_out);
    return manet_check_launch("manet_local_match_full_arg_f32");
}

int manet_local_match_full_backward_f32(const float *prev_chw, const float *cur_chw, const int32_t *arg, const float *grad_out,
                                        int h, int w, int C, int n_ids, int max_distance, float *grad_prev_chw,
                                        float *grad_cur_chw, float *dv_ws, manet_stream_t stream)
{
    int rc = check_local(h, w, C, max_distance, 0);
    if (rc) return rc;
    if (!cur_chw || !prev_chw || !arg || !grad_out || !grad_prev_chw || !grad_cur_chw || !dv_ws)
        return manet_set_error(MANET_E_INVALID, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int P = 2 * max_distance + 1;
    const long plane = (long)h * w, nv = plane * P * P;
    {
        unsigned blocks = (unsigned)((nv + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, st, dv_ws, 0.0f, nv);
    }
    long tot = plane * n_ids;
    hipLaunchKernelGGL(local_bwd_scatter_full_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, arg, grad_out, plane,
                       n_ids, dv_ws);
    long n = plane * C;
    hipLaunchKernelGGL(local_bwd_dist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cur_chw, prev_chw,
                       (const float *)dv_ws, C, h, w, max_distance, grad_cur_chw, grad_prev_chw);
    return manet_check_launch("manet_local_match_full_backward_f32");
}

}  // extern "C"
