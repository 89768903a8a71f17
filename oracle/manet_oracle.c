/*
 * manet_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the MANet matching hot path of the reference
 * (lightas/CVPR2020_MANet, networks/IntVOS.py).  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
 * HIP path against something that follows the reference algorithm line by line.
 * Nothing under cvpr2020_manet_amd/ may import, link or call this file.
 *
 * Parity pinning: the reference ships no golden vectors for this path
 * (SURVEY.md 4); the oracle is pinned against outputs of the reference itself,
 * generated in the build container by oracle/gen_golden.py (which imports
 * /root/reference/networks/IntVOS.py on CPU) and committed under tests/golden/.
 * tests/test_oracle_golden.py checks every function below against them.
 *
 * Arithmetic conventions (so that the HIP fp32 path can be compared bit-for-bit):
 *   - dot products and squared norms are k-ascending fmaf chains starting at 0
 *     (this is what v_mfma_f32_32x32x2_f32 computes, and a valid evaluation
 *     order for the reference's torch.matmul / torch.sum, whose order is
 *     unspecified);
 *   - d = (xs + ys) - 2*mm is evaluated as fmaf(-2, mm, xs + ys): 2*mm is exact,
 *     so this is the reference's expression with its two roundings;
 *   - everything else is evaluated with one rounding per reference op
 *     (compile with -ffp-contract=off).
 *
 * Layouts: every tensor argument is a raw pointer plus element strides, so the
 * oracle can be fed the same non-contiguous HWC views of C-major storage that
 * the reference's callers produce (IntVOS.py:605-606,625).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WRONG_LABEL_PADDING_DISTANCE 1e20f /* IntVOS.py:17 */

static inline float sq_norm_chain(const float *v, long stride, int C)
{
    float s = 0.0f;
    for (int c = 0; c < C; ++c) {
        float x = v[(long)c * stride];
        s = fmaf(x, x, s);
    }
    return s;
}

/* round-to-nearest-even fp32 -> bf16 -> fp32 (what the bf16 HIP path feeds the MFMA) */
static inline float bf16_round(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) { /* NaN: keep quiet NaN */
        u |= 0x00400000u;
        u &= 0xffff0000u;
    } else {
        u += 0x7fffu + ((u >> 16) & 1u);
        u &= 0xffff0000u;
    }
    float r;
    memcpy(&r, &u, 4);
    return r;
}

float oracle_bf16_round(float x) { return bf16_round(x); }

/* (sigmoid(x) - 0.5) * 2   -- IntVOS.py:612, :294 */
static inline float normalize_dist(float x)
{
    float s = 1.0f / (1.0f + expf(-x));
    return (s - 0.5f) * 2.0f;
}

/*
 * Global nearest-neighbour matching per object.
 *   IntVOS.py:160-210 nearest_neighbor_features_per_object (entry, ids = 0..n_ids-1 :200)
 *   IntVOS.py:113-157 chunk loop (results are chunk-invariant, so no chunking here)
 *   IntVOS.py:100-109 _selected_pixel (TEST_MODE drops label == -1 rows; a dropped row
 *                     and a row masked for every object give the same minimum)
 *   IntVOS.py:23-40   d = xs + ys - 2 x.y^T
 *   IntVOS.py:81-85   + wrong_label_mask * 1e20, min over the bank      (k_nn == 1)
 *   IntVOS.py:87-94   top-k smallest, invalid -> max valid, mean        (k_nn  > 1)
 *
 * q    [N][C]   element strides (qs_n, qs_c)
 * bank [M0][C]  element strides (bs_m, bs_c)
 * labels[M0] int32;  out [N][n_ids] float  (= the reference's [1,h,w,n_ids,1])
 * quant_bf16 != 0: inputs are rounded to bf16 first (oracle for the bf16 HIP path).
 * returns 0, or -1 when the top-k path has fewer than k_nn bank rows (torch.topk raises).
 */
/* k = 1 fast path of the CPU baseline (the scalar loop ran one dependent fmaf chain per
 * (n, m) pair, ~50 GFLOP/s on 256 threads).  Same arithmetic, vectorised ACROSS bank rows: the bank is held
 * channel-major kT[c][m], a block of QBLK queries x 16 bank rows keeps QBLK x 16 running dot products in vector
 * registers, every pair still accumulates q[c]*k[c] with fmaf in ascending c -- bit-identical results, the bank
 * block is reused by QBLK queries from L1.  target_clones: one binary, AVX-512 / AVX2 / baseline picked at
 * load time (the library is built in the build container and runs on the GPU box's host). */
#define QBLK 8
#define MBLK 16
#define K1_MAX_IDS 64
__attribute__((target_clones("avx512f", "avx2", "default")))
static void k1_block(const float *qv /* [QBLK][C] */, const float *xs /* [QBLK] */, int nq, const float *kT, long Mpad,
                     const float *ys /* [Mpad], +inf beyond M */, const int32_t *lab /* [Mpad], -1 beyond M */, int C,
                     int n_ids, float *bestv /* [QBLK][n_ids][MBLK] lane-wise running minima */)
{
    for (long m0 = 0; m0 < Mpad; m0 += MBLK) {
        float mm[QBLK][MBLK];
        for (int qi = 0; qi < QBLK; ++qi)
#pragma omp simd
            for (int j = 0; j < MBLK; ++j) mm[qi][j] = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float *kv = kT + (long)c * Mpad + m0;
            for (int qi = 0; qi < QBLK; ++qi) {
                const float qc = qv[qi * C + c];
#pragma omp simd
                for (int j = 0; j < MBLK; ++j) mm[qi][j] = fmaf(qc, kv[j], mm[qi][j]);
            }
        }
        for (int qi = 0; qi < nq; ++qi) {
            float d[MBLK];
#pragma omp simd
            for (int j = 0; j < MBLK; ++j) d[j] = fmaf(-2.0f, mm[qi][j], xs[qi] + ys[m0 + j]); /* IntVOS.py:39 */
            for (int o = 0; o < n_ids; ++o) {
                float *b = bestv + ((long)qi * n_ids + o) * MBLK;
#pragma omp simd
                for (int j = 0; j < MBLK; ++j) {
                    /* IntVOS.py:81-85: dists + wrong_label_mask * 1e20, min */
                    const float dd = d[j] + ((lab[m0 + j] != o) ? 1.0f : 0.0f) * WRONG_LABEL_PADDING_DISTANCE;
                    b[j] = (dd < b[j]) ? dd : b[j];
                }
            }
        }
    }
}

static void global_match_k1_fast(const float *q, long qs_n, long qs_c, const float *kb, const float *ys_in,
                                 const long *rows, const int32_t *labels, long N, long M, int C, int n_ids,
                                 int quant_bf16, float *out)
{
    const long Mpad = (M + MBLK - 1) / MBLK * MBLK;
    float *kT = (float *)calloc((size_t)(Mpad > 0 ? Mpad : 1) * (size_t)C, sizeof(float));
    float *ys = (float *)malloc(sizeof(float) * (size_t)(Mpad > 0 ? Mpad : 1));
    int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Mpad > 0 ? Mpad : 1));
    for (long m = 0; m < Mpad; ++m) {
        ys[m] = (m < M) ? ys_in[m] : INFINITY; /* padding rows: d = +inf, never the minimum */
        lab[m] = (m < M) ? labels[rows[m]] : -1;
    }
    for (long m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c) kT[(long)c * Mpad + m] = kb[m * C + c];
#pragma omp parallel
    {
        float *qv = (float *)calloc((size_t)QBLK * (size_t)C, sizeof(float));
        float *bestv = (float *)malloc(sizeof(float) * (size_t)QBLK * (size_t)n_ids * MBLK);
        float xs[QBLK];
#pragma omp for schedule(dynamic, 4)
        for (long n0 = 0; n0 < N; n0 += QBLK) {
            const int nq = (N - n0) < QBLK ? (int)(N - n0) : QBLK;
            for (int qi = 0; qi < nq; ++qi) {
                for (int c = 0; c < C; ++c) {
                    float v = q[(n0 + qi) * qs_n + (long)c * qs_c];
                    qv[qi * C + c] = quant_bf16 ? bf16_round(v) : v;
                }
                xs[qi] = sq_norm_chain(qv + qi * C, 1, C); /* IntVOS.py:32 */
            }
            for (long i = 0; i < (long)QBLK * n_ids * MBLK; ++i) bestv[i] = INFINITY;
            k1_block(qv, xs, nq, kT, Mpad, ys, lab, C, n_ids, bestv);
            for (int qi = 0; qi < nq; ++qi)
                for (int o = 0; o < n_ids; ++o) {
                    const float *b = bestv + ((long)qi * n_ids + o) * MBLK;
                    float best = INFINITY;
                    for (int j = 0; j < MBLK; ++j) best = (b[j] < best) ? b[j] : best;
                    out[(n0 + qi) * n_ids + o] = (M == 0) ? WRONG_LABEL_PADDING_DISTANCE : best;
                }
        }
        free(qv);
        free(bestv);
    }
    free(kT);
    free(ys);
    free(lab);
}

int oracle_global_match_f32(const float *q, long qs_n, long qs_c,
                            const float *bank, long bs_m, long bs_c,
                            const int32_t *labels, long N, long M0, int C,
                            int n_ids, int k_nn, int test_mode, int quant_bf16,
                            float *out)
{
    /* compaction (IntVOS.py:135-136) -- only changes which rows exist, which matters
       for the top-k row count; for k=1 it is a no-op on the result. */
    long M = 0;
    long *rows = (long *)malloc(sizeof(long) * (size_t)(M0 > 0 ? M0 : 1));
    for (long m = 0; m < M0; ++m)
        if (!test_mode || labels[m] != -1) rows[M++] = m;
    if (k_nn > 1 && M < k_nn) { free(rows); return -1; }

    float *kb = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1) * (size_t)C);
    float *ys = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    for (long m = 0; m < M; ++m) {
        const float *src = bank + rows[m] * bs_m;
        for (int c = 0; c < C; ++c) {
            float v = src[(long)c * bs_c];
            kb[m * C + c] = quant_bf16 ? bf16_round(v) : v;
        }
        ys[m] = sq_norm_chain(kb + m * C, 1, C); /* IntVOS.py:35 */
    }
    if (k_nn == 1 && !getenv("MANET_ORACLE_SCALAR")) { /* vectorised across bank rows, bit-identical to the loop below */
        global_match_k1_fast(q, qs_n, qs_c, kb, ys, rows, labels, N, M, C, n_ids, quant_bf16, out);
        free(kb);
        free(ys);
        free(rows);
        return 0;
    }

#pragma omp parallel
    {
        float *qv = (float *)malloc(sizeof(float) * (size_t)C);
        float *best = (float *)malloc(sizeof(float) * (size_t)n_ids * (size_t)(k_nn > 0 ? k_nn : 1));
#pragma omp for schedule(dynamic, 16)
        for (long n = 0; n < N; ++n) {
            for (int c = 0; c < C; ++c) {
                float v = q[n * qs_n + (long)c * qs_c];
                qv[c] = quant_bf16 ? bf16_round(v) : v;
            }
            float xs = sq_norm_chain(qv, 1, C); /* IntVOS.py:32 */
            for (int i = 0; i < n_ids * k_nn; ++i) best[i] = INFINITY;
            for (long m = 0; m < M; ++m) {
                const float *kv = kb + m * C;
                float mm = 0.0f;
                for (int c = 0; c < C; ++c) mm = fmaf(qv[c], kv[c], mm);
                float d = fmaf(-2.0f, mm, xs + ys[m]); /* IntVOS.py:39 */
                int lab = labels[rows[m]];
                for (int o = 0; o < n_ids; ++o) {
                    /* IntVOS.py:81-83: dists + wrong_label_mask * 1e20 */
                    float dd = d + ((lab != o) ? 1.0f : 0.0f) * WRONG_LABEL_PADDING_DISTANCE;
                    float *b = best + o * k_nn;
                    /* keep the k_nn smallest, ascending (== -topk(-d), IntVOS.py:87-88) */
                    if (dd < b[k_nn - 1]) {
                        int j = k_nn - 1;
                        while (j > 0 && b[j - 1] > dd) { b[j] = b[j - 1]; --j; }
                        b[j] = dd;
                    }
                }
            }
            for (int o = 0; o < n_ids; ++o) {
                float *b = best + o * k_nn;
                if (k_nn == 1) {
                    /* M == 0 cannot reach torch.min (it raises); report the padding distance */
                    out[n * n_ids + o] = (M == 0) ? WRONG_LABEL_PADDING_DISTANCE : b[0];
                } else {
                    /* IntVOS.py:89-94 */
                    float pad = -INFINITY;
                    for (int j = 0; j < k_nn; ++j) {
                        float valid = (b[j] < WRONG_LABEL_PADDING_DISTANCE) ? 1.0f : 0.0f;
                        float masked = b[j] * valid;
                        if (masked > pad) pad = masked;
                    }
                    float s = 0.0f;
                    for (int j = 0; j < k_nn; ++j)
                        s += (b[j] < WRONG_LABEL_PADDING_DISTANCE) ? b[j] : pad;
                    out[n * n_ids + o] = s / (float)k_nn;
                }
            }
        }
        free(qv);
        free(best);
    }
    free(kb);
    free(ys);
    free(rows);
    return 0;
}

/*
 * Normalise + aggregate with the stored per-frame global map.
 *   IntVOS.py:611-612  g = (sigmoid(g) - 0.5) * 2            (normalize != 0)
 *   IntVOS.py:620-622  g = where(g <= mem, g, mem); mem = g   (mem != NULL)
 *   IntVOS.py:718-723  same merge in int_seghead (no normalise)
 * x is updated in place; mem (same length) is updated in place when given.
 */
void oracle_normalize_merge_f32(float *x, float *mem, long n, int normalize)
{
    for (long i = 0; i < n; ++i) {
        float g = x[i];
        if (normalize) g = normalize_dist(g);
        if (mem) {
            g = (g <= mem[i]) ? g : mem[i];
            mem[i] = g;
        }
        x[i] = g;
    }
}

/* bilinear, align_corners=True, as aten/native/UpSample.h area_pixel_compute_scale +
 * upsample_bilinear2d: scale = (in-1)/(out-1) (0 when out == 1), src = scale*dst,
 * i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1. */
static inline void bilin_coeff(int dst, int in_size, int out_size, int *i0, int *i1,
                               float *l0, float *l1)
{
    float scale = (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
    float src = scale * (float)dst;
    int a = (int)src;
    if (a > in_size - 1) a = in_size - 1;
    *i0 = a;
    *i1 = a + ((a < in_size - 1) ? 1 : 0);
    *l1 = src - (float)a;
    *l0 = 1.0f - *l1;
}

/*
 * Local (2d+1)^2 window squared-L2 distances between the current (query) frame x and
 * the previous frame y.            IntVOS.py:266-315 local_pairwise_distances2
 *   downsample != 0 (live default, :279-296): avg_pool2d 2x2 of both (floor sizes),
 *     pad y with 1e20, offsets l = dy*P+dx <-> displacement (dy-d, dx-d) on the pooled
 *     grid, sum_c (x - y_off)^2, (sigmoid-0.5)*2, bilinear(align_corners) up to (h,w).
 *   downsample == 0 (:299-313): same on the full-resolution grid, no sigmoid, no resize.
 * x, y: [h][w][C] with element strides (s_y, s_x, s_c) each.   out: [h][w][P*P].
 */
void oracle_local_dist_f32(const float *x, long xs_y, long xs_x, long xs_c,
                           const float *y, long ys_y, long ys_x, long ys_c,
                           int h, int w, int C, int d, int downsample, float *out)
{
    const int P = 2 * d + 1;
    const int PP = P * P;
    if (!downsample) {
#pragma omp parallel for schedule(static)
        for (int py = 0; py < h; ++py)
            for (int px = 0; px < w; ++px)
                for (int dy = 0; dy < P; ++dy)
                    for (int dx = 0; dx < P; ++dx) {
                        int yy = py + dy - d, xx = px + dx - d;
                        int oob = (yy < 0 || yy >= h || xx < 0 || xx >= w);
                        float acc = 0.0f;
                        for (int c = 0; c < C; ++c) {
                            float a = x[py * xs_y + px * xs_x + c * xs_c];
                            float b = oob ? 1e20f : y[yy * ys_y + xx * ys_x + c * ys_c];
                            float df = a - b;
                            acc = fmaf(df, df, acc); /* one rounding less than torch.pow(.,2)+sum; same as the HIP kernel */
                        }
                        out[((long)py * w + px) * PP + dy * P + dx] = acc;
                    }
        return;
    }
    const int hp = h / 2, wp = w / 2; /* avg_pool2d floors odd sizes */
    float *xp = (float *)malloc(sizeof(float) * (size_t)C * hp * wp);
    float *yp = (float *)malloc(sizeof(float) * (size_t)C * hp * wp);
    for (int c = 0; c < C; ++c)
        for (int py = 0; py < hp; ++py)
            for (int px = 0; px < wp; ++px) {
                const float *a = x + (2 * py) * xs_y + (2 * px) * xs_x + c * xs_c;
                const float *b = y + (2 * py) * ys_y + (2 * px) * ys_x + c * ys_c;
                /* window accumulated row-major from 0, then / 4 */
                float sa = ((a[0] + a[xs_x]) + a[xs_y]) + a[xs_y + xs_x];
                float sb = ((b[0] + b[ys_x]) + b[ys_y]) + b[ys_y + ys_x];
                xp[((long)c * hp + py) * wp + px] = sa / 4.0f;
                yp[((long)c * hp + py) * wp + px] = sb / 4.0f;
            }
    float *dp = (float *)malloc(sizeof(float) * (size_t)PP * hp * wp); /* [l][hp][wp] */
#pragma omp parallel for schedule(static)
    for (int py = 0; py < hp; ++py)
        for (int px = 0; px < wp; ++px)
            for (int dy = 0; dy < P; ++dy)
                for (int dx = 0; dx < P; ++dx) {
                    int yy = py + dy - d, xx = px + dx - d;
                    int oob = (yy < 0 || yy >= hp || xx < 0 || xx >= wp);
                    float acc = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        float a = xp[((long)c * hp + py) * wp + px];
                        float b = oob ? 1e20f : yp[((long)c * hp + yy) * wp + xx];
                        float df = a - b;
                        acc = fmaf(df, df, acc);
                    }
                    dp[((long)(dy * P + dx) * hp + py) * wp + px] = normalize_dist(acc);
                }
#pragma omp parallel for schedule(static)
    for (int oy = 0; oy < h; ++oy) {
        int y0, y1;
        float hl0, hl1;
        bilin_coeff(oy, hp, h, &y0, &y1, &hl0, &hl1);
        for (int ox = 0; ox < w; ++ox) {
            int x0, x1;
            float wl0, wl1;
            bilin_coeff(ox, wp, w, &x0, &x1, &wl0, &wl1);
            for (int l = 0; l < PP; ++l) {
                const float *pl = dp + (long)l * hp * wp;
                float v = hl0 * (wl0 * pl[y0 * wp + x0] + wl1 * pl[y0 * wp + x1]) +
                          hl1 * (wl0 * pl[y1 * wp + x0] + wl1 * pl[y1 * wp + x1]);
                out[((long)oy * w + ox) * PP + l] = v;
            }
        }
    }
    free(xp);
    free(yp);
    free(dp);
}

/*
 * Masked minimum over the local window, per object.    IntVOS.py:398-408, :428-432
 *   offset_labels[y][x][l=(by,bx)] = labels[y + 2(by-d)][x + 2(bx-d)], zero outside the
 *   image (F.pad default 0 + unfold stride 2);  mask = (offset_label == id);
 *   out[y][x][o] = min_l (mask ? dist[y][x][l] : 1.0)
 * dist: [h][w][P*P] contiguous; labels: [h][w] int32; ids = 0..n_ids-1; out: [h][w][n_ids].
 */
void oracle_local_masked_min_f32(const float *dist, const int32_t *labels, int h, int w,
                                 int d, int n_ids, float *out)
{
    const int P = 2 * d + 1;
    const int PP = P * P;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int o = 0; o < n_ids; ++o) {
                float m = INFINITY;
                for (int by = 0; by < P; ++by)
                    for (int bx = 0; bx < P; ++bx) {
                        int yy = y + 2 * (by - d), xx = x + 2 * (bx - d);
                        float lab = 0.0f; /* zero padding == background id 0 */
                        if (yy >= 0 && yy < h && xx >= 0 && xx < w) lab = (float)labels[yy * w + xx];
                        float v = (lab == (float)o) ? dist[((long)y * w + x) * PP + by * P + bx] : 1.0f;
                        if (v < m) m = v;
                    }
                out[((long)y * w + x) * n_ids + o] = m;
            }
}

/* convenience: a11 end to end (IntVOS.py:345-434, USE_CORRELATION_COST=False, MODEL_UNFOLD=True) */
void oracle_local_match_f32(const float *prev, long ps_y, long ps_x, long ps_c,
                            const float *cur, long cs_y, long cs_x, long cs_c,
                            const int32_t *labels, int h, int w, int C, int d, int n_ids,
                            int downsample, float *out)
{
    const int P = 2 * d + 1;
    float *dist = (float *)malloc(sizeof(float) * (size_t)h * w * P * P);
    /* IntVOS.py:370: local_pairwise_distances2(query_embedding, prev_frame_embedding) */
    oracle_local_dist_f32(cur, cs_y, cs_x, cs_c, prev, ps_y, ps_x, ps_c, h, w, C, d, downsample, dist);
    oracle_local_masked_min_f32(dist, labels, h, w, d, n_ids, out);
    free(dist);
}

/*
 * correlation_package forward (FlowNet2-style), fp32.
 *   correlation_cuda.cc:25-34          output shape rule
 *   correlation_cuda_kernel.cu:46-70   zero-padded NHWC copies of both inputs
 *   correlation_cuda_kernel.cu:73-147  out[n][tc][oy][ox] = sum / (K*K*C), fp32 accumulate,
 *                                      tc = (tj+r)*(2r+1) + (ti+r), r = max_disp/stride2,
 *                                      centre (y1,x1) = (oy*s1 + max_disp, ox*s1 + max_disp)
 *                                      in padded coordinates, second input displaced by
 *                                      (tj*s2, ti*s2), kernel window [-kr, kr]^2.
 * in1,in2: [B][C][H][W] contiguous.  out: [B][(2r+1)^2][outH][outW] contiguous.
 * The accumulation order here is channel-ascending inside kernel-window row-major; the
 * reference sums 32 strided partials per warp and reduces by shuffles, so agreement with a
 * real run of it is to rounding only (and that CUDA code cannot be built here: SURVEY 8c).
 */
int oracle_correlation_out_dims(int H, int W, int pad, int K, int max_disp, int s1, int s2,
                                int *out_c, int *out_h, int *out_w)
{
    int kr = (K - 1) / 2;
    int border = kr + max_disp;
    int ph = H + 2 * pad, pw = W + 2 * pad;
    int r = max_disp / s2;
    *out_c = (2 * r + 1) * (2 * r + 1);
    *out_h = (int)ceilf((float)(ph - 2 * border) / (float)s1);
    *out_w = (int)ceilf((float)(pw - 2 * border) / (float)s1);
    return (*out_h > 0 && *out_w > 0) ? 0 : -1;
}

int oracle_correlation_forward_f32(const float *in1, const float *in2, int B, int C, int H, int W,
                                   int pad, int K, int max_disp, int s1, int s2, float *out)
{
    int oc, oh, ow;
    if (oracle_correlation_out_dims(H, W, pad, K, max_disp, s1, s2, &oc, &oh, &ow)) return -1;
    const int kr = (K - 1) / 2;
    const int r = max_disp / s2;
    const int D = 2 * r + 1;
    const float nelems = (float)(K * K * C);
#pragma omp parallel for schedule(static) collapse(2)
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox) {
                int y1 = oy * s1 + max_disp, x1 = ox * s1 + max_disp; /* padded coords */
                for (int tj = -r; tj <= r; ++tj)
                    for (int ti = -r; ti <= r; ++ti) {
                        int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
                        float acc = 0.0f;
                        for (int j = -kr; j <= kr; ++j)
                            for (int i = -kr; i <= kr; ++i) {
                                int ya = y1 + j - pad, xa = x1 + i - pad; /* unpadded */
                                int yb = y2 + j - pad, xb = x2 + i - pad;
                                int ina = (ya >= 0 && ya < H && xa >= 0 && xa < W);
                                int inb = (yb >= 0 && yb < H && xb >= 0 && xb < W);
                                if (!ina || !inb) continue; /* a zero factor adds +0 */
                                for (int c = 0; c < C; ++c) {
                                    float a = in1[(((long)n * C + c) * H + ya) * W + xa];
                                    float b = in2[(((long)n * C + c) * H + yb) * W + xb];
                                    acc = fmaf(a, b, acc);
                                }
                            }
                        int tc = (tj + r) * D + (ti + r);
                        out[(((long)n * oc + tc) * oh + oy) * ow + ox] = acc / nelems;
                    }
            }
    return 0;
}

/*
 * Mask step of the driver:
 *   test.py:253-255   F.interpolate(logits, (H,W), bilinear, align_corners=True) then argmax over ids
 *                     (first maximum wins)
 *   IntVOS.py:598-599 F.interpolate(mask.float(), (h,w), 'nearest').int(): source index
 *                     min(floor(dst * (in/out)), in-1) with the scale in float (aten UpSample.h)
 * logits [n_ids][h][w]; mask [H][W] int64; small [h][w] int32.
 */
void oracle_upsample_argmax(const float *logits, int n_ids, int h, int w, int H, int W, int64_t *mask,
                            int32_t *small)
{
    for (int Y = 0; Y < H; ++Y) {
        int y0, y1;
        float hl0, hl1;
        bilin_coeff(Y, h, H, &y0, &y1, &hl0, &hl1);
        for (int X = 0; X < W; ++X) {
            int x0, x1;
            float wl0, wl1;
            bilin_coeff(X, w, W, &x0, &x1, &wl0, &wl1);
            int best = 0;
            float bv = 0.0f;
            for (int o = 0; o < n_ids; ++o) {
                const float *p = logits + (long)o * h * w;
                float v = hl0 * (wl0 * p[y0 * w + x0] + wl1 * p[y0 * w + x1]) +
                          hl1 * (wl0 * p[y1 * w + x0] + wl1 * p[y1 * w + x1]);
                if (o == 0 || v > bv) { bv = v; best = o; }
            }
            mask[(long)Y * W + X] = best;
        }
    }
    float sy = (float)H / (float)h, sx = (float)W / (float)w;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int Y = (int)floorf((float)y * sy), X = (int)floorf((float)x * sx);
            if (Y > H - 1) Y = H - 1;
            if (X > W - 1) X = W - 1;
            small[y * w + x] = (int32_t)mask[(long)Y * W + X];
        }
}

/*
 * Depthwise 7x7 conv (padding 3) + bias + BatchNorm(eval, folded to scale/shift) + ReLU
 * (IntVOS.py:491-493,500-502: conv1 -> bn1 -> relu1 of _split_separable_conv2d).
 * in/out [B][C][h][w], weight [C][7][7]; taps accumulated row-major with fmaf, as the HIP kernel.
 */
void oracle_dwconv7x7_bn_relu_f32(const float *in, int B, int C, int h, int w, const float *weight,
                                  const float *bias, const float *scale, const float *shift, int relu, float *out)
{
#pragma omp parallel for schedule(static)
    for (int p = 0; p < B * C; ++p) {
        int c = p % C;
        const float *src = in + (long)p * h * w;
        const float *wk = weight + (long)c * 49;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float acc = 0.0f;
                for (int ky = 0; ky < 7; ++ky)
                    for (int kx = 0; kx < 7; ++kx) {
                        int yy = y + ky - 3, xx = x + kx - 3;
                        float v = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? src[yy * w + xx] : 0.0f;
                        acc = fmaf(v, wk[ky * 7 + kx], acc);
                    }
                float v = fmaf(acc + (bias ? bias[c] : 0.0f), scale ? scale[c] : 1.0f, shift ? shift[c] : 0.0f);
                out[(long)p * h * w + y * w + x] = relu ? (v > 0.0f ? v : 0.0f) : v;
            }
    }
}
