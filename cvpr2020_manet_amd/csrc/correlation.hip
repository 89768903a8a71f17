// correlation_package forward for MI355X (gfx950).
//
// Replaces correlation_package/correlation_cuda.cc:10-87 (pybind `forward`) and the kernels
// correlation_cuda_kernel.cu:46-70 (channels_first: zero-padded NHWC copies) and :73-147
// (correlation_forward).  Same op contract:
//     out[n][tc][oy][ox] = 1/(K*K*C) * sum_{j,i in [-kr,kr]} sum_c in1p[n][y1+j][x1+i][c] * in2p[n][y2+j][x2+i][c]
// with (y1,x1) = (oy*s1 + max_disp, ox*s1 + max_disp) in zero-padded coordinates, (y2,x2) displaced
// by (tj*s2, ti*s2), tc = (tj+r)*(2r+1) + (ti+r), r = max_disp/s2, fp32 accumulate.
//
// The reference launches one 32-thread block per output pixel, strides channels over the warp
// and shuffles the partial sums together, after materialising padded NHWC copies of both inputs.
// Two kernels: an LDS-tiled one (below; float / double / half) that stages both patches once per channel chunk
// and reuses them for every displacement, and -- as the fallback for displacement grids / halos the tile cannot
// hold -- correlation_forward_kernel, where a thread owns one output element and walks channels serially (lanes
// along ox: every load is a coalesced row segment of the NCHW planes).  No padded copies, no cross-lane traffic.
// (This op is dead code in the reference's live tree -- SURVEY.md 0 -- and is not on the benchmarked path.)
#include <hip/hip_fp16.h>
#include <math.h>

#include "manet_common.h"

namespace {

__global__ __launch_bounds__(256) void correlation_forward_kernel(
    const float *__restrict__ in1, const float *__restrict__ in2, int B, int C, int H, int W, int pad, int kr,
    int max_disp, int s1, int s2, int r, int oh, int ow, float nelems,
    float *__restrict__ out)
{
    const int D = 2 * r + 1;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * D * D * oh * ow;
    if (i >= total) return;
    int ox = (int)(i % ow);
    int oy = (int)((i / ow) % oh);
    int tc = (int)((i / ((long)ow * oh)) % (D * D));
    int n = (int)(i / ((long)ow * oh * D * D));
    int tj = tc / D - r, ti = tc % D - r;
    int y1 = oy * s1 + max_disp, x1 = ox * s1 + max_disp;
    int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
    const long plane = (long)H * W;
    const float *a0 = in1 + (long)n * C * plane;
    const float *b0 = in2 + (long)n * C * plane;
    float acc = 0.0f;
    for (int j = -kr; j <= kr; ++j)
        for (int ii = -kr; ii <= kr; ++ii) {
            int ya = y1 + j - pad, xa = x1 + ii - pad;
            int yb = y2 + j - pad, xb = x2 + ii - pad;
            bool in_a = (ya >= 0 && ya < H && xa >= 0 && xa < W);
            bool in_b = (yb >= 0 && yb < H && xb >= 0 && xb < W);
            if (!in_a || !in_b) continue;  // a zero-padded factor contributes +0
            const float *pa = a0 + (long)ya * W + xa;
            const float *pb = b0 + (long)yb * W + xb;
            for (int c = 0; c < C; ++c) acc = fmaf(pa[c * plane], pb[c * plane], acc);
        }
    out[i] = acc / nelems;
}

// ---- LDS-tiled forward, float / double / half (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// correlation_cuda_kernel.cu:386-415) -------------------------------------------------------------------------
// Workgroup = 16 consecutive output columns of one output row, all (2r+1)^2 displacements.  Per chunk of channels
// the first input's K x (15 s1 + K) patch and the second input's halo (K + 2 r s2 rows, 15 s1 + K + 2 r s2 columns),
// zero outside the image (= the reference's zero-padded copies, correlation_cuda_kernel.cu:46-70), are staged in
// LDS once and reused by every displacement: global traffic drops from one load pair per multiply-add to one load
// per staged element.  Thread = (output column, displacement row tj [+16]); it keeps the DMAX displacement columns
// of its row(s) as running sums.  Arithmetic as the reference's kernel (:121-127): the product is formed in the
// tensor's type (half: rounded to half; double: then narrowed), accumulated in fp32, the mean written in the
// tensor's type.  float: fmaf chain -- for kernel_size 1 the same order as the one-thread-per-output kernel above
// and the oracle (bit-identical); for larger kernels the channel chunks reorder the sum (fp32 rounding).
__device__ __forceinline__ float corr_prod(float a, float b, float acc) { return fmaf(a, b, acc); }
__device__ __forceinline__ float corr_prod(double a, double b, float acc) { return acc + (float)(a * b); }
__device__ __forceinline__ float corr_prod(__half a, __half b, float acc) { return acc + __half2float(__hmul(a, b)); }
__device__ __forceinline__ void corr_store(float *p, float v) { *p = v; }
__device__ __forceinline__ void corr_store(double *p, float v) { *p = (double)v; }
__device__ __forceinline__ void corr_store(__half *p, float v) { *p = __float2half(v); }
template <typename T> __device__ __forceinline__ T corr_zero();
template <> __device__ __forceinline__ float corr_zero<float>() { return 0.0f; }
template <> __device__ __forceinline__ double corr_zero<double>() { return 0.0; }
template <> __device__ __forceinline__ __half corr_zero<__half>() { return __float2half(0.0f); }

template <typename T, int DMAX>
__global__ __launch_bounds__(256) void correlation_forward_tiled_kernel(
    const T *__restrict__ in1, const T *__restrict__ in2, int C, int H, int W, int pad, int kr, int max_disp, int s1,
    int s2, int r, int oh, int ow, int CC, float nelems, T *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char corr_smem[];
    constexpr int NP = DMAX > 16 ? 2 : 1;  // displacement rows per thread
    const int D = 2 * r + 1, K = 2 * kr + 1;
    const int W1 = 15 * s1 + K, R2 = K + 2 * r * s2, W2 = W1 + 2 * r * s2;
    T *A = (T *)corr_smem;            // [CC][K][W1]
    T *Bh = A + (size_t)CC * K * W1;  // [CC][R2][W2]
    const int tid = threadIdx.x, oxl = tid & 15, slot = tid >> 4;
    const int ox0 = blockIdx.x * 16, oy = blockIdx.y, n = blockIdx.z;
    // padded coordinates of the tile's first-input patch origin and of the halo origin
    const int y1 = oy * s1 + max_disp - kr, x1 = ox0 * s1 + max_disp - kr;
    const int y2 = y1 - r * s2, x2 = x1 - r * s2;
    const long plane = (long)H * W;
    const T *a0 = in1 + (long)n * C * plane, *b0 = in2 + (long)n * C * plane;
    float acc[NP][DMAX];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int t = 0; t < DMAX; ++t) acc[p][t] = 0.0f;
    for (int c0 = 0; c0 < C; c0 += CC) {
        const int cn = (C - c0) < CC ? (C - c0) : CC;
        __syncthreads();  // the previous chunk is consumed
        for (int e = tid; e < cn * K * W1; e += 256) {
            const int c = e / (K * W1), rem = e - c * (K * W1), yy = rem / W1, xx = rem - yy * W1;
            const int iy = y1 + yy - pad, ix = x1 + xx - pad;
            A[e] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? a0[(long)(c0 + c) * plane + (long)iy * W + ix] : corr_zero<T>();
        }
        for (int e = tid; e < cn * R2 * W2; e += 256) {
            const int c = e / (R2 * W2), rem = e - c * (R2 * W2), yy = rem / W2, xx = rem - yy * W2;
            const int iy = y2 + yy - pad, ix = x2 + xx - pad;
            Bh[e] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? b0[(long)(c0 + c) * plane + (long)iy * W + ix] : corr_zero<T>();
        }
        __syncthreads();
        for (int j = 0; j < K; ++j)
            for (int i = 0; i < K; ++i)
                for (int c = 0; c < cn; ++c) {
                    const T a = A[(c * K + j) * W1 + oxl * s1 + i];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int tj = slot + 16 * p;
                        if (tj < D) {
                            const T *brow = Bh + ((size_t)c * R2 + tj * s2 + j) * W2 + oxl * s1 + i;
#pragma unroll
                            for (int t = 0; t < DMAX; ++t)
                                if (t < D) acc[p][t] = corr_prod(a, brow[t * s2], acc[p][t]);
                        }
                    }
                }
    }
    const int ox = ox0 + oxl;
    if (ox >= ow) return;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int tj = slot + 16 * p;
        if (tj >= D) continue;
#pragma unroll
        for (int t = 0; t < DMAX; ++t)
            if (t < D) corr_store(out + (((long)n * D * D + (long)tj * D + t) * oh + oy) * ow + ox, acc[p][t] / nelems);
    }
}

template <typename T>
static bool launch_corr_tiled(hipStream_t st, const T *in1, const T *in2, int B, int C, int H, int W, int pad, int kr,
                              int md, int s1, int s2, int r, int oh, int ow, float nelems, T *out)
{
    const int D = 2 * r + 1, K = 2 * kr + 1;
    if (D > 32) return false;
    const size_t per_c = ((size_t)K * (15 * s1 + K) + (size_t)(K + 2 * r * s2) * (15 * s1 + K + 2 * r * s2)) * sizeof(T);
    int CC = (int)(65536 / per_c);
    if (CC < 1) return false;  // halo of one channel does not fit: the one-thread-per-output kernel handles it
    if (CC > C) CC = C;
    if (CC > 32) CC = 32;
    const size_t lds = per_c * CC;
    dim3 grid((unsigned)((ow + 15) / 16), (unsigned)oh, (unsigned)B);
#define MANET_CORR_LAUNCH(DM_)                                                                                          \
    {                                                                                                                   \
        (void)hipFuncSetAttribute((const void *)correlation_forward_tiled_kernel<T, DM_>,                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
        hipLaunchKernelGGL((correlation_forward_tiled_kernel<T, DM_>), grid, dim3(256), lds, st, in1, in2, C, H, W, pad, \
                           kr, md, s1, s2, r, oh, ow, CC, nelems, out);                                                 \
    }
    if (D <= 8) MANET_CORR_LAUNCH(8)
    else if (D <= 16) MANET_CORR_LAUNCH(16)
    else MANET_CORR_LAUNCH(32)
#undef MANET_CORR_LAUNCH
    return true;
}

// Backward (reference: correlation_cuda_kernel.cu:150-241 grad wrt input1, :243-334 grad wrt input2; host wrapper
// correlation_cuda.cc:89-167).  The reference gathers with one 32-thread block per input element and a serial
// final reduce; here one thread owns one input element (lanes along x) and walks the kernel window and the
// displacement grid: no atomics, no scratch, no padded copies.
//   gin1[n][c][ya][xa] = 1/(K*K*C) * sum_{j,i} sum_{tj,ti} gout[n][tc][oy][ox] * in2[n][c][ya + tj*s2][xa + ti*s2]
//      with oy*s1 = ya + pad - max_disp - j,  ox*s1 = xa + pad - max_disp - i   (integral, inside the output)
//   gin2[n][c][yb][xb] = the same with in1 sampled at (yb - tj*s2, xb - ti*s2) and
//      oy*s1 = yb + pad - max_disp - j - tj*s2
// T = float (half inputs are widened by the caller: fp32 gradients rounded once) or double (the reference dispatches a double
// backward as well, correlation_cuda_kernel.cu:495-541; arithmetic in the tensor's type like its scalar_t kernels)
template <bool SECOND, typename T>
__global__ __launch_bounds__(256) void correlation_backward_kernel(
    const T *__restrict__ other, const T *__restrict__ gout, int B, int C, int H, int W, int pad, int kr,
    int max_disp, int s1, int s2, int r, int oh, int ow, T nelems, T *__restrict__ gin)
{
    const int D = 2 * r + 1;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long plane = (long)H * W;
    if (idx >= (long)B * C * plane) return;
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const int c = (int)((idx / plane) % C);
    const int n = (int)(idx / (plane * C));
    const T *oth = other + ((long)n * C + c) * plane;
    const T *go = gout + (long)n * D * D * oh * ow;
    T acc = (T)0;
    for (int tj = -r; tj <= r; ++tj)
        for (int ti = -r; ti <= r; ++ti) {
            const int tc = (tj + r) * D + (ti + r);
            // the other input's sample that multiplies this element at displacement (tj, ti)
            const int yo = SECOND ? y - tj * s2 : y + tj * s2;
            const int xo = SECOND ? x - ti * s2 : x + ti * s2;
            if (yo < 0 || yo >= H || xo < 0 || xo >= W) continue;
            const T ov = oth[(long)yo * W + xo];
            // first-input position of the product: (y, x) itself, or (yo, xo) when this is the second input
            const int y1p = (SECOND ? yo : y) + pad - max_disp, x1p = (SECOND ? xo : x) + pad - max_disp;
            T gs = (T)0;
            for (int j = -kr; j <= kr; ++j) {
                const int ys = y1p - j;
                if (ys < 0 || ys % s1) continue;
                const int oy = ys / s1;
                if (oy >= oh) continue;
                for (int i = -kr; i <= kr; ++i) {
                    const int xs = x1p - i;
                    if (xs < 0 || xs % s1) continue;
                    const int ox = xs / s1;
                    if (ox >= ow) continue;
                    gs += go[((long)tc * oh + oy) * ow + ox];
                }
            }
            acc = fma(gs, ov, acc);
        }
    gin[idx] = acc / nelems;
}

}  // namespace

extern "C" {

// correlation_cuda.cc:25-34
int manet_correlation_out_dims(int H, int W, int pad_size, int kernel_size, int max_displacement, int stride1,
                               int stride2, int *out_c, int *out_h, int *out_w)
{
    if (!out_c || !out_h || !out_w) return manet_set_error(MANET_E_INVALID, "null pointer");
    if (H <= 0 || W <= 0 || pad_size < 0 || kernel_size < 1 || max_displacement < 0 || stride1 < 1 || stride2 < 1)
        return manet_set_error(MANET_E_INVALID, "bad correlation parameters");
    int kr = (kernel_size - 1) / 2;
    int border = kr + max_displacement;
    int ph = H + 2 * pad_size, pw = W + 2 * pad_size;
    int r = max_displacement / stride2;
    *out_c = (2 * r + 1) * (2 * r + 1);
    *out_h = (int)ceilf((float)(ph - 2 * border) / (float)stride1);
    *out_w = (int)ceilf((float)(pw - 2 * border) / (float)stride1);
    if (*out_h <= 0 || *out_w <= 0) return manet_set_error(MANET_E_INVALID, "empty correlation output");
    return MANET_OK;
}

int manet_correlation_forward_f32(const float *in1, const float *in2, int B, int C, int H, int W, int pad_size,
                                  int kernel_size, int max_displacement, int stride1, int stride2, float *out,
                                  manet_stream_t stream)
{
    int oc, oh, ow;
    int rc = manet_correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
    if (rc) return rc;
    if (!in1 || !in2 || !out || B <= 0 || C <= 0) return manet_set_error(MANET_E_INVALID, "bad arguments");
    int kr = (kernel_size - 1) / 2;
    int r = max_displacement / stride2;
    long total = (long)B * oc * oh * ow;
    float nelems = (float)(kernel_size * kernel_size * C);
    if (!launch_corr_tiled<float>((hipStream_t)stream, in1, in2, B, C, H, W, pad_size, kr, max_displacement, stride1,
                                  stride2, r, oh, ow, nelems, out))
        hipLaunchKernelGGL(correlation_forward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, in1, in2, B, C, H, W, pad_size, kr, max_displacement, stride1, stride2, r,
                           oh, ow, nelems, out);
    return manet_check_launch("manet_correlation_forward_f32");
}

int manet_correlation_forward(const void *in1, const void *in2, int dtype, int B, int C, int H, int W, int pad_size,
                              int kernel_size, int max_displacement, int stride1, int stride2, void *out,
                              manet_stream_t stream)
{
    if (dtype == MANET_CORR_F32)
        return manet_correlation_forward_f32((const float *)in1, (const float *)in2, B, C, H, W, pad_size, kernel_size,
                                             max_displacement, stride1, stride2, (float *)out, stream);
    int oc, oh, ow;
    int rc = manet_correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
    if (rc) return rc;
    if (!in1 || !in2 || !out || B <= 0 || C <= 0) return manet_set_error(MANET_E_INVALID, "bad arguments");
    int kr = (kernel_size - 1) / 2;
    int r = max_displacement / stride2;
    float nelems = (float)(kernel_size * kernel_size * C);
    bool ok = false;
    if (dtype == MANET_CORR_F16)
        ok = launch_corr_tiled<__half>((hipStream_t)stream, (const __half *)in1, (const __half *)in2, B, C, H, W, pad_size, kr,
                                       max_displacement, stride1, stride2, r, oh, ow, nelems, (__half *)out);
    else if (dtype == MANET_CORR_F64)
        ok = launch_corr_tiled<double>((hipStream_t)stream, (const double *)in1, (const double *)in2, B, C, H, W, pad_size, kr,
                                       max_displacement, stride1, stride2, r, oh, ow, nelems, (double *)out);
    else
        return manet_set_error(MANET_E_INVALID, "correlation dtype %d (MANET_CORR_F32 / _F16 / _F64)", dtype);
    if (!ok)
        return manet_set_error(MANET_E_INVALID, "half / double correlation: displacement grid (2r+1 = %d) or halo too "
                                                "large for the tiled kernel (float has a fallback)", 2 * r + 1);
    return manet_check_launch("manet_correlation_forward");
}

}  // extern "C"

template <typename T>
static int correlation_backward_t(const T *in1, const T *in2, const T *grad_out, int B, int C, int H, int W, int pad_size,
                                  int kernel_size, int max_displacement, int stride1, int stride2, T *grad_in1, T *grad_in2,
                                  manet_stream_t stream)
{
    int oc, oh, ow;
    int rc = manet_correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
    if (rc) return rc;
    if (!in1 || !in2 || !grad_out || !grad_in1 || !grad_in2 || B <= 0 || C <= 0)
        return manet_set_error(MANET_E_INVALID, "bad arguments");
    int kr = (kernel_size - 1) / 2;
    int r = max_displacement / stride2;
    long total = (long)B * C * H * W;
    T nelems = (T)(kernel_size * kernel_size * C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((correlation_backward_kernel<false, T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in2,
                       grad_out, B, C, H, W, pad_size, kr, max_displacement, stride1, stride2, r, oh, ow, nelems, grad_in1);
    hipLaunchKernelGGL((correlation_backward_kernel<true, T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in1,
                       grad_out, B, C, H, W, pad_size, kr, max_displacement, stride1, stride2, r, oh, ow, nelems, grad_in2);
    return manet_check_launch("manet_correlation_backward");
}

extern "C" {

int manet_correlation_backward_f32(const float *in1, const float *in2, const float *grad_out, int B, int C, int H, int W,
                                   int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                   float *grad_in1, float *grad_in2, manet_stream_t stream)
{
    return correlation_backward_t<float>(in1, in2, grad_out, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                         grad_in1, grad_in2, stream);
}

int manet_correlation_backward_f64(const double *in1, const double *in2, const double *grad_out, int B, int C, int H, int W,
                                   int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                   double *grad_in1, double *grad_in2, manet_stream_t stream)
{
    return correlation_backward_t<double>(in1, in2, grad_out, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                          grad_in1, grad_in2, stream);
}

}  // extern "C"
