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
// Here a thread owns one output element and walks channels serially: lanes run along ox, so
// every load is a coalesced row segment of the NCHW planes, there are no copies, no cross-lane
// traffic and no scratch.  (This op is dead code in the reference's live tree -- SURVEY.md 0 --
// and is kept for API coverage; it is not on the benchmarked path.)
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

// Backward (reference: correlation_cuda_kernel.cu:150-241 grad wrt input1, :243-334 grad wrt input2; host wrapper
// correlation_cuda.cc:89-167).  The reference gathers with one 32-thread block per input element and a serial
// final reduce; here one thread owns one input element (lanes along x) and walks the kernel window and the
// displacement grid: no atomics, no scratch, no padded copies.
//   gin1[n][c][ya][xa] = 1/(K*K*C) * sum_{j,i} sum_{tj,ti} gout[n][tc][oy][ox] * in2[n][c][ya + tj*s2][xa + ti*s2]
//      with oy*s1 = ya + pad - max_disp - j,  ox*s1 = xa + pad - max_disp - i   (integral, inside the output)
//   gin2[n][c][yb][xb] = the same with in1 sampled at (yb - tj*s2, xb - ti*s2) and
//      oy*s1 = yb + pad - max_disp - j - tj*s2
template <bool SECOND>
__global__ __launch_bounds__(256) void correlation_backward_kernel(
    const float *__restrict__ other, const float *__restrict__ gout, int B, int C, int H, int W, int pad, int kr,
    int max_disp, int s1, int s2, int r, int oh, int ow, float nelems, float *__restrict__ gin)
{
    const int D = 2 * r + 1;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long plane = (long)H * W;
    if (idx >= (long)B * C * plane) return;
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const int c = (int)((idx / plane) % C);
    const int n = (int)(idx / (plane * C));
    const float *oth = other + ((long)n * C + c) * plane;
    const float *go = gout + (long)n * D * D * oh * ow;
    float acc = 0.0f;
    for (int tj = -r; tj <= r; ++tj)
        for (int ti = -r; ti <= r; ++ti) {
            const int tc = (tj + r) * D + (ti + r);
            // the other input's sample that multiplies this element at displacement (tj, ti)
            const int yo = SECOND ? y - tj * s2 : y + tj * s2;
            const int xo = SECOND ? x - ti * s2 : x + ti * s2;
            if (yo < 0 || yo >= H || xo < 0 || xo >= W) continue;
            const float ov = oth[(long)yo * W + xo];
            // first-input position of the product: (y, x) itself, or (yo, xo) when this is the second input
            const int y1p = (SECOND ? yo : y) + pad - max_disp, x1p = (SECOND ? xo : x) + pad - max_disp;
            float gs = 0.0f;
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
            acc = fmaf(gs, ov, acc);
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
    hipLaunchKernelGGL(correlation_forward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, in1, in2, B, C, H, W, pad_size, kr, max_displacement, stride1, stride2, r,
                       oh, ow, nelems, out);
    return manet_check_launch("manet_correlation_forward_f32");
}

int manet_correlation_backward_f32(const float *in1, const float *in2, const float *grad_out, int B, int C, int H, int W,
                                   int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                   float *grad_in1, float *grad_in2, manet_stream_t stream)
{
    int oc, oh, ow;
    int rc = manet_correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow);
    if (rc) return rc;
    if (!in1 || !in2 || !grad_out || !grad_in1 || !grad_in2 || B <= 0 || C <= 0)
        return manet_set_error(MANET_E_INVALID, "bad arguments");
    int kr = (kernel_size - 1) / 2;
    int r = max_displacement / stride2;
    long total = (long)B * C * H * W;
    float nelems = (float)(kernel_size * kernel_size * C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(correlation_backward_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in2,
                       grad_out, B, C, H, W, pad_size, kr, max_displacement, stride1, stride2, r, oh, ow, nelems, grad_in1);
    hipLaunchKernelGGL(correlation_backward_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in1,
                       grad_out, B, C, H, W, pad_size, kr, max_displacement, stride1, stride2, r, oh, ow, nelems, grad_in2);
    return manet_check_launch("manet_correlation_backward_f32");
}

}  // extern "C"
