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

}  // extern "C"
