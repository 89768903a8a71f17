// Depthwise 7x7 convolution + BatchNorm(eval) + ReLU, fused -- SURVEY.md 8f rank 1 (the step right after
// the matching path): the building block of the reference's DynamicSegHead (IntVOS.py:488-506,
// _split_separable_conv2d: conv1 (groups = channels, 7x7, padding 3) -> bn1 -> relu1).
//
// Why it is here: on MI355X the framework's conv library has no tuned depthwise-7x7 fp32 kernel and falls
// back to a naive one (0.83 ms per call at [3,256,120,214]; 55 % of the GPU time of a propagated frame,
// profiles/r01_e2e_kernel_stats.csv) although the op is a pure HBM stream: 2 x 4 B per element.
//
//   out[b][c][y][x] = relu( (sum_{ky,kx} in[b][c][y+ky-3][x+kx-3] * w[c][ky][kx] + bias[c]) * scale[c] + shift[c] )
// with zero padding, scale = gamma / sqrt(var + eps), shift = beta - mean * scale (BatchNorm in eval mode).
//
// Workgroup = one (b, c) plane tile of 16 x 64 outputs; the (16+6) x (64+6) input tile is staged in LDS
// (row-contiguous global reads), each thread produces 4 adjacent outputs of one row from a 7 x 10 window
// (10 LDS floats feed 28 FMAs per kernel row); the 49 weights are wave-uniform (scalar loads).
#include "manet_common.h"

namespace {

constexpr int DW_K = 7, DW_R = 3;
constexpr int DW_TY = 16, DW_TX = 64;
constexpr int DW_LW = 72;  // LDS row stride (>= DW_TX + 6, multiple of 4)

__global__ __launch_bounds__(256) void dwconv7x7_bn_relu_kernel(const float *__restrict__ in, int C, int h, int w,
                                                                const float *__restrict__ weight,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift, int relu,
                                                                float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float tile[(DW_TY + 2 * DW_R) * DW_LW];
    const int plane_id = blockIdx.z;  // b * C + c
    const int c = plane_id % C;
    const int x0 = blockIdx.x * DW_TX, y0 = blockIdx.y * DW_TY;
    const float *src = in + (long)plane_id * h * w;
    const int tid = threadIdx.x;
    for (int i = tid; i < (DW_TY + 2 * DW_R) * (DW_TX + 2 * DW_R); i += 256) {
        int r = i / (DW_TX + 2 * DW_R), col = i - r * (DW_TX + 2 * DW_R);
        int yy = y0 - DW_R + r, xx = x0 - DW_R + col;
        tile[r * DW_LW + col] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? src[(long)yy * w + xx] : 0.0f;
    }
    __syncthreads();
    const int ty = tid >> 4, tg = tid & 15;  // output row, group of 4 columns
    const float *wk = weight + (long)c * DW_K * DW_K;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int ky = 0; ky < DW_K; ++ky) {
        const float *row = tile + (ty + ky) * DW_LW + 4 * tg;
        f32x4 v0 = *(const f32x4 *)row, v1 = *(const f32x4 *)(row + 4);
        float v8 = row[8], v9 = row[9];
        float win[10] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], v8, v9};
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) {
            const float wv = wk[ky * DW_K + kx];
            a0 = fmaf(win[kx], wv, a0);
            a1 = fmaf(win[kx + 1], wv, a1);
            a2 = fmaf(win[kx + 2], wv, a2);
            a3 = fmaf(win[kx + 3], wv, a3);
        }
    }
    const int y = y0 + ty, x = x0 + 4 * tg;
    if (y >= h) return;
    const float bc = bias ? bias[c] : 0.0f, sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
    float r[4] = {a0, a1, a2, a3};
    float *dst = out + (long)plane_id * h * w + (long)y * w + x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (x + j < w) {
            float v = fmaf(r[j] + bc, sc, sh);
            dst[j] = relu ? fmaxf(v, 0.0f) : v;
        }
    }
}

}  // namespace

extern "C" int manet_dwconv7x7_bn_relu_f32(const float *in, int B, int C, int h, int w, const float *weight,
                                           const float *bias, const float *bn_scale, const float *bn_shift, int relu,
                                           float *out, manet_stream_t stream)
{
    if (!in || !weight || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || (long)B * C > 65535)
        return manet_set_error(MANET_E_INVALID, "bad arguments (B*C must be <= 65535)");
    dim3 grid((unsigned)((w + DW_TX - 1) / DW_TX), (unsigned)((h + DW_TY - 1) / DW_TY), (unsigned)(B * C));
    hipLaunchKernelGGL(dwconv7x7_bn_relu_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, C, h, w, weight, bias,
                       bn_scale, bn_shift, relu, out);
    return manet_check_launch("manet_dwconv7x7_bn_relu_f32");
}
