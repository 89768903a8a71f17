// Mask step between two propagated frames, MI355X (gfx950).
//
// The reference's driver turns the head's logits into the next frame's inputs with three framework
// calls per frame, on the sequential chain that limits multi-GPU scaling:
//   test.py:253     pred = F.interpolate(logits [1,n_ids,h,w], size=(H,W), mode='bilinear', align_corners=True)
//   test.py:255     pred = torch.argmax(pred, dim=1)                       -> [1,H,W] int64
//   IntVOS.py:598-599 (next call)  F.interpolate(prev_mask.float(), size=(h,w), mode='nearest').int()
// i.e. a [1,n_ids,H,W] fp32 intermediate (16 n_ids MB at 480p) is written and re-read only to take an
// argmax.  Here one launch produces both results straight from the [n_ids,h,w] logits:
//   mask  [H][W]  int64  = argmax_o bilinear(logits[o])(Y,X)            (first maximum wins, as torch)
//   small [h][w]  int32  = mask[min(floor(y*H/h),H-1)][min(floor(x*W/w),W-1)]   ('nearest' source index)
// The small grid recomputes its few source pixels instead of waiting for the big one: no second pass.
#include "manet_common.h"

namespace {

struct Bil {
    int i0, i1;
    float l0, l1;
};
// aten area_pixel_compute_scale (align_corners=True) + linear source index, as in local_match.hip
__device__ __forceinline__ Bil bil(int dst, int in_size, int out_size)
{
    float scale = (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
    float src = scale * (float)dst;
    int a = (int)src;
    if (a > in_size - 1) a = in_size - 1;
    Bil b;
    b.i0 = a;
    b.i1 = a + ((a < in_size - 1) ? 1 : 0);
    b.l1 = src - (float)a;
    b.l0 = 1.0f - b.l1;
    return b;
}

__device__ __forceinline__ int argmax_at(const float *__restrict__ logits, int n_ids, int h, int w, int Y, int X,
                                         int H, int W)
{
    const Bil by = bil(Y, h, H), bx = bil(X, w, W);
    const long plane = (long)h * w;
    int best = 0;
    float bv = -INFINITY;
    for (int o = 0; o < n_ids; ++o) {
        const float *p = logits + o * plane;
        float v = by.l0 * (bx.l0 * p[by.i0 * w + bx.i0] + bx.l1 * p[by.i0 * w + bx.i1]) +
                  by.l1 * (bx.l0 * p[by.i1 * w + bx.i0] + bx.l1 * p[by.i1 * w + bx.i1]);
        if (v > bv || o == 0) {  // strict: the first maximum wins (torch.argmax)
            bv = v;
            best = o;
        }
    }
    return best;
}

__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float *__restrict__ logits, int n_ids, int h,
                                                              int w, int H, int W, long long *__restrict__ mask,
                                                              int *__restrict__ small)
{
    const long big = (long)H * W;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < big) {
        if (!mask) return;
        int Y = (int)(i / W), X = (int)(i - (long)Y * W);
        mask[i] = argmax_at(logits, n_ids, h, w, Y, X, H, W);
        return;
    }
    i -= big;
    if (!small || i >= (long)h * w) return;
    int y = (int)(i / w), x = (int)(i - (long)y * w);
    // aten nearest_neighbor_compute_source_index: min(floor(dst * (in/out)), in - 1), scale in float
    float sy = (float)H / (float)h, sx = (float)W / (float)w;
    int Y = (int)floorf((float)y * sy), X = (int)floorf((float)x * sx);
    if (Y > H - 1) Y = H - 1;
    if (X > W - 1) X = W - 1;
    small[i] = argmax_at(logits, n_ids, h, w, Y, X, H, W);
}

// F.interpolate(mask.float(), size=(h, w), mode='nearest').int() (IntVOS.py:598-599) on an int64 mask, in one launch
__global__ __launch_bounds__(256) void label_resize_kernel(const long long *__restrict__ mask, int H, int W, int h, int w,
                                                           int *__restrict__ small)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)h * w) return;
    const int y = (int)(i / w), x = (int)(i - (long)y * w);
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;  // aten nearest_neighbor_compute_source_index
    int Y = (int)floorf((float)y * sy), X = (int)floorf((float)x * sx);
    if (Y > H - 1) Y = H - 1;
    if (X > W - 1) X = W - 1;
    small[i] = (int)mask[(long)Y * W + X];
}

// manet_frame_begin: the label resize + the two small fills a propagated frame needs before its matches, one launch
__global__ __launch_bounds__(256) void frame_begin_kernel(const long long *__restrict__ mask, int H, int W, int h, int w,
                                                          int *__restrict__ small, float *__restrict__ fill, long fill_n, float fill_v,
                                                          float *__restrict__ scalar_dst, float scalar_v)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && scalar_dst) *scalar_dst = scalar_v;
    if (i < fill_n) fill[i] = fill_v;
    if (i >= (long)h * w) return;
    const int y = (int)(i / w), x = (int)(i - (long)y * w);
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;  // as label_resize_kernel
    int Y = (int)floorf((float)y * sy), X = (int)floorf((float)x * sx);
    if (Y > H - 1) Y = H - 1;
    if (X > W - 1) X = W - 1;
    small[i] = (int)mask[(long)Y * W + X];
}

// the per-object channels of the head's input (IntVOS.py:663-669): out[o] = (global map of o, local map of o, label == o)
__global__ __launch_bounds__(256) void head_inputs_kernel(const float *__restrict__ gmap, const float *__restrict__ lmap,
                                                          const int *__restrict__ labels, long HW, int n_ids,
                                                          float *__restrict__ out)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int lab = labels[p];
    for (int o = 0; o < n_ids; ++o) {
        float *dst = out + (long)o * 3 * HW + p;
        dst[0] = gmap[p * n_ids + o];
        dst[HW] = lmap[p * n_ids + o];
        dst[2 * HW] = lab == o ? 1.0f : 0.0f;
    }
}

}  // namespace

extern "C" int manet_label_resize_nearest(const int64_t *mask_hw, int H, int W, int h, int w, int32_t *label_small_hw,
                                          manet_stream_t stream)
{
    if (!mask_hw || !label_small_hw || h <= 0 || w <= 0 || H <= 0 || W <= 0) return manet_set_error(MANET_E_INVALID, "bad arguments");
    hipLaunchKernelGGL(label_resize_kernel, dim3((unsigned)(((long)h * w + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)mask_hw, H, W, h, w, (int *)label_small_hw);
    return manet_check_launch("manet_label_resize_nearest");
}

extern "C" int manet_frame_begin(const int64_t *mask_hw, int H, int W, int h, int w, int32_t *label_small_hw, float *fill,
                                 int64_t fill_n, float fill_value, float *scalar_dst, float scalar_value, manet_stream_t stream)
{
    if (!mask_hw || !label_small_hw || h <= 0 || w <= 0 || H <= 0 || W <= 0 || fill_n < 0 || (fill_n > 0 && !fill))
        return manet_set_error(MANET_E_INVALID, "bad arguments");
    const long n = (long)h * w > fill_n ? (long)h * w : (long)fill_n;
    hipLaunchKernelGGL(frame_begin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long *)mask_hw, H, W, h, w, (int *)label_small_hw, fill, (long)fill_n, fill_value, scalar_dst,
                       scalar_value);
    return manet_check_launch("manet_frame_begin");
}

extern "C" int manet_head_inputs_f32(const float *global_map, const float *local_map, const int32_t *labels, int64_t HW,
                                     int n_ids, float *out, manet_stream_t stream)
{
    if (!global_map || !local_map || !labels || !out || HW <= 0 || n_ids <= 0) return manet_set_error(MANET_E_INVALID, "bad arguments");
    hipLaunchKernelGGL(head_inputs_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, global_map,
                       local_map, (const int *)labels, (long)HW, n_ids, out);
    return manet_check_launch("manet_head_inputs_f32");
}

extern "C" int manet_upsample_argmax(const float *logits, int n_ids, int h, int w, int H, int W, int64_t *mask_hw,
                                     int32_t *label_small_hw, manet_stream_t stream)
{
    if (!logits || n_ids <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (!mask_hw && !label_small_hw))
        return manet_set_error(MANET_E_INVALID, "bad arguments");
    long total = (long)H * W + (long)h * w;
    hipLaunchKernelGGL(upsample_argmax_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, n_ids, h, w, H, W, (long long *)mask_hw, (int *)label_small_hw);
    return manet_check_launch("manet_upsample_argmax");
}
