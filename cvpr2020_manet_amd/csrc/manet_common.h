// Shared helpers of libmanet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "manet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MANET_WRONG_LABEL_PADDING_DISTANCE 1e20f  // IntVOS.py:17
#define MANET_MAX_IDS 64                          // object ids 0..63
#define MANET_MAX_C 128                           // embedding width (reference: 100)
#define MANET_MAX_LOCAL_DISTANCE 12               // reference default (config.py:50)

// error text of the last failure on this thread (manet_last_error_string)
int manet_set_error(int code, const char *fmt, ...);

// status of the launches enqueued so far; never synchronises
static inline int manet_check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return manet_set_error(MANET_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MANET_OK;
}

// opt-in launch timing (manet_profile_begin/_end, used by bench.py): when enabled, the launcher of the global-match
// main kernel (channel 0) and the local-window stage (channel 1) are bracketed with two HIP events each.
void manet_profile_record(hipStream_t st, bool start, int channel = 0);

// tuning knobs (manet_tune_set; defaults are the shipped configuration)
enum { MANET_TUNE_BLOCK_MAP = 0, MANET_TUNE_SPLITS = 1, MANET_TUNE_BF16_VARIANT = 2, MANET_TUNE_ABLATION = 3,
       MANET_TUNE_LOCAL_UNFUSED = 4, MANET_TUNE_F32_UNPIPED = 5, MANET_TUNE_FRAME_XC = 6, MANET_TUNE_REFINE_SUB = 7,
       MANET_TUNE_CONV1X1 = 8 /* 1: the LDS-weights 1x1 kernel even where the resident-weights one applies; 2-4: forms of the latter (seg_head.hip) */,
       MANET_TUNE_RESCUE_SPLITS = 9 /* bank splits per listed tile of bf16r's rescue launch (experiments) */,
       MANET_TUNE_ONE_ROUND = 10 /* 1: the fp32 kernel keeps the host's split count whatever the bank's real size (A/B timing) */,
       MANET_TUNE_RW_GROUPS = 11 /* resident-weights 1x1: pixel-range groups (default 256 = one per CU, two workgroups each) */,
       MANET_TUNE_RW_LDS_PAD = 12 /* ... and KiB of unused dynamic LDS added to its launch (caps its workgroups per CU; experiments) */,
       MANET_TUNE_DW_NARROW = 13 /* 0: the depthwise kernel tiles a narrow last column with standard 60 x 64 tiles (r2-r5; A/B timing and tests) */,
       MANET_TUNE_COUNT = 15 };
int manet_tune_get(int key, int dflt);

static inline size_t manet_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Per-frame operands written by manet_frame_prepare (csrc/global_match.hip), read by manet_global_match_prepared_ex
// (the query operand image, at offset 0: a frame workspace IS a MANET_EMB_PACKED query) and by manet_local_match_frames
// (csrc/local_match.hip: the 2x2-average-pooled plane with its border of the reference's padding value, and the tile
// table of the fused kernel).  max_distance < 0: no pooled plane (global match only).
struct ManetFrameLayout {
    size_t off_image, image_bytes, off_plane, plane_bytes, off_tab, tab_bytes, total;
    int hp, wp, HPAD, WS, TY, TX, nty, ntx;
    long PS;     // floats per channel plane (HPAD * WS)
    long N_pad;  // query rows of the image (N padded to whole query tiles)
};
ManetFrameLayout manet_frame_layout(int h, int w, int C, int compute, int max_distance);

// (sigmoid(x) - 0.5) * 2      IntVOS.py:612, :294
__device__ __forceinline__ float manet_normalize_dist(float x)
{
    float s = 1.0f / (1.0f + expf(-x));
    return (s - 0.5f) * 2.0f;
}
// LDS-DMA of one 1 KiB piece: lane l's 16 bytes at `gsrc` land at LDS byte address lds_dst + 16 l.
// Issued from inline asm on purpose: hipcc cannot tell the DMA's LDS destination from the ds_reads of the
// OTHER staging buffer and would drain vmcnt(0) in front of them, which serialises the prefetch (r1).
// Hidden in asm, the compiler does not count these loads; the kernel does (one s_waitcnt vmcnt(0) per
// step, right before the barrier that publishes the buffer).  M0 (the DMA's LDS base) is saved/restored
// inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (saddr form): one VGPR of address
// state instead of a pair -- for kernels at the register limit.
__device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// The local-window kernels' form: hardware exp2 / reciprocal (1 ulp each) instead of libm expf and an IEEE divide --
// 6 instructions instead of ~25 per window entry (the fused kernel normalises (2d+1)^2 entries per pooled pixel:
// 4.3 us of its 56 at d=12).  Exact where it matters: x = 0 -> 0, x = inf -> 1.  |error| <= 3e-7 absolute; the
// reference's own torch.sigmoid is no closer to the real value.  Every local kernel uses this one, so the fused
// and the materialised-volume paths stay bit-identical with each other.
__device__ __forceinline__ float manet_normalize_dist_local(float x)
{
    float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
    return (s - 0.5f) * 2.0f;
}
