/*
 * manet_hip.h -- C ABI of libmanet_hip.so: the MI355X (gfx950) implementation of MANet's
 * per-frame matching path.  This is the drop-in boundary: every entry point takes raw device
 * pointers, explicit sizes / element strides and a HIP stream, returns an int status, never
 * throws, never allocates, never synchronises the device.  Work is enqueued on `stream`
 * (pass the caller's current stream; the reference's native op does the same:
 * correlation_cuda.cc:76).  Scratch memory is caller-provided; its size comes from the matching
 * *_workspace_bytes query (the reference's native op resizes caller-provided scratch tensors:
 * correlation_cuda.cc:36-42).
 *
 * Reference interfaces replaced (paths relative to the reference repo root):
 *   manet_global_match_*      networks/IntVOS.py:160-210  nearest_neighbor_features_per_object
 *                             (+ :113-157 chunk loop, :100-109 pixel selection, :62-97 masked
 *                             min / top-k, :23-40 pairwise distances) and, fused as an epilogue,
 *                             :611-612 normalisation and :615-622 / :716-723 min-aggregation.
 *   manet_normalize_merge_f32 networks/IntVOS.py:611-622, :718-723 (stand-alone form).
 *   manet_local_dist_f32      networks/IntVOS.py:266-315  local_pairwise_distances2
 *   manet_local_match_f32     networks/IntVOS.py:345-434
 *                             local_previous_frame_nearest_neighbor_features_per_object
 *   manet_correlation_forward_f32
 *                             correlation_package/correlation_cuda.cc:10-87 (pybind `forward`)
 *                             + correlation_cuda_kernel.cu:46-147.
 *   manet_*_arg_f32 / manet_*_backward_f32
 *                             torch.autograd through the two functions above (train_stage1.py:126-156)
 *                             and correlation_cuda.backward (correlation_cuda.cc:89-167)
 *   manet_upsample_argmax     test.py:253-255 + networks/IntVOS.py:598-599
 *   manet_dwconv7x7_bn_relu_f32  networks/IntVOS.py:491-493,500-502
 *   manet_relu_conv1x1_c1_f32    networks/IntVOS.py:519,525
 *
 * NaN inputs (outside the reference's contract, documented deviation): the global match propagates a NaN
 * distance to the output like torch.min does, but a NaN bank row poisons only its own object (in the reference
 * the +1e20 label mask spreads it to every object); the local masked min (fminf) and the mask-step argmax ignore
 * NaN candidates where torch.min / torch.argmax would return them.
 *
 * Status codes: 0 = ok; negative = error (see MANET_E_*); manet_last_error_string() gives the
 * text of the last error raised on the calling thread.
 * Thread-safety: the data-path functions are re-entrant and keep no state between calls (the
 * opt-in manet_profile_* / manet_tune_* hooks at the end are process-wide and not on the data path).
 */
#ifndef MANET_HIP_H
#define MANET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *manet_stream_t; /* a hipStream_t */

#define MANET_OK 0
#define MANET_E_INVALID (-1)   /* bad argument (shape, null pointer, unsupported value) */
#define MANET_E_WORKSPACE (-2) /* workspace too small */
#define MANET_E_LAUNCH (-3)    /* hipGetLastError() after a launch was not hipSuccess */
#define MANET_E_NODEVICE (-4)  /* no usable HIP device */

/* arithmetic type of the QK^T contraction */
#define MANET_COMPUTE_F32 0     /* v_mfma_f32_32x32x2_f32, exact fp32 (bit-equal to an fmaf chain) */
#define MANET_COMPUTE_BF16 1    /* v_mfma_f32_32x32x16_bf16 on inputs rounded to bf16, fp32 accumulate */
#define MANET_COMPUTE_BF16X3 2  /* split-bf16 (hi+lo, 3 MFMAs): fp32-class accuracy at bf16 rate */
/* bf16 filter + exact fp32 re-rank (k_nn = 1, C <= 106): IntVOS.py:81-85 is a minimum, so a filter that provably keeps the
 * arg-min row may drop every other one.  A bf16 pre-pass over every 8th bank tile bounds the minimum, a bf16 pass over
 * the whole bank keeps the rows whose bf16 distance is within the rounding bound of that, and those few are re-evaluated in
 * the reference's fp32 arithmetic (the fmaf chains of MANET_COMPUTE_F32): the result EQUALS MANET_COMPUTE_F32's bit for bit
 * at about 1.4x the cost of MANET_COMPUTE_BF16 on embeddings the bf16 pass can tell apart.  Where it cannot -- a 32-query x
 * 32-row block of which more than an eighth qualifies -- the block is listed whole and its 1 024 distances are re-evaluated on
 * the fp32 matrix pipe by the re-rank (spatially smooth embeddings: ~1.8x the cost of MANET_COMPUTE_BF16); a 32-query block with
 * more than 1 024 such blocks, or more than 128 listed rows per pair on average, sends its 256-query tile through the exact
 * fp32 kernel instead (dealt to the listed tiles only; if that is every tile -- e.g. all rows of an object identical -- the
 * frame costs the fp32 path's time plus the filter's: manet_global_match_refine_stats2, MANET_EPI_REFINE_EXACT).
 * NaN embeddings propagate as in MANET_COMPUTE_F32: a NaN bank row makes its own object's minimum NaN for every query, a NaN
 * query row its own pixel's, every other pair keeps the fp32 bits. */
#define MANET_COMPUTE_BF16_REFINE 3

/* storage type of an embedding operand of the *_ex entry points (take the producer's layout) */
#define MANET_EMB_F32 0    /* float */
#define MANET_EMB_BF16 1   /* bfloat16 as raw 16-bit words (what a bf16 encoder epilogue stores) */
#define MANET_EMB_PACKED 2 /* query only: the MFMA operand image written by manet_query_pack */

/* flags of the fused epilogue of manet_global_match */
#define MANET_EPI_NORMALIZE 1 /* g = (sigmoid(g) - 0.5) * 2            IntVOS.py:611-612 */
/* manet_global_match_prepared_ex with a MANET_EMB_PACKED query and k_nn = 1 only: the caller promises that `match_ws`
 * holds 0xff bytes throughout (e.g. it filled it once and has only used it for calls with this flag since); the call
 * then skips its fill launch and leaves the workspace in that state again (the epilogue re-arms what it reads). */
#define MANET_EPI_KEYS_ARMED 2
/* MANET_COMPUTE_BF16_REFINE only: skip the bf16 filter and let the rescue pass (the exact fp32 kernel) take every query tile --
 * the same bits at the fp32 path's cost (+ ~0.03 ms), for a caller that knows from the previous frame
 * (manet_global_match_refine_rescued_async) that the bf16 pass cannot tell these embeddings apart; without it such a frame costs
 * the filter pass AND the fp32 kernel. */
#define MANET_EPI_REFINE_EXACT 4

const char *manet_version(void);
const char *manet_last_error_string(void);

/* ------------------------------------------------------------------------------------------ */
/* Global nearest-neighbour matching (IntVOS.py:160-210).
 *
 *   query  [N][C]  fp32, element strides (q_stride_n, q_stride_c)  -- the caller's view of its
 *                  C-major [C,h,w] embedding is (1, h*w); a row-major [N,C] tensor is (C, 1).
 *   bank   [M0][C] fp32, element strides (b_stride_m, b_stride_c)  -- the reference pixel set
 *                  (a T-frame memory bank is T frames stacked along M0).
 *   labels [M0]    int32; rows whose label is outside 0..n_ids-1 (e.g. -1 = unlabelled) never
 *                  match any object (IntVOS.py:100-109 drops them, :137 masks them).
 *   n_ids          number of object ids, ids = 0..n_ids-1 (IntVOS.py:200: arange(0, gt_ids+1)).
 *   k_nn           1 = masked minimum (IntVOS.py:84-85); 2..MANET_MAX_KNN = mean of the k smallest
 *                  with invalid entries replaced by the largest valid one (IntVOS.py:87-94).
 *   out    [N][n_ids] fp32 contiguous (= the reference's [1,h,w,n_ids,1]).
 *   mem_inout      NULL, or [N][n_ids] fp32: the stored global map of this frame; the result is
 *                  min-merged with it and written back to both (IntVOS.py:620-622).
 *   epilogue_flags MANET_EPI_* bits.
 *   compute        MANET_COMPUTE_*.
 *   workspace      device scratch of at least manet_global_match_workspace_bytes(...) bytes,
 *                  256-byte aligned.  Contents are undefined on entry and on return.
 */
#define MANET_MAX_KNN 8

int manet_global_match_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, int k_nn,
                                       int compute, size_t *bytes);

int manet_global_match(const float *query, int64_t q_stride_n, int64_t q_stride_c,
                       const float *bank, int64_t b_stride_m, int64_t b_stride_c,
                       const int32_t *labels, int64_t N, int64_t M0, int C, int n_ids, int k_nn,
                       int compute, float *out, float *mem_inout, int epilogue_flags,
                       void *workspace, size_t workspace_bytes, manet_stream_t stream);

/* Two-step form of the same operation, for callers that match many query frames against one
 * bank (test.py:237-259 propagates a whole clip against one annotated frame):
 * prepare once (sort rows by object, pad, pre-compute |k|^2), then match per frame.
 * `bank_ws` must stay untouched between the two calls; `match_ws` is per-call scratch. */
int manet_bank_workspace_bytes(int64_t M0, int C, int n_ids, int compute, size_t *bytes);
int manet_match_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, int k_nn, int compute,
                                size_t *bytes);
int manet_bank_prepare(const float *bank, int64_t b_stride_m, int64_t b_stride_c,
                       const int32_t *labels, int64_t M0, int C, int n_ids, int compute,
                       void *bank_ws, size_t bank_ws_bytes, manet_stream_t stream);
int manet_global_match_prepared(const float *query, int64_t q_stride_n, int64_t q_stride_c,
                                const void *bank_ws, int64_t N, int64_t M0, int C, int n_ids,
                                int k_nn, int compute, float *out, float *mem_inout,
                                int epilogue_flags, void *match_ws, size_t match_ws_bytes,
                                manet_stream_t stream);

/* The same three entry points for embeddings in the PRODUCER's storage (the embeddings of a
 * whole clip are computed once, test.py:143-154, and live in HBM as [F,C,h,w]): `*_dtype` = MANET_EMB_*.
 * With bf16 storage the path reads 2-byte embeddings end to end (for MANET_COMPUTE_BF16 the values are used as
 * they are; MANET_COMPUTE_F32 widens them exactly).
 * manet_query_pack writes a frame's query operand image once (e.g. right behind the encoder, for every frame
 * of the clip); manet_global_match_prepared_ex(query = that image, MANET_EMB_PACKED, ...) then skips the
 * per-frame pack pass entirely.  The image depends on (N, C, compute) only. */
int manet_bank_prepare_ex(const void *bank, int emb_dtype, int64_t b_stride_m, int64_t b_stride_c,
                          const int32_t *labels, int64_t M0, int C, int n_ids, int compute,
                          void *bank_ws, size_t bank_ws_bytes, manet_stream_t stream);
int manet_query_pack_bytes(int64_t N, int C, int compute, size_t *bytes);
int manet_query_pack(const void *query, int emb_dtype, int64_t q_stride_n, int64_t q_stride_c, int64_t N,
                     int C, int compute, void *packed, size_t packed_bytes, manet_stream_t stream);
int manet_global_match_prepared_ex(const void *query, int emb_dtype, int64_t q_stride_n,
                                   int64_t q_stride_c, const void *bank_ws, int64_t N, int64_t M0, int C,
                                   int n_ids, int k_nn, int compute, float *out, float *mem_inout,
                                   int epilogue_flags, void *match_ws, size_t match_ws_bytes,
                                   manet_stream_t stream);
int manet_global_match_ex(const void *query, int q_dtype, int64_t q_stride_n, int64_t q_stride_c,
                          const void *bank, int b_dtype, int64_t b_stride_m, int64_t b_stride_c,
                          const int32_t *labels, int64_t N, int64_t M0, int C, int n_ids, int k_nn,
                          int compute, float *out, float *mem_inout, int epilogue_flags, void *workspace,
                          size_t workspace_bytes, manet_stream_t stream);

/* Per-frame operands, made ONCE when a frame's embedding is produced (the reference computes every
 * frame's embedding up front, test.py:143-154, batch 14): from ONE read of the C-major embedding,
 *   - the query operand image of the global match (the frame workspace's first bytes: pass the workspace itself as a
 *     MANET_EMB_PACKED query to manet_global_match_prepared_ex), and, when max_distance >= 0,
 *   - the 2x2-average-pooled plane of the local match (IntVOS.py:282-284) with a border of the reference's padding
 *     value 1e20 (IntVOS.py:287) and the fused local kernel's tile table, read by manet_local_match_frames.
 * A propagated frame then reads its own embedding once and the previous frame's not at all (that frame's plane was
 * made when IT was the current frame; test.py:259 `prev_embedding = current_embedding`).
 *   emb            n_frames x [h][w][C] in MANET_EMB_F32 / MANET_EMB_BF16 storage, element strides (s_f, s_y, s_x, s_c);
 *                  extract_feature's [B,C,h,w] batch (IntVOS.py:578-581) is (C*h*w, w, 1, h*w): one launch for the batch.
 *   frames_ws      n_frames workspaces, frame_ws_stride bytes apart (>= manet_frame_workspace_bytes, multiple of 1024).
 *   max_distance   the local window radius the plane is padded for (cfg.MODEL_MAX_LOCAL_DISTANCE), or -1: no plane.
 *   fill_ptr       optional: fill_words 32-bit words at fill_ptr are set to fill_value by the same launch (the caller's
 *                  next local-match `out`, pre-set to 1.0f = 0x3f800000, see manet_local_match_frames). */
int manet_frame_workspace_bytes(int h, int w, int C, int compute, int max_distance, size_t *bytes);
int manet_frame_prepare(const void *emb, int emb_dtype, int64_t s_f, int64_t s_y, int64_t s_x, int64_t s_c,
                        int n_frames, int h, int w, int C, int compute, int max_distance, void *frames_ws,
                        size_t frame_ws_stride, void *fill_ptr, int64_t fill_words, uint32_t fill_value,
                        manet_stream_t stream);

/* The embedding layer's epilogue fused with manet_frame_prepare (IntVOS.py:537-543, :578-581): `conv_out`
 * is the raw fp32 output of the 1x1 embedding convolution (embedding_conv, bias included), n_frames x [h][w][C] with element
 * strides as above.  One launch computes y = relu(x * scale[c] + shift[c]) (eval-mode bn2 folded by the caller + relu2),
 * stores the embedding in emb_out [n_frames][C][h][w] in emb_out_dtype storage (MANET_EMB_F32 / MANET_EMB_BF16, contiguous,
 * 8-byte aligned) and, from the same registers, the frames' operands -- exactly what manet_frame_prepare would make from the
 * embedding AS STORED.  The batch-norm, ReLU, storage-cast and prepare launches of the producer (three elementwise passes and a
 * re-read of the embedding) collapse into this one. */
int manet_embed_finish(const float *conv_out, int64_t s_f, int64_t s_y, int64_t s_x, int64_t s_c, const float *scale,
                       const float *shift, int relu, void *emb_out, int emb_out_dtype, int n_frames, int h, int w, int C,
                       int compute, int max_distance, void *frames_ws, size_t frame_ws_stride, manet_stream_t stream);

/* MANET_COMPUTE_BF16_REFINE on a prepared bank when the query's packed image exists already (manet_frame_prepare /
 * manet_query_pack with MANET_COMPUTE_BF16 or _BF16_REFINE): the fp32 re-rank also needs the query as stored, so this entry
 * point takes both.  query_image == NULL: same as manet_global_match_prepared_ex(..., MANET_COMPUTE_BF16_REFINE, ...).
 * manet_global_match_refine_stats reads back (blocking copy on the NULL stream -- tests / benchmarks; it does not wait for work on
 * non-blocking streams: synchronise the stream the match ran on first, as ops.PreparedBank.refine_stats does) what the last call on `match_ws` did:
 * candidate rows the filter pass listed (+, per 32 x 32 block it listed whole, the number of (query, half pass) lanes that held
 * a qualifying row: a lower bound of that block's qualifying rows), and whether some 32-query block's candidate bucket (128 rows per pair on
 * average, 1 024 whole 32 x 32 blocks) was incomplete -- 1: the 256-query tiles of those blocks also went through the exact fp32 kernel (the bank
 * workspace of this mode carries the fp32 operand image beside the bf16 one for that), which bounds the cost of any input at
 * about the fp32 path's; 0: the usual case. */
int manet_global_match_refine(const void *query, int emb_dtype, int64_t q_stride_n, int64_t q_stride_c,
                              const void *query_image, const void *bank_ws, int64_t N, int64_t M0, int C, int n_ids,
                              float *out, float *mem_inout, int epilogue_flags, void *match_ws, size_t match_ws_bytes,
                              manet_stream_t stream);
int manet_global_match_refine_stats(const void *match_ws, int64_t N, int C, int n_ids, int64_t *candidates,
                                    int64_t *list_overflowed);
/* ... and in full (blocking copies; benchmarks): stats4[0] = candidate rows (as above), [1] = the flag above, [2] = 256-query tiles
 * that went through the rescue pass, [3] = 256-query tiles of the frame -- [2] / [3] is the share of the frame that cost the
 * fp32 kernel's time ON TOP of the filter pass (distribution-dependent: 0 on embeddings the bf16 pass can tell apart). */
int manet_global_match_refine_stats2(const void *match_ws, int64_t N, int C, int n_ids, int64_t *stats4);
/* ... and without blocking: a one-workgroup launch on `stream` writes {tiles rescued, tiles} of the last filter pass on `match_ws`
 * to two int32 of device-visible memory (device memory, or pinned host memory: then nothing needs copying -- record an event
 * behind the call and read the two values once it has completed: ops.PreparedBank's adaptive policy). */
int manet_global_match_refine_rescued_async(const void *match_ws, int64_t N, int C, int n_ids, int32_t *out2_device,
                                            manet_stream_t stream);

/* Stand-alone normalise / min-merge (IntVOS.py:611-622, :718-723), in place on x[n]
 * (and on mem_inout[n] when not NULL). */
int manet_normalize_merge_f32(float *x, float *mem_inout, int64_t n, int normalize,
                              manet_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Local (2d+1)^2 window matching against the previous frame.
 *
 * Embeddings are [h][w][C] fp32 with element strides (s_y, s_x, s_c); the reference's callers
 * pass permute(1,2,0) views of C-major storage, i.e. (w, 1, h*w).
 *
 * manet_local_dist_f32 (IntVOS.py:266-315, local_pairwise_distances2(x=cur, y=prev)):
 *   downsample != 0: 2x2 average pooling, window on the pooled grid, padding distance inf,
 *                    (sigmoid-0.5)*2, bilinear (align_corners) resize back to (h, w);
 *   downsample == 0: window on the full grid, raw distances (inf outside the image).
 *   out [h][w][(2d+1)^2] fp32 contiguous.
 *
 * manet_local_match_f32 (IntVOS.py:345-434): the same distances, never materialised at full
 *   resolution, masked by the previous frame's labels gathered at stride-2 offsets
 *   (labels[y+2(by-d)][x+2(bx-d)], 0 outside the image) and reduced by min per object id;
 *   unmatched -> 1.0.   labels [h][w] int32 contiguous; out [h][w][n_ids] fp32 contiguous.
 */
int manet_local_workspace_bytes(int h, int w, int C, int max_distance, int downsample,
                                size_t *bytes);

int manet_local_dist_f32(const float *cur, int64_t c_sy, int64_t c_sx, int64_t c_sc,
                         const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc, int h, int w,
                         int C, int max_distance, int downsample, float *out, void *workspace,
                         size_t workspace_bytes, manet_stream_t stream);

int manet_local_match_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc,
                          const float *cur, int64_t c_sy, int64_t c_sx, int64_t c_sc,
                          const int32_t *prev_labels, int h, int w, int C, int n_ids,
                          int max_distance, int downsample, float *out, void *workspace,
                          size_t workspace_bytes, manet_stream_t stream);

/* manet_local_match_f32 for embeddings in the producer's storage (MANET_EMB_F32 / MANET_EMB_BF16; bf16 only in the
 * downsample configuration): the pooling pass reads the 2-byte embeddings, everything after it is fp32 --
 * i.e. exactly manet_local_match_f32 on the bf16-rounded embeddings. */
int manet_local_match_ex(const void *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc, const void *cur,
                         int64_t c_sy, int64_t c_sx, int64_t c_sc, int emb_dtype,
                         const int32_t *prev_labels, int h, int w, int C, int n_ids, int max_distance,
                         int downsample, float *out, void *workspace, size_t workspace_bytes,
                         manet_stream_t stream);

/* manet_local_match_f32 (downsample configuration) on two PREPARED frames (manet_frame_prepare with the same h, w, C,
 * compute, max_distance): one launch, the fused window / min kernel reading the two pooled planes -- no pooling pass, no
 * full-resolution read.  For max_distance >= 11 the kernel combines partial minima in `out` by atomicMin and needs it
 * pre-set to 1.0f: out_is_preset != 0 says the caller did that (manet_frame_prepare's fill), else a fill launch is
 * enqueued first.  Bit-identical to manet_local_match_ex on the same embeddings. */
int manet_local_match_frames(const void *prev_frame_ws, const void *cur_frame_ws, const int32_t *prev_labels,
                             int h, int w, int C, int compute, int n_ids, int max_distance, float *out,
                             int out_is_preset, manet_stream_t stream);

/* The label-INDEPENDENT half of the local match, kept per frame pair (r6).  The window distances of
 * local_pairwise_distances2 (IntVOS.py:266-296: pooled frames -> (2d+1)^2 squared distances -> (sigmoid - 0.5) * 2) depend on
 * the two embeddings only, and a clip's embeddings are extracted once per sequence (test.py:137-154) while every interaction round
 * walks the clip again (test.py:237-259, :276-295): the volume of a frame pair is computed once -- many pairs per launch
 * (manet_local_volume_frames: 32 pairs per launch, several waves of workgroups) -- and the per-frame sequential chain then runs
 * only the label-dependent tail (IntVOS.py:398-432: stride-2 label unfold, where(mask, d, 1), min over the window:
 * manet_local_match_volume).  A volume is stored as the fused kernel's per-workgroup LDS images (tile aprons and 16-byte cell padding
 * included: manet_local_volume_bytes, 25.8 MB per 480p frame pair at d = 12).  Same arithmetic as manet_local_match_frames: the same bits.
 *   prev_frame_ws / cur_frame_ws / volumes   HOST arrays of n_pairs DEVICE pointers (prepared frames of one geometry; volumes 16-byte
 *                                            aligned, manet_local_volume_bytes each)
 *   manet_local_match_volume                 `cur_frame_ws` = the current frame's prepared workspace (its tile table); `out` as in
 *                                            manet_local_match_frames (out_is_preset for max_distance >= 11). */
int manet_local_volume_bytes(int h, int w, int max_distance, size_t *bytes);
int manet_local_volume_frames(const void *const *prev_frame_ws, const void *const *cur_frame_ws, float *const *volumes,
                              int n_pairs, int h, int w, int C, int compute, int max_distance, manet_stream_t stream);
int manet_local_match_volume(const float *volume, const void *cur_frame_ws, const int32_t *prev_labels, int h, int w, int C,
                             int compute, int n_ids, int max_distance, float *out, int out_is_preset, manet_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* correlation_package forward (correlation_cuda.cc:10-87).
 *   in1, in2 [B][C][H][W] fp32 contiguous; out [B][(2r+1)^2][outH][outW] fp32 contiguous with
 *   r = max_displacement / stride2 and the shape rule of correlation_cuda.cc:25-34, which
 *   manet_correlation_out_dims reproduces.  out = sum / (kernel_size^2 * C), fp32 accumulate.
 *   No scratch is needed (the reference's padded NHWC copies rInput1/2 are not materialised). */
int manet_correlation_out_dims(int H, int W, int pad_size, int kernel_size, int max_displacement,
                               int stride1, int stride2, int *out_c, int *out_h, int *out_w);

int manet_correlation_forward_f32(const float *in1, const float *in2, int B, int C, int H, int W,
                                  int pad_size, int kernel_size, int max_displacement, int stride1,
                                  int stride2, float *out, manet_stream_t stream);
/* the same for the three tensor types the reference's forward dispatches (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * correlation_cuda_kernel.cu:386-415): in1 / in2 / out all of `dtype`; product in that type, fp32 accumulate,
 * mean stored in that type (correlation_cuda_kernel.cu:121-143). */
#define MANET_CORR_F32 0
#define MANET_CORR_F16 1 /* IEEE half */
#define MANET_CORR_F64 2
int manet_correlation_forward(const void *in1, const void *in2, int dtype, int B, int C, int H, int W,
                              int pad_size, int kernel_size, int max_displacement, int stride1,
                              int stride2, void *out, manet_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Mask step between two propagated frames (the driver side of the path):
 *   test.py:253-255   F.interpolate(logits, (H,W), 'bilinear', align_corners=True) -> argmax(dim=1)
 *   IntVOS.py:598-599 F.interpolate(prev_mask.float(), (h,w), 'nearest').int()   (next frame's input)
 * logits [n_ids][h][w] fp32 contiguous -> mask_hw [H][W] int64 (may be NULL) and label_small_hw
 * [h][w] int32 (may be NULL), in one launch, without the [n_ids][H][W] intermediate. */
int manet_upsample_argmax(const float *logits, int n_ids, int h, int w, int H, int W, int64_t *mask_hw,
                          int32_t *label_small_hw, manet_stream_t stream);

/* The two glue steps on either side of the matches in a propagated frame, one launch each:
 *   manet_label_resize_nearest: F.interpolate(previous_frame_mask.float(), size=(h, w), mode='nearest').int()
 *     (networks/IntVOS.py:598-599) on the int64 [H][W] mask -> int32 [h][w] (aten's nearest source index:
 *     min(floor(dst * (in / out)), in - 1), the ratio in float);
 *   manet_head_inputs_f32: the per-object channels of the head's input (networks/IntVOS.py:663-669, the three tensors
 *     torch.cat joins behind the repeated embedding): out [n_ids][3][HW] = (global_map[p][o], local_map[p][o], labels[p] == o)
 *     from global_map / local_map [HW][n_ids] fp32 and labels [HW] int32. */
int manet_label_resize_nearest(const int64_t *mask_hw, int H, int W, int h, int w, int32_t *label_small_hw,
                               manet_stream_t stream);
/* r5: manet_label_resize_nearest plus the two small device writes a propagated frame makes before its matches, in the same launch
 * (each was a ~5 us launch of its own): `fill` [fill_n] = fill_value (the local map's slot pre-set to 1.0 for the wide windows
 * whose workgroups meet by atomic min, networks/IntVOS.py:627-634 / manet_local_match_frames' `out_is_preset`) and
 * *scalar_dst = scalar_value (the frame's distance weight in the caller's table, networks/IntVOS.py:641).  fill_n = 0 and
 * scalar_dst = NULL leave either out. */
int manet_frame_begin(const int64_t *mask_hw, int H, int W, int h, int w, int32_t *label_small_hw, float *fill, int64_t fill_n,
                      float fill_value, float *scalar_dst, float scalar_value, manet_stream_t stream);
/* r5: DynamicSegHead layer 1, per-object half, in ONE launch (networks/IntVOS.py:663-669 input assembly -> :491-494 depthwise 7x7 +
 * bn1 + relu1 and the 1x1 + bn2 of the three per-object channels -> + `term`, the shared-embedding half's [256][HW] contribution ->
 * relu2): global_map / local_map [HW][n_ids], labels [HW] int32; dw_weight [3][49], dw_bias / bn_scale / bn_shift [3] (NULL: 0 / 1 / 0);
 * w2t_object [3][256], b2 [256] (bn2 folded) -> out [n_ids][256][HW].  The same bits as manet_head_inputs_f32 +
 * manet_dwconv7x7_bn_relu_f32 + manet_conv1x1_add_f32 on those channels. */
int manet_head_layer1_object_f32(const float *global_map, const float *local_map, const int32_t *labels, int h, int w, int n_ids,
                                 const float *dw_weight, const float *dw_bias, const float *bn_scale, const float *bn_shift,
                                 const float *w2t_object, const float *b2, const float *term, int relu_out, float *out,
                                 manet_stream_t stream);
int manet_head_inputs_f32(const float *global_map, const float *local_map, const int32_t *labels, int64_t HW, int n_ids,
                          float *out, manet_stream_t stream);

/* Depthwise 7x7 convolution (padding 3, one filter per channel) + bias + BatchNorm(eval) + ReLU, fused:
 * the first half of the reference's _split_separable_conv2d (IntVOS.py:491-493,500-502), the building block
 * of DynamicSegHead.  in/out [B][C][h][w] fp32 contiguous, weight [C][7][7],
 * bias / bn_scale / bn_shift [C] or NULL (0 / 1 / 0); bn_scale = gamma / sqrt(var + eps),
 * bn_shift = beta - mean * bn_scale.  out = relu?((conv + bias) * bn_scale + bn_shift). */
int manet_dwconv7x7_bn_relu_f32(const float *in, int B, int C, int h, int w, const float *weight,
                                const float *bias, const float *bn_scale, const float *bn_shift, int relu,
                                float *out, manet_stream_t stream);
/* The same with relu_in: when non-zero the input is read through max(x, 0) -- relu2 of the PRECEDING
 * _split_separable_conv2d (networks/IntVOS.py:503-505) folded into this pass, so the caller can leave that block's
 * 1x1 convolution output un-rectified and save one elementwise pass over the activation. */
int manet_dwconv7x7_bn_relu_ex(const float *in, int B, int C, int h, int w, const float *weight,
                               const float *bias, const float *bn_scale, const float *bn_shift, int relu,
                               int relu_in, float *out, manet_stream_t stream);

/* DynamicSegHead's output layer, fused (networks/IntVOS.py:519,525: Conv2d(embed_dim, 1, kernel 1) on layer4's ReLU
 * output): out[b][p] = bias[0] + sum_c weight[c] * (relu_in ? max(in[b][c][p], 0) : in[b][c][p]).
 * in [B][C][HW] fp32 contiguous, weight [C], bias [1] or NULL, out [B][HW]. */
int manet_relu_conv1x1_c1_f32(const float *in, int B, int C, long HW, const float *weight, const float *bias,
                              int relu_in, float *out, manet_stream_t stream);

/* The 1x1 convolution of a _split_separable_conv2d block (networks/IntVOS.py:494,503-505: conv2 -> bn2 [-> relu2]) as an
 * fp32-MFMA contraction, operands fed by LDS-DMA:
 *     out[b][co][p] = b2[co] + sum_ci w2t[ci][co] * in[b][ci][p]      [max(., 0) if relu_out]
 *   in   [B][Cin][HW] fp32, batch stride in_batch_stride elements (0 = one tensor for every batch item; a multiple of 4),
 *        16-byte aligned; any Cin >= 1, HW a multiple of 4
 *   w2t  [Cin][Cout] fp32, 16-byte aligned: the 1x1 weight TRANSPOSED with eval-mode bn2 folded in
 *        (w2t[ci][co] = conv2.weight[co][ci] * bn2_scale[co]);  b2 [Cout] = conv2.bias * bn2_scale + bn2_shift;  Cout = 256
 *   out  [B][Cout][HW] fp32 contiguous.
 * The sum is the ascending-channel fp32 fmaf chain of v_mfma_f32_32x32x2_f32. */
int manet_conv1x1_f32(const float *in, int64_t in_batch_stride, int B, int Cin, int64_t HW, const float *w2t,
                      const float *b2, int Cout, int relu_out, float *out, manet_stream_t stream);
/* The same with DynamicSegHead's output layer (networks/IntVOS.py:519,525: Conv2d(256, 1, 1) on layer4's ReLU output) fused
 * into the epilogue when head_w != NULL:  head_out[b][p] = head_b[0] + sum_co head_w[co] * max(out[b][co][p], 0);
 * `out` is then NOT written (layer4's [B,256,h,w] activation never leaves the compute units); head_w [Cout], head_b [1] or
 * NULL, head_out [B][HW].  head_w == NULL: exactly manet_conv1x1_f32. */
int manet_conv1x1_head_f32(const float *in, int64_t in_batch_stride, int B, int Cin, int64_t HW, const float *w2t,
                           const float *b2, int Cout, int relu_out, float *out, const float *head_w,
                           const float *head_b, float *head_out, manet_stream_t stream);
/* ... or with add [Cout][HW] fp32 (NULL = none) added to every batch entry's output before relu_out: layer1 of the
 * shared-embedding form (the embedding half of the contraction is computed once per frame and added here, instead of a
 * broadcast-add pass over the [n_objects,256,h,w] activation). */
int manet_conv1x1_add_f32(const float *in, int64_t in_batch_stride, int B, int Cin, int64_t HW, const float *w2t,
                          const float *b2, const float *add, int Cout, int relu_out, float *out, manet_stream_t stream);

/* The same layer in SPLIT-bf16 arithmetic: each fp32 factor = hi + lo (two bf16 pieces, 16 significand bits), a product =
 * hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (<= 2^-16 relative per product; the reference's
 * cuDNN path is TF32 -- 10 bits -- under torch's default allow_tf32).  3/16 of the fp32 matrix pipe's time: the layer
 * becomes a stream over its activation.
 *   manet_conv1x1_x3_weight_bytes(Cin) / manet_conv1x1_x3_pack: w2t [Cin][256] fp32 (as for manet_conv1x1_f32) -> the packed
 *     MFMA A-operand image `wpk` (hi and lo), once per fold of the layer's constants; wpk 16-byte aligned.
 *   manet_conv1x1_x3_f32: in as above but ANY Cin >= 1 (HW a multiple of 4); add = NULL or [256][HW] fp32 added to every batch
 *     entry's output before relu_out (layer1 of the shared-embedding form: the embedding half of the contraction, computed
 *     once per frame); head_w / head_b / head_out as manet_conv1x1_head_f32 (exclusive with add). */
/* ... and with THREE pieces per factor ("split3": hi + mid + lo = 24 significand bits, six products per pair -- hi*hi + hi*mid +
 * mid*hi + mid*mid + hi*lo + lo*hi, the dropped terms below 2^-24 of a product): fp32-class results (not the fmaf chain's bits:
 * the sum is taken in another order) at 6/16 of the fp32 matrix pipe's time.  manet_conv1x1_x6_weight_bytes / _pack / _f32: as
 * the _x3_ trio, 24 KiB of packed weights per 16 input channels. */
int64_t manet_conv1x1_x6_weight_bytes(int Cin);
int manet_conv1x1_x6_pack(const float *w2t, int Cin, int Cout, void *wpk, manet_stream_t stream);
int manet_conv1x1_x6_f32(const float *in, int64_t in_batch_stride, int B, int Cin, int64_t HW, const void *wpk, const float *b2,
                         const float *add, int Cout, int relu_out, float *out, const float *head_w, const float *head_b,
                         float *head_out, manet_stream_t stream);
int64_t manet_conv1x1_x3_weight_bytes(int Cin);
int manet_conv1x1_x3_pack(const float *w2t, int Cin, int Cout, void *wpk, manet_stream_t stream);
int manet_conv1x1_x3_f32(const float *in, int64_t in_batch_stride, int B, int Cin, int64_t HW, const void *wpk, const float *b2,
                         const float *add, int Cout, int relu_out, float *out, const float *head_w, const float *head_b,
                         float *head_out, manet_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Training path: what torch.autograd does for the reference's pure-PyTorch path
 * (train_stage1.py:126-156 back-propagates through IntVOS.forward) and what
 * correlation_cuda.backward does for the native op (correlation_cuda.cc:89-167,
 * correlation_cuda_kernel.cu:150-334).  The Python side wraps these in torch.autograd.Function
 * (cvpr2020_manet_amd/autograd.py).  fp32, k_nearest_neighbors = 1, downsample configuration.
 *
 * manet_global_match_arg_f32: manet_global_match (MANET_COMPUTE_F32, k_nn = 1, no epilogue; `out` is
 *   bit-identical to it) that also returns arg_out [N][n_ids] int32 = the row of the caller's bank
 *   whose distance is the minimum (the element torch.min differentiates through, IntVOS.py:84);
 *   equal distances resolve to the first such row of the object-sorted bank, -1 if the object has no row.
 * manet_global_match_backward_f32: with m* = arg[n][o], g = grad_out[n][o]:
 *   grad_query[n] = sum_o 2 g (q_n - k_m*),  grad_bank[m] = sum over (n,o) with m* = m of 2 g (k_m - q_n)
 *   (d = |q|^2 + |k|^2 - 2 q.k, IntVOS.py:32-39).  Both gradient tensors take element strides and are
 *   fully overwritten (grad_bank is zeroed, then accumulated with atomicAdd: summation order is not
 *   deterministic, like torch's own scatter-add backward).  Either of grad_query / grad_bank may be NULL: that
 *   gradient is not computed (a frozen reference frame needs no zero-fill and no scatter). */
int manet_global_match_arg_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, size_t *bytes);
int manet_global_match_arg_f32(const float *query, int64_t q_stride_n, int64_t q_stride_c,
                               const float *bank, int64_t b_stride_m, int64_t b_stride_c,
                               const int32_t *labels, int64_t N, int64_t M0, int C, int n_ids,
                               float *out, int32_t *arg_out, void *workspace, size_t workspace_bytes,
                               manet_stream_t stream);
int manet_global_match_backward_f32(const float *query, int64_t q_stride_n, int64_t q_stride_c,
                                    const float *bank, int64_t b_stride_m, int64_t b_stride_c,
                                    const int32_t *arg, const float *grad_out, int64_t N, int64_t M0,
                                    int C, int n_ids, float *grad_query, int64_t gq_stride_n,
                                    int64_t gq_stride_c, float *grad_bank, int64_t gb_stride_m,
                                    int64_t gb_stride_c, manet_stream_t stream);

/* Training with k_nearest_neighbors > 1 (r5; IntVOS.py:87-94: topk(-d, k) per (query, object), entries past the object's row
 * count (>= 1e20) replaced by the farthest real neighbour, mean -- autograd sends 1/k of the gradient to each selected row and
 * the replaced entries' share to the farthest real one).  manet_global_match_topk_arg_f32 returns the k_nn smallest distances
 * per (query, object) in ascending order WITH their rows: out / arg_out [k_nn][N][n_ids]; beyond an object's row count the
 * distance is 1e20 and the row -1.  k_nn passes of the arg-min kernel, pass j bounded from below by pass j - 1's (distance,
 * bank slot) pair: exact; equal distances are ordered by their position in the object-sorted bank.  The backward is k_nn calls
 * of manet_global_match_backward_f32 (one per rank j, with that rank's rows and weights; cvpr2020_manet_amd/autograd.py). */
int manet_global_match_topk_arg_workspace_bytes(int64_t N, int64_t M0, int C, int n_ids, size_t *bytes);
int manet_global_match_topk_arg_f32(const float *query, int64_t q_stride_n, int64_t q_stride_c,
                                    const float *bank, int64_t b_stride_m, int64_t b_stride_c,
                                    const int32_t *labels, int64_t N, int64_t M0, int C, int n_ids, int k_nn,
                                    float *out, int32_t *arg_out, void *workspace, size_t workspace_bytes,
                                    manet_stream_t stream);

/* manet_local_match_arg_f32: manet_local_match_f32 (downsample on) that also returns
 *   arg_out [h][w][n_ids] int32 = the window offset l = dy*(2d+1)+dx whose masked value is the minimum
 *   (first offset on ties), -1 when the constant 1.0 of IntVOS.py:429-430 wins, and keeps
 *   vol_out [(2d+1)^2][h/2][w/2] = the normalised pooled distance volume for the backward.
 * manet_local_match_backward_f32: gradient of `out` w.r.t. both embeddings through
 *   min -> where -> bilinear(align_corners) -> (sigmoid-0.5)*2 -> sum_c (x - y_off)^2 -> avg_pool2d
 *   (IntVOS.py:266-296, :398-432).  grad_prev / grad_cur: [h][w][C] with element strides, fully
 *   overwritten. */
int manet_local_match_arg_workspace_bytes(int h, int w, int C, int max_distance, size_t *bytes);
int manet_local_match_arg_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc,
                              const float *cur, int64_t c_sy, int64_t c_sx, int64_t c_sc,
                              const int32_t *prev_labels, int h, int w, int C, int n_ids,
                              int max_distance, float *out, int32_t *arg_out, float *vol_out,
                              void *workspace, size_t workspace_bytes, manet_stream_t stream);
int manet_local_match_backward_workspace_bytes(int h, int w, int C, int max_distance, size_t *bytes);
int manet_local_match_backward_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc,
                                   const float *cur, int64_t c_sy, int64_t c_sx, int64_t c_sc,
                                   const float *vol, const int32_t *arg, const float *grad_out, int h,
                                   int w, int C, int n_ids, int max_distance, float *grad_prev,
                                   int64_t gp_sy, int64_t gp_sx, int64_t gp_sc, float *grad_cur,
                                   int64_t gc_sy, int64_t gc_sx, int64_t gc_sc, void *workspace,
                                   size_t workspace_bytes, manet_stream_t stream);

/* The same for MODEL_LOCAL_DOWNSAMPLE = False (r5; IntVOS.py:299-313 raw full-resolution distances, :398-432 labels gathered
 * at stride 2 and the constant 1.0 -- the reference's own combination, kept).  manet_local_match_full_arg_f32: out [h][w][n_ids]
 * (= manet_local_match_f32 with downsample 0), arg_out = the winning window offset (-1: the constant won), vol_out
 * [h][w][(2d+1)^2] the caller's scratch for the volume.  manet_local_match_full_backward_f32: embeddings and gradients as
 * CONTIGUOUS [C][h][w] planes (fully overwritten); dv_ws [(2d+1)^2][h*w] floats of scratch. */
int manet_local_match_full_arg_f32(const float *prev, int64_t p_sy, int64_t p_sx, int64_t p_sc, const float *cur,
                                   int64_t c_sy, int64_t c_sx, int64_t c_sc, const int32_t *prev_labels, int h, int w,
                                   int C, int n_ids, int max_distance, float *out, int32_t *arg_out, float *vol_out,
                                   manet_stream_t stream);
int manet_local_match_full_backward_f32(const float *prev_chw, const float *cur_chw, const int32_t *arg,
                                        const float *grad_out, int h, int w, int C, int n_ids, int max_distance,
                                        float *grad_prev_chw, float *grad_cur_chw, float *dv_ws, manet_stream_t stream);

/* correlation_package backward (correlation_cuda.cc:89-167): gradients w.r.t. both inputs,
 * [B][C][H][W] fp32 contiguous, fully overwritten. */
int manet_correlation_backward_f32(const float *in1, const float *in2, const float *grad_out, int B,
                                   int C, int H, int W, int pad_size, int kernel_size,
                                   int max_displacement, int stride1, int stride2, float *grad_in1,
                                   float *grad_in2, manet_stream_t stream);
/* ... and for double tensors (the reference dispatches float, double and half backwards, correlation_cuda_kernel.cu:495-541;
 * half gradients are this library's fp32 kernel on widened inputs, rounded once by the caller). */
int manet_correlation_backward_f64(const double *in1, const double *in2, const double *grad_out, int B,
                                   int C, int H, int W, int pad_size, int kernel_size,
                                   int max_displacement, int stride1, int stride2, double *grad_in1,
                                   double *grad_in2, manet_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Opt-in measurement hook (not part of the data path, used by bench.py): between _begin and _end
 * every launch of the dominant kernel (the global-match MFMA kernel) is bracketed by two HIP
 * events on its own stream.  manet_profile_end synchronises on those events (the only call in
 * this library that blocks) and returns the per-launch durations in milliseconds. */
int manet_profile_begin(int max_launches);
/* Tuning knobs for experiments (process-wide; the defaults are the shipped configuration).  Refused
 * (MANET_E_INVALID) unless the environment has MANET_TUNING=1: without that opt-in nothing can change them, and the
 * data path keeps no state between calls.
 * value INT32_MIN puts a key back to "not set".
 * key 0 = block -> (query tile, bank split) mapping of the global-match kernels (not set: XCD-aware with as many splits
 *         fastest as fit an L2 side by side; 0 XCD-aware tile-fastest, 1 tile fastest, 2 split fastest, 4..7 XCD-aware with
 *         2..5 splits fastest),
 * key 1 = forced number of bank splits (0 = automatic),
 * key 2 = form of the bf16 kernels (bit field, see launch_main_bf16 in csrc/global_match.hip),
 * key 3 = timing ablations (results are garbage; -DMANET_ABLATION builds only; for MANET_COMPUTE_BF16_REFINE's filter pass:
 *         16 no listing path, 32 no threshold exchange, 64 no bucket reservations, 64 + 128 no per-register listing, 256 cycle
 *         counters of the listing path printed after each filter launch),
 * key 4 = 1: the r1 three-launch local match, key 5 = un-pipelined fp32 kernel, key 6 = frame-prepare channel block,
 * key 7 = pre-pass sampling of MANET_COMPUTE_BF16_REFINE (every value-th bank tile), key 8 = 1: the LDS-weights fp32 1x1
 * kernel everywhere (2: the resident-weights kernel with 32-channel stages, 3 / 4: its pixel ranges in whole tiles / half-tile
 * units whatever the launch size), key 9 = bank splits per listed tile of MANET_COMPUTE_BF16_REFINE's rescue launch, key 10 = 1: the fp32 kernel
 * keeps the host's split count whatever the bank's real size (see split_of_block in csrc/global_match.hip).
 * Keys 2, 3, 5, 6 select kernels that only -DMANET_ABLATION builds contain and are refused otherwise.
 * No knob changes a workspace layout. */
int manet_tune_set(int key, int value);
int manet_profile_end(float *ms_out, int capacity, int *n_launches);
/* the same, also returning the durations of the local-window stage (pooling pass + fused kernel of
 * manet_local_match_*), the HBM-bound stage of the path */
int manet_profile_end2(float *ms_out, int capacity, int *n_launches, float *local_ms_out,
                       int local_capacity, int *n_local);
/* durations recorded so far on one channel, without ending the session (call before manet_profile_end*; blocks on the
 * recorded events): 0 = global-match main kernel, 1 = local-window stage, 2 = manet_frame_prepare, 3 = the exact
 * re-rank kernel of MANET_COMPUTE_BF16_REFINE */
int manet_profile_read(int channel, float *ms_out, int capacity, int *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* MANET_HIP_H */
