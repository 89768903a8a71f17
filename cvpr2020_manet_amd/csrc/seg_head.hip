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
// r2 kernel.  The op is 49 FMAs per output on a pure stream (2 x 4 B per element): at [3,256,120,214] that is
// 158 MB (20 us at 8 TB/s) and 15 M scalar wave-FMAs (~28 us at the measured v_fma_f32 rate) -- the r1 kernel (one
// output row x 4 columns per thread, scalar FMAs, 16 x 64 tiles) took 58 us.  Here:
//  * a thread owns 2 rows x 8 columns; its running sums are VERTICAL pairs (out[y][x], out[y+1][x]), so every tap is one
//    v_pk_fma_f32 with the tap weight broadcast: half the VALU instructions (a wave issues one VALU instruction per
//    ~2.5 ns whatever its width: tools/ubench/valu_rate.hip);
//  * the packed operand must be an aligned register pair holding (in[r][x], in[r+1][x]): the input tile is kept in LDS
//    twice, rows interleaved in pairs starting at even rows (E) and at odd rows (O); a kernel row ky reads pair
//    (2t + ky, 2t + ky + 1) from E or O by parity, 14 columns = 7 ds_read_b128 for 56 packed FMAs;
//  * workgroup = 60 x 64 outputs (halo amplification 1.2x instead of 1.5x), 256 threads, 39 KB LDS -> 4 per CU;
//  * `relu_in`: max(x, 0) applied while staging, so the caller can drop the separate ReLU pass after the preceding
//    1x1 convolution (IntVOS.py:503-505: relu2 of the previous block) -- one HBM round trip less per block.
// Tap order (ky outer, kx inner, fmaf chain per output) is the r1 kernel's and the oracle's: bit-identical.
#include "manet_common.h"

namespace {

constexpr int DW_K = 7, DW_R = 3;
constexpr int DW_TY = 60, DW_TX = 64;      // 60 rows: 480p's 120 and 720p's 180 grid rows tile exactly; 33 row pairs of LDS
// LDS columns: image columns x0-4 .. x0+67 (even start: aligned float2 loads) = 72, padded to 74: the row-pair stride
// 148 floats = 20 (mod 64 banks) puts the four (row pair, column group) quads of a ds_read_b128 lane group on disjoint
// banks (stride 144: 4-way conflicts on every window read)
constexpr int DW_LW = 74;
constexpr int DW_NP = (DW_TY + 2 * DW_R) / 2;  // 35 row pairs per image (rows y0-3 .. y0+66)
constexpr int DW_IMG = DW_NP * DW_LW * 2;    // floats per interleaved image

// FAST: w even (every aligned column pair is inside or outside the image as a whole, rows are 8-byte aligned)
template <bool FAST>
__global__ __launch_bounds__(256, 4) void dwconv7x7_bn_relu_kernel(const float *__restrict__ in, int C, int h, int w,
                                                                   const float *__restrict__ weight,
                                                                   const float *__restrict__ bias,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift, int relu,
                                                                   int relu_in, float *__restrict__ out)
{
    // E: element (r, col) at ((r >> 1) * LW + col) * 2 + (r & 1), r = row - (y0 - 3); O: the same for r - 1
    __shared__ __attribute__((aligned(16))) float tile[2 * DW_IMG];
    const int plane_id = blockIdx.z;  // b * C + c
    const int c = plane_id % C;
    const int x0 = blockIdx.x * DW_TX, y0 = blockIdx.y * DW_TY;
    const float *src = in + (long)plane_id * h * w;
    const int tid = threadIdx.x;
    // staging item = (row pair p, column pair q): rows 2p, 2p+1, 2p+2 of the tile, two columns -- one b128 store into E
    // (rows 2p, 2p+1) and one into O (rows 2p+1, 2p+2).  Branch-free: clamped addresses, values selected afterwards.
    constexpr int NQ = 36, NITEM = DW_NP * NQ, KI = (NITEM + 255) / 256;  // 36 column pairs = image columns x0-4 .. x0+67
    // pass 1: every load of the thread is issued (clamped addresses, no predicate anywhere near them); pass 2 selects,
    // rectifies and stores.  Written as one loop, hipcc sinks each load into the branch of its select and waits
    // vmcnt(0) there: 15 serial L2 round trips per thread (r2 trace: this, not the 392 FMAs, was the kernel's time).
    f32x2 ld[KI][3];
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int i = tid + 256 * k;
        const int p = i / NQ, q = i - p * NQ;
        const int xx = x0 - 4 + 2 * q;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int yc = min(max(y0 - DW_R + 2 * p + e, 0), h - 1);
            // (unsigned element offsets: the loads take the scalar-base + 32-bit-offset form, no 64-bit address VALU)
            if (FAST) {
                ld[k][e] = *(const f32x2 *)(src + (unsigned)(yc * w + min(max(xx, 0), w - 2)));
            } else {
                ld[k][e][0] = src[(unsigned)(yc * w + min(max(xx, 0), w - 1))];
                ld[k][e][1] = src[(unsigned)(yc * w + min(max(xx + 1, 0), w - 1))];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) asm volatile("" : "+v"(ld[k][0]), "+v"(ld[k][1]), "+v"(ld[k][2]));
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int i = tid + 256 * k;
        const int p = i / NQ, q = i - p * NQ;
        const int xx = x0 - 4 + 2 * q;
        const bool iok = i < NITEM;
        f32x2 rv[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int yy = y0 - DW_R + 2 * p + e;
            const bool yok = yy >= 0 && yy < h && iok;
            rv[e][0] = (yok && xx >= 0 && xx < w) ? ld[k][e][0] : 0.0f;
            rv[e][1] = (yok && xx + 1 >= 0 && xx + 1 < w) ? ld[k][e][1] : 0.0f;
            if (relu_in) rv[e] = __builtin_elementwise_max(rv[e], f32x2{0.0f, 0.0f});
        }
        if (iok) {
            float *d = tile + (p * DW_LW + 2 * q) * 2;
            *(f32x4 *)d = f32x4{rv[0][0], rv[1][0], rv[0][1], rv[1][1]};
            *(f32x4 *)(d + DW_IMG) = f32x4{rv[1][0], rv[2][0], rv[1][1], rv[2][1]};
        }
    }
    __syncthreads();
    const int t = tid >> 3, tg = tid & 7;  // output row pair (2t, 2t+1), group of 8 columns
    if (t >= DW_TY / 2) return;            // (no barrier below)
    const float *wk = weight + (long)c * DW_K * DW_K;
    f32x2 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int ky = 0; ky < DW_K; ++ky) {
        // input rows (2t + ky, 2t + ky + 1): pair t + ky/2 of E (ky even) or pair t + (ky-1)/2 of O (ky odd); output
        // column 8 tg + j, tap kx reads LDS column 8 tg + j + kx + 1 (LDS column 0 is image column x0 - 4)
        const float *row = tile + ((ky & 1) ? DW_IMG : 0) + ((t + (ky >> 1)) * DW_LW + 8 * tg) * 2;
        // (ties the reads of this kernel row behind the previous row's arithmetic: the fully unrolled loop otherwise
        // holds all the reads in flight and the register count halves the occupancy)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])::"memory");
        f32x2 win[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 u = *(const f32x4 *)(row + 4 * i);
            win[2 * i] = f32x2{u[0], u[1]};
            win[2 * i + 1] = f32x2{u[2], u[3]};
        }
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) {
            const float wv = wk[ky * DW_K + kx];
            const f32x2 w2 = {wv, wv};
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_elementwise_fma(win[kx + j + 1], w2, acc[j]);
        }
    }
    const float bc = bias ? bias[c] : 0.0f, sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
    const int x = x0 + 8 * tg;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int y = y0 + 2 * t + e;
        if (y >= h) continue;
        float *dst = out + (long)plane_id * h * w + (unsigned)(y * w + x);
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float o = fmaf(acc[j][e] + bc, sc, sh);
            r[j] = relu ? fmaxf(o, 0.0f) : o;
        }
        if (FAST && x + 7 < w) {  // w even, plane 8-byte aligned: float2 stores are always aligned
#pragma unroll
            for (int j = 0; j < 8; j += 2) *(f32x2 *)(dst + j) = f32x2{r[j], r[j + 1]};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (x + j < w) dst[j] = r[j];
        }
    }
}

// DynamicSegHead's output layer (IntVOS.py:519,525: conv = Conv2d(embed_dim, 1, 1) applied to layer4's ReLU output):
//   out[b][p] = bias + sum_c w[c] * max(in[b][c][p], 0)
// a pure stream over the [B][C][HW] activation (79 MB at [3,256,120,214]: 10 us at 8 TB/s).  The framework runs it as
// clamp (22 us) + two layout transposes (31 us) + an implicit-GEMM with N = 1 (19 us).  Here: a workgroup owns 256 pixels
// (64 lanes x float4), its 4 waves split the channels; 8 independent float4 loads in flight per lane; the waves' partial
// sums meet in LDS.  Per output the sum is ((w0 + w1) + (w2 + w3)) of four ascending fmaf chains.
constexpr int RC_WAVES = 4;
__global__ __launch_bounds__(64 * RC_WAVES) void relu_conv1x1_c1_kernel(const float *__restrict__ in, int C, long HW,
                                                                       const float *__restrict__ weight,
                                                                       const float *__restrict__ bias, int relu_in,
                                                                       float *__restrict__ out)
{
    __shared__ f32x4 part[RC_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const long p0 = ((long)blockIdx.x * 64 + lane) * 4;
    const bool vec = (HW % 4 == 0) && p0 + 3 < HW;  // rows of whole float4 (16-byte aligned planes)
    const float *src = in + (long)b * C * HW;
    const int cper = (C + RC_WAVES - 1) / RC_WAVES, c0 = wave * cper, c1 = (c0 + cper < C) ? c0 + cper : C;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (vec) {
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f32x4 *)(src + (long)(c + u) * HW + p0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float wv = weight[c + u];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(relu_in ? fmaxf(v[u][j], 0.0f) : v[u][j], wv, acc[j]);
            }
        }
        for (; c < c1; ++c) {
            const f32x4 v = *(const f32x4 *)(src + (long)c * HW + p0);
            const float wv = weight[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(relu_in ? fmaxf(v[j], 0.0f) : v[j], wv, acc[j]);
        }
    } else {
        for (int c = c0; c < c1; ++c) {
            const float wv = weight[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (p0 + j < HW) {
                    const float v = src[(long)c * HW + p0 + j];
                    acc[j] = fmaf(relu_in ? fmaxf(v, 0.0f) : v, wv, acc[j]);
                }
            }
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        const f32x4 s01 = part[0][lane] + part[1][lane], s23 = part[2][lane] + part[3][lane];
        const float bz = bias ? bias[0] : 0.0f;
        const f32x4 r = (s01 + s23) + f32x4{bz, bz, bz, bz};
        float *dst = out + (long)b * HW + p0;
        if (vec) *(f32x4 *)dst = r;
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < HW) dst[j] = r[j];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// One whole _split_separable_conv2d (IntVOS.py:488-506) in ONE launch (VERDICT r2 "next" #3):
//     y = relu(bn1(dwconv7x7(x)))          -- per channel, VALU
//     z = bn2(conv1x1(y)) [, relu]         -- a [256 x Cin] x [Cin x pixels] contraction, fp32 MFMA (exact fp32 chain)
// r2 ran them as two kernels with the [B,256,h,w] activation y (79 MB at 3 objects, 480p) written and re-read in between
// and the contraction in the framework's GEMM: 60 + 108 us per block, 45 % of an end-to-end frame.  Here y never leaves
// the CU: a workgroup owns a 4 x 16 pixel tile of one batch item and ALL 256 output channels, and is split by role --
//   waves 0-3  matrix waves: 64 output channels x 64 pixels each (2 x 2 blocks of v_mfma_f32_32x32x2_f32, 64 accumulator
//              VGPRs); per chunk of 16 input channels 8 k-steps x 4 MFMAs with A = folded 1x1 weights and B = y, both read
//              from LDS as one dword per lane (conflict-free);
//   waves 4-7  depthwise waves: per chunk each wave computes 4 channels x 64 pixels, a lane owns 4 neighbouring pixels of
//              one channel: per kernel row three LDS reads (10 window floats) feed 28 fmaf whose tap weights sit in the
//              lane's registers (the channel's 49 taps + bias + bn1 scale / shift, one padded 52-float row, prefetched a
//              chunk ahead); tap order (ky outer, kx inner) and bn1 expression of dwconv7x7_bn_relu_kernel, so y is
//              bit-identical to the two-kernel path.  They also stage the NEXT chunks' halo tiles (global -> registers ->
//              LDS, zero padding and the preceding block's deferred ReLU applied on the way);
// so the matrix pipe and the vector pipe of every SIMD run side by side (MI355X_MICROARCH.md: an MFMA-only and a
// VALU-only wave on one SIMD overlap), one barrier per chunk; per chunk the matrix waves need 32 x 64 = 2 048 cycles, the
// depthwise waves ~450 VALU / LDS instructions -- about the same time: both pipes are busy.  The 1x1 weights arrive by
// LDS-DMA from a pre-transposed [Cin_pad][256] copy (bn2 folded in), double buffered.  The input may come from TWO tensors
// (channels [0, Ca) from `in_a`, the rest from `in_b`, each with its own batch stride): layer 1 reads the C-channel
// embedding with batch stride 0 next to the per-object maps, i.e. IntVOS.py:665-670's repeat / cat is never built.
constexpr int SC_TY = 4, SC_TX = 16, SC_P = SC_TY * SC_TX, SC_KC = 16, SC_CO = 256;
constexpr int SC_IR = SC_TY + 2 * DW_R, SC_IC = SC_TX + 2 * DW_R;  // halo tile 10 x 22
constexpr int SC_IW = 24, SC_ICH = SC_IR * SC_IW;                  // LDS row stride / floats per channel
constexpr int SC_NMM = 4, SC_NDW = 4, SC_NT = 64 * (SC_NMM + SC_NDW);
constexpr int SC_NLD = (SC_KC * SC_IR * SC_IC + 64 * SC_NMM - 1) / (64 * SC_NMM);  // staged elements per matrix-wave thread
constexpr int SC_WROW = 64;  // floats per row of the padded depthwise parameter table: 49 taps, bias, bn1 scale, bn1 shift, 0..
constexpr int SC_WPIECES = SC_KC * SC_CO / 256, SC_PPIECES = SC_KC * SC_WROW / 256;  // 1 KiB LDS-DMA pieces per chunk
struct SepConv {
    const float *in_a, *in_b;
    long sa, sb;  // batch strides (elements); 0 = the same tensor for every batch item
    int Ca, Cin, nchunks, h, w;
    const float *dwp;  // [Cin_pad][SC_WROW]
    int relu_in;
    const float *w2t, *b2;
    int relu_out;
    float *out;
};
__global__ __launch_bounds__(SC_NT) void sepconv7x7_pw_kernel(const SepConv A)
{
    __shared__ __attribute__((aligned(1024))) float wbuf[2][SC_KC * SC_CO];    // 1x1 weights of a chunk, [k][co]
    __shared__ __attribute__((aligned(1024))) float pbuf[2][SC_KC * SC_WROW];  // depthwise parameters of a chunk
    __shared__ __attribute__((aligned(16))) float inbuf[2][SC_KC * SC_ICH];    // halo tiles of a chunk's 16 channels
    __shared__ __attribute__((aligned(16))) float dbuf[2][SC_KC * SC_P];       // y of a chunk, [k][pixel]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mm = wave < SC_NMM;
    const int x0 = blockIdx.x * SC_TX, y0 = blockIdx.y * SC_TY, b = blockIdx.z;
    const int h = A.h, w = A.w, n = A.nchunks;
    const long plane = (long)h * w;

    // ---- matrix waves, side job: staging of the halo tiles (their VALU is idle between MFMAs).  A thread's SC_NLD elements
    // keep their place for the whole kernel: pixel offset, validity and LDS slot are computed once, a chunk only moves the
    // channel base.
    int s_pix[SC_NLD], s_lds[SC_NLD];  // s_lds: LDS slot | channel << 20, < 0: no element
    unsigned s_ok = 0;
    if (mm) {
#pragma unroll
        for (int j = 0; j < SC_NLD; ++j) {
            const int e = tid + 64 * SC_NMM * j;
            const int ch = e / (SC_IR * SC_IC), rem = e - ch * (SC_IR * SC_IC);
            const int r = rem / SC_IC, col = rem - r * SC_IC;
            const int y = y0 - DW_R + r, x = x0 - DW_R + col;
            const bool have = e < SC_KC * SC_IR * SC_IC;
            if (have && y >= 0 && y < h && x >= 0 && x < w) s_ok |= 1u << j;
            s_pix[j] = min(max(y, 0), h - 1) * w + min(max(x, 0), w - 1);
            s_lds[j] = have ? (ch * SC_ICH + r * SC_IW + col) | (ch << 20) : -1;
        }
    }
    float stg[SC_NLD];
    auto stage_load1 = [&](int c, int j) __attribute__((always_inline)) {
        const int ch = (s_lds[j] >> 20) & 15;
        const int ci = c * SC_KC + ch;
        const int cc = ci < A.Cin ? ci : A.Cin - 1;
        const float *src = cc < A.Ca ? A.in_a + (long)b * A.sa + (long)cc * plane
                                     : A.in_b + (long)b * A.sb + (long)(cc - A.Ca) * plane;
        float v = src[(unsigned)s_pix[j]];  // unconditional load, selected afterwards
        v = (((s_ok >> j) & 1u) && ci < A.Cin) ? v : 0.0f;
        stg[j] = A.relu_in ? fmaxf(v, 0.0f) : v;
    };
    auto stage_load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SC_NLD; ++j) stage_load1(c, j);
    };
    auto stage_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SC_NLD; ++j)
            if (s_lds[j] >= 0) inbuf[buf][s_lds[j] & 0xfffff] = stg[j];
    };
    // the chunk's 1x1 weights [16][256] and depthwise parameters [16][64] by LDS-DMA
    const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&wbuf[0][0]);
    const unsigned pbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&pbuf[0][0]);
    auto w_dma = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SC_WPIECES / SC_NMM; ++i) {
            const int pc = wave * (SC_WPIECES / SC_NMM) + i;
            lds_dma16(A.w2t + (long)c * (SC_KC * SC_CO) + pc * 256 + lane * 4,
                      wbase + (unsigned)buf * (unsigned)(SC_KC * SC_CO * 4) + (unsigned)pc * 1024u);
        }
    };
    auto p_dma = [&](int c, int buf) __attribute__((always_inline)) {
        static_assert(SC_PPIECES == SC_NMM, "one parameter piece per matrix wave");
        lds_dma16(A.dwp + (long)c * (SC_KC * SC_WROW) + wave * 256 + lane * 4,
                  pbase + (unsigned)buf * (unsigned)(SC_KC * SC_WROW * 4) + (unsigned)wave * 1024u);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int co0 = wave * 64;
    // 8 k-steps x 4 MFMAs on chunk `buf`; between the k-steps the staging loads of chunk `cs` (if >= 0) are issued
    auto mma = [&](int buf, int cs) __attribute__((always_inline)) {
        const float *Wt = &wbuf[buf][(lane >> 5) * SC_CO + co0 + (lane & 31)];
        const float *D = &dbuf[buf][(lane >> 5) * SC_P + (lane & 31)];
#pragma unroll
        for (int kk = 0; kk < SC_KC / 2; ++kk) {
            const float a0 = Wt[2 * kk * SC_CO], a1 = Wt[2 * kk * SC_CO + 32];
            const float b0 = D[2 * kk * SC_P], b1 = D[2 * kk * SC_P + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            if (cs >= 0) {
#pragma unroll
                for (int j = 2 * kk; j < 2 * kk + 2; ++j)
                    if (j < SC_NLD) stage_load1(cs, j);
            }
        }
    };

    // ---- depthwise waves: lane -> (channel of the wave's four, row, group of 4 columns)
    const int chl = lane >> 4, py = (lane >> 2) & 3, pxg = (lane & 3) * 4;
    const int ch_dw = (wave - SC_NMM) * 4 + chl;  // channel inside the chunk (waves 4-7)
    auto dw_compute = [&](int buf) __attribute__((always_inline)) {
        const float *src = &inbuf[buf][ch_dw * SC_ICH + py * SC_IW + pxg];
        const float *wr = &pbuf[buf][ch_dw * SC_WROW];  // the 16 lanes of a channel read the same words: broadcast
        float a4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ky = 0; ky < DW_K; ++ky) {
            // (ties this row's reads behind the previous row's arithmetic: the unrolled loop otherwise keeps 70 window
            // registers in flight)
            asm volatile("" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]));
            const f32x4 u0 = *(const f32x4 *)(src + ky * SC_IW), u1 = *(const f32x4 *)(src + ky * SC_IW + 4);
            const f32x2 u2 = *(const f32x2 *)(src + ky * SC_IW + 8);
            const float v[10] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3], u2[0], u2[1]};
#pragma unroll
            for (int kx = 0; kx < DW_K; ++kx) {
                const float wt = wr[ky * DW_K + kx];
#pragma unroll
                for (int p = 0; p < 4; ++p) a4[p] = fmaf(v[kx + p], wt, a4[p]);
            }
        }
        const float bc = wr[49], sc = wr[50], sh = wr[51];
        f32x4 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = fmaxf(fmaf(a4[p] + bc, sc, sh), 0.0f);
        *(f32x4 *)&dbuf[buf][ch_dw * SC_P + py * SC_TX + pxg] = o;
    };

    // ---- schedule.  Iteration c: the matrix waves multiply chunk c (wbuf / dbuf [c & 1]) while the depthwise waves
    // produce y of chunk c + 1 (inbuf / pbuf [(c + 1) & 1] -> dbuf [(c + 1) & 1]); under their MFMAs the matrix waves fetch
    // what the NEXT iterations need into the buffers the previous iteration released: the 1x1 weights of chunk c + 1
    // (wbuf [(c + 1) & 1], last read in iteration c - 1), the depthwise parameters and the halo tile of chunk c + 2
    // (pbuf / inbuf [c & 1], last read in iteration c - 1).  One barrier per iteration.
    if (mm) {
        w_dma(0, 0);
        p_dma(0, 0);
        if (n > 1) p_dma(1, 1);
        stage_load(0);
        stage_store(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (mm) {
        if (n > 1) {
            stage_load(1);
            stage_store(1);
        }
    } else {
        dw_compute(0);
    }
    __syncthreads();
    for (int c = 0; c < n; ++c) {
        if (mm) {
            if (c + 1 < n) w_dma(c + 1, (c + 1) & 1);
            if (c + 2 < n) p_dma(c + 2, c & 1);
            mma(c & 1, c + 2 < n ? c + 2 : -1);
            if (c + 2 < n) stage_store(c & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
        } else if (c + 1 < n) {
            dw_compute((c + 1) & 1);
        }
        __syncthreads();
    }
    if (!mm) return;
    // ---- epilogue: + folded bias [, ReLU], store.  C/D layout: column = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2)
    // + 4 (lane >> 5) (output channel)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int p = pb * 32 + (lane & 31);
        const int y = y0 + p / SC_TX, x = x0 + p % SC_TX;
        if (y >= h || x >= w) continue;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[cb][pb][r] + A.b2[co];
                if (A.relu_out) v = fmaxf(v, 0.0f);
                A.out[((long)b * SC_CO + co) * plane + (unsigned)(y * w + x)] = v;
            }
    }
}

}  // namespace

extern "C" int manet_relu_conv1x1_c1_f32(const float *in, int B, int C, long HW, const float *weight, const float *bias,
                                         int relu_in, float *out, manet_stream_t stream)
{
    if (!in || !weight || !out || B <= 0 || C <= 0 || HW <= 0 || B > 65535)
        return manet_set_error(MANET_E_INVALID, "bad arguments");
    dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(relu_conv1x1_c1_kernel, grid, dim3(64 * RC_WAVES), 0, (hipStream_t)stream, in, C, HW, weight, bias,
                       relu_in, out);
    return manet_check_launch("manet_relu_conv1x1_c1_f32");
}

extern "C" int manet_dwconv7x7_bn_relu_f32(const float *in, int B, int C, int h, int w, const float *weight,
                                           const float *bias, const float *bn_scale, const float *bn_shift, int relu,
                                           float *out, manet_stream_t stream)
{
    return manet_dwconv7x7_bn_relu_ex(in, B, C, h, w, weight, bias, bn_scale, bn_shift, relu, 0, out, stream);
}

// relu_in != 0: the input is read through max(x, 0) (the preceding block's ReLU folded into this pass)
extern "C" int manet_dwconv7x7_bn_relu_ex(const float *in, int B, int C, int h, int w, const float *weight,
                                          const float *bias, const float *bn_scale, const float *bn_shift, int relu,
                                          int relu_in, float *out, manet_stream_t stream)
{
    if (!in || !weight || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || (long)B * C > 65535)
        return manet_set_error(MANET_E_INVALID, "bad arguments (B*C must be <= 65535)");
    dim3 grid((unsigned)((w + DW_TX - 1) / DW_TX), (unsigned)((h + DW_TY - 1) / DW_TY), (unsigned)(B * C));
    if (w % 2 == 0 && w >= 2 && ((size_t)in & 7) == 0 && ((size_t)out & 7) == 0)
        hipLaunchKernelGGL(dwconv7x7_bn_relu_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, C, h, w, weight,
                           bias, bn_scale, bn_shift, relu, relu_in, out);
    else
        hipLaunchKernelGGL(dwconv7x7_bn_relu_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, C, h, w, weight,
                           bias, bn_scale, bn_shift, relu, relu_in, out);
    return manet_check_launch("manet_dwconv7x7_bn_relu_f32");
}

// One _split_separable_conv2d block in one launch (sepconv7x7_pw_kernel); out channels fixed at 256 (the reference's
// MODEL_HEAD_EMBEDDING_DIM, config.py:48).  dw_params: [Cin_pad][64] = per input channel the 49 taps, the depthwise bias,
// bn1 scale and bn1 shift (rows beyond Cin: anything).
extern "C" int manet_sepconv7x7_pw_f32(const float *in_a, int64_t batch_stride_a, int Ca, const float *in_b,
                                       int64_t batch_stride_b, int Cb, int B, int h, int w, const float *dw_params,
                                       int relu_in, const float *w2t, int Cin_pad, const float *b2, int Cout, int relu_out,
                                       float *out, manet_stream_t stream)
{
    const int Cin = Ca + Cb;
    if (!in_a || Ca <= 0 || Cb < 0 || (Cb > 0 && !in_b) || !dw_params || !w2t || !b2 || !out || B <= 0 || B > 65535 ||
        h <= 0 || w <= 0)
        return manet_set_error(MANET_E_INVALID, "bad arguments");
    if (Cout != SC_CO) return manet_set_error(MANET_E_INVALID, "Cout=%d (this kernel is built for %d output channels)", Cout, SC_CO);
    const int nchunks = (Cin + SC_KC - 1) / SC_KC;
    if (Cin_pad != nchunks * SC_KC)
        return manet_set_error(MANET_E_INVALID, "w2t / dw_params must have %d rows (Cin padded to whole chunks of %d), got %d",
                               nchunks * SC_KC, SC_KC, Cin_pad);
    if (((size_t)w2t & 15) != 0 || ((size_t)dw_params & 15) != 0)
        return manet_set_error(MANET_E_INVALID, "w2t and dw_params must be 16-byte aligned");
    if ((long)h * w >= (1L << 30)) return manet_set_error(MANET_E_INVALID, "plane too large");
    SepConv A;
    A.in_a = in_a; A.in_b = in_b ? in_b : in_a; A.sa = (long)batch_stride_a; A.sb = (long)batch_stride_b;
    A.Ca = Ca; A.Cin = Cin; A.nchunks = nchunks; A.h = h; A.w = w;
    A.dwp = dw_params; A.relu_in = relu_in;
    A.w2t = w2t; A.b2 = b2; A.relu_out = relu_out; A.out = out;
    dim3 grid((unsigned)((w + SC_TX - 1) / SC_TX), (unsigned)((h + SC_TY - 1) / SC_TY), (unsigned)B);
    hipLaunchKernelGGL(sepconv7x7_pw_kernel, grid, dim3(SC_NT), 0, (hipStream_t)stream, A);
    return manet_check_launch("manet_sepconv7x7_pw_f32");
}
