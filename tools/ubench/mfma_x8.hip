// Is v_mfma_f32_32x32x8_bf16_1k half the cost of v_mfma_f32_32x32x16_bf16 on gfx950?  (VERDICT r3 next #5 proposes to run the
// C = 100 contraction as 6 x K16 + 1 x K8 = 104 k-slots instead of 7 x K16 = 112: that only pays if the K8 instruction -- the
// CDNA3 opcode, kept on gfx950 -- retires in half the passes.)  Pure register-resident MFMA streams, 4 independent accumulators,
// constant operands (no DVFS effect): ns and s_memtime ticks per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 20000;

template <int MODE>  // 0: 7 x K16 per pass; 1: 6 x K16 + 1 x K8; 2: K8 only (7 per pass); 3: 6 x K16 only
__global__ __launch_bounds__(512) void k(float *out, long long *ticks)
{
    const long long tk0 = clock64();
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    u32x2 a2 = {0x3f803f80u, 0x3f803f80u}, b2 = a2;
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) c[j] = f32x16{0};
    for (int p = 0; p < ITERS; ++p) {
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
            const bool k8 = (MODE == 2) || (MODE == 1 && kk == 6);
            if (MODE == 3 && kk == 6) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k8) asm volatile("v_mfma_f32_32x32x8bf16_1k %0, %1, %2, %0" : "+v"(c[j]) : "v"(a2), "v"(b2));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += c[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = clock64() - tk0;
}

template <int MODE>
void run(const char *name, int per_pass, float *out, long long *ticks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        if (rep == 1) hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, ticks);
    }
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    long long ht[256]; hipMemcpy(ht, ticks, sizeof(ht), hipMemcpyDeviceToHost);
    double avgt = 0; for (int i = 0; i < 256; ++i) avgt += ht[i]; avgt /= 256;
    const double passes = (double)ITERS;  // one wave per SIMD
    printf("%-40s %.1f ns per pass of %d MFMAs per SIMD (%.2f ns each), %.0f s_memtime ticks per pass\n", name, ms * 1e6 / passes,
           per_pass, ms * 1e6 / passes / per_pass, avgt / passes);
}

int main()
{
    float *out; long long *ticks;
    hipMalloc(&out, 4 << 20); hipMalloc(&ticks, 8 * 256);
    run<0>("7 x K16 (112 k-slots, shipped)", 28, out, ticks);
    run<1>("6 x K16 + 1 x K8 (104 k-slots)", 28, out, ticks);
    run<3>("6 x K16 (96 k-slots)", 24, out, ticks);
    run<2>("7 x K8", 28, out, ticks);
    return 0;
}
