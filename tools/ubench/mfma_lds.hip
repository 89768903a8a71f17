// Does an LDS read stream slow a bf16 MFMA stream on gfx950?  One workgroup per CU, W waves per SIMD; each wave loops
// over a "pass" of 28 v_mfma_f32_32x32x16_bf16 (4 accumulators x 7 k-steps, as the wide global-match kernel) with
// R ds_read_b128 interleaved one per k-step (R = 0 or 7), destinations either never read by an MFMA (MODE 1) or used as the
// next pass's A operand (MODE 2).   Build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int PASSES = 2000;

template <int MODE>  // 4: interleaved reads into AGPRs; 3: burst of 7 reads, then 28 MFMAs; 0: no reads; 1: 7 reads per pass into registers nobody uses; 2: 7 reads per pass feeding the next pass
__global__ __launch_bounds__(512) void k(float *out, const unsigned *seed, long long *ticks)
{
    const long long tk0 = clock64();
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed[i & 255] + i;
    __syncthreads();
    u32x4 a[7], b[4][7], spare[7];
    for (int k = 0; k < 7; ++k) {
        a[k] = *(const u32x4 *)(lds + ((threadIdx.x & 63) * 4 + k * 256));
        spare[k] = a[k];
        for (int j = 0; j < 4; ++j) b[j][k] = *(const u32x4 *)(lds + ((threadIdx.x & 63) * 4 + (k + j) * 256 + 1024));
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned *base = lds + (threadIdx.x & 63) * 4;
    for (int p = 0; p < PASSES; ++p) {
        if (MODE == 3) {  // burst: all 7 reads of the next pass first, then the 28 MFMAs
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) spare[kk] = *(const volatile u32x4 *)(base + ((p + kk) & 7) * 256);
            __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[0][kk]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[1][kk]), c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[2][kk]), c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[3][kk]), c3, 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 28, 0);
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) { u32x4 t = a[kk]; a[kk] = spare[kk]; spare[kk] = t; }
            continue;
        }
        if (MODE == 8 || MODE == 9) {
            // 8 = B (query) operands in AccVGPRs, accumulators and fragment reads in ArchVGPRs; 9 = B and fragments in AccVGPRs
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                const unsigned addr = (unsigned)(size_t)(base + ((p + kk) & 7) * 256);
                if (MODE == 8)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %8, %3\n\t"
                                 "ds_read_b128 %4, %9"
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a[kk])
                                 : "a"(b[0][kk]), "a"(b[1][kk]), "a"(b[2][kk]), "a"(b[3][kk]), "v"(addr) : "memory");
                else
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %8, %3\n\t"
                                 "ds_read_b128 %4, %9"
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+a"(a[kk])
                                 : "a"(b[0][kk]), "a"(b[1][kk]), "a"(b[2][kk]), "a"(b[3][kk]), "v"(addr) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
        if (MODE == 5 || MODE == 6 || MODE == 7) {
            // explicit register classes: 5 = accumulators in ArchVGPRs, fragments read into AccVGPRs;
            // 6 = accumulators in AccVGPRs, fragments read into ArchVGPRs; 7 = accumulators in AccVGPRs, no reads
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                const unsigned addr = (unsigned)(size_t)(base + ((p + kk) & 7) * 256);
                if (MODE == 5) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %8, %3\n\t"
                                 "ds_read_b128 %4, %9"
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+a"(a[kk])
                                 : "v"(b[0][kk]), "v"(b[1][kk]), "v"(b[2][kk]), "v"(b[3][kk]), "v"(addr) : "memory");
                } else if (MODE == 6) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %8, %3\n\t"
                                 "ds_read_b128 %4, %9"
                                 : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+v"(a[kk])
                                 : "v"(b[0][kk]), "v"(b[1][kk]), "v"(b[2][kk]), "v"(b[3][kk]), "v"(addr) : "memory");
                } else {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %8, %3"
                                 : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
                                 : "v"(a[kk]), "v"(b[0][kk]), "v"(b[1][kk]), "v"(b[2][kk]), "v"(b[3][kk]));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[0][kk]), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[1][kk]), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[2][kk]), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]), __builtin_bit_cast(bf16x8, b[3][kk]), c3, 0, 0, 0);
            if (MODE == 1) spare[kk] = *(const volatile u32x4 *)(base + ((p + kk) & 7) * 256);
            if (MODE == 2) a[kk] = *(const volatile u32x4 *)(base + ((p + kk) & 7) * 256);
            if (MODE == 4) {  // the same refill, but the LDS data lands in AccVGPRs (the MFMA reads its A operand from there)
                const unsigned addr = (unsigned)(size_t)(base + ((p + kk) & 7) * 256);
                asm volatile("ds_read_b128 %0, %1" : "=a"(a[kk]) : "v"(addr) : "memory");
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int kk = 0; kk < 7; ++kk) s += (float)spare[kk][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = clock64() - tk0;
}

int main()
{
    float *out; unsigned *seed; long long *ticks; hipMalloc(&ticks, 8 * 256);
    hipMalloc(&out, 4 << 20); hipMalloc(&seed, 1024); hipMemset(seed, 0x11, 1024);
    const char *names[10] = {"no LDS reads", "7 ds_read_b128 per pass, results unused by MFMA", "7 ds_read_b128 per pass feeding the next pass's A operand", "7 ds_read_b128 in one burst per pass, feeding the next pass", "7 ds_read_b128 per pass into AccVGPRs, feeding the next pass", "asm: acc in ArchVGPR, 7 reads per pass into AccVGPR", "asm: acc in AccVGPR, 7 reads per pass into ArchVGPR", "asm: acc in AccVGPR, no reads", "asm: B (query) in AccVGPR, acc + 7 reads per pass in ArchVGPR", "asm: B and the 7 reads per pass in AccVGPR, acc in ArchVGPR"};
    for (int mode = 0; mode < 10; ++mode)
        for (int wps = 1; wps <= 2; ++wps) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
                if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(256), dim3(256 * wps), 0, 0, out, seed, ticks);
            }
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long ht[256]; hipMemcpy(ht, ticks, sizeof(ht), hipMemcpyDeviceToHost);
            double avgt = 0; for (int i = 0; i < 256; ++i) avgt += ht[i]; avgt /= 256;
            double mfma_per_simd = (double)PASSES * 28 * wps;
            double tf = 256.0 * 4 * mfma_per_simd * 32768.0 / (ms * 1e-3) / 1e12;
            printf("%-62s waves/SIMD %d: %.3f ms, %.2f ns per MFMA per SIMD, %.0f TFLOP/s; s_memtime %.2f ticks per ns of kernel, %.1f ticks per MFMA per SIMD\n", names[mode], wps, ms, ms * 1e6 / mfma_per_simd, tf, avgt / (ms * 1e6), avgt / mfma_per_simd);
        }
    return 0;
}
