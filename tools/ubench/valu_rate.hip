// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for scalar / packed fp32 FMA chains at
// 1, 2, 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, NACC = 16;

template <int MODE>  // 0: scalar fma, 1: packed fma, 2: scalar sub + fma (dependent pair), 3: packed sub + fma
__global__ void k(float *out, long long *cyc, float seed)
{
    float a[NACC];
    f32x2 p[NACC];
    for (int i = 0; i < NACC; ++i) { a[i] = seed + i; p[i] = f32x2{seed + i, seed - i}; }
    float x = seed * 1.0001f + threadIdx.x;
    f32x2 xx = {x, x + 1.0f};
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(x));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i]) : "v"(xx));
            if (MODE == 2) { float d; asm volatile("v_sub_f32 %0, %1, %2\n\tv_fma_f32 %3, %0, %0, %3" : "=&v"(d), "+v"(x) , "+v"(a[(i + 1) % NACC]), "+v"(a[i])); }
            if (MODE == 3) { f32x2 d; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_fma_f32 %3, %0, %0, %3" : "=&v"(d), "+v"(xx), "+v"(p[(i + 1) % NACC]), "+v"(p[i])); }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8 * 4096);
    const char *names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_sub_f32+v_fma_f32 (dependent)", "v_pk_add_f32+v_pk_fma_f32 (dependent)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps = 1; wps <= 4; wps *= 2) {
            int threads = 256 * wps;  // wps waves per SIMD, one workgroup per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
            }
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long h[256];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < 256; ++i) avg += h[i];
            avg /= 256;
            int per_it = (mode < 2 ? 1 : 2) * NACC;
            double per_wave = avg / ((double)ITERS * per_it);           // clock64 ticks per instruction as one wave sees it
            double ns_per_instr_wave = ms * 1e6 / ((double)ITERS * per_it);
            printf("%-42s waves/SIMD %d: %.2f ticks, %.2f ns per instr per wave; %.2f ns per instr per SIMD (kernel %.1f us, %.2f ticks/ns)\n", names[mode], wps, per_wave, ns_per_instr_wave, ns_per_instr_wave / wps, ms * 1e3, avg / (ms * 1e6));
        }
    return 0;
}
