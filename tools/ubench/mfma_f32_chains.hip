// How many independent accumulator chains does a wave need to keep the fp32 matrix pipe full?  v_mfma_f32_32x32x2_f32 streams
// with operands in registers (no LDS, no memory): NACC accumulators per wave, issued round robin; 1 or 2 waves per SIMD.
// (The register-resident-weights 1x1 kernel has 2 chains per wave -- 32 output channels x 2 pixel blocks -- and its bare MFMA
// loop measured 0.72 of the peak.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 4000;

template <int NACC>
__global__ __launch_bounds__(256, 2) void k(float *out, const float *data)
{
    const int lane = threadIdx.x & 63;
    float a[16], b[NACC][16];
    for (int s = 0; s < 16; ++s) {
        a[s] = data[s * 64 + lane];
        for (int j = 0; j < NACC; ++j) b[j][s] = data[1024 + (j * 16 + s) * 64 + lane];
    }
    f32x16 c[NACC];
    for (int j = 0; j < NACC; ++j) c[j] = f32x16{0};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[j][s], c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += c[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu, float *out, float *data)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        if (rep == 1) hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, data);
    }
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    const double n = (double)ITERS * 16 * NACC * wgs_per_cu;  // MFMAs per SIMD
    printf("%d accumulator chain(s) per wave, %d wave(s) per SIMD: %.2f ns per MFMA per SIMD = %.1f TFLOP/s (%.3f of 157.3)\n", NACC,
           wgs_per_cu, ms * 1e6 / n, 1024.0 * n * 4096.0 / (ms * 1e-3) / 1e12, 1024.0 * n * 4096.0 / (ms * 1e-3) / 1e12 / 157.3);
}

int main()
{
    float *out, *data;
    hipMalloc(&out, 4 << 20); hipMalloc(&data, 4 * 65536);
    float *h = (float *)malloc(4 * 65536);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 4.0f - 2.0f;
    hipMemcpy(data, h, 4 * 65536, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run<1>(w, out, data);
        run<2>(w, out, data);
        run<4>(w, out, data);
    }
    return 0;
}
