// The fp32 headline kernel's MFMA stream alone (v_mfma_f32_32x32x2_f32: 2x2 accumulator blocks per wave, 50 k-steps per tile,
// A fragments refilled from LDS, two waves per SIMD), on constant and on random operands: what the fp32 matrix pipe sustains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TILES = 400;

__global__ __launch_bounds__(256, 2) void k(float *out, const float *data, long long *ticks)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 13 * 2 * 64 * 4];  // two tile images: 13 groups x 2 halves x 64 rows x float4
    for (int i = threadIdx.x; i < 2 * 13 * 2 * 64 * 4; i += blockDim.x) lds[i] = data[i];
    __syncthreads();
    const long long tk0 = clock64();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    f32x4 q0[13], q1[13];
    for (int g = 0; g < 13; ++g) {
        q0[g] = *(const f32x4 *)(data + 16384 + (g * 64 + lane) * 4);
        q1[g] = *(const f32x4 *)(data + 32768 + (g * 64 + lane) * 4);
    }
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    for (int t = 0; t < TILES; ++t) {
        const f32x4 *A = (const f32x4 *)(lds + (t & 1) * 13 * 2 * 64 * 4);
#pragma unroll
        for (int g = 0; g < 13; ++g) {
            const f32x4 a0 = A[(g * 2 + h) * 64 + l31], a1 = A[(g * 2 + h) * 64 + 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (g * 4 + j < 50) {
                    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], q0[g][j], c00, 0, 0, 0);
                    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], q1[g][j], c01, 0, 0, 0);
                    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], q0[g][j], c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], q1[g][j], c11, 0, 0, 0);
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c00[i] + c01[i] + c10[i] + c11[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = clock64() - tk0;
}

int main()
{
    float *out, *data; long long *ticks;
    hipMalloc(&out, 4 << 20); hipMalloc(&data, 4 * 65536); hipMalloc(&ticks, 8 * 512);
    float *h = (float *)malloc(4 * 65536);
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (int i = 0; i < 65536; ++i) h[i] = pass == 0 ? 1.0f : (float)rand() / RAND_MAX * 4.0f - 2.0f;
        hipMemcpy(data, h, 4 * 65536, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 1) hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, out, data, ticks);  // 2 workgroups per CU: 2 waves per SIMD
        }
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
        long long ht[512]; hipMemcpy(ht, ticks, sizeof(ht), hipMemcpyDeviceToHost);
        double avgt = 0; for (int i = 0; i < 512; ++i) avgt += ht[i]; avgt /= 512;
        double n = (double)TILES * 200 * 2;  // MFMAs per SIMD
        printf("%-28s %.2f ns per MFMA per SIMD = %.1f TFLOP/s (%.3f of 157.3); wave clock %.2f GHz\n",
               pass == 0 ? "operands constant (1.0):" : "operands random in (-2, 2):", ms * 1e6 / n,
               256.0 * 4 * n * 4096.0 / (ms * 1e-3) / 1e12, 256.0 * 4 * n * 4096.0 / (ms * 1e-3) / 1e12 / 157.3, avgt / (ms * 1e6));
    }
    return 0;
}
