// Store-pattern microbenchmark for gfx950: 768 planes of 25 680 floats (the head's [3, 256, 120, 214] activation, 79 MB), written cold
// (ten rotating buffers: nothing stays in the 256 MB memory-side cache) by workgroups that own a RUN of pixels and walk the planes --
// as the head's kernels do -- for run lengths 64 px (256 B per plane and workgroup, one wave instruction) .. 2 048 px (8 KB), 4 or 16
// bytes per lane; and a plain linear fill.  Optionally each plane row is also READ first (the 1:1 stream of the depthwise kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -o store_runs store_runs.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr long HW = 25680, PLANES = 768;

// a workgroup of 256 threads owns `run` consecutive pixels (run = 64 * k) and writes them in every plane; VEC = floats per lane
template <int VEC, bool READ>
__global__ void walk(float *__restrict__ out, const float *__restrict__ in, int run, int planes_per_wg)
{
    const int nrun = (int)((HW + run - 1) / run);
    const int r = blockIdx.x % nrun, pg = blockIdx.x / nrun;  // run index, plane group
    const long p0 = (long)r * run;
    const int lanes = run / VEC;                               // lanes that cover the run once
    const int rows_per_pass = 256 / lanes > 0 ? 256 / lanes : 1;
    const int lane = threadIdx.x % lanes, sub = threadIdx.x / lanes;
    if (sub >= rows_per_pass) return;
    for (int c = pg * planes_per_wg + sub; c < (pg + 1) * planes_per_wg; c += rows_per_pass) {
        const long p = p0 + (long)lane * VEC;
        if (p + VEC > HW) continue;
        float *d = out + c * HW + p;
        if (VEC == 4) {
            f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
            if (READ) v += *(const f32x4 *)(in + c * HW + p);
            *(f32x4 *)d = v;
        } else {
            float v = 1.0f;
            if (READ) v += in[c * HW + p];
            *d = v;
        }
    }
}

template <bool READ>
__global__ void linear(float *__restrict__ out, const float *__restrict__ in, long n4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    if (READ) v += ((const f32x4 *)in)[i];
    ((f32x4 *)out)[i] = v;
}

int main()
{
    const long n = HW * PLANES;
    constexpr int NB = 10;
    float *buf[NB], *src[NB];
    for (int i = 0; i < NB; ++i) { hipMalloc(&buf[i], n * 4); hipMalloc(&src[i], n * 4); hipMemset(src[i], 0, n * 4); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char *name) {
        for (int i = 0; i < 20; ++i) launch(buf[i % NB], src[(i + 5) % NB]);
        hipEventRecord(e0, 0);
        const int reps = 100;
        for (int i = 0; i < reps; ++i) launch(buf[i % NB], src[(i + 5) % NB]);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-72s %6.1f us\n", name, ms * 1e3 / reps);
    };
    for (int rd = 0; rd < 2; ++rd) {
        printf("--- %s\n", rd ? "read 79 MB + write 79 MB" : "write 79 MB");
        if (rd) time([&](float *o, float *s) { hipLaunchKernelGGL(linear<true>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, o, s, n / 4); }, "linear, 16 B per lane");
        else time([&](float *o, float *s) { hipLaunchKernelGGL(linear<false>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, o, s, n / 4); }, "linear, 16 B per lane");
        for (int run : {64, 128, 256, 512, 1024, 2048}) {
            for (int ppw : {128, 768}) {
                const int nrun = (int)((HW + run - 1) / run);
                char name[128];
                snprintf(name, sizeof name, "runs of %4d px (%5d B), 4 B per lane, %3d planes per workgroup", run, run * 4, ppw);
                if (run <= 256) {
                    if (rd) time([&](float *o, float *s) { hipLaunchKernelGGL((walk<1, true>), dim3(nrun * (PLANES / ppw)), dim3(256), 0, 0, o, s, run, ppw); }, name);
                    else time([&](float *o, float *s) { hipLaunchKernelGGL((walk<1, false>), dim3(nrun * (PLANES / ppw)), dim3(256), 0, 0, o, s, run, ppw); }, name);
                }
                snprintf(name, sizeof name, "runs of %4d px (%5d B), 16 B per lane, %3d planes per workgroup", run, run * 4, ppw);
                if (run <= 1024) {
                    if (rd) time([&](float *o, float *s) { hipLaunchKernelGGL((walk<4, true>), dim3(nrun * (PLANES / ppw)), dim3(256), 0, 0, o, s, run, ppw); }, name);
                    else time([&](float *o, float *s) { hipLaunchKernelGGL((walk<4, false>), dim3(nrun * (PLANES / ppw)), dim3(256), 0, 0, o, s, run, ppw); }, name);
                }
            }
        }
    }
    return 0;
}
