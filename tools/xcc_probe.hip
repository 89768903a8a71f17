// Which XCD does workgroup b run on?  (speed-only knowledge for the block -> tile mapping)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *xcc, int *cu)
{
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        xcc[blockIdx.x] = v & 0xf;
        unsigned w;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(w));
        cu[blockIdx.x] = (int)w;
    }
    // burn a little time so blocks overlap
    float x = threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) xcc[0] = -1;
}
int main(int argc, char **argv)
{
    int nb = argc > 1 ? atoi(argv[1]) : 64;
    int threads = argc > 2 ? atoi(argv[2]) : 256;
    int *dx, *dc;
    hipMalloc(&dx, nb * 4);
    hipMalloc(&dc, nb * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), 0, 0, dx, dc);
    std::vector<int> x(nb), c(nb);
    hipMemcpy(x.data(), dx, nb * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, nb * 4, hipMemcpyDeviceToHost);
    int match = 0;
    for (int b = 0; b < nb; ++b) match += (x[b] == (b % 8));
    printf("blocks=%d threads=%d  xcc==b%%8 for %d blocks\n", nb, threads, match);
    for (int b = 0; b < (nb < 48 ? nb : 48); ++b) printf("%d ", x[b]);
    printf("\n");
    return 0;
}
