// Which SIMD does wave w of a workgroup land on?  (ka_run_items deals the first items to waves 0..3 of every
// workgroup on the assumption that those sit on four different SIMDs.)
// build: hipcc --offload-arch=gfx950 -O2 -o hw_id hw_id.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
}
int main(int argc, char** argv)
{
        const int nb = 6;
        for (int nt : {256, 512}) {
                unsigned* d; unsigned h[64];
                hipMalloc(&d, sizeof(h));
                hipLaunchKernelGGL(k, dim3(nb), dim3(nt), nt == 512 ? 150000 : 30000, 0, d);
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                printf("%d threads per workgroup (dynamic LDS %d)\n", nt, nt == 512 ? 150000 : 30000);
                for (int b = 0; b < nb; ++b) {
                        printf("  wg %d:", b);
                        for (int w = 0; w < nt / 64; ++w) {
                                const unsigned x = h[b * (nt / 64) + w];
                                printf("  w%d simd %u cu %u se %u", w, (x >> 4) & 3, (x >> 8) & 15, (x >> 13) & 7);
                        }
                        printf("\n");
                }
                hipFree(d);
        }
        return 0;
}
