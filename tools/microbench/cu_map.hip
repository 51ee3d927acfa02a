// Where do the workgroups of a launch land?  block -> (XCC_ID, SE, SH, CU) for a launch that fills the GPU two workgroups per CU
// (the queued launch's shape): the map behind the CU reservation of round 6 (ka_task_queue_entry: queue workgroups leave the CUs
// kept for the chained launch).  build: hipcc --offload-arch=gfx950 -O2 -o cu_map cu_map.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
__global__ void k(unsigned* out, int spin)
{
        extern __shared__ char lds[];
        unsigned id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (threadIdx.x == 0) { out[2 * blockIdx.x] = id; out[2 * blockIdx.x + 1] = xcc; lds[0] = 1; }
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);        // (stay resident so that the launch spreads)
}
int main()
{
        const int nb = 512;
        unsigned* d; static unsigned h[2 * nb];
        hipMalloc(&d, sizeof(h));
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 79000);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 79000, 0, d, 2000);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_cu;
        std::set<unsigned> cus_of_xcc[16];
        int mism = 0;
        for (int b = 0; b < nb; ++b) {
                const unsigned x = h[2 * b], xcc = h[2 * b + 1] & 15;
                const unsigned cu = (x >> 8) & 15, sh = (x >> 12) & 1, se = (x >> 13) & 7;
                const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
                per_cu[key]++;
                cus_of_xcc[xcc].insert(key & 0xfff);
                if ((int)xcc != b % 8) mism++;
                if (b < 24) printf("block %3d: xcc %u se %u sh %u cu %u (raw %08x)\n", b, xcc, se, sh, cu, x);
        }
        printf("distinct CUs %zu; blocks whose XCC != block %% 8: %d\n", per_cu.size(), mism);
        for (int x = 0; x < 8; ++x) {
                printf("xcc %d: %zu CUs:", x, cus_of_xcc[x].size());
                for (unsigned kk : cus_of_xcc[x]) printf(" %x", kk);
                printf("\n");
        }
        int h1 = 0, h2 = 0, h3 = 0;
        for (auto& kv : per_cu) { if (kv.second == 1) h1++; else if (kv.second == 2) h2++; else h3++; }
        printf("CUs holding 1 / 2 / more workgroups: %d / %d / %d\n", h1, h2, h3);
        return 0;
}
