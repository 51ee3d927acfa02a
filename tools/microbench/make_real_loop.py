"""Generates tools/microbench/real_loop.hip: the compiler's OWN steady loop of ka_strip (two wavefront steps; cut out of the ISA
listing `make -C kalign_amd/csrc asm`) as inline assembly in a kernel of its own, timed with s_memtime on 1..8 waves of one
workgroup -- with and without its LDS reads, DPP moves and packed operations.  What the step costs when nothing but the step runs:
390 cycles per step at 4.4 cycles per instruction (profiles/r03b_real_loop_microbench.log), whatever else sits on the CU.
usage: make_real_loop.py <ka_kernels.s> <label of the loop, e.g. .LBB8_16812>  >  real_loop.hip"""
import sys
lines = open(sys.argv[1]).read().split('\n')
label = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(label + ':'))
body = []
for l in lines[start + 1:]:
    t = l.strip()
    if not t or t.startswith((';', '.', '//')):
        continue
    if t.startswith('s_cbranch') and label in t:
        break
    body.append(t)
def variant(name, keep):
    ls = [keep(l) for l in body]
    ls = [l for l in ls if l]
    return name, len(ls), '\\n\\t'.join(ls)
variants = [variant('real loop (2 steps)', lambda l: l),
            variant('without ds_read', lambda l: None if l.startswith('ds_read') else l),
            variant('without DPP', lambda l: None if 'wave_' in l else l),
            variant('without v_pk_*', lambda l: None if l.startswith('v_pk_') else l),
            variant('pk without op_sel', lambda l: l.replace(' op_sel:[0,1]', '').replace(' op_sel_hi:[1,0]', ''))]
clob = ','.join('"v%d"' % i for i in range(256))
print('#include <hip/hip_runtime.h>\n#include <cstdio>\n#define N_IT 3000')
def kernel(name, txt, bounds, sig, pre, post):
    return '''
__global__ __launch_bounds__(%s) void %s(%s)
{
        extern __shared__ char lds[];
        unsigned long long t0 = 0, t1 = 0;
%s        asm volatile("v_mbcnt_lo_u32_b32 v130, -1, 0\\n\\tv_mbcnt_hi_u32_b32 v130, -1, v130\\n\\t"
                     "s_mov_b32 s54, 1.0\\n\\ts_movk_i32 s57, 0x7f0\\n\\tv_mov_b32 v125, 0\\n\\tv_mov_b32 v74, 0\\n\\t"
                     "s_mov_b32 s28, 0\\n\\ts_mov_b32 s13, 0\\n\\ts_movk_i32 s12, %%2\\n\\t"
                     "s_memtime %%0\\n\\ts_waitcnt lgkmcnt(0)\\n\\t"
                     "1:\\n\\t"
                     "%s\\n\\t"
                     "s_cbranch_scc1 1b\\n\\t"
                     "s_waitcnt lgkmcnt(0)\\n\\ts_memtime %%1\\n\\ts_waitcnt lgkmcnt(0)"
                     : "=s"(t0), "=s"(t1) : "n"(2 * N_IT)
                     : %s, "s12", "s13", "s28", "s54", "s57", "vcc", "scc", "memory");
%s        if (threadIdx.x == 0) cyc[0] = (long long)(t1 - t0);
}''' % (bounds, name, sig, pre, txt, clob, post)
for i, (name, n, txt) in enumerate(variants):
    print(kernel('k%d' % i, txt, '64', 'long long* cyc', '', ''))
print(kernel('kpark', variants[0][2], '512', 'long long* cyc, int active', '        if ((int)threadIdx.x < active) {\n', '        }\n        __syncthreads();\n'))
print('''
int main()
{
        long long* cyc; (void)hipMalloc(&cyc, 64);
        struct { const char* name; void (*fn)(long long*); int n; } ks[] = {''')
for i, (name, n, txt) in enumerate(variants):
    print('                {"%s", k%d, %d},' % (name, i, n))
print('''        };
        for (auto& k : ks) {
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k.fn, dim3(1), dim3(64), 65536, 0, cyc);
                long long c = 0;
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) { printf("%s: %s\\n", k.name, hipGetErrorString(e)); return 1; }
                (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
                printf("%-28s %4d instr/iter %8.1f clk/iter  %6.2f clk/instr\\n", k.name, k.n, (double)c / N_IT, (double)c / N_IT / k.n);
        }
        for (int active = 64; active <= 512; active *= 2)
        for (int threads = active; threads <= 512; threads *= 2) {
                for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kpark, dim3(1), dim3(threads), 65536, 0, cyc, active);
                long long c = 0;
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
                printf("real loop, %d wave(s) running, %d in the workgroup (rest parked at s_barrier): %8.1f clk/iter\\n", active / 64, threads / 64, (double)c / N_IT);
        }
        return 0;
}''')
