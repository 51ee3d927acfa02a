// Issue / latency microbenchmarks for ONE wave64 on one SIMD (gfx950): what does a dependent fp32 chain cost, what do
// packed ops, DPP moves and LDS reads in the pattern of ka_strip's step cost?  hipcc --offload-arch=gfx950 -O3 -o lone_wave lone_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define N_IT 4000

#define KERNEL(name, body)                                                                     \
        __global__ void name(float* out, long long* cyc, int nwaves)                           \
        {                                                                                      \
                __shared__ float4v lds[1024];                                                  \
                const int lane = threadIdx.x & 63;                                             \
                for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = float4v{1.0f * i, 0.5f, 0.25f, 2.0f}; \
                __syncthreads();                                                               \
                float acc = out[lane], x = 1.0e-7f * lane, y = 3.0e-7f;                        \
                float2v acc2 = {acc, acc + 1.0f}, a2 = {x, y}, b2 = {y, x}, p2 = {0.0f, 0.0f}, q2 = {0.0f, 0.0f}; \
                float4v r0 = lds[lane], r1 = lds[lane + 64], r2 = lds[lane + 128], r3 = lds[lane + 192], r4 = r0, r5 = r1, r6 = r2; \
                unsigned addr = (unsigned)(unsigned long long)&lds[0] + lane * 16;             \
                float d0 = acc, d1 = x, d2 = y;                                                \
                (void)a2; (void)b2; (void)p2; (void)q2; (void)addr; (void)d0; (void)d1; (void)d2; (void)r3; (void)r4; (void)r5; (void)r6; \
                const long long t0 = __builtin_amdgcn_s_memtime();                             \
                for (int it = 0; it < N_IT; ++it) { body }                                     \
                const long long t1 = __builtin_amdgcn_s_memtime();                             \
                out[threadIdx.x] = acc + acc2.x + acc2.y + d0 + d1 + d2 + r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + p2.x + q2.y; \
                if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                               \
        }

// 1: 20 dependent v_add_f32
KERNEL(k_add20, asm volatile(".rept 20\n v_add_f32 %0, %0, %1\n .endr" : "+v"(acc) : "v"(x));)
// 2: 40 independent-pair v_add (two chains interleaved)
KERNEL(k_add20x2, asm volatile(".rept 20\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n .endr" : "+v"(acc), "+v"(d0) : "v"(x));)
// 3: 20 x (v_pk_mul independent, v_pk_add dependent) -- the two-row dot product
KERNEL(k_pk20, asm volatile(".rept 10\n v_pk_mul_f32 %1, %3, %4\n v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %2, %3, %4\n v_pk_add_f32 %0, %0, %1\n .endr"
                            : "+v"(acc2), "+v"(p2), "+v"(q2) : "v"(a2), "v"(b2));)
// 4: 20 dependent v_pk_add back to back
KERNEL(k_pkadd20, asm volatile(".rept 20\n v_pk_add_f32 %0, %0, %1\n .endr" : "+v"(acc2) : "v"(a2));)
// 5: the one-row dot product: 10 v_pk_mul + 20 dependent v_add
KERNEL(k_q1dot, asm volatile(".rept 5\n v_pk_mul_f32 v[40:41], %1, %2\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v42\n v_pk_mul_f32 v[42:43], %1, %2\n v_add_f32 %0, %0, v41\n v_add_f32 %0, %0, v40\n .endr"
                             : "+v"(acc) : "v"(a2), "v"(b2) : "v40", "v41", "v42", "v43");)
// 6: 20 independent v_mul (issue rate of plain VALU)
KERNEL(k_indep20, asm volatile(".rept 10\n v_mul_f32 %0, %2, %3\n v_mul_f32 %1, %2, %3\n .endr" : "=v"(d1), "=v"(d2) : "v"(x), "v"(y));)
// 7: 9 DPP moves of values just written (hazards handled by the s_nop the assembler does NOT add: explicit)
KERNEL(k_dpp9, asm volatile(".rept 3\n v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n v_add_f32 %2, %2, %3\n s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n .endr"
                            : "+v"(d0), "+v"(d1), "+v"(d2) : "v"(x));)
// 8: 7 ds_read_b128 issued, then 40 VALU, then wait (the step's pattern: reads one step ahead)
KERNEL(k_lds7_40, asm volatile("ds_read_b128 %1, %8\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:4096\n ds_read_b128 %4, %8 offset:6144\n"
                               "ds_read_b128 %5, %8 offset:8192\n ds_read_b128 %6, %8 offset:10240\n ds_read_b128 %7, %8 offset:12288\n"
                               ".rept 40\n v_add_f32 %0, %0, %9\n .endr\n s_waitcnt lgkmcnt(0)"
                               : "+v"(acc), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6) : "v"(addr), "v"(x));)
// 9: the same with 15 VALU between issue and wait (the one-row step)
KERNEL(k_lds7_15, asm volatile("ds_read_b128 %1, %8\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:4096\n ds_read_b128 %4, %8 offset:6144\n"
                               "ds_read_b128 %5, %8 offset:8192\n ds_read_b128 %6, %8 offset:10240\n ds_read_b128 %7, %8 offset:12288\n"
                               ".rept 15\n v_add_f32 %0, %0, %9\n .endr\n s_waitcnt lgkmcnt(0)"
                               : "+v"(acc), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6) : "v"(addr), "v"(x));)
// 10: 7 ds_read_b128 + immediate wait (raw LDS latency of the batch)
KERNEL(k_lds7_0, asm volatile("ds_read_b128 %1, %8\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:4096\n ds_read_b128 %4, %8 offset:6144\n"
                              "ds_read_b128 %5, %8 offset:8192\n ds_read_b128 %6, %8 offset:10240\n ds_read_b128 %7, %8 offset:12288\n"
                              "s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %9"
                              : "+v"(acc), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6) : "v"(addr), "v"(x));)
// 11: 1 ds_read_b128 + immediate wait
KERNEL(k_lds1_0, asm volatile("ds_read_b128 %1, %2\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %3" : "+v"(acc), "=&v"(r0) : "v"(addr), "v"(x));)
// 12: 20 dependent v_max_f32 / v_add alternating (the gap recurrences)
KERNEL(k_addmax, asm volatile(".rept 10\n v_add_f32 %0, %0, %1\n v_max_f32 %0, %0, %1\n .endr" : "+v"(acc) : "v"(x));)
// 13: 20 s_nop 0 (scalar issue)
KERNEL(k_salu20, asm volatile(".rept 20\n s_add_u32 s20, s20, 1\n .endr" ::: "s20");)
// 14: v_pk_mul (independent) x20
KERNEL(k_pkmul20, asm volatile(".rept 10\n v_pk_mul_f32 %0, %2, %3\n v_pk_mul_f32 %1, %2, %3\n .endr" : "=v"(p2), "=v"(q2) : "v"(a2), "v"(b2));)

// 15..18: the steady step of ka_strip<KA_PP, 20, 0, 2> as an instruction mix (round 3, second session): wait, 3 v_mul, 2 x (2 v_add + v_max3),
// 20 x (v_pk_mul, dependent v_pk_add), 3 address VALU + 7 ds_read_b128, 9 DPP wave shifts, 4 x (2 v_add + v_max) -- 80 instructions.
// Variants leave out the LDS reads / the DPP moves / both: which part of the step costs more than its issue slots?
#define MIX_HEAD "s_waitcnt lgkmcnt(0)\n v_mul_f32 v60, %3, v45\n v_mul_f32 v61, %3, v46\n v_mul_f32 v62, %3, v47\n" \
                 "v_add_f32 v63, %0, v60\n v_add_f32 v64, %0, v61\n v_max3_f32 v40, %0, v63, v64\n v_add_f32 v63, %0, v60\n v_add_f32 v64, %0, v61\n v_max3_f32 v41, %0, v63, v64\n"
#define MIX_CHAIN ".rept 10\n v_pk_mul_f32 v[50:51], %1, v[20:21]\n v_pk_add_f32 v[40:41], v[40:41], v[52:53]\n v_pk_mul_f32 v[52:53], %1, v[24:25]\n v_pk_add_f32 v[40:41], v[40:41], v[50:51]\n .endr\n"
#define MIX_LDS "v_sub_u32 v65, %4, %4\n v_lshl_add_u32 v65, v65, 4, 16\n v_and_or_b32 v65, v65, 63, %4\n" \
                "ds_read_b128 v[20:23], v65\n ds_read_b128 v[24:27], v65 offset:2048\n ds_read_b128 v[28:31], v65 offset:4096\n ds_read_b128 v[32:35], v65 offset:6144\n" \
                "ds_read_b128 v[36:39], v65 offset:8192\n ds_read_b128 v[42:45], v65 offset:10240\n ds_read_b128 v[46:49], v65 offset:12288\n"
#define MIX_DPP "s_nop 1\n v_mov_b32_dpp v66, v40 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v67, v41 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v68, v60 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
                "v_mov_b32_dpp v69, v61 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v70, v62 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v71, v63 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_mov_b32_dpp v72, v40 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v73, v41 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v74, v64 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
#define MIX_GAP ".rept 4\n v_add_f32 v63, %0, v60\n v_add_f32 v64, v40, v61\n v_max_f32 %0, v63, v64\n .endr\n"
#define MIX_CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49", \
                 "v50","v51","v52","v53","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","memory"
KERNEL(k_mix_full, asm volatile(MIX_HEAD MIX_CHAIN MIX_LDS MIX_DPP MIX_GAP : "+v"(acc) : "v"(a2), "v"(b2), "v"(x), "v"(addr) : MIX_CLOB);)
KERNEL(k_mix_nolds, asm volatile(MIX_HEAD MIX_CHAIN MIX_DPP MIX_GAP : "+v"(acc) : "v"(a2), "v"(b2), "v"(x), "v"(addr) : MIX_CLOB);)
KERNEL(k_mix_nodpp, asm volatile(MIX_HEAD MIX_CHAIN MIX_LDS MIX_GAP : "+v"(acc) : "v"(a2), "v"(b2), "v"(x), "v"(addr) : MIX_CLOB);)
KERNEL(k_mix_chain, asm volatile(MIX_HEAD MIX_CHAIN MIX_GAP : "+v"(acc) : "v"(a2), "v"(b2), "v"(x), "v"(addr) : MIX_CLOB);)
// the LDS reads FIRST (right behind the wait), the chain behind them: do returning reads slow the VALU down?
KERNEL(k_mix_ldsfirst, asm volatile(MIX_HEAD MIX_LDS MIX_CHAIN MIX_DPP MIX_GAP : "+v"(acc) : "v"(a2), "v"(b2), "v"(x), "v"(addr) : MIX_CLOB);)

__global__ void k_clock(long long* cyc)
{
        const long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime(), c0 = clock64();
        float a = (float)threadIdx.x;
        for (int i = 0; i < 2000; ++i) asm volatile(".rept 50\n v_add_f32 %0, %0, %0\n .endr" : "+v"(a));
        const long long w1 = wall_clock64(), t1 = __builtin_amdgcn_s_memtime(), c1 = clock64();
        if (threadIdx.x == 0) { cyc[0] = w1 - w0; cyc[1] = t1 - t0; cyc[2] = c1 - c0; cyc[3] = (long long)a; }
}

struct K { const char* name; void (*fn)(float*, long long*, int); int ninstr; };

int main()
{
        setvbuf(stdout, NULL, _IONBF, 0);
        float* out; long long* cyc;
        (void)hipMalloc(&out, 4096 * sizeof(float)); (void)hipMalloc(&cyc, 64 * sizeof(long long));
        (void)hipMemset(out, 0, 4096 * sizeof(float));
        K ks[] = { {"20 dependent v_add_f32", k_add20, 20}, {"2 x 20 interleaved v_add_f32 (two chains)", k_add20x2, 40},
                   {"20 x (v_pk_mul + dependent v_pk_add)", k_pk20, 40}, {"20 dependent v_pk_add_f32", k_pkadd20, 20},
                   {"10 v_pk_mul + 20 dependent v_add (one-row dot)", k_q1dot, 30}, {"20 independent v_mul_f32", k_indep20, 20},
                   {"9 v_add + 9 DPP wave_shr + 3 s_nop", k_dpp9, 21}, {"7 ds_read_b128, 40 v_add, wait", k_lds7_40, 48},
                   {"7 ds_read_b128, 15 v_add, wait", k_lds7_15, 23}, {"7 ds_read_b128, wait", k_lds7_0, 9}, {"1 ds_read_b128, wait", k_lds1_0, 3},
                   {"10 x (v_add, v_max) dependent", k_addmax, 20}, {"20 independent v_pk_mul_f32", k_pkmul20, 20},
                   {"step mix: head + chain + 7 LDS + 9 DPP + gap", k_mix_full, 82}, {"step mix without the LDS reads", k_mix_nolds, 72},
                   {"step mix without the DPP moves", k_mix_nodpp, 72}, {"step mix: head + chain + gap only", k_mix_chain, 62},
                   {"step mix with the LDS reads ahead of the chain", k_mix_ldsfirst, 82} };
        for (int waves = 1; waves <= 8; waves *= 2) {
                // waves per workgroup: 1 (lone wave), 2, 4 (one per SIMD), 8 (two per SIMD)
                printf("---- %d wave(s) in the workgroup ----\n", waves);
                for (auto& k : ks) {
                        hipLaunchKernelGGL(k.fn, dim3(1), dim3(64 * waves), 0, 0, out, cyc, waves);
                        hipLaunchKernelGGL(k.fn, dim3(1), dim3(64 * waves), 0, 0, out, cyc, waves);
                        long long c = 0;
                        hipError_t e = hipDeviceSynchronize();
                        if (e != hipSuccess) { printf("%s: %s\n", k.name, hipGetErrorString(e)); return 1; }
                        (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
                        printf("%-52s %8.1f clk/iter  %6.2f clk/instr (s_memtime ticks)\n", k.name, (double)c / N_IT, (double)c / N_IT / k.ninstr);
                }
        }
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, cyc);
        long long c4[4];
        (void)hipMemcpy(c4, cyc, sizeof(c4), hipMemcpyDeviceToHost);
        int wrate = 0; (void)hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
        printf("100000 dependent v_add: wall_clock64 %lld ticks (%d kHz), s_memtime %lld, clock64 %lld -> s_memtime runs at %.1f MHz, %.2f s_memtime ticks per v_add\n",
               c4[0], wrate, c4[1], c4[2], (double)c4[1] / ((double)c4[0] / (wrate * 1e3)) / 1e6, (double)c4[1] / 1e5);
        return 0;
}
