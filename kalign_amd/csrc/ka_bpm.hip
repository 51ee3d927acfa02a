// ka_bpm.hip -- distance estimation for the guide tree: the reference's block-wise Myers
// bit-vector edit distance (bpm_block, lib/src/bpm.c:356-582) for a batch of sequence pairs
// (d_estimation / calc_distance, lib/src/sequence_distance.c:37-162: numseq x num_samples pairs).
//
// Integer-exact restatement, including the reference's band rules.  One THREAD per pair: the recurrence is a
// carry chain over the pattern's 64-bit blocks for every text position, so there is nothing to spread over
// lanes inside a pair; 64 pairs per wave run in lock-step (blocks beyond a pair's band are masked off).
//   * Peq[c][block] (which pattern positions hold residue c) depends on the pattern only: a first kernel
//     builds it once per sequence (13 x 16 x 8 B = 1664 B each, L2-resident), the scan kernel gathers it.
//   * the per-pair vertical state P[16], M[16], score[16] lives in LDS, block-major ([block][thread]) so
//     that the lock-stepped block loop is bank-conflict free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KA_BPM_SIGMA 13
#define KA_BPM_BLOCKS 16
#define KA_BPM_THREADS 128

// one workgroup per sequence, one thread per (residue, block)
__global__ __launch_bounds__(256) void ka_bpm_peq_kernel(const uint8_t* __restrict__ codes, const int* __restrict__ off,
                                                        const int* __restrict__ lens, unsigned long long* __restrict__ peq)
{
        const int s = blockIdx.x;
        const int x = threadIdx.x;
        if (x >= KA_BPM_SIGMA * KA_BPM_BLOCKS) return;
        const int c = x / KA_BPM_BLOCKS, b = x % KA_BPM_BLOCKS;
        int m = lens[s];
        if (m > 1024) m = 1024;                                       // bpm.c:367-369
        const int b_max = (m == 0) ? 1 : (m / 64 + ((m % 64) ? 1 : 0));
        const uint8_t* p = codes + off[s];
        unsigned long long w = 0ull;
        if (b < b_max) {
                for (int k = 0; k < 64; ++k) {
                        const int i = b * 64 + k;
                        if (i >= m || p[i] == c) w |= (1ull << k);       // beyond m: matches anything (bpm.c:430-433)
                }
        }
        peq[((long long)s * KA_BPM_SIGMA + c) * KA_BPM_BLOCKS + b] = w;
}

__global__ __launch_bounds__(KA_BPM_THREADS) void ka_bpm_scan_kernel(const uint8_t* __restrict__ codes, const int* __restrict__ off,
                                                                    const int* __restrict__ lens,
                                                                    const unsigned long long* __restrict__ peq,
                                                                    const int* __restrict__ ia, const int* __restrict__ ib,
                                                                    const int npairs, int* __restrict__ dist)
{
        __shared__ unsigned long long sP[KA_BPM_BLOCKS][KA_BPM_THREADS];
        __shared__ unsigned long long sM[KA_BPM_BLOCKS][KA_BPM_THREADS];
        __shared__ int sS[KA_BPM_BLOCKS][KA_BPM_THREADS];
        const int tid = threadIdx.x;
        const int pair = blockIdx.x * KA_BPM_THREADS + tid;
        const bool live = pair < npairs;
        int a = live ? ia[pair] : 0, bq = live ? ib[pair] : 0;
        // calc_distance: the longer sequence is the text (sequence_distance.c:153-157)
        int st = a, sp = bq;
        if (!(lens[a] > lens[bq])) { st = bq; sp = a; }
        const int n = live ? lens[st] : 0;
        int m = lens[sp];
        if (m > 1024) m = 1024;
        const int b_max = (m == 0) ? 1 : (m / 64 + ((m % 64) ? 1 : 0));
        const int wpad = 64 * b_max - m;
        const int maxd = m;
        int k = m;
        int y = b_max - 1;                                            // DIV_CEIL(maxd, 64) - 1 with maxd = m
        const uint8_t* t = codes + off[st];
        const unsigned long long* pq = peq + (long long)sp * KA_BPM_SIGMA * KA_BPM_BLOCKS;
        const unsigned long long ONE = 1ull, HIGH = 1ull << 63;
#pragma unroll
        for (int b = 0; b < KA_BPM_BLOCKS; ++b) { sP[b][tid] = (b <= y) ? ~0ull : 0ull; sM[b][tid] = 0ull; sS[b][tid] = (b <= y) ? (b + 1) * 64 : 0; }
        const int nsteps = live ? (n + wpad) : 0;
        // lock-step over the text positions of the longest scan in the wave
        int maxsteps = nsteps;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxsteps = max(maxsteps, __shfl_xor(maxsteps, o, 64));
        for (int i = 0; i < maxsteps; ++i) {
                const bool act = i < nsteps;
                const int c = (act && i < n) ? t[i] : 0;              // positions >= n: padding with code 0 (bpm.c:455-461)
                const unsigned long long* pc = pq + c * KA_BPM_BLOCKS;
                int carry = 0;
                int ymax = act ? y : -1;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) ymax = max(ymax, __shfl_xor(ymax, o, 64));
                for (int b = 0; b <= ymax; ++b) {
                        if (act && b <= y) {
                                unsigned long long Pv = sP[b][tid], Mv = sM[b][tid], Eq = pc[b];
                                const int hin = carry;
                                int hout = 0;
                                const unsigned long long Xv = Eq | Mv;
                                if (hin < 0) Eq |= ONE;
                                const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                                unsigned long long Ph = Mv | ~(Xh | Pv);
                                unsigned long long Mh = Pv & Xh;
                                if (Ph & HIGH) hout += 1;
                                if (Mh & HIGH) hout -= 1;
                                Ph <<= 1; Mh <<= 1;
                                if (hin < 0) Mh |= ONE; else if (hin > 0) Ph |= ONE;
                                sP[b][tid] = Mh | ~(Xv | Ph);
                                sM[b][tid] = Ph & Xv;
                                carry = hout;
                                sS[b][tid] += carry;
                        }
                }
                if (act) {
                        const int sy = sS[y][tid];
                        if ((sy - carry <= maxd) && (y < b_max - 1) && ((pc[y + 1] & ONE) || (carry < 0))) {
                                // re-open the next block with P = ~0, M = 0 and run it on this position (bpm.c:505-549)
                                y += 1;
                                unsigned long long Pv = ~0ull, Mv = 0ull, Eq = pc[y];
                                const int hin = carry;
                                int hout = 0;
                                const unsigned long long Xv = Eq | Mv;
                                if (hin < 0) Eq |= ONE;
                                const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                                unsigned long long Ph = Mv | ~(Xh | Pv);
                                unsigned long long Mh = Pv & Xh;
                                if (Ph & HIGH) hout += 1;
                                if (Mh & HIGH) hout -= 1;
                                Ph <<= 1; Mh <<= 1;
                                if (hin < 0) Mh |= ONE; else if (hin > 0) Ph |= ONE;
                                sP[y][tid] = Mh | ~(Xv | Ph);
                                sM[y][tid] = Ph & Xv;
                                sS[y][tid] = sy + 64 - carry + hout;
                        } else {
                                while (sS[y][tid] >= maxd + 64) {
                                        if (y == 0) break;
                                        y -= 1;
                                }
                        }
                        const int s2 = sS[y][tid];
                        if (s2 < k) k = s2;
                }
        }
        if (live) dist[pair] = k;
}

extern "C" void ka_launch_bpm(const uint8_t* codes, const int* off, const int* lens, int numseq, unsigned long long* peq,
                              const int* ia, const int* ib, int npairs, int* dist, hipStream_t stream)
{
        hipLaunchKernelGGL(ka_bpm_peq_kernel, dim3(numseq), dim3(256), 0, stream, codes, off, lens, peq);
        hipLaunchKernelGGL(ka_bpm_scan_kernel, dim3((npairs + KA_BPM_THREADS - 1) / KA_BPM_THREADS), dim3(KA_BPM_THREADS), 0, stream,
                           codes, off, lens, peq, ia, ib, npairs, dist);
}
