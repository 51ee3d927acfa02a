// Small byte/integer kernels either side of the DP.
//
// Aligned rows from the residue->column tables: finalise_alignment / make_linear_sequence
// (reference lib/src/msa_op.c:546-598) with the loop over residues spread over threads.  Byte work, bound by the
// N x alnlen bytes it writes; one workgroup per sequence.
#include <hip/hip_runtime.h>
#include <stdint.h>

// Thread p owns residue p of its sequence: the gap run in front of it (columns colof[p-1]+1 .. colof[p]-1), the
// letter itself, and -- the last residue -- the trailing gaps and the terminator.  Every byte of the row is
// written exactly once.
__global__ void __launch_bounds__(256) ka_rows_kernel(const uint8_t* __restrict__ letters, const int* __restrict__ off,
                                                      const int* __restrict__ lens, const int* __restrict__ colof,
                                                      const int* __restrict__ alnlen, uint8_t gap,
                                                      uint8_t* __restrict__ rows, long long stride)
{
        const int i = blockIdx.x;
        const int len = lens[i];
        const int n = alnlen[i];
        const uint8_t* s = letters + off[i];
        const int* col = colof + off[i];
        uint8_t* row = rows + (long long)i * stride;
        for (int p = threadIdx.x; p < len; p += blockDim.x) {
                const int c = col[p];
                const int first = p ? col[p - 1] + 1 : 0;
                for (int j = first; j < c; j++) row[j] = gap;
                row[c] = s[p];
                if (p == len - 1) {
                        for (int j = c + 1; j < n; j++) row[j] = gap;
                        row[n] = 0;
                }
        }
}

extern "C" void ka_launch_rows(const uint8_t* letters, const int* off, const int* lens, const int* colof, const int* alnlen,
                               int numseq, uint8_t gap, uint8_t* rows, long long stride, hipStream_t stream)
{
        hipLaunchKernelGGL(ka_rows_kernel, dim3(numseq), dim3(256), 0, stream, letters, off, lens, colof, alnlen, gap, rows, stride);
}

// Position maps of anchor consistency (reference lib/src/anchor_consistency.c:86-114): the coded path of the
// alignment (sequence i, anchor k) becomes map[p] = anchor position aligned to residue p of i, or -1.
// One wave per (i, k): 64 path elements per step, the running positions are ballot prefix counts.
// pair_of[i*K+k]: index of the pair in the batch, -1: i is the anchor itself (identity), -2: no table (all -1).
__global__ void __launch_bounds__(256) ka_posmap_kernel(const int* __restrict__ paths, const long long* __restrict__ poff,
                                                        const int* __restrict__ pair_of, const int* __restrict__ lens,
                                                        const long long* __restrict__ map_off, int n_entries, int K,
                                                        int* __restrict__ maps)
{
        const int lane = threadIdx.x & 63;
        const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (e >= n_entries) return;
        const int i = e / K, k = e - i * K;
        const int len = lens[i];
        int* map = maps + map_off[i] + (long long)k * len;
        const int pk = pair_of[e];
        if (pk < 0) {
                for (int p = lane; p < len; p += 64) map[p] = (pk == -1) ? p : -1;
                return;
        }
        const int* path = paths + poff[pk];
        const int plen = path[0];
        int pos_a = 0, pos_b = 0;                                    // wave-uniform
        for (int x0 = 1; x0 <= plen; x0 += 64) {
                const int x = x0 + lane;
                const int v = (x <= plen) ? path[x] : 3;
                const bool live = v != 3;
                const bool match = live && v == 0;
                const bool adv_b = live && (match || (v & 1));
                const bool adv_a = live && (match || (!(v & 1) && (v & 2)));
                const unsigned long long ma = __ballot(adv_a), mb = __ballot(adv_b);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int pa = pos_a + __popcll(ma & below), pb = pos_b + __popcll(mb & below);
                if (adv_a && pa < len) map[pa] = match ? pb : -1;
                pos_a += __popcll(ma);
                pos_b += __popcll(mb);
        }
        for (int p = pos_a + lane; p < len; p += 64) map[p] = -1;   // (a path that does not cover i: cannot happen)
}

extern "C" void ka_launch_posmaps(const int* paths, const long long* poff, const int* pair_of, const int* lens,
                                  const long long* map_off, int numseq, int K, int* maps, hipStream_t stream)
{
        const int n = numseq * K;
        hipLaunchKernelGGL(ka_posmap_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, paths, poff, pair_of, lens, map_off, n, K, maps);
}
