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

// ---- realignment: a guide tree from the finished alignment (reference lib/src/aln_wrap.c:449-495) ----
//
// compute_aln_pairwise_dist (lib/src/aln_apair_dist.c:9-86): for every pair of rows the columns where both have a
// residue and those where the residues are equal; distance = 1 - matches / aligned (1 when nothing is aligned).
// One workgroup per 16 x 16 tile of pairs (upper triangle), 128 columns of both row sets staged in LDS per step;
// rows padded to 33 words so that the 16 rows a wave touches at one column sit in different banks.
#define KA_AD_TILE 16
#define KA_AD_COLS 128
__global__ void __launch_bounds__(256) ka_aln_dist_kernel(const uint8_t* __restrict__ rows, long long stride, int alnlen, int n,
                                                          uint8_t gap, float* __restrict__ dm)
{
        if (blockIdx.x < blockIdx.y) return;                         // tile (y = i tile, x = j tile), j tile >= i tile
        __shared__ uint32_t A[KA_AD_TILE][KA_AD_COLS / 4 + 1], B[KA_AD_TILE][KA_AD_COLS / 4 + 1];
        const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
        const int i = blockIdx.y * KA_AD_TILE + ti, j = blockIdx.x * KA_AD_TILE + tj;
        int matches = 0, aligned = 0;
        for (int c0 = 0; c0 < alnlen; c0 += KA_AD_COLS) {
                // 256 threads x 2 words: row r = tid / 16 of each tile, words (tid % 16) and (tid % 16) + 16
                for (int h = 0; h < 2; ++h) {
                        const int w = tj + 16 * h;
                        uint32_t va = 0, vb = 0;
                        const int ra = blockIdx.y * KA_AD_TILE + ti, rb = blockIdx.x * KA_AD_TILE + ti;
                        for (int b = 0; b < 4; ++b) {                // bytes past the row's end read as gaps
                                const int c = c0 + 4 * w + b;
                                const uint32_t ca = (ra < n && c < alnlen) ? rows[(long long)ra * stride + c] : gap;
                                const uint32_t cb = (rb < n && c < alnlen) ? rows[(long long)rb * stride + c] : gap;
                                va |= ca << (8 * b);
                                vb |= cb << (8 * b);
                        }
                        A[ti][w] = va;
                        B[ti][w] = vb;
                }
                __syncthreads();
#pragma unroll 4
                for (int w = 0; w < KA_AD_COLS / 4; ++w) {
                        const uint32_t a = A[ti][w], b = B[tj][w];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                                const uint32_t x = (a >> (8 * k)) & 0xffu, y = (b >> (8 * k)) & 0xffu;
                                const bool both = (x != gap) && (y != gap);
                                aligned += both;
                                matches += both && (x == y);
                        }
                }
                __syncthreads();
        }
        if (i < n && j < n && i < j) {
                const float d = (aligned == 0) ? 1.0f : 1.0f - (float)matches / (float)aligned;
                dm[(long long)i * n + j] = d;
                dm[(long long)j * n + i] = d;
        }
        if (i < n && j < n && i == j) dm[(long long)i * n + i] = 0.0f;
}

// build_tree_from_pairwise (lib/src/bisectingKmeans.c:1150-1200): mean distance of every sequence to the others,
// summed in column order like the reference, before UPGMA overwrites the matrix
__global__ void ka_row_mean_kernel(const float* __restrict__ dm, int n, float* __restrict__ out)
{
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        // the matrix is symmetric: walk column i (coalesced across the threads of a wave) instead of row i -- the same
        // values in the same order
        float sum = 0.0f;
        for (int j = 0; j < n; ++j)
                if (j != i) sum += dm[(long long)j * n + i];
        out[i] = (n > 1) ? sum / (float)(n - 1) : 0.0f;
}

// upgma (lib/src/bisectingKmeans.c:974-1053), one launch per merge: every workgroup scans a band of rows for the
// smallest active dm[i][j], i < j (ties: the first in row-major order, the reference's strict '<' scan); the last
// workgroup to finish reduces the candidates, records the pair and folds row/column b into a:
// dm[a][j] = (dm[a][j] + dm[b][j]) * 0.5 + 0.001 for every j != b.
struct KaUpgma { float* dm; int* active; unsigned long long* cand; unsigned int* done; int2* merges; int n; };

__device__ __forceinline__ unsigned long long ka_upgma_key(float v, unsigned int idx)
{
        // distances are >= 0: their bit patterns order like the values
        return ((unsigned long long)__float_as_uint(v) << 32) | idx;
}

__global__ void __launch_bounds__(256) ka_upgma_step_kernel(KaUpgma U, int step)
{
        __shared__ unsigned long long red[256];
        __shared__ int s_last, s_a, s_b;
        const int n = U.n, tid = threadIdx.x;
        unsigned long long best = ~0ull;
        for (int i = blockIdx.x; i < n - 1; i += gridDim.x) {
                if (!U.active[i]) continue;
                const float* row = U.dm + (long long)i * n;
                for (int j = i + 1 + tid; j < n; j += 256)
                        if (U.active[j]) {
                                const unsigned long long k = ka_upgma_key(row[j], (unsigned int)(i * n + j));
                                best = k < best ? k : best;
                        }
        }
        red[tid] = best;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
                if (tid < s) red[tid] = red[tid + s] < red[tid] ? red[tid + s] : red[tid];
                __syncthreads();
        }
        if (tid == 0) {
                U.cand[blockIdx.x] = red[0];
                __threadfence();
                s_last = (atomicAdd(U.done, 1u) == gridDim.x - 1);
        }
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        best = ~0ull;
        for (int b = tid; b < (int)gridDim.x; b += 256) {
                const unsigned long long k = __hip_atomic_load(&U.cand[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                best = k < best ? k : best;
        }
        red[tid] = best;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
                if (tid < s) red[tid] = red[tid + s] < red[tid] ? red[tid + s] : red[tid];
                __syncthreads();
        }
        if (tid == 0) {
                const unsigned int idx = (unsigned int)(red[0] & 0xffffffffull);
                s_a = (int)(idx / (unsigned int)n); s_b = (int)(idx % (unsigned int)n);
                U.merges[step] = make_int2(s_a, s_b);
                U.active[s_b] = 0;
                *U.done = 0u;
        }
        __syncthreads();
        const int a = s_a, b = s_b;
        float* ra = U.dm + (long long)a * n;
        const float* rb = U.dm + (long long)b * n;
        for (int j = tid; j < n; j += 256)
                if (j != b) ra[j] = (ra[j] + rb[j]) * 0.5f + 0.001f;
        __syncthreads();
        if (tid == 0) ra[a] = 0.0f;
        __syncthreads();
        for (int j = tid; j < n; j += 256) U.dm[(long long)j * n + a] = ra[j];
}

extern "C" void ka_launch_aln_dist(const uint8_t* rows, long long stride, int alnlen, int n, uint8_t gap, float* dm, float* means,
                                   hipStream_t stream)
{
        const int t = (n + KA_AD_TILE - 1) / KA_AD_TILE;
        hipLaunchKernelGGL(ka_aln_dist_kernel, dim3(t, t), dim3(256), 0, stream, rows, stride, alnlen, n, gap, dm);
        hipLaunchKernelGGL(ka_row_mean_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dm, n, means);
}

// cand: room for 256 candidates, done: one zeroed counter, active: n ones, merges: n - 1 pairs
extern "C" void ka_launch_upgma(float* dm, int* active, unsigned long long* cand, unsigned int* done, int2* merges, int n,
                                hipStream_t stream)
{
        KaUpgma U{ dm, active, cand, done, merges, n };
        const int blocks = n < 512 ? 32 : 256;
        for (int step = 0; step < n - 1; ++step)
                hipLaunchKernelGGL(ka_upgma_step_kernel, dim3(blocks), dim3(256), 0, stream, U, step);
}
