// Small byte/integer kernels either side of the DP.
//
// Aligned rows from the residue->column tables: finalise_alignment / make_linear_sequence
// (reference lib/src/msa_op.c:546-598) with the loop over residues spread over threads.  Byte work, bound by the
// N x alnlen bytes it writes; one workgroup per sequence.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kalign_amd.h"

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
// pair_of[i*K+k]: index of the pair in the batch, -1: i is the anchor itself (identity), -2: no table (all -1),
// -3: not this rank's part of a sharded table (left alone).
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
        if (pk == -3) return;                                        // another rank's share of a sharded table
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

// upgma (lib/src/bisectingKmeans.c:974-1053), one launch per merge.  The reference scans all active pairs for the
// smallest dm[i][j], i < j, taking the first in row-major order on ties (strict '<'); here every active row keeps its
// own minimum as a 64-bit key (value bits << 32 | i * n + j: distances are >= 0, so keys order like (value, i, j)),
// and a merge costs O(n) instead of O(n^2):
//   A  every workgroup reduces the n row keys of the previous step to the pair (a, b), a < b;
//   B  it then updates its own rows: row i folds column b into column a from its own symmetric copies,
//      dm[i][a] = (dm[i][a] + dm[i][b]) * 0.5 + 0.001 -- the value the reference writes to dm[a][i] and mirrors --,
//      row a is rebuilt from row b (not written in this step), and a row key is rescanned only when its minimum sat in
//      column a or b.  Keys are double-buffered: phase A of slow workgroups still reads the old ones.
struct KaUpgma { float* dm; int* active; unsigned long long* key[2]; int2* merges; int n; };

__device__ __forceinline__ unsigned long long ka_upgma_key(float v, unsigned int idx)
{
        return ((unsigned long long)__float_as_uint(v) << 32) | idx;
}

__device__ __forceinline__ unsigned long long ka_block_min(unsigned long long v, unsigned long long* red)
{
        const int tid = threadIdx.x;
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
                if (tid < s) red[tid] = red[tid + s] < red[tid] ? red[tid + s] : red[tid];
                __syncthreads();
        }
        return red[0];
}

// key of row i over the active columns j > i (block-wide); d0, d1: columns merged away in this and the previous step,
// whose flags are not down yet
__device__ __forceinline__ unsigned long long ka_upgma_scan_row(const KaUpgma& U, int i, int d0, int d1, unsigned long long* red)
{
        const int n = U.n;
        const float* row = U.dm + (long long)i * n;
        unsigned long long best = ~0ull;
        for (int j = i + 1 + threadIdx.x; j < n; j += 256)
                if (j != d0 && j != d1 && U.active[j]) {
                        const unsigned long long k = ka_upgma_key(row[j], (unsigned int)(i * n + j));
                        best = k < best ? k : best;
                }
        return ka_block_min(best, red);
}

__global__ void __launch_bounds__(256) ka_upgma_init_kernel(KaUpgma U)
{
        __shared__ unsigned long long red[256];
        for (int i = blockIdx.x; i < U.n; i += gridDim.x) {
                const unsigned long long k = ka_upgma_scan_row(U, i, -1, -1, red);
                if (threadIdx.x == 0) U.key[0][i] = k;
        }
}

__device__ __forceinline__ void ka_upgma_step(const KaUpgma& U, const int step, unsigned long long* red)
{
        const int n = U.n, tid = threadIdx.x;
        const unsigned long long* kin = U.key[step & 1];
        unsigned long long* kout = U.key[(step + 1) & 1];
        // A: the pair to merge
        unsigned long long best = ~0ull;
        for (int i = tid; i < n; i += 256) {
                const unsigned long long k = __hip_atomic_load(&kin[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                best = k < best ? k : best;
        }
        best = ka_block_min(best, red);
        const unsigned int idx = (unsigned int)(best & 0xffffffffull);
        const int a = (int)(idx / (unsigned int)n), b = (int)(idx % (unsigned int)n);
        if (blockIdx.x == 0 && tid == 0) U.merges[step] = make_int2(a, b);
        // the row merged away one step ago: its flag goes down in this launch (below), so nobody may rely on it yet
        const int bprev = step ? U.merges[step - 1].y : -1;
        // B: this workgroup's rows
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
                if (i == b || i == bprev || !U.active[i]) { if (tid == 0) kout[i] = ~0ull; continue; }
                float* row = U.dm + (long long)i * n;
                if (i == a) {
                        const float* rb = U.dm + (long long)b * n;
                        for (int j = tid; j < n; j += 256)
                                if (j != b) row[j] = (j == a) ? 0.0f : (row[j] + rb[j]) * 0.5f + 0.001f;
                        __syncthreads();                               // the scan below reads what other threads wrote
                        const unsigned long long k = ka_upgma_scan_row(U, a, b, bprev, red);
                        if (tid == 0) kout[a] = k;
                        continue;
                }
                const unsigned long long k = kin[i];
                const int jmin = (int)((unsigned int)(k & 0xffffffffull) % (unsigned int)n);
                float via = 0.0f;
                if (tid == 0) { via = (row[a] + row[b]) * 0.5f + 0.001f; row[a] = via; }
                const bool rescan = (k != ~0ull) && (jmin == a || jmin == b) && i < b;
                if (rescan) {
                        __syncthreads();                               // row[a] is part of the scan when i < a
                        const unsigned long long k2 = ka_upgma_scan_row(U, i, b, bprev, red);
                        if (tid == 0) kout[i] = k2;
                } else if (tid == 0) {
                        unsigned long long k2 = k;
                        if (i < a) { const unsigned long long c = ka_upgma_key(via, (unsigned int)(i * n + a)); k2 = c < k2 ? c : k2; }
                        kout[i] = k2;
                }
        }
        // readers of this step treat bprev as gone whatever its flag says; from the next step on the flag is down
        if (blockIdx.x == 0 && tid == 0 && bprev >= 0) U.active[bprev] = 0;
}

__global__ void __launch_bounds__(256) ka_upgma_step_kernel(KaUpgma U, int step)
{
        __shared__ unsigned long long red[256];
        ka_upgma_step(U, step, red);
}

extern "C" void ka_launch_aln_dist(const uint8_t* rows, long long stride, int alnlen, int n, uint8_t gap, float* dm, float* means,
                                   hipStream_t stream)
{
        const int t = (n + KA_AD_TILE - 1) / KA_AD_TILE;
        hipLaunchKernelGGL(ka_aln_dist_kernel, dim3(t, t), dim3(256), 0, stream, rows, stride, alnlen, n, gap, dm);
        hipLaunchKernelGGL(ka_row_mean_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dm, n, means);
}

// ---- all n - 1 merges in ONE launch of ONE workgroup ----
// The merges depend on each other, so the loop is a latency chain: per launch it costs the dependent dispatch (~11 us),
// per cluster-wide barrier in HBM a release/acquire pair across the XCDs' L2s (~21 us, measured in round 2).  One
// workgroup of 8 waves needs neither: its waves share one L1 and meet at s_barrier.  Row keys, activity flags and the
// rescan list live in LDS; the matrix is kept symmetric, so the two COLUMNS a step reads (a and b, a stride-n walk)
// are read as ROWS a and b (coalesced).  One step = LDS work + ONE round trip to the matrix:
//   A  all threads reduce the n row keys to the pair (a, b)                                             [LDS]
//   L  thread i lists row i for a rescan if its minimum sat in column a or b                            [LDS]
//   B  loads in flight together: rows a and b (thread i: elements i), and for each listed row its columns > i.
//      v = (dm[a][i] + dm[b][i]) * 0.5 + 0.001 (the operands and their order are the reference's
//      dm[i][a] + dm[i][b], bisectingKmeans.c:1027-1040) goes to dm[a][i] (coalesced) and to its mirror dm[i][a]
//      (the step's only strided access, a store nobody waits for until the next step's loads); row i's key takes
//      the new candidate (i < a); row a's key is the minimum over the v of the columns > a
//   C  one wave per listed row: the key from the columns loaded in B, with v in place of column a.
// Barriers between the phases wait for LDS only; the one in front of B also waits for the previous step's stores.
#define KA_UPGMA_NT 512
#define KA_UPGMA_ONE_WG_MAX 6144
#define KA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long ka_dpp_min_u64(const unsigned long long v)
{
        // (lanes without a source, or in a masked row, keep the identity)
        const unsigned int olo = (unsigned int)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)(unsigned int)v, CTRL, ROW_MASK, 0xf, false);
        const unsigned int ohi = (unsigned int)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)(unsigned int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
        const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
        return o < v ? o : v;
}
// minimum over the wave, in every lane (VALU only: a butterfly of 64-bit shuffles is 12 trips through the LDS crossbar)
__device__ __forceinline__ unsigned long long ka_wave_min_u64(unsigned long long v)
{
        v = ka_dpp_min_u64<0x111, 0xf>(v);                            // row_shr:1
        v = ka_dpp_min_u64<0x112, 0xf>(v);                            // row_shr:2
        v = ka_dpp_min_u64<0x114, 0xf>(v);                            // row_shr:4
        v = ka_dpp_min_u64<0x118, 0xf>(v);                            // row_shr:8   -> lane 15 of every row: the row's minimum
        v = ka_dpp_min_u64<0x142, 0xa>(v);                            // row_bcast:15 into rows 1 and 3
        v = ka_dpp_min_u64<0x143, 0xc>(v);                            // row_bcast:31 into rows 2 and 3 -> lane 63: everything
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, 63);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), 63);
        return ((unsigned long long)hi << 32) | lo;
}

// PER: rows per thread (n <= PER * KA_UPGMA_NT); SV * 64: columns of a listed row per round trip
template <int PER, int SV>
__global__ void __launch_bounds__(KA_UPGMA_NT) ka_upgma_one_wg_kernel(KaUpgma U)
{
        extern __shared__ __attribute__((aligned(16))) unsigned char ka_upgma_lds[];
        __shared__ unsigned long long red[KA_UPGMA_NT / 64], red_a[KA_UPGMA_NT / 64];
        __shared__ int nlist;
        const int n = U.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        constexpr int nw = KA_UPGMA_NT / 64;
        unsigned long long* keys = (unsigned long long*)ka_upgma_lds;  // n row keys
        int* list = (int*)(keys + n);                                 // rows to rescan in this step
        float* listv = (float*)(list + n);                            // [row] their new value in column a
        unsigned char* act = (unsigned char*)(listv + n);             // [row] still a cluster of its own
        for (int i = tid; i < n; i += KA_UPGMA_NT) act[i] = U.active[i] != 0;
        __syncthreads();
        // key of row i from the columns in sv (loaded by the caller) and, beyond SV * 64 columns, from the matrix
        auto row_key = [&](const int i, const float* sv, const int a) {
                const float* row = U.dm + (long long)i * n;
                unsigned long long best = ~0ull;
                // (flags first, all reads in flight together: a flag read inside a branch costs an LDS round trip per column)
                unsigned char ac[SV];
#pragma unroll
                for (int u = 0; u < SV; ++u) { const int j = i + 1 + lane + 64 * u; ac[u] = act[j < n ? j : n - 1]; }
                const float lv = a >= 0 ? listv[i] : 0.0f;
#pragma unroll
                for (int u = 0; u < SV; ++u) {
                        const int j = i + 1 + lane + 64 * u;
                        const unsigned long long k = ka_upgma_key(j == a ? lv : sv[u], (unsigned int)(i * n + j));
                        const unsigned long long kk = (j < n && ac[u]) ? k : ~0ull;
                        best = kk < best ? kk : best;
                }
                for (int j0 = i + 1 + lane + 64 * SV; j0 < n; j0 += 64 * SV) {
                        float v[SV];
#pragma unroll
                        for (int u = 0; u < SV; ++u) { const int j = j0 + 64 * u; v[u] = row[j < n ? j : n - 1]; }
#pragma unroll
                        for (int u = 0; u < SV; ++u) { const int j = j0 + 64 * u; ac[u] = act[j < n ? j : n - 1]; }
#pragma unroll
                        for (int u = 0; u < SV; ++u) {
                                const int j = j0 + 64 * u;
                                const unsigned long long k = ka_upgma_key(j == a ? lv : v[u], (unsigned int)(i * n + j));
                                const unsigned long long kk = (j < n && ac[u]) ? k : ~0ull;
                                best = kk < best ? kk : best;
                        }
                }
                best = ka_wave_min_u64(best);
                if (lane == 0) keys[i] = best;
        };
        auto row_load = [&](const int i, float* sv) {
                const float* row = U.dm + (long long)i * n;
#pragma unroll
                for (int u = 0; u < SV; ++u) { const int j = i + 1 + lane + 64 * u; sv[u] = row[j < n ? j : n - 1]; }
        };
        for (int i = wave; i < n; i += nw) { float sv[SV]; row_load(i, sv); row_key(i, sv, -1); }
        __syncthreads();
        for (int step = 0; step < n - 1; ++step) {
                // A
                unsigned long long best = ~0ull;
                for (int i = tid; i < n; i += KA_UPGMA_NT) { const unsigned long long k = keys[i]; best = k < best ? k : best; }
                best = ka_wave_min_u64(best);
                if (lane == 0) red[wave] = best;
                if (tid == 0) nlist = 0;
                KA_LDS_BARRIER();
                best = ka_wave_min_u64(red[lane & (nw - 1)]);
                const unsigned int idx = (unsigned int)(best & 0xffffffffull);
                const int a = (int)(idx / (unsigned int)n), b = (int)(idx % (unsigned int)n);
                // L
                unsigned long long kq[PER]; unsigned char aq[PER];
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                        const int i = tid + u * KA_UPGMA_NT, ii = i < n ? i : n - 1;
                        kq[u] = keys[ii]; aq[u] = act[ii];
                }
                bool listed[PER];
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                        const int i = tid + u * KA_UPGMA_NT;
                        const int jmin = (int)(unsigned int)(kq[u] & 0xffffffffull) - i * n;   // (row i's key: index i * n + j)
                        listed[u] = i < n && i != a && i != b && aq[u] && kq[u] != ~0ull && (jmin == a || jmin == b);
                        if (listed[u]) list[atomicAdd(&nlist, 1)] = i;
                }
                if (tid == 0) { U.merges[step] = make_int2(a, b); act[b] = 0; keys[b] = ~0ull; }
                __syncthreads();                                      // (also: the previous step's stores have landed)
                // B
                const int nl = nlist;
                float sv[SV];
                if (wave < nl) row_load(list[wave], sv);
                float* ra = U.dm + (long long)a * n;
                const float* rb = U.dm + (long long)b * n;
                float va[PER], vb[PER];
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                        const int i = tid + u * KA_UPGMA_NT, ii = i < n ? i : n - 1;
                        va[u] = ra[ii]; vb[u] = rb[ii];
                }
                unsigned long long best_a = ~0ull;
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                        const int i = tid + u * KA_UPGMA_NT;
                        if (i >= n || i == b) continue;
                        if (i == a) { ra[a] = 0.0f; continue; }
                        const float v = (va[u] + vb[u]) * 0.5f + 0.001f;
                        ra[i] = v;
                        if (!aq[u]) continue;
                        U.dm[(long long)i * n + a] = v;
                        if (i > a) {
                                const unsigned long long c = ka_upgma_key(v, (unsigned int)(a * n + i));
                                best_a = c < best_a ? c : best_a;
                        }
                        if (listed[u]) {
                                listv[i] = v;
                        } else if (i < a) {
                                const unsigned long long c = ka_upgma_key(v, (unsigned int)(i * n + a));
                                keys[i] = c < kq[u] ? c : kq[u];
                        }
                }
                best_a = ka_wave_min_u64(best_a);
                if (lane == 0) red_a[wave] = best_a;
                KA_LDS_BARRIER();
                // C
                if (wave == 0) {
                        const unsigned long long ka = ka_wave_min_u64(red_a[lane & (nw - 1)]);
                        if (lane == 0) keys[a] = ka;
                }
                for (int x = wave; x < nl; x += nw) {
                        if (x != wave) row_load(list[x], sv);
                        row_key(list[x], sv, a);
                }
                KA_LDS_BARRIER();
        }
}

// ---- launching the n - 1 dependent merge steps, one launch per step ----
// ~11 us each, of which the step's own work is a few.  Measured alternatives on MI355X, both bit-identical and neither
// faster: the loop replayed as one hipGraph of n kernel nodes (12 us per node: the cost is the dependent dispatch on the
// GPU, not the host-side launch), and one persistent kernel of 64 workgroups with a barrier in HBM between steps (21 us
// per step).  keys: 2 * n words, active: n ones, merges: n - 1 pairs.
// mode 0: one workgroup for all merges when n <= KA_UPGMA_ONE_WG_MAX; 1: always one launch per merge (KA_UPGMA_LAUNCHES=1)
extern "C" void ka_launch_upgma(float* dm, int* active, unsigned long long* keys, int2* merges, int n, int mode, hipStream_t stream)
{
        KaUpgma U{ dm, active, { keys, keys + n }, merges, n };
        if (mode == 0 && n <= KA_UPGMA_ONE_WG_MAX) {
                const int lds = n * 17 + 16;                          // keys, list, listv, act
                // (the opt-in to more than 64 KB of LDS can fail -- another GPU, a smaller limit: then the per-merge launches below)
                auto go = [&](auto kernel) -> bool {
                        if (lds > 65536 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KA_UPGMA_ONE_WG_MAX * 17 + 16) != hipSuccess) {
                                (void)hipGetLastError();
                                return false;
                        }
                        hipLaunchKernelGGL(kernel, dim3(1), dim3(KA_UPGMA_NT), lds, stream, U);
                        return true;
                };
                bool ok;
                if (n <= KA_UPGMA_NT) ok = go(ka_upgma_one_wg_kernel<1, 8>);
                else if (n <= 2 * KA_UPGMA_NT) ok = go(ka_upgma_one_wg_kernel<2, 16>);
                else if (n <= 4 * KA_UPGMA_NT) ok = go(ka_upgma_one_wg_kernel<4, 32>);
                else if (n <= 8 * KA_UPGMA_NT) ok = go(ka_upgma_one_wg_kernel<8, 32>);
                else ok = go(ka_upgma_one_wg_kernel<KA_UPGMA_ONE_WG_MAX / KA_UPGMA_NT, 32>);
                if (ok) return;
        }
        const int blocks = n < 256 ? n : 256;
        hipLaunchKernelGGL(ka_upgma_init_kernel, dim3(blocks), dim3(256), 0, stream, U);
        for (int step = 0; step < n - 1; ++step)
                hipLaunchKernelGGL(ka_upgma_step_kernel, dim3(blocks), dim3(256), 0, stream, U, step);
}

// ------------------------------------------------------------------------------------------------
// Helpers of the sharded tree (ka_dist_*, ka_api.cpp): all byte / index work, one launch each.
// ------------------------------------------------------------------------------------------------
// The residue -> column table of a subtree's member sequences (scattered over `colof` like the sequences are), packed
// into one contiguous buffer for the hand-over to another GPU, or unpacked from it.  One workgroup per member.
__global__ void __launch_bounds__(256) ka_cols_pack_kernel(int* __restrict__ colof, const int* __restrict__ seq_off, const int* __restrict__ seq_len,
                                                           const int* __restrict__ members, const long long* __restrict__ moff, int* __restrict__ buf,
                                                           const int unpack)
{
        const int m = members[blockIdx.x];
        int* tbl = colof + seq_off[m];
        int* b = buf + moff[blockIdx.x];
        const int n = seq_len[m];
        if (unpack) for (int i = threadIdx.x; i < n; i += blockDim.x) tbl[i] = b[i];
        else for (int i = threadIdx.x; i < n; i += blockDim.x) b[i] = tbl[i];
}
extern "C" void ka_launch_cols_pack(int* colof, const int* seq_off, const int* seq_len, const int* members, const long long* moff, int nmem,
                                    int* buf, int unpack, hipStream_t stream)
{
        if (nmem > 0) hipLaunchKernelGGL(ka_cols_pack_kernel, dim3(nmem), dim3(256), 0, stream, colof, seq_off, seq_len, members, moff, buf, unpack);
}

// plen + 2 of every task this rank ran (0 for the others): what the all-reduce turns into every rank's path layout
__global__ void __launch_bounds__(256) ka_path_counts_kernel(const ka_task_rec* __restrict__ recs, const char* __restrict__ mine, int n_tasks, int* __restrict__ counts)
{
        const int t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t < n_tasks) counts[t] = mine[t] ? recs[t].plen + 2 : 0;
}
// this rank's coded paths from its own arena into the job-wide layout (goff[t], task order); other ranks' ranges stay zero
__global__ void __launch_bounds__(256) ka_path_scatter_kernel(const ka_task_rec* __restrict__ recs, const char* __restrict__ mine,
                                                              const int* __restrict__ arena, const long long* __restrict__ goff, int* __restrict__ out)
{
        const int t = blockIdx.x;
        if (!mine[t]) return;
        const int n = recs[t].plen + 2;
        const int* src = arena + recs[t].path_off;
        int* dst = out + goff[t];
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
extern "C" void ka_launch_path_counts(const ka_task_rec* recs, const char* mine, int n_tasks, int* counts, hipStream_t stream)
{
        hipLaunchKernelGGL(ka_path_counts_kernel, dim3((n_tasks + 255) / 256), dim3(256), 0, stream, recs, mine, n_tasks, counts);
}
extern "C" void ka_launch_path_scatter(const ka_task_rec* recs, const char* mine, int n_tasks, const int* arena, const long long* goff, int* out, hipStream_t stream)
{
        hipLaunchKernelGGL(ka_path_scatter_kernel, dim3(n_tasks), dim3(256), 0, stream, recs, mine, arena, goff, out);
}
