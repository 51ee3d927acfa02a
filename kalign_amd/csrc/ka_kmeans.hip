// ka_kmeans.hip -- the 2-means bisection of build_tree_kmeans on the device (round 4; SURVEY.md 8f rank 4, closes row f4).
//
// Reference: bisecting_kmeans (lib/src/bisectingKmeans.c:273-402) splits a set of sequences in two with split2 (:766-971) --
// a 2-means on the N x 32 matrix of distances to the anchors, started from one seed sample and its mirror image through the
// centroid -- tries up to 40 seeds in groups of four, keeps the split with the lowest score and recurses until a set has fewer
// than 50 members.  Everything in split2 that adds over the samples does so in SAMPLE ORDER in binary32: the centroid of the
// set (32 chains), the two new centroids of every iteration (32 chains each, over the members of the cluster in order) and the
// score (one chain).  The tree must be the reference's tree bit for bit, so these stay serial chains here too; what the
// device adds is width:
//   * a workgroup per (set, seed): all of the reference's up to 40 candidates of a set side by side (the reference stops after
//     the first group of four seeds that brings no improvement; the extra candidates are wasted work, the acceptance rule is
//     applied afterwards on the host in the reference's order and gives the same winner), and every set of a recursion level
//     in the same launch;
//   * inside a candidate, per iteration: the two distances of every sample (edist_256's eight-lane order, euclidean_dist.c)
//     and its assignment in parallel over the workgroup; an order-preserving compaction into the two member lists; then ONE
//     wave walks both lists at once -- lanes 0..31 carry the 32 chains of the left centroid, lanes 32..63 those of the right
//     one.  The score is only ever read after the last iteration: its chain runs once, at the end.
// Host side (ka_kmeans_device): level-synchronous over the recursion -- per level one launch for the centroids of the level's
// sets, one for all candidates, the acceptance rule on the host (40 scores per set), one launch that gathers the winners'
// lists into the next level's sample buffer.
#include <hip/hip_runtime.h>
#include <mutex>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ka_kmeans.h"

#define KM_PAD 32                        // floats per row of the distance matrix (32 anchors, pick_anchor.c:25)
#define KM_TRIES 40
#define KM_UPGMA_BELOW 50                // KALIGN_KMEANS_UPGMA_THRESHOLD

struct KmSet { int start, n, cand0, pad; };          // slice of the level's sample buffer; first candidate slot of the set

__device__ __forceinline__ int km_cmp(float a, float b)          // cmp_floats, bisectingKmeans.c:63-73
{
        if (fabsf(a - b) < 1e-6f) return 0;
        return a > b ? 1 : -1;
}

// edist_256 (euclidean_dist.c): eight running lane sums over i = k, k+8, k+16, k+24, then (l0+l4 + l1+l5) + (l2+l6 + l3+l7), sqrtf
__device__ __forceinline__ float km_edist(const float* __restrict__ row, const float* __restrict__ c)
{
        float lane[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) lane[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < KM_PAD; i += 8)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                        float t = row[i + k] - c[i + k];
                        t = t * t;
                        lane[k] = lane[k] + t;
                }
        const float v0 = lane[0] + lane[4], v1 = lane[1] + lane[5], v2 = lane[2] + lane[6], v3 = lane[3] + lane[7];
        const float s01 = v0 + v1, s23 = v2 + v3;
        return sqrtf(s01 + s23);
}

// A serial chain over rows that sit all over the distance matrix: fetched row by row the chain pays a trip to L2 for every
// sample (a first version did: 4 ms per 2-means iteration on a 16 384-sequence set).  So the WORKGROUP stages the rows of the
// next KM_TILE list entries in LDS (all threads, hundreds of loads in flight), then one wave adds them up in list order.
// Two shapes: 512 threads with 512-row tiles (128 KiB of LDS) for the big sets near the root, where a candidate's
// latency is the level's; 128 threads with 32-row tiles (8 KiB) for levels of hundreds of small sets times 40 candidates,
// where the number of workgroups a CU holds is what counts.

// stage rows lst[base .. base+KM_TILE) of up to two lists (side 1 optional) into tile[side][row][32]
template <int KM_TILE>
__device__ __forceinline__ void km_stage(float* tile, const float* __restrict__ dm, const int* lst0, int cnt0, const int* lst1, int cnt1, int base)
{
        const int nsides = lst1 ? 2 : 1;
        for (int x = threadIdx.x; x < nsides * KM_TILE * (KM_PAD / 4); x += blockDim.x) {
                const int side = x / (KM_TILE * (KM_PAD / 4));
                const int r = (x / (KM_PAD / 4)) % KM_TILE, q = x % (KM_PAD / 4);
                const int idx = base + r;
                const int* lst = side ? lst1 : lst0;
                const int cnt = side ? cnt1 : cnt0;
                if (idx < cnt) ((float4*)tile)[x] = ((const float4*)(dm + (size_t)lst[idx] * KM_PAD))[q];
        }
}

// The centroid of every set of the level: w[j] = (sum over the samples in order of row[j]) / n (split2, :790-800) -- a
// workgroup per set, lane j of its first wave carries chain j.
template <int KM_BLOCK, int KM_TILE>
__global__ __launch_bounds__(KM_BLOCK) void km_centroid_kernel(const float* __restrict__ dm, const int* __restrict__ samples, const KmSet* __restrict__ sets, float* __restrict__ wmean)
{
        extern __shared__ __attribute__((aligned(16))) float km_tile[];
        const KmSet S = sets[blockIdx.x];
        const int* smp = samples + S.start;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float w = 0.0f;
        for (int base = 0; base < S.n; base += KM_TILE) {
                km_stage<KM_TILE>(km_tile, dm, smp, S.n, nullptr, 0, base);
                __syncthreads();
                if (wave == 0 && lane < 32) {
                        const int m = min(KM_TILE, S.n - base);
                        int r = 0;
                        for (; r + 8 <= m; r += 8) {
                                float v[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) v[u] = km_tile[(r + u) * KM_PAD + lane];
#pragma unroll
                                for (int u = 0; u < 8; ++u) w += v[u];
                        }
                        for (; r < m; ++r) w += km_tile[r * KM_PAD + lane];
                }
                __syncthreads();
        }
        if (wave == 0 && lane < 32) wmean[(size_t)blockIdx.x * KM_PAD + lane] = w / (float)S.n;
}

// One candidate split (split2 with seed `cand * step`): blockIdx.x = the set's cand0 + cand.
// lists: per candidate slot 2 * n ints at lists[(size_t)2 * (S.start * KM_TRIES) ... ] -- see the offset below; assign / mind:
// per candidate n bytes / n floats of scratch.
template <int KM_BLOCK, int KM_TILE>
__global__ __launch_bounds__(KM_BLOCK) void km_split_kernel(const float* __restrict__ dm, const int* __restrict__ samples, const KmSet* __restrict__ sets,
                                                            const int* __restrict__ cand_set, const float* __restrict__ wmean,
                                                            int* __restrict__ lists, const long long* __restrict__ list_off,
                                                            unsigned char* __restrict__ assign_all, float* __restrict__ mind_all,
                                                            float* __restrict__ cand_score, int2* __restrict__ cand_n)
{
        const int slot = blockIdx.x;
        const int set = cand_set[slot];
        const KmSet S = sets[set];
        const int cand = slot - S.cand0;
        const int n = S.n;
        const int tries = n < KM_TRIES ? n : KM_TRIES;
        const int step = n / tries;
        const int* smp = samples + S.start;
        const long long lo = list_off[slot];                          // ints: sl at lo, sr at lo + n
        int* sl = lists + lo;
        int* sr = sl + n;
        unsigned char* asg = assign_all + lo / 2;                     // (lo = 2 * (samples before this slot): n bytes / floats per slot)
        float* mind = mind_all + lo / 2;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

        constexpr int KM_TILE_FLOATS = 2 * KM_TILE * KM_PAD;
        extern __shared__ __attribute__((aligned(16))) float km_tile[];
        __shared__ float cen[2][2][KM_PAD];                           // [buffer][left / right][j]: current centroids in cen[cur]
        __shared__ int wcount[KM_BLOCK / 64 + 1];
        __shared__ int s_moved, s_num_r;
        int cur = 0;
        if (tid < KM_PAD) {
                const float w = wmean[(size_t)set * KM_PAD + tid];
                const float cl = dm[(size_t)smp[cand * step] * KM_PAD + tid];
                cen[0][0][tid] = cl;
                cen[0][1][tid] = w - (cl - w);
        }
        __syncthreads();

        // every thread owns a contiguous chunk of the samples (the compaction keeps their order)
        const int per = (n + KM_BLOCK - 1) / KM_BLOCK;
        const int i0 = min(tid * per, n), i1 = min(i0 + per, n);
        bool degenerate = false;
        for (int stop = 0; stop < 500; ++stop) {
                // ---- distances and assignment (:843-880) ----
                float cl[KM_PAD], cr[KM_PAD];
#pragma unroll
                for (int j = 0; j < KM_PAD; ++j) { cl[j] = cen[cur][0][j]; cr[j] = cen[cur][1][j]; }
                int mine_r = 0;
                auto one = [&](const int i, const float* row) {
                        const float dl = km_edist(row, cl), dr = km_edist(row, cr);
                        const int c = km_cmp(dr, dl);
                        const int right = (c == -1 || (c == 0 && (i & 1))) ? 1 : 0;
                        asg[i] = (unsigned char)right;
                        mind[i] = (dl < dr) ? dl : dr;
                        mine_r += right;
                };
                auto fetch = [&](const int i, float* row) {
                        const float4* rp = (const float4*)(dm + (size_t)smp[i] * KM_PAD);
#pragma unroll
                        for (int q = 0; q < KM_PAD / 4; ++q) { const float4 v = rp[q]; row[4 * q] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w; }
                };
                {
                        // (two rows in flight per thread: a row is a trip to L2 behind an index load)
                        int i = i0;
                        for (; i + 2 <= i1; i += 2) {
                                float ra[KM_PAD], rb[KM_PAD];
                                fetch(i, ra); fetch(i + 1, rb);
                                one(i, ra); one(i + 1, rb);
                        }
                        if (i < i1) { float ra[KM_PAD]; fetch(i, ra); one(i, ra); }
                }
                // ---- order-preserving compaction: exclusive scan of the per-thread counts ----
                int incl = mine_r;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
                if (lane == 63) wcount[wave] = incl;
                __syncthreads();
                if (tid == 0) {
                        int run = 0;
                        for (int w = 0; w < KM_BLOCK / 64; ++w) { const int t = wcount[w]; wcount[w] = run; run += t; }
                        wcount[KM_BLOCK / 64] = run;
                        s_num_r = run;
                }
                __syncthreads();
                const int num_r = s_num_r, num_l = n - num_r;
                if (num_l == 0 || num_r == 0) { degenerate = true; break; }
                {
                        int pr = wcount[wave] + incl - mine_r;                // rights before my chunk
                        int pl = i0 - pr;                                     // lefts before my chunk
                        for (int i = i0; i < i1; ++i) {
                                const int s = smp[i];
                                if (asg[i]) sr[pr++] = s; else sl[pl++] = s;
                        }
                }
                __syncthreads();
                // ---- the new centroids: 2 x 32 chains in member order (:881-905): lanes 0..31 of the first wave the left list,
                // lanes 32..63 the right one, rows staged tile by tile by the whole workgroup ----
                {
                        const int side = lane >> 5, j = lane & 31;
                        const int cnt = side ? num_r : num_l;
                        float acc = 0.0f;
                        for (int base = 0; base < max(num_l, num_r); base += KM_TILE) {
                                km_stage<KM_TILE>(km_tile, dm, sl, num_l, sr, num_r, base);
                                __syncthreads();
                                if (wave == 0) {
                                        const float* t = km_tile + side * KM_TILE * KM_PAD;
                                        const int m = min(KM_TILE, cnt - base);
                                        int r = 0;
                                        for (; r + 8 <= m; r += 8) {
                                                float v[8];
#pragma unroll
                                                for (int u = 0; u < 8; ++u) v[u] = t[(r + u) * KM_PAD + j];
#pragma unroll
                                                for (int u = 0; u < 8; ++u) acc += v[u];
                                        }
                                        for (; r < m; ++r) acc += t[r * KM_PAD + j];
                                }
                                __syncthreads();
                        }
                        if (wave == 0) {
                                acc = acc / (float)cnt;
                                cen[cur ^ 1][side][j] = acc;
                                const bool diff = km_cmp(acc, cen[cur][side][j]) != 0;
                                const unsigned long long any = __ballot(diff);
                                if (lane == 0) s_moved = any != 0ull;
                        }
                }
                __syncthreads();
                if (!s_moved) break;
                cur ^= 1;
        }
        if (degenerate) {
                // (:906-925) no sample on one side: cut the list in the middle, score 0
                for (int i = tid; i < n; i += KM_BLOCK) { if (i < n / 2) sl[i] = smp[i]; else sr[i - n / 2] = smp[i]; }
                if (tid == 0) { cand_score[slot] = 0.0f; cand_n[slot] = make_int2(n / 2, n - n / 2); }
                return;
        }
        {
                // the score of the last iteration: one chain over the samples in order, min(dl, dr) staged through LDS
                float score = 0.0f;
                for (int base = 0; base < n; base += KM_TILE_FLOATS) {
                        __syncthreads();
                        for (int x = tid; x < KM_TILE_FLOATS && base + x < n; x += KM_BLOCK) km_tile[x] = mind[base + x];
                        __syncthreads();
                        if (tid == 0) {
                                const int m = min(KM_TILE_FLOATS, n - base);
                                int i = 0;
                                for (; i + 8 <= m; i += 8) {
                                        float v[8];
#pragma unroll
                                        for (int u = 0; u < 8; ++u) v[u] = km_tile[i + u];
#pragma unroll
                                        for (int u = 0; u < 8; ++u) score += v[u];
                                }
                                for (; i < m; ++i) score += km_tile[i];
                        }
                }
                if (tid == 0) {
                        cand_score[slot] = score;
                        cand_n[slot] = make_int2(n - s_num_r, s_num_r);
                }
        }
}

// The winners' lists become the next level's sample buffer: set k's left list at dst_off[2k], its right list at dst_off[2k+1].
__global__ void km_gather_kernel(const int* __restrict__ lists, const long long* __restrict__ src_off, const int2* __restrict__ src_n,
                                 const long long* __restrict__ dst_off, int* __restrict__ next_samples)
{
        const int k = blockIdx.x;
        const int2 nn = src_n[k];
        const int* sl = lists + src_off[k];
        const int* sr = sl + (nn.x + nn.y);
        int* dl = next_samples + dst_off[2 * k];
        int* dr = next_samples + dst_off[2 * k + 1];
        for (int i = threadIdx.x; i < nn.x; i += blockDim.x) dl[i] = sl[i];
        for (int i = threadIdx.x; i < nn.y; i += blockDim.x) dr[i] = sr[i];
}

namespace {
template <typename T>
struct Buf {
        T* p = nullptr;
        size_t n = 0;
        bool alloc(size_t count) { if (count <= n && p) return true; if (p) (void)hipFree(p); p = nullptr; n = 0; if (hipMalloc((void**)&p, (count ? count : 1) * sizeof(T)) != hipSuccess) return false; n = count; return true; }
        void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
        ~Buf() { release(); }
        Buf() = default;
        Buf(const Buf&) = delete;
        Buf& operator=(const Buf&) = delete;
};
}  // namespace

#define KM_MAX_DEVICES 64
#define KMCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return 1; } } while (0)

// dm: numseq x 32 floats (host).  nodes[0] is the root; a node with left < 0 is a leaf cluster (its members in `cluster`, in
// the reference's order).  Returns 0, or 1 with `err` set.
int ka_kmeans_device(int device, hipStream_t stream, const float* dm, int numseq, std::vector<KaKmNode>& nodes, std::string& err)
{
        KMCHK(hipSetDevice(device));
        if (device < 0 || device >= KM_MAX_DEVICES) { err = "ka_kmeans_device: device index out of range"; return 1; }
        // One pool of buffers and one LDS opt-in PER DEVICE, process-wide, used under the device's lock: the guide tree may be asked
        // for from several threads and devices at once (ka_multi_*, ensemble members side by side), the opt-in is a per-device
        // attribute, and buffers owned by a thread would be freed from its destructors -- possibly after the runtime is gone.
        // (Never freed: they live as long as the process, like the runtime they belong to.)
        static std::mutex dev_mu[KM_MAX_DEVICES];
        static bool dev_opted[KM_MAX_DEVICES];
        std::lock_guard<std::mutex> dev_lock(dev_mu[device]);
        if (!dev_opted[device]) {
                // the row tiles of the big shape take 128 KiB of dynamic LDS
                KMCHK(hipFuncSetAttribute((const void*)km_centroid_kernel<512, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * KM_PAD * sizeof(float)));
                KMCHK(hipFuncSetAttribute((const void*)km_split_kernel<512, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * KM_PAD * sizeof(float)));
                dev_opted[device] = true;
        }
        // (the buffers live as long as the process: a guide tree is built per alignment, the allocations cost more than a level)
        struct Pool {
                Buf<float> d_dm, d_wmean, d_mind, d_score;
                Buf<int> d_samples[2], d_lists, d_cand_set;
                Buf<KmSet> d_sets;
                Buf<long long> d_list_off, d_src_off, d_dst_off;
                Buf<unsigned char> d_assign;
                Buf<int2> d_cand_n, d_src_n;
        };
        static Pool* pools[KM_MAX_DEVICES];
        if (!pools[device]) pools[device] = new Pool();
        Pool& pool = *pools[device];
        Buf<float>&d_dm = pool.d_dm, &d_wmean = pool.d_wmean, &d_mind = pool.d_mind, &d_score = pool.d_score;
        Buf<int>(&d_samples)[2] = pool.d_samples; Buf<int>&d_lists = pool.d_lists, &d_cand_set = pool.d_cand_set;
        Buf<KmSet>& d_sets = pool.d_sets;
        Buf<long long>&d_list_off = pool.d_list_off, &d_src_off = pool.d_src_off, &d_dst_off = pool.d_dst_off;
        Buf<unsigned char>& d_assign = pool.d_assign;
        Buf<int2>&d_cand_n = pool.d_cand_n, &d_src_n = pool.d_src_n;
        const size_t N = (size_t)numseq;
        if (!d_dm.alloc(N * KM_PAD) || !d_samples[0].alloc(N) || !d_samples[1].alloc(N) || !d_lists.alloc(2 * N * KM_TRIES) ||
            !d_assign.alloc(N * KM_TRIES) || !d_mind.alloc(N * KM_TRIES)) { err = "hipMalloc failed (2-means bisection)"; return 1; }
        KMCHK(hipMemcpyAsync(d_dm.p, dm, sizeof(float) * N * KM_PAD, hipMemcpyHostToDevice, stream));
        std::vector<int> level_samples(N);
        for (int i = 0; i < numseq; i++) level_samples[i] = i;
        KMCHK(hipMemcpyAsync(d_samples[0].p, level_samples.data(), sizeof(int) * N, hipMemcpyHostToDevice, stream));

        const auto t_start = std::chrono::steady_clock::now();
        nodes.clear();
        nodes.push_back(KaKmNode());
        struct Open { int node, start, n; };
        std::vector<Open> open(1, Open{ 0, 0, numseq });               // the level's sets that still have to be split
        int cur = 0;
        for (int depth = 0; !open.empty(); depth++) {
                if (depth > 4096) { err = "bisecting k-means degenerated (recursion deeper than 4096)"; return 1; }
                // sets below the threshold are leaf clusters: their members come from the level's sample buffer
                std::vector<Open> split;
                bool need_samples = false;
                for (const Open& o : open) { if (o.n < KM_UPGMA_BELOW) need_samples = true; else split.push_back(o); }
                if (need_samples) {
                        KMCHK(hipMemcpyAsync(level_samples.data(), d_samples[cur].p, sizeof(int) * N, hipMemcpyDeviceToHost, stream));
                        KMCHK(hipStreamSynchronize(stream));
                        for (const Open& o : open)
                                if (o.n < KM_UPGMA_BELOW) nodes[o.node].cluster.assign(level_samples.begin() + o.start, level_samples.begin() + o.start + o.n);
                }
                if (split.empty()) break;
                const int nsets = (int)split.size();
                std::vector<KmSet> sets(nsets);
                std::vector<int> cand_set;
                std::vector<long long> list_off;
                long long before = 0;
                for (int k = 0; k < nsets; k++) {
                        const int tries = std::min(KM_TRIES, split[k].n);
                        sets[k] = KmSet{ split[k].start, split[k].n, (int)cand_set.size(), 0 };
                        for (int c = 0; c < tries; c++) { cand_set.push_back(k); list_off.push_back(2 * before); before += split[k].n; }
                }
                const int nslots = (int)cand_set.size();
                if (!d_sets.alloc(nsets) || !d_cand_set.alloc(nslots) || !d_list_off.alloc(nslots) || !d_wmean.alloc((size_t)nsets * KM_PAD) ||
                    !d_score.alloc(nslots) || !d_cand_n.alloc(nslots) || !d_src_off.alloc(nsets) || !d_src_n.alloc(nsets) || !d_dst_off.alloc(2 * (size_t)nsets)) { err = "hipMalloc failed (2-means bisection)"; return 1; }
                KMCHK(hipMemcpyAsync(d_sets.p, sets.data(), sizeof(KmSet) * nsets, hipMemcpyHostToDevice, stream));
                KMCHK(hipMemcpyAsync(d_cand_set.p, cand_set.data(), sizeof(int) * nslots, hipMemcpyHostToDevice, stream));
                KMCHK(hipMemcpyAsync(d_list_off.p, list_off.data(), sizeof(long long) * nslots, hipMemcpyHostToDevice, stream));
                int largest = 0;
                for (const Open& o : split) largest = std::max(largest, o.n);
                if (largest > 1024) {
                        hipLaunchKernelGGL((km_centroid_kernel<512, 512>), dim3(nsets), dim3(512), 2 * 512 * KM_PAD * sizeof(float), stream, d_dm.p, d_samples[cur].p, d_sets.p, d_wmean.p);
                        hipLaunchKernelGGL((km_split_kernel<512, 512>), dim3(nslots), dim3(512), 2 * 512 * KM_PAD * sizeof(float), stream, d_dm.p, d_samples[cur].p, d_sets.p, d_cand_set.p, d_wmean.p,
                                           d_lists.p, d_list_off.p, d_assign.p, d_mind.p, d_score.p, d_cand_n.p);
                } else {
                        hipLaunchKernelGGL((km_centroid_kernel<128, 32>), dim3(nsets), dim3(128), 2 * 32 * KM_PAD * sizeof(float), stream, d_dm.p, d_samples[cur].p, d_sets.p, d_wmean.p);
                        hipLaunchKernelGGL((km_split_kernel<128, 32>), dim3(nslots), dim3(128), 2 * 32 * KM_PAD * sizeof(float), stream, d_dm.p, d_samples[cur].p, d_sets.p, d_cand_set.p, d_wmean.p,
                                           d_lists.p, d_list_off.p, d_assign.p, d_mind.p, d_score.p, d_cand_n.p);
                }
                KMCHK(hipGetLastError());
                if (getenv("KA_KMEANS_VERBOSE")) {
                        KMCHK(hipStreamSynchronize(stream));
                        int mx = 0; for (const Open& o : split) mx = std::max(mx, o.n);
                        fprintf(stderr, "kmeans level %d: %d sets to split (largest %d), %d candidates, %.3f ms since start\n", depth, nsets, mx, nslots,
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
                }
                std::vector<float> score(nslots);
                std::vector<int2> cn(nslots);
                KMCHK(hipMemcpyAsync(score.data(), d_score.p, sizeof(float) * nslots, hipMemcpyDeviceToHost, stream));
                KMCHK(hipMemcpyAsync(cn.data(), d_cand_n.p, sizeof(int2) * nslots, hipMemcpyDeviceToHost, stream));
                KMCHK(hipStreamSynchronize(stream));
                // the reference's acceptance rule (:318-352): seeds in groups of four, a candidate replaces the best one on a
                // strictly lower score, the first group that changes nothing ends the search
                std::vector<long long> src_off(nsets), dst_off(2 * (size_t)nsets);
                std::vector<int2> src_n(nsets);
                std::vector<Open> next;
                long long fill = 0;
                for (int k = 0; k < nsets; k++) {
                        const int tries = std::min(KM_TRIES, split[k].n);
                        int best = -1;
                        for (int i = 0; i < tries; i += 4) {
                                int change = 0;
                                for (int j = 0; j < 4 && i + j < tries; j++) {
                                        const int c = sets[k].cand0 + i + j;
                                        if (best < 0 || score[best] > score[c]) { best = c; change++; }
                                }
                                if (!change) break;
                        }
                        src_off[k] = list_off[best]; src_n[k] = cn[best];
                        const int nl = cn[best].x, nr = cn[best].y;
                        const int ln = (int)nodes.size();
                        nodes.push_back(KaKmNode()); nodes.push_back(KaKmNode());
                        nodes[split[k].node].left = ln; nodes[split[k].node].right = ln + 1;
                        dst_off[2 * k] = fill; next.push_back(Open{ ln, (int)fill, nl }); fill += nl;
                        dst_off[2 * k + 1] = fill; next.push_back(Open{ ln + 1, (int)fill, nr }); fill += nr;
                }
                KMCHK(hipMemcpyAsync(d_src_off.p, src_off.data(), sizeof(long long) * nsets, hipMemcpyHostToDevice, stream));
                KMCHK(hipMemcpyAsync(d_src_n.p, src_n.data(), sizeof(int2) * nsets, hipMemcpyHostToDevice, stream));
                KMCHK(hipMemcpyAsync(d_dst_off.p, dst_off.data(), sizeof(long long) * 2 * nsets, hipMemcpyHostToDevice, stream));
                hipLaunchKernelGGL(km_gather_kernel, dim3(nsets), dim3(256), 0, stream, d_lists.p, d_src_off.p, d_src_n.p, d_dst_off.p, d_samples[cur ^ 1].p);
                KMCHK(hipGetLastError());
                KMCHK(hipStreamSynchronize(stream));                    // (the host vectors above are about to go out of scope)
                cur ^= 1;
                open.swap(next);
        }
        return 0;
}
