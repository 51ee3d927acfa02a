// ka_cons.cpp -- anchor consistency on the device (round 5: out of ka_api.cpp): anchor_consistency_build (lib/src/anchor_consistency.c:122-275:
// anchor selection on the host, the N x K seq-seq batch and the position maps on the device, whole or one rank's part), the seq-seq pair
// batch as a call of its own (pairwise_align_map, :19-120), and the distance batch of the guide tree (calc_distance / bpm_block,
// lib/src/sequence_distance.c:37-162).
#include "ka_ctx.h"

// residue -> column tables and member lists on the device (consistency votes, device-side gap arrays)
int setup_colof(ka_ctx* c)
{
        const int N = c->numseq;
        std::vector<int> ident((size_t)c->h_codes.size(), 0);
        for (int i = 0; i < N; i++) for (int p = 0; p < c->lens[i]; p++) ident[(size_t)c->off[i] + p] = p;
        if (c->d_colof.alloc(ident.size()) || c->d_colof_init.alloc(ident.size()) || c->d_sip.alloc(c->sip_flat.size()) ||
            c->d_sip_off.alloc(c->sip_off.size()))
                return fail("hipMalloc failed");
        c->colof_n = ident.size();
        HIPCHK(hipMemcpy(c->d_colof_init.p, ident.data(), sizeof(int) * ident.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_sip.p, c->sip_flat.data(), sizeof(int) * c->sip_flat.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_sip_off.p, c->sip_off.data(), sizeof(long long) * c->sip_off.size(), hipMemcpyHostToDevice));
        c->have_colof = true;
        return KA_OK;
}

// ---- anchor consistency: anchor_consistency_build (anchor_consistency.c:122-275) ----
// Anchor selection (farthest-first over |seq_distances[i] - seq_distances[anchor]|) runs on the host, the
// N x K seq-seq alignments on the device (ka_pairwise_batch), the paths become position maps on the host
// (:86-114) and the maps go back to HBM for the per-task bonus construction inside the task kernels.
static void select_anchors(const std::vector<float>& dist, int K, std::vector<int>& ids)
{
        const int N = (int)dist.size();
        std::vector<float> min_dist(N);
        float sum = 0.0f;
        for (int i = 0; i < N; i++) sum += dist[i];
        const float mean = sum / (float)N;
        float best_diff = 3.402823466e+38f;
        int best = 0;
        for (int i = 0; i < N; i++) {
                float diff = dist[i] - mean;
                if (diff < 0) diff = -diff;
                if (diff < best_diff) { best_diff = diff; best = i; }
        }
        ids.assign(K, 0);
        ids[0] = best;
        for (int i = 0; i < N; i++) {
                float d = dist[i] - dist[ids[0]];
                if (d < 0) d = -d;
                min_dist[i] = d;
        }
        for (int k = 1; k < K; k++) {
                float best_min = -1.0f;
                best = 0;
                for (int i = 0; i < N; i++) {
                        bool skip = false;
                        for (int j = 0; j < k; j++) if (ids[j] == i) { skip = true; break; }
                        if (skip) continue;
                        if (min_dist[i] > best_min) { best_min = min_dist[i]; best = i; }
                }
                ids[k] = best;
                for (int i = 0; i < N; i++) {
                        float d = dist[i] - dist[best];
                        if (d < 0) d = -d;
                        if (d < min_dist[i]) min_dist[i] = d;
                }
        }
}

// Sequences [lo, hi) of part `part` of `nparts`: contiguous ranges with balanced total length (every sequence is
// aligned to the same K anchors, so a sequence's share of the N x K batch is proportional to its length).
static void cons_part_seqs(const ka_ctx* c, int part, int nparts, int* lo, int* hi)
{
        const int N = c->numseq;
        auto cut = [&](int r) -> int {
                if (r <= 0) return 0;
                if (r >= nparts) return N;
                const long long target = c->sum_len * (long long)r / nparts;
                long long acc = 0;
                int i = 0;
                while (i < N && acc < target) acc += c->lens[i++];
                return i;
        };
        *lo = cut(part); *hi = cut(part + 1);
}

extern "C" int ka_tree_build_consistency(ka_ctx* c, int n_anchors, float weight)
{
        return ka_tree_build_consistency_part(c, n_anchors, weight, 0, 1);
}

extern "C" int ka_tree_consistency_part_range(ka_ctx* c, int part, int nparts, long long* lo, long long* hi)
{
        if (!c || !c->have_job || c->cons_K <= 0) return fail("no consistency table on this context");
        if (nparts < 1 || part < 0 || part >= nparts || !lo || !hi) return fail("bad part");
        int s0, s1;
        cons_part_seqs(c, part, nparts, &s0, &s1);
        *lo = s0 < c->numseq ? c->cons_map_off[s0] : c->cons_maps_total;
        *hi = s1 < c->numseq ? c->cons_map_off[s1] : c->cons_maps_total;
        return KA_OK;
}

extern "C" int ka_tree_consistency_maps_dev(ka_ctx* c, void** maps_dev, long long* total_ints)
{
        if (!c || !c->have_job || c->cons_K <= 0) return fail("no consistency table on this context");
        if (maps_dev) *maps_dev = c->d_cons_maps.p;
        if (total_ints) *total_ints = c->cons_maps_total;
        c->cons_maps.clear();                                         // the caller may write the table: drop the host copy
        return KA_OK;
}

extern "C" int ka_tree_build_consistency_part(ka_ctx* c, int n_anchors, float weight, int part, int nparts)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (nparts < 1 || part < 0 || part >= nparts) return fail("bad part");
        HIPCHK(hipSetDevice(c->device));
        c->cons_K = 0;
        const int N = c->numseq;
        int part_lo = 0, part_hi = N;
        cons_part_seqs(c, part, nparts, &part_lo, &part_hi);
        // the reference silently declines in these cases (anchor_consistency.c:206-217)
        if (n_anchors <= 0 || N < 3 || c->seq_dist.empty()) return KA_OK;
        if (n_anchors > KA_CONS_MAX_ANCHORS) return fail("this build takes at most 128 consistency anchors (KA_CONS_MAX_ANCHORS)");
        // One table per alignment.  A forest job holds several: every tree selects its own anchors among its own
        // sequences (in ascending index order = that alignment's own order); map k of a sequence is always against
        // anchor k of ITS tree, so the kernels need no notion of trees.
        std::vector<std::vector<int>> trees;
        {
                std::vector<char> seen(N, 0);
                for (int t = 0; t < c->n_tasks; t++) {
                        if (!c->descs[t].is_root) continue;
                        long long lo, hi;
                        node_members(c, c->descs[t].c, &lo, &hi);
                        std::vector<int> m(c->sip_flat.begin() + lo, c->sip_flat.begin() + hi);
                        std::sort(m.begin(), m.end());
                        for (int x : m) seen[x] = 1;
                        trees.push_back(m);
                }
                std::sort(trees.begin(), trees.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a[0] < b[0]; });
        }
        int K = n_anchors;
        for (auto& m : trees) if ((int)m.size() >= 3) K = std::min(K, (int)m.size());
        for (auto& m : trees)
                if ((int)m.size() >= 3 && (int)m.size() < n_anchors && trees.size() > 1)
                        return fail("forest job: every alignment with a consistency table needs at least n_anchors sequences");
        std::vector<int> anchor_of((size_t)N * K, -1);               // anchor k of the tree sequence i belongs to (-1: no table)
        c->cons_anchor_ids.clear();
        bool any = false;
        for (auto& m : trees) {
                if ((int)m.size() < 3) continue;
                std::vector<float> d(m.size());
                for (size_t x = 0; x < m.size(); x++) d[x] = c->seq_dist[m[x]];
                std::vector<int> ids;
                select_anchors(d, K, ids);
                for (int k = 0; k < K; k++) { ids[k] = m[ids[k]]; c->cons_anchor_ids.push_back(ids[k]); }
                for (int x : m) for (int k = 0; k < K; k++) anchor_of[(size_t)x * K + k] = ids[k];
                any = true;
        }
        if (!any) return KA_OK;

        // pairs (i, anchor_k of i's tree), i != anchor
        std::vector<int> ia, ib;
        std::vector<long long> poff;
        long long ptotal = 0;
        for (int i = part_lo; i < part_hi; i++)                       // (this part's sequences; all of them when nparts == 1)
                for (int k = 0; k < K; k++) {
                        const int ak = anchor_of[(size_t)i * K + k];
                        if (ak < 0 || i == ak) continue;
                        ia.push_back(i); ib.push_back(ak); poff.push_back(ptotal);
                        ptotal += (long long)c->lens[i] + c->lens[ak] + 3;
                }
        if (ia.empty() && nparts == 1) return KA_OK;
        if (ia.empty() && part_hi > part_lo) return fail("a part of the consistency batch holds only anchors: use fewer parts");
        // the N x K alignments on the device; their coded paths become position maps there as well
        // (anchor_consistency.c:86-114) and never leave HBM unless ka_tree_get_consistency asks for them
        long long used = 0;
        if (!ia.empty() &&
            pairwise_on_device(c, c->h_codes.data(), c->off.data(), c->lens.data(), N, ia.data(), ib.data(), (int)ia.size(),
                               c->subm, c->scal[0], c->scal[1], c->scal[2], poff.data(), &used))
                return KA_FAIL;
        c->cons_map_off.assign(N, 0);
        long long mt = 0;
        for (int i = 0; i < N; i++) { c->cons_map_off[i] = mt; mt += (long long)K * c->lens[i]; }
        // pair index, -1: the anchor itself, -2: no table, -3: another part's sequence (its maps arrive from the rank
        // that aligned it: ka_tree_consistency_maps_dev / _part_range)
        std::vector<int> pair_of((size_t)N * K, -2);
        {
                int pk = 0;
                for (int i = 0; i < N; i++)
                        for (int k = 0; k < K; k++) {
                                const int ak = anchor_of[(size_t)i * K + k];
                                if (ak < 0) continue;
                                if (i < part_lo || i >= part_hi) pair_of[(size_t)i * K + k] = -3;
                                else pair_of[(size_t)i * K + k] = (i == ak) ? -1 : pk++;
                        }
        }
        if (c->d_cons_maps.alloc((size_t)mt) || c->d_cons_map_off.alloc(N) || c->d_pair_of.alloc(pair_of.size())) return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->d_cons_map_off.p, c->cons_map_off.data(), sizeof(long long) * N, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_pair_of.p, pair_of.data(), sizeof(int) * pair_of.size(), hipMemcpyHostToDevice, c->stream));
        if (!ia.empty()) ka_launch_posmaps(c->p_paths.p, c->p_poff.p, c->d_pair_of.p, c->p_len.p, c->d_cons_map_off.p, N, K, c->d_cons_maps.p, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        c->cons_maps.clear();                                         // host copy on demand
        c->cons_maps_total = mt;
        if (!c->have_colof && setup_colof(c)) return KA_FAIL;
        c->cons_K = K; c->cons_weight = weight;
        c->ran = false; c->synced = false; c->state_valid = false;
        if (K > KA_NB - 1) {
                // More than five anchors: the streamed kernels carve KA_NB_BIG bonus entries per DP row and K-sized position / confidence /
                // vote tables -- ~3.6 x the bytes per row at K = 32 -- while the plan sized the scratch arena for the K <= 5 kernels
                // (shared contexts overflowed, doubled twice and ran their first tree three times).  Grow it here, once K is known.
                const double ratio = (double)ka_scratch_bytes_host_big(c->sum_len, c->sum_len, c->max_len, K) / (double)std::max<long long>(ka_scratch_bytes_host(c->sum_len, c->sum_len, c->max_len), 1);
                const long long want = (long long)((double)c->scratch_cap * std::max(ratio, 1.0)) + 4096;
                if (want > c->scratch_cap) {
                        c->scratch_cap = want;
                        c->d_scratch.release();
                        if (c->d_scratch.alloc((size_t)c->scratch_cap)) return fail("hipMalloc failed while growing the scratch arena for the consistency table");
                }
        }
        // the launch plan knows about the table (cluster limit of big jobs, plan_launches): plan again if it would come out differently
        if (c->env.max_cluster <= 0 && !c->shared_gpu && N >= 2048 && c->max_cluster < 32) {
                if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
        }
        return KA_OK;
}

extern "C" int ka_tree_get_consistency(ka_ctx* c, int* anchor_ids, int* maps_out)
{
        if (!c || !c->have_job) return -1;
        if (c->cons_K <= 0) return 0;
        if (anchor_ids) memcpy(anchor_ids, c->cons_anchor_ids.data(), sizeof(int) * c->cons_anchor_ids.size());
        if (maps_out) {
                if (c->cons_maps.empty() && c->cons_maps_total > 0) {
                        c->cons_maps.resize((size_t)c->cons_maps_total);
                        if (hipSetDevice(c->device) != hipSuccess ||
                            hipMemcpy(c->cons_maps.data(), c->d_cons_maps.p, sizeof(int) * c->cons_maps.size(), hipMemcpyDeviceToHost) != hipSuccess) {
                                c->cons_maps.clear();
                                fail("ka_tree_get_consistency: copying the position maps back failed");
                                return -1;
                        }
                }
                memcpy(maps_out, c->cons_maps.data(), sizeof(int) * c->cons_maps.size());
        }
        return c->cons_K;
}

extern "C" int ka_msa_tree(ka_ctx* c, int numseq, const uint8_t* codes, const int* off, const int* lens,
                           const float* seq_distances, int n_tasks, const int* abc,
                           const float* subm, const float* scal, int flags,
                           ka_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out)
{
        if (ka_tree_upload(c, numseq, codes, off, lens, seq_distances, n_tasks, abc, subm, scal, flags | (gaps_out ? KA_FLAG_DEVICE_GAPS : 0))) return KA_FAIL;
        if (ka_tree_run(c)) return KA_FAIL;
        if (ka_tree_sync(c)) return KA_FAIL;
        return ka_tree_download(c, recs, paths_out, paths_cap, gaps_out);
}

// The batch up to and including the kernel: coded paths stay in c->p_paths (pair k at poff[k]), scores in c->p_scores.
int pairwise_on_device(ka_ctx* c, const uint8_t* codes, const int* off, const int* lens, int numseq,
                              const int* ia, const int* ib, int npairs,
                              const float* subm, float gpo, float gpe, float tgpe, const long long* poff, long long* ptotal_out)
{
        HIPCHK(hipSetDevice(c->device));
        long long codes_bytes = 0, stride = 0, ptotal = 0;
        for (int i = 0; i < numseq; i++) codes_bytes = std::max<long long>(codes_bytes, (long long)off[i] + lens[i]);
        for (int k = 0; k < npairs; k++) {
                if (ia[k] < 0 || ia[k] >= numseq || ib[k] < 0 || ib[k] >= numseq) return fail("pair index out of range");
                const long long li = lens[ia[k]], lj = lens[ib[k]];
                if (li < 1 || lj < 1) return fail("zero-length sequence");
                stride = std::max(stride, ka_scratch_bytes_host(li, lj, 0));
                ptotal = std::max(ptotal, poff[k] + li + lj + 3);
        }
        stride = (stride + 255) / 256 * 256;
        DevBuf<uint8_t>& d_codes = c->p_codes; DevBuf<int>& d_off = c->p_off; DevBuf<int>& d_len = c->p_len;
        DevBuf<int>& d_ia = c->p_ia; DevBuf<int>& d_ib = c->p_ib; DevBuf<int>& d_paths = c->p_paths; DevBuf<int>& d_err = c->p_err;
        DevBuf<float>& d_subm = c->p_subm; DevBuf<float>& d_scores = c->p_scores;
        DevBuf<long long>& d_poff = c->p_poff; DevBuf<char>& d_scr = c->p_scr;
        if (d_codes.alloc((size_t)codes_bytes) || d_off.alloc(numseq) || d_len.alloc(numseq) || d_ia.alloc(npairs) ||
            d_ib.alloc(npairs) || d_paths.alloc((size_t)ptotal) || d_subm.alloc(23 * 23) || d_scores.alloc(npairs) ||
            d_poff.alloc(npairs) || d_scr.alloc((size_t)(stride * npairs)) || d_err.alloc(1))
                return fail("hipMalloc failed");
        auto cleanup = [&]() {};
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
        PCHK(hipMemcpyAsync(d_codes.p, codes, (size_t)codes_bytes, hipMemcpyHostToDevice, c->stream));
        PCHK(hipMemcpyAsync(d_off.p, off, sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        PCHK(hipMemcpyAsync(d_len.p, lens, sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        PCHK(hipMemcpyAsync(d_ia.p, ia, sizeof(int) * npairs, hipMemcpyHostToDevice, c->stream));
        PCHK(hipMemcpyAsync(d_ib.p, ib, sizeof(int) * npairs, hipMemcpyHostToDevice, c->stream));
        PCHK(hipMemcpyAsync(d_subm.p, subm, sizeof(float) * 23 * 23, hipMemcpyHostToDevice, c->stream));
        PCHK(hipMemcpyAsync(d_poff.p, poff, sizeof(long long) * npairs, hipMemcpyHostToDevice, c->stream));
        KaPairDev P;
        P.codes = d_codes.p; P.seq_off = d_off.p; P.seq_len = d_len.p; P.ia = d_ia.p; P.ib = d_ib.p;
        P.subm = d_subm.p; P.gpo = gpo; P.gpe = gpe; P.tgpe = tgpe;
        P.scratch = d_scr.p; P.scratch_stride = stride;
        P.paths_out = d_paths.p; P.poff = d_poff.p; P.scores = d_scores.p; P.npairs = npairs;
        P.error = d_err.p; P.pw = c->env.pw; P.reuse = c->env.reuse;
        PCHK(hipMemsetAsync(d_err.p, 0, sizeof(int), c->stream));
        PCHK(hipEventRecord(c->ev0, c->stream));
        ka_launch_pairs(&P, c->stream);
        PCHK(hipGetLastError());
        PCHK(hipEventRecord(c->ev1, c->stream));
        PCHK(hipStreamSynchronize(c->stream));
        PCHK(hipEventElapsedTime(&c->pair_ms, c->ev0, c->ev1));
        {
                int err = 0;
                PCHK(hipMemcpy(&err, d_err.p, sizeof(int), hipMemcpyDeviceToHost));
                if (err) { cleanup(); return fail("device watchdog: a strip pipeline inside a workgroup stopped making progress"); }
        }
#undef PCHK
        cleanup();
        *ptotal_out = ptotal;
        return KA_OK;
}

extern "C" int ka_pairwise_batch(ka_ctx* c, const uint8_t* codes, const int* off, const int* lens, int numseq,
                                 const int* ia, const int* ib, int npairs,
                                 const float* subm, float gpo, float gpe, float tgpe,
                                 int* paths_out, const long long* poff, float* scores_out)
{
        if (!c) return fail("null ctx");
        if (npairs <= 0) return KA_OK;
        long long ptotal = 0;
        if (pairwise_on_device(c, codes, off, lens, numseq, ia, ib, npairs, subm, gpo, gpe, tgpe, poff, &ptotal)) return KA_FAIL;
        if (copy_to_host(c, paths_out, c->p_paths.p, sizeof(int) * (size_t)ptotal)) return KA_FAIL;
        if (scores_out) HIPCHK(hipMemcpy(scores_out, c->p_scores.p, sizeof(float) * npairs, hipMemcpyDeviceToHost));
        return KA_OK;
}

// ---- distance estimation: calc_distance / bpm_block for a batch of pairs ----
extern "C" int ka_bpm_batch(ka_ctx* c, const uint8_t* codes, const int* off, const int* lens, int numseq,
                            const int* ia, const int* ib, int npairs, int* dist_out)
{
        if (!c) return fail("null ctx");
        if (npairs <= 0) return KA_OK;
        HIPCHK(hipSetDevice(c->device));
        long long codes_bytes = 0;
        for (int i = 0; i < numseq; i++) {
                if (lens[i] < 1) return fail("zero-length sequence");
                codes_bytes = std::max<long long>(codes_bytes, (long long)off[i] + lens[i]);
                for (int j = 0; j < lens[i]; j++) if (codes[off[i] + j] >= 13) return fail("bpm: sequence code out of range (the distance alphabet has 13 letters, bpm.c:11)");
        }
        for (int k = 0; k < npairs; k++)
                if (ia[k] < 0 || ia[k] >= numseq || ib[k] < 0 || ib[k] >= numseq) return fail("pair index out of range");
        if (c->p_codes.alloc((size_t)codes_bytes) || c->p_off.alloc(numseq) || c->p_len.alloc(numseq) || c->p_ia.alloc(npairs) ||
            c->p_ib.alloc(npairs) || c->b_peq.alloc((size_t)numseq * 13 * 16) || c->b_dist.alloc(npairs))
                return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->p_codes.p, codes, (size_t)codes_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->p_off.p, off, sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->p_len.p, lens, sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->p_ia.p, ia, sizeof(int) * npairs, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->p_ib.p, ib, sizeof(int) * npairs, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        ka_launch_bpm(c->p_codes.p, c->p_off.p, c->p_len.p, numseq, c->b_peq.p, c->p_ia.p, c->p_ib.p, npairs, c->b_dist.p, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipEventElapsedTime(&c->pair_ms, c->ev0, c->ev1));
        HIPCHK(hipMemcpy(dist_out, c->b_dist.p, sizeof(int) * npairs, hipMemcpyDeviceToHost));
        return KA_OK;
}

