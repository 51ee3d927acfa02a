// Guide tree: build_tree_kmeans (reference lib/src/bisectingKmeans.c:177-271) -- the stage that produces the task
// list and msa->seq_distances the dispatcher consumes.
//
// The distances (N x 32 against the anchors, then all pairs inside every leaf cluster) are the two GPU batches
// (ka_bpm_batch); what sits between them -- anchor choice, bisecting k-means on the N x 32 matrix, UPGMA inside
// clusters of < 50 sequences, node labels, task list -- is small, branchy host work and stays on the host, in the
// reference's fp32 evaluation order so that the tree is the same tree, bit for bit.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <stdexcept>
#include <string>
#include <system_error>
#include <vector>

#include "kalign_amd.h"
#include "ka_kmeans.h"

namespace {

constexpr int KA_UPGMA_BELOW = 50;            // KALIGN_KMEANS_UPGMA_THRESHOLD, reference CMakeLists.txt:71
constexpr int KA_MAX_ANCHORS = 32;            // pick_anchor.c:25

struct Node { int left = -1, right = -1, id = -1; };

struct Tree {
        std::vector<Node> nodes;
        int add(int l, int r, int id) { nodes.push_back(Node{ l, r, id }); return (int)nodes.size() - 1; }
};

// pick_anchor.c:34-70: sort by length, longest first, and take every (numseq / n)-th.  The reference's comparator
// never reports equality, so the order of equal lengths is whatever the C library's qsort makes of it: use the same
// qsort on the same element type (pointers to {len, id}) so that both see the same comparisons.
struct LenId { int len, id; };
int by_len_desc(const void* a, const void* b)
{
        const LenId* const* x = (const LenId* const*)a;
        const LenId* const* y = (const LenId* const*)b;
        return ((*x)->len > (*y)->len) ? -1 : 1;
}

void pick_anchors(int numseq, const int* lens, std::vector<int>& anchors)
{
        const int n = std::min(KA_MAX_ANCHORS, numseq);
        std::vector<LenId> recs(numseq);
        std::vector<LenId*> ptr(numseq);
        for (int i = 0; i < numseq; i++) { recs[i] = LenId{ lens[i], i }; ptr[i] = &recs[i]; }
        qsort(ptr.data(), numseq, sizeof(LenId*), by_len_desc);
        const int stride = numseq / n;
        anchors.resize(n);
        for (int i = 0; i < n; i++) anchors[i] = ptr[(size_t)i * stride]->id;
}

// d_estimation's length term (sequence_distance.c:66-69,118-120); the quotient is formed in double there
inline float with_length_term(int dist, int l1, int l2)
{
        const int s = (l1 + l2) / 2;
        const float add = (float)((10000.0 < (double)s ? 10000.0 : (double)s) / 10000.0);
        float d = (float)dist;
        d += add;
        return d;
}

// edist_256 (euclidean_dist.c): eight running lane sums, then (l0+l4 + l1+l5) + (l2+l6 + l3+l7), then sqrtf.
// `a`, `b` hold `padded` floats, zero beyond the anchors.
inline float edist(const float* a, const float* b, int padded)
{
        float lane[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int i = 0; i < padded; i += 8)
                for (int k = 0; k < 8; k++) {
                        float t = a[i + k] - b[i + k];
                        t = t * t;
                        lane[k] = lane[k] + t;
                }
        const float v0 = lane[0] + lane[4], v1 = lane[1] + lane[5], v2 = lane[2] + lane[6], v3 = lane[3] + lane[7];
        const float s01 = v0 + v1, s23 = v2 + v3;
        return sqrtf(s01 + s23);
}

inline int cmp_floats(float a, float b)          // bisectingKmeans.c:63-73
{
        const float epsilon = 1e-6f;
        if (fabsf(a - b) < epsilon) return 0;
        return a > b ? 1 : -1;
}

struct Split { std::vector<int> sl, sr; float score = FLT_MAX; };

// split2 (bisectingKmeans.c:766-971): 2-means from one seed sample and its mirror image through the centroid
void split2(const float* dm, int padded, const std::vector<int>& samples, int num_anchors, int seed_pick, Split& res)
{
        const int num_samples = (int)samples.size();
        std::vector<float> w(padded, 0.0f), wl(padded, 0.0f), wr(padded, 0.0f), cl(padded, 0.0f), cr(padded, 0.0f);
        for (int i = 0; i < num_samples; i++) {
                const float* row = dm + (size_t)samples[i] * padded;
                for (int j = 0; j < num_anchors; j++) w[j] += row[j];
        }
        for (int j = 0; j < num_anchors; j++) w[j] /= (float)num_samples;
        {
                const float* row = dm + (size_t)samples[seed_pick] * padded;
                for (int j = 0; j < num_anchors; j++) cl[j] = row[j];
                for (int j = 0; j < num_anchors; j++) cr[j] = w[j] - (cl[j] - w[j]);
        }
        res.sl.resize(num_samples);
        res.sr.resize(num_samples);
        int num_l = 0, num_r = 0;
        float score = 0.0f;
        float *pcl = cl.data(), *pcr = cr.data(), *pwl = wl.data(), *pwr = wr.data();
        for (int stop = 0; stop < 500; stop++) {
                num_l = num_r = 0;
                for (int i = 0; i < num_anchors; i++) { pwr[i] = 0.0f; pwl[i] = 0.0f; }
                score = 0.0f;
                for (int i = 0; i < num_samples; i++) {
                        const int s = samples[i];
                        const float* row = dm + (size_t)s * padded;
                        const float dl = edist(row, pcl, padded);
                        const float dr = edist(row, pcr, padded);
                        score += (dl < dr) ? dl : dr;
                        const int c = cmp_floats(dr, dl);
                        float* acc;
                        if (c == -1 || (c == 0 && (i & 1))) { acc = pwr; res.sr[num_r++] = s; }
                        else { acc = pwl; res.sl[num_l++] = s; }
                        for (int j = 0; j < num_anchors; j++) acc[j] += row[j];
                }
                if (num_l == 0 || num_r == 0) {                  // degenerate: cut the list in the middle, score 0
                        score = 0.0f;
                        num_l = num_r = 0;
                        for (int i = 0; i < num_samples / 2; i++) res.sl[num_l++] = samples[i];
                        for (int i = num_samples / 2; i < num_samples; i++) res.sr[num_r++] = samples[i];
                        break;
                }
                for (int j = 0; j < num_anchors; j++) { pwl[j] /= (float)num_l; pwr[j] /= (float)num_r; }
                bool moved = false;
                for (int j = 0; j < num_anchors; j++)
                        if (cmp_floats(pwl[j], pcl[j]) != 0 || cmp_floats(pwr[j], pcr[j]) != 0) { moved = true; break; }
                if (!moved) break;
                std::swap(pcl, pwl);
                std::swap(pcr, pwr);
        }
        res.sl.resize(num_l);
        res.sr.resize(num_r);
        res.score = score;
}

struct Builder {
        int numseq = 0, num_anchors = 0, padded = 0;
        const float* dm = nullptr;
        int n_threads = 1;
};

// A subtree under construction.  Children are either finished subtrees or leaf clusters waiting for their
// pairwise distances.
struct Sub {
        std::unique_ptr<Sub> l, r;
        std::vector<int> cluster;            // non-empty: a leaf cluster (l, r empty)
};

// bisecting_kmeans (bisectingKmeans.c:273-402)
std::unique_ptr<Sub> bisect(const Builder& B, std::vector<int> samples, int depth)
{
        std::unique_ptr<Sub> out(new Sub());
        const int num_samples = (int)samples.size();
        if (num_samples < KA_UPGMA_BELOW) { out->cluster = std::move(samples); return out; }
        const int tries = std::min(40, num_samples);
        const int step = num_samples / tries;
        Split best;
        bool have_best = false;
        for (int i = 0; i < tries; i += 4) {
                Split cand[4];
                if (B.n_threads > 1 && num_samples >= 128) {     // the reference: four OpenMP tasks (:325-340)
                        std::future<void> f[3];
                        bool started[3] = { false, false, false };
                        for (int k = 1; k < 4; k++) {
                                try {
                                        f[k - 1] = std::async(std::launch::async, [&, k] { split2(B.dm, B.padded, samples, B.num_anchors, (i + k) * step, cand[k]); });
                                        started[k - 1] = true;
                                } catch (const std::system_error&) {}     // no thread to be had: this candidate runs here
                        }
                        split2(B.dm, B.padded, samples, B.num_anchors, i * step, cand[0]);
                        for (int k = 0; k < 3; k++) {
                                if (started[k]) f[k].get();
                                else split2(B.dm, B.padded, samples, B.num_anchors, (i + k + 1) * step, cand[k + 1]);
                        }
                } else {
                        for (int k = 0; k < 4; k++) split2(B.dm, B.padded, samples, B.num_anchors, (i + k) * step, cand[k]);
                }
                int change = 0;
                for (int k = 0; k < 4; k++)
                        if (!have_best || best.score > cand[k].score) { best = std::move(cand[k]); have_best = true; change++; }
                if (!change) break;
        }
        samples.clear();
        samples.shrink_to_fit();
        std::future<std::unique_ptr<Sub>> fut;
        bool forked = false;
        // (a degenerate input can peel one sequence off per split: bound the recursion -- the C ABI wrapper turns
        // this into KA_FAIL -- instead of overflowing the stack)
        if (depth > 4096) throw std::runtime_error("bisecting k-means degenerated (recursion deeper than 4096)");
        if (depth < 30 && (1 << depth) < B.n_threads) {                // the two halves are independent (the reference: OpenMP tasks)
                try {
                        fut = std::async(std::launch::async, [&] { return bisect(B, std::move(best.sl), depth + 1); });
                        forked = true;
                } catch (const std::system_error&) {}             // no thread to be had: serial
        }
        if (forked) {
                out->r = bisect(B, std::move(best.sr), depth + 1);
                out->l = fut.get();
        } else {
                out->l = bisect(B, std::move(best.sl), depth + 1);
                out->r = bisect(B, std::move(best.sr), depth + 1);
        }
        return out;
}

// upgma (bisectingKmeans.c:974-1053) on an n x n matrix whose upper triangle is valid; returns the root node index
int upgma(Tree& T, std::vector<float>& dm, const std::vector<int>& samples)
{
        const int n = (int)samples.size();
        std::vector<int> active(n, 1), tree(n);
        for (int i = 0; i < n; i++) tree[i] = T.add(-1, -1, samples[i]);
        int node_a = 0, node_b = 0;
        for (int merges = 0; merges < n - 1; merges++) {
                float best = FLT_MAX;
                for (int i = 0; i < n - 1; i++) {
                        if (!active[i]) continue;
                        for (int j = i + 1; j < n; j++)
                                if (active[j] && dm[(size_t)i * n + j] < best) { best = dm[(size_t)i * n + j]; node_a = i; node_b = j; }
                }
                tree[node_a] = T.add(tree[node_a], tree[node_b], -1);
                tree[node_b] = -1;
                active[node_b] = 0;
                for (int j = n; j--;)
                        if (j != node_b) dm[(size_t)node_a * n + j] = (dm[(size_t)node_a * n + j] + dm[(size_t)node_b * n + j]) * 0.5f + 0.001f;
                dm[(size_t)node_a * n + node_a] = 0.0f;
                for (int j = n; j--;) dm[(size_t)j * n + node_a] = dm[(size_t)node_a * n + j];
        }
        return tree[node_a];
}

// label_internal + create_tasks + sort_tasks(TASK_ORDER_TREE) (bisectingKmeans.c:1067-1115, task.c:114-136):
// internal nodes numbered in post-order from numseq, one task per internal node, c ascending.  Returns the task count.
int emit_tasks(Tree& T, int top, int numseq, int* tasks_abc)
{
        int label = numseq, n_tasks = 0;
        struct Walk { int node; int state; };
        std::vector<Walk> ws;
        ws.push_back(Walk{ top, 0 });
        while (!ws.empty()) {
                Walk& w = ws.back();
                Node& nd = T.nodes[w.node];
                if (nd.left < 0) { ws.pop_back(); continue; }
                if (w.state == 0) { w.state = 1; ws.push_back(Walk{ nd.left, 0 }); continue; }
                if (w.state == 1) { w.state = 2; ws.push_back(Walk{ nd.right, 0 }); continue; }
                nd.id = label++;
                tasks_abc[3 * n_tasks] = T.nodes[nd.left].id;
                tasks_abc[3 * n_tasks + 1] = T.nodes[nd.right].id;
                tasks_abc[3 * n_tasks + 2] = nd.id;
                n_tasks++;
                ws.pop_back();
        }
        return n_tasks;
}

void collect_leaves(Sub* root, std::vector<Sub*>& out)
{
        std::vector<Sub*> st(1, root);                   // iterative pre-order, left before right
        while (!st.empty()) {
                Sub* s = st.back(); st.pop_back();
                if (!s->l) { out.push_back(s); continue; }
                st.push_back(s->r.get());
                st.push_back(s->l.get());
        }
}

}  // namespace

int ka_fail_message(const char* m);      // ka_api.cpp: sets what ka_last_error() returns

// the library's own distance source (ka_bpm_batch) has already said why: keep its text
static int dist_failed()
{
        const std::string inner = ka_last_error();
        return ka_fail_message(("ka_guide_tree_from: the distance source failed" + (inner.empty() ? std::string() : ": " + inner)).c_str());
}

static int guide_tree_from(int numseq, const int* lens, ka_dist_fn dist, void* user, int n_threads,
                           const float* dm_scale, int* tasks_abc, float* seq_distances);

// The 2-means bisection on the device (ka_kmeans.hip) for the callers that have one: ka_guide_tree sets the device and stream
// of its context here for the duration of the call.  KA_KMEANS in the environment: "0" host, "1" device whatever the size;
// default: the device from 2048 sequences (below that the host's 40 x 2-means take less than the launches).
namespace {
struct KmDevice { bool on = false; int device = 0; hipStream_t stream = nullptr; };
thread_local KmDevice g_km;
double g_last_bisect_ms = 0.0;
int g_last_bisect_device = 0;
}
int ka_ctx_device_stream(ka_ctx* c, int* device, hipStream_t* stream);    // ka_api.cpp
extern "C" double ka_guide_last_bisect_ms(int* on_device)
{
        if (on_device) *on_device = g_last_bisect_device;
        return g_last_bisect_ms;
}

extern "C" int ka_guide_tree_from(int numseq, const int* lens, ka_dist_fn dist, void* user, int n_threads,
                                  const float* dm_scale, int* tasks_abc, float* seq_distances)
{
        try {                                            // no exception may cross the C ABI
                return guide_tree_from(numseq, lens, dist, user, n_threads, dm_scale, tasks_abc, seq_distances);
        } catch (const std::bad_alloc&) {
                return ka_fail_message("ka_guide_tree_from: out of memory");
        } catch (const std::exception& e) {
                return ka_fail_message((std::string("ka_guide_tree_from: ") + e.what()).c_str());
        }
}

static int guide_tree_from(int numseq, const int* lens, ka_dist_fn dist, void* user, int n_threads,
                           const float* dm_scale, int* tasks_abc, float* seq_distances)
{
        if (numseq < 2 || !lens || !dist || !tasks_abc) return ka_fail_message("ka_guide_tree_from: bad arguments");
        for (int i = 0; i < numseq; i++)
                if (lens[i] < 1) return ka_fail_message("ka_guide_tree_from: zero-length sequence");

        // ---- anchors and the N x A distance matrix (pick_anchor, d_estimation with pair = 0) ----
        std::vector<int> anchors;
        pick_anchors(numseq, lens, anchors);
        const int A = (int)anchors.size();
        const int padded = ((A + 7) / 8) * 8;
        std::vector<int> ia((size_t)numseq * A), ib((size_t)numseq * A), d((size_t)numseq * A);
        for (int i = 0; i < numseq; i++)
                for (int j = 0; j < A; j++) { ia[(size_t)i * A + j] = i; ib[(size_t)i * A + j] = anchors[j]; }
        if (dist(user, numseq * A, ia.data(), ib.data(), d.data())) return dist_failed();
        std::vector<float> dm((size_t)numseq * padded, 0.0f);
        for (int i = 0; i < numseq; i++)
                for (int j = 0; j < A; j++) dm[(size_t)i * padded + j] = with_length_term(d[(size_t)i * A + j], lens[i], lens[anchors[j]]);
        // build_tree_kmeans_noisy (:103-115): the caller's multiplicative noise on the anchor distances
        if (dm_scale)
                for (int i = 0; i < numseq; i++)
                        for (int j = 0; j < A; j++) dm[(size_t)i * padded + j] *= dm_scale[(size_t)i * A + j];

        // ---- bisecting k-means down to clusters of < 50 sequences ----
        Builder B;
        B.numseq = numseq; B.num_anchors = A; B.padded = padded; B.dm = dm.data(); B.n_threads = std::max(1, n_threads);
        std::unique_ptr<Sub> root;
        const auto t_bisect = std::chrono::steady_clock::now();
        const char* km_env = getenv("KA_KMEANS");
        const bool km_dev = g_km.on && A == 32 && padded == 32 && (km_env ? atoi(km_env) != 0 : numseq >= 2048);
        g_last_bisect_device = km_dev ? 1 : 0;
        if (km_dev) {
                std::vector<KaKmNode> kn;
                std::string why;
                if (ka_kmeans_device(g_km.device, g_km.stream, dm.data(), numseq, kn, why)) return ka_fail_message(("ka_guide_tree: " + why).c_str());
                // the device's node table as the builder's tree (iteratively: lopsided splits make it deep)
                std::vector<std::unique_ptr<Sub>> made(kn.size());
                for (size_t k = kn.size(); k--;) {                  // children have larger indices than their parent
                        made[k].reset(new Sub());
                        if (kn[k].left < 0) made[k]->cluster = std::move(kn[k].cluster);
                        else { made[k]->l = std::move(made[kn[k].left]); made[k]->r = std::move(made[kn[k].right]); }
                }
                root = std::move(made[0]);
        } else {
                std::vector<int> all(numseq);
                for (int i = 0; i < numseq; i++) all[i] = i;
                root = bisect(B, std::move(all), 0);
        }
        g_last_bisect_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bisect).count();

        // ---- all pairs inside every leaf cluster in one batch (d_estimation with pair = 1): the value the reference
        //      keeps for i < j is the one it computes last, calc_distance(seq[samples[j]], seq[samples[i]]) ----
        std::vector<Sub*> leaves;
        collect_leaves(root.get(), leaves);
        std::vector<size_t> first(leaves.size() + 1, 0);
        for (size_t k = 0; k < leaves.size(); k++) {
                const size_t n = leaves[k]->cluster.size();
                first[k + 1] = first[k] + n * (n - 1) / 2;
        }
        ia.assign(first.back(), 0); ib.assign(first.back(), 0); d.assign(first.back(), 0);
        for (size_t k = 0; k < leaves.size(); k++) {
                const std::vector<int>& s = leaves[k]->cluster;
                size_t p = first[k];
                for (size_t i = 0; i < s.size(); i++)
                        for (size_t j = i + 1; j < s.size(); j++, p++) { ia[p] = s[j]; ib[p] = s[i]; }
        }
        if (!ia.empty() && dist(user, (int)ia.size(), ia.data(), ib.data(), d.data())) return dist_failed();

        // ---- UPGMA inside the clusters, then labels and tasks in post-order (label_internal, create_tasks,
        //      sort_tasks(TASK_ORDER_TREE): c ascending == post-order) ----
        Tree T;
        std::vector<int> leaf_root(leaves.size());
        for (size_t k = 0; k < leaves.size(); k++) {
                const std::vector<int>& s = leaves[k]->cluster;
                const int n = (int)s.size();
                std::vector<float> pd((size_t)n * n, 0.0f);
                size_t p = first[k];
                for (int i = 0; i < n; i++)
                        for (int j = i + 1; j < n; j++, p++) {
                                const float v = with_length_term(d[p], lens[s[j]], lens[s[i]]);
                                pd[(size_t)i * n + j] = v;
                                pd[(size_t)j * n + i] = v;
                        }
                leaf_root[k] = upgma(T, pd, s);
        }
        // stitch the k-means levels above the clusters (iteratively: the recursion can be N deep when splits are lopsided)
        {
                size_t next_leaf = 0;
                struct Frame { Sub* s; int state; int l; };
                std::vector<Frame> st;
                std::vector<int> done;           // node indices of finished subtrees
                st.push_back(Frame{ root.get(), 0, -1 });
                while (!st.empty()) {
                        Frame& f = st.back();
                        if (!f.s->l) { done.push_back(leaf_root[next_leaf++]); st.pop_back(); continue; }
                        if (f.state == 0) { f.state = 1; st.push_back(Frame{ f.s->l.get(), 0, -1 }); continue; }
                        if (f.state == 1) { f.state = 2; st.push_back(Frame{ f.s->r.get(), 0, -1 }); continue; }
                        const int r = done.back(); done.pop_back();
                        const int l = done.back(); done.pop_back();
                        done.push_back(T.add(l, r, -1));
                        st.pop_back();
                }
                if (emit_tasks(T, done.back(), numseq, tasks_abc) != numseq - 1) return ka_fail_message("ka_guide_tree_from: internal error (task count)");
        }

        // ---- msa->seq_distances (bisectingKmeans.c:244-255) ----
        if (seq_distances)
                for (int i = 0; i < numseq; i++) {
                        float sum = 0.0f;
                        for (int j = 0; j < A; j++) sum += dm[(size_t)i * padded + j];
                        const float mean_dist = sum / (float)A;
                        seq_distances[i] = mean_dist / (float)lens[i];
                }
        return KA_OK;
}

// upgma()'s bookkeeping (bisectingKmeans.c:996-1047) for a merge sequence found on the device: merge k joins the
// current subtrees of slots a < b into slot a; the last merge's slot holds the root.  Library-internal.
__attribute__((visibility("hidden"))) int ka_tasks_from_merges(int numseq, const int* merges_ab, int* tasks_abc)
{
        Tree T;
        std::vector<int> slot(numseq);
        for (int i = 0; i < numseq; i++) slot[i] = T.add(-1, -1, i);
        int top = slot[0];
        for (int k = 0; k < numseq - 1; k++) {
                const int a = merges_ab[2 * k], b = merges_ab[2 * k + 1];
                if (a < 0 || b < 0 || a >= numseq || b >= numseq || a == b || slot[a] < 0 || slot[b] < 0)
                        return ka_fail_message("realignment tree: inconsistent merge sequence");
                slot[a] = T.add(slot[a], slot[b], -1);
                slot[b] = -1;
                top = slot[a];
        }
        return emit_tasks(T, top, numseq, tasks_abc) == numseq - 1 ? KA_OK : ka_fail_message("realignment tree: internal error (task count)");
}

// The distance source of the product: the two batches run on the device (ka_bpm.hip).
namespace {
struct DeviceDist { ka_ctx* ctx; const uint8_t* codes; const int* off; const int* lens; int numseq; };
int device_dist(void* user, int npairs, const int* ia, const int* ib, int* out)
{
        DeviceDist* D = (DeviceDist*)user;
        return ka_bpm_batch(D->ctx, D->codes, D->off, D->lens, D->numseq, ia, ib, npairs, out);
}
}  // namespace

extern "C" int ka_guide_tree(ka_ctx* ctx, int numseq, const uint8_t* codes, const int* off, const int* lens,
                             int n_threads, const float* dm_scale, int* tasks_abc, float* seq_distances)
{
        if (!ctx || !codes || !off) return ka_fail_message("ka_guide_tree: bad arguments");
        DeviceDist D{ ctx, codes, off, lens, numseq };
        g_km = KmDevice();
        g_km.on = ka_ctx_device_stream(ctx, &g_km.device, &g_km.stream) == 0;
        const int rc = ka_guide_tree_from(numseq, lens, device_dist, &D, n_threads, dm_scale, tasks_abc, seq_distances);
        g_km = KmDevice();
        return rc;
}
