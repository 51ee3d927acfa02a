// ka_multi.cpp -- the GPUs of one node under ONE caller (round 4): what a single-process C program -- the reference's own
// kalign_run behind the drop-in glue -- needs to put a whole node under create_msa_tree / anchor_consistency_build.
//
// north_star: "independent pairwise tasks at each guide-tree level are sharded across the 8 GPUs of one node".  The sharded
// path itself is ka_dist_* (ka_api.cpp: the cut, a rank's subtrees as one planned run, RCCL from C, the gather); it wants one
// caller PER RANK, all making the same calls.  bench.py provides them as processes (torch.distributed.run).  Here they are
// threads of the calling process, one context + one ka_dist per device, behind calls shaped like ka_msa_tree /
// ka_tree_build_consistency: the caller hands over host buffers once and gets records, coded paths and gap arrays back --
// identical, bit for bit, to what one GPU returns (tasks are position-addressed; lib/src/aln_run.c:95-109: subtrees are
// independent, the result does not depend on who runs them; tests/dssim_test.c:41-86 checks the same of thread counts).
//
// Built on the public C ABI only (include/kalign_amd.h).  `loopback` puts every rank on device 0 over the in-process
// transport (RCCL refuses two ranks on one device): how the one-GPU test boxes run worlds of 2 and 4 through this layer.
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "kalign_amd.h"

struct ka_multi {
        int world = 1;
        bool loopback = false;
        void* loop = nullptr;
        std::vector<ka_ctx*> ctx;
        std::vector<ka_dist*> dist;
        bool cons_resident = false;              // every rank's context holds the consistency table of the sequences uploaded last
        // the ranks agree on the outcome of every rank-local stage (upload, planning) BEFORE they enter a stage that talks
        // (ka_dist_consistency, ka_dist_tree_run): a rank that failed alone would leave the others inside a collective for good
        std::mutex agree_mu;
        std::condition_variable agree_cv;
        int agree_arrived = 0, agree_failed = 0, agree_verdict = 0;
        long long agree_round = 0;
        long long runs = 0;                      // sharded tree runs so far (the drop-in's seam counter reads it)
        std::string err;
};

static thread_local std::string g_multi_err;
extern "C" const char* ka_multi_last_error(void) { return g_multi_err.c_str(); }

// every rank's part of a call on a thread of its own (the ranks meet inside ka_dist_*); the first failure is reported
template <typename F>
static int run_ranks(ka_multi* m, F f)
{
        std::vector<int> rc(m->world, 0);
        std::vector<std::string> why(m->world);
        std::vector<std::thread> th;
        for (int r = 0; r < m->world; r++)
                th.emplace_back([&, r]() {
                        rc[r] = f(r);
                        if (rc[r]) why[r] = ka_last_error();          // (the library's error text is per thread)
                });
        for (auto& t : th) t.join();
        for (int r = 0; r < m->world; r++)
                if (rc[r]) { g_multi_err = "rank " + std::to_string(r) + ": " + why[r]; return rc[r]; }
        return KA_OK;
}

// Barrier + OR over the ranks' local return codes: every rank leaves with the same answer (true: all fine).
static bool ranks_agree(ka_multi* m, int local_rc)
{
        std::unique_lock<std::mutex> lk(m->agree_mu);
        const long long round = m->agree_round;
        if (local_rc) m->agree_failed++;
        if (++m->agree_arrived == m->world) {
                m->agree_verdict = m->agree_failed;
                m->agree_arrived = 0; m->agree_failed = 0;
                m->agree_round++;
                m->agree_cv.notify_all();
        } else {
                m->agree_cv.wait(lk, [&] { return m->agree_round != round; });
        }
        return m->agree_verdict == 0;
}

extern "C" void ka_multi_destroy(ka_multi* m)
{
        if (!m) return;
        for (auto d : m->dist) if (d) ka_dist_destroy(d);
        for (auto c : m->ctx) if (c) ka_ctx_destroy(c);
        if (m->loop) ka_dist_loopback_free(m->loop);
        delete m;
}

// devices[world] (NULL: 0 .. world-1).  loopback != 0: every rank on devices[0] (or device 0), in-process transport.
extern "C" int ka_multi_create(int world, const int* devices, int loopback, ka_multi** out)
{
        if (world < 1 || !out) { g_multi_err = "ka_multi_create: bad arguments"; return KA_FAIL; }
        ka_multi* m = new ka_multi();
        m->world = world; m->loopback = loopback != 0;
        m->ctx.assign(world, nullptr); m->dist.assign(world, nullptr);
        for (int r = 0; r < world; r++) {
                const int dev = loopback ? (devices ? devices[0] : 0) : (devices ? devices[r] : r);
                if (ka_ctx_create(dev, &m->ctx[r])) { g_multi_err = std::string("ka_multi_create: ") + ka_last_error(); ka_multi_destroy(m); return KA_FAIL; }
                // several contexts on one GPU: no assumption that workgroups of one launch are all resident
                if (loopback && world > 1 && ka_ctx_set_shared(m->ctx[r], 1)) { g_multi_err = ka_last_error(); ka_multi_destroy(m); return KA_FAIL; }
        }
        unsigned char id[128];
        memset(id, 0, sizeof(id));
        if (loopback) {
                m->loop = ka_dist_loopback_new(world);
                if (!m->loop) { g_multi_err = "ka_multi_create: no loopback"; ka_multi_destroy(m); return KA_FAIL; }
        } else if (world > 1 && ka_dist_unique_id(id)) { g_multi_err = std::string("ka_multi_create: ") + ka_last_error(); ka_multi_destroy(m); return KA_FAIL; }
        // (ncclCommInitRank blocks until every rank has called it: one thread per rank)
        const int rc = run_ranks(m, [&](int r) {
                return loopback ? ka_dist_create_loopback(m->ctx[r], r, world, m->loop, &m->dist[r])
                                : ka_dist_create(m->ctx[r], r, world, world > 1 ? id : nullptr, &m->dist[r]);
        });
        if (rc) { ka_multi_destroy(m); return rc; }
        *out = m;
        return KA_OK;
}

extern "C" int ka_device_count(void)
{
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) return 0;
        return n;
}

extern "C" int ka_multi_world(ka_multi* m) { return m ? m->world : -1; }
extern "C" long long ka_multi_runs(ka_multi* m) { return m ? m->runs : -1; }

// anchor_consistency_build (lib/src/anchor_consistency.c:200-275) over the node: the N x K seq-seq batch sharded over the
// ranks, every rank's share of the position maps broadcast in place (ka_dist_consistency); the host copy of the table
// (anchor_ids_out[K], maps_out: ka_tree_get_consistency's layout) from rank 0.  tasks_abc: any valid task list over the
// sequences (the stage only needs them on the device).  Returns the number of anchors, 0 when the job declines, < 0 on error.
extern "C" int ka_multi_consistency(ka_multi* m, int numseq, const uint8_t* codes, const int* off, const int* lens, const float* seq_distances,
                                    int n_tasks, const int* tasks_abc, const float* subm, const float* scal, int flags,
                                    int n_anchors, float weight, int* anchor_ids_out, int* maps_out)
{
        if (!m) { g_multi_err = "ka_multi_consistency: null"; return -1; }
        m->cons_resident = false;
        int rc = run_ranks(m, [&](int r) {
                const int up = ka_tree_upload(m->ctx[r], numseq, codes, off, lens, seq_distances, n_tasks, tasks_abc, subm, scal, flags);
                if (!ranks_agree(m, up)) return up ? up : (int)KA_FAIL;       // (one rank could not upload -- out of memory, say: nobody enters the collectives)
                return ka_dist_consistency(m->dist[r], n_anchors, weight);
        });
        if (rc) return -1;
        const int K = ka_tree_get_consistency(m->ctx[0], anchor_ids_out, maps_out);
        if (K < 0) { g_multi_err = ka_last_error(); return -1; }
        m->cons_resident = K > 0;
        return K;
}

// create_msa_tree (lib/src/aln_run.c:43-78) over the node, staged like ka_tree_upload / run / download: ka_multi_tree_run takes
// ka_tree_upload's arguments (every rank uploads the job, the tree is cut, every rank runs its subtrees and its share of the
// tasks above the cut, records and coded paths are gathered on every rank), ka_multi_paths_size sizes the path buffer,
// ka_multi_download returns rank 0's copy with the gap arrays woven on the host (weave_alignment.c:41-112).
// n_anchors > 0: default mode -- the consistency table is built (sharded) unless `flags` carries KA_FLAG_KEEP_CONSISTENCY and
// the ranks still hold the table of these sequences from ka_multi_consistency.
extern "C" int ka_multi_tree_run(ka_multi* m, int numseq, const uint8_t* codes, const int* off, const int* lens, const float* seq_distances,
                                 int n_tasks, const int* tasks_abc, const float* subm, const float* scal, int flags,
                                 int n_anchors, float weight)
{
        if (!m) { g_multi_err = "ka_multi_tree_run: null"; return KA_FAIL; }
        const bool keep = (flags & KA_FLAG_KEEP_CONSISTENCY) && m->cons_resident && n_anchors > 0;
        const int up_flags = (flags & ~KA_FLAG_KEEP_CONSISTENCY) | (keep ? KA_FLAG_KEEP_CONSISTENCY : 0);
        if (!keep) m->cons_resident = false;
        const int rc = run_ranks(m, [&](int r) {
                const int up = ka_tree_upload(m->ctx[r], numseq, codes, off, lens, seq_distances, n_tasks, tasks_abc, subm, scal, up_flags);
                if (!ranks_agree(m, up)) return up ? up : (int)KA_FAIL;
                // (ka_dist_consistency fails on every rank alike when one part cannot be built)
                const int co = (n_anchors > 0 && !keep) ? ka_dist_consistency(m->dist[r], n_anchors, weight) : 0;
                const int pl = co ? co : ka_dist_plan(m->dist[r]);
                if (!ranks_agree(m, pl)) return pl ? pl : (int)KA_FAIL;
                return ka_dist_tree_run(m->dist[r]);
        });
        if (rc) { m->cons_resident = false; return rc; }
        if (n_anchors > 0) m->cons_resident = true;
        m->runs++;
        return KA_OK;
}

// The context of rank r (its device's single-GPU context): what the caller continues on after a sharded run, and what an
// ensemble member that runs on device r alone uses (kalign_ensemble's member loop, lib/src/ensemble.c:286-339).
extern "C" ka_ctx* ka_multi_ctx(ka_multi* m, int rank) { return (m && rank >= 0 && rank < m->world) ? m->ctx[rank] : nullptr; }

// After ka_multi_tree_run + ka_multi_download: rank 0's context -- it holds the job, the complete consistency table and the
// gathered records -- takes the finished alignment over (ka_tree_adopt_alignment), so that the stages behind the dispatcher
// (refine_alignment, finalise_alignment, the identity distances of a realignment pass) stay on a device with more than one rank.
extern "C" int ka_multi_adopt(ka_multi* m, const ka_task_rec* recs, const int* gaps)
{
        if (!m || !recs || !gaps) { g_multi_err = "ka_multi_adopt: bad arguments"; return KA_FAIL; }
        if (ka_tree_adopt_alignment(m->ctx[0], recs, gaps)) { g_multi_err = ka_last_error(); return KA_FAIL; }
        return KA_OK;
}

// ints the coded paths of the last ka_multi_tree_run take
extern "C" long long ka_multi_paths_size(ka_multi* m) { return (m && m->dist[0]) ? ka_dist_paths_size(m->dist[0]) : -1; }

extern "C" int ka_multi_download(ka_multi* m, int numseq, const int* lens, int n_tasks, ka_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out)
{
        if (!m || !recs || !paths_out) { g_multi_err = "ka_multi_download: bad arguments"; return KA_FAIL; }
        long long used = 0;
        const int rc = ka_dist_download(m->dist[0], recs, paths_out, paths_cap, &used);
        if (rc) { g_multi_err = ka_last_error(); return rc; }
        if (gaps_out && ka_weave_gaps(numseq, lens, n_tasks, recs, paths_out, gaps_out)) { g_multi_err = ka_last_error(); return KA_FAIL; }
        return KA_OK;
}
