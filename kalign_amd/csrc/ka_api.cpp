// ka_api.cpp -- host side of the C ABI (include/kalign_amd.h): the guide-tree level scheduler
// that replaces create_msa_tree / recursive_aln (aln_run.c:43-124) and the gap weaving that
// follows each task (weave_alignment.c:41-112).
//
// The reference walks the guide tree post-order with OpenMP tasks, one do_align per node.
// Here the tree is cut into dependency levels on the host (level(c) = 1 + max(level(a),
// level(b))); every level is ONE kernel launch with one workgroup per task, and profiles,
// paths and all per-task state stay in HBM between levels.  Only the coded paths and the
// per-task records come back to the host, once, at the end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ka_device.h"

// the task kernels live in four translation units (ka_kernels.hip, -DKA_UNIT=0..3)
extern "C" void ka_unit0_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream);   // 8 waves
extern "C" void ka_unit1_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream);   // 8 waves + consistency
extern "C" void ka_unit2_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, int nqueue, hipStream_t stream);   // half (4 waves, 2 per CU)
extern "C" void ka_unit3_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, hipStream_t stream);    // lean (seq-seq levels)
extern "C" void ka_unit4_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, hipStream_t stream);    // refinement pass (one workgroup per task)
// the consistency kernels once more with room for ten anchors per DP row (units 6..9; K > KA_NB - 1)
extern "C" void ka_unit6_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream);
extern "C" void ka_unit7_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int nqueue, hipStream_t stream);
extern "C" void ka_unit8_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream);
extern "C" void ka_unit9_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream);
static bool ka_cons_big(const KaTreeDev* D) { return D->cons_K > KA_NB - 1; }
// kind: 0 = 8-wave kernel, 1 = lean (seq-seq only), 2 = half (4 waves, two workgroups per CU)
static void ka_launch_task_level(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int kind, int chain, hipStream_t stream)
{
        const int cons = D->cons_K > 0;
        if (ka_cons_big(D)) {
                if (kind == 1) ka_unit8_launch(D, blocks_dev, nblocks, stream);
                else if (kind == 2) ka_unit7_launch(D, blocks_dev, nblocks, 0, stream);
                else ka_unit6_launch(D, blocks_dev, nblocks, chain, stream);
                return;
        }
        if (kind == 1) ka_unit3_launch(D, blocks_dev, nblocks, cons, stream);
        else if (kind == 2) ka_unit2_launch(D, blocks_dev, nblocks, cons, 0, stream);
        else if (cons) ka_unit1_launch(D, blocks_dev, nblocks, chain, stream);
        else ka_unit0_launch(D, blocks_dev, nblocks, chain, stream);
}
extern "C" int ka_max_g_host(void);
extern "C" void ka_launch_posmaps(const int* paths, const long long* poff, const int* pair_of, const int* lens, const long long* map_off,
                                  int numseq, int K, int* maps, hipStream_t stream);
extern "C" void ka_launch_aln_dist(const uint8_t* rows, long long stride, int alnlen, int n, uint8_t gap, float* dm, float* means,
                                   hipStream_t stream);
extern "C" void ka_launch_upgma(float* dm, int* active, unsigned long long* keys, int2* merges, int n, int mode, hipStream_t stream);
int ka_tasks_from_merges(int numseq, const int* merges_ab, int* tasks_abc);      // ka_guide.cpp
extern "C" void ka_launch_rows(const uint8_t* letters, const int* off, const int* lens, const int* colof, const int* alnlen,
                               int numseq, uint8_t gap, uint8_t* rows, long long stride, hipStream_t stream);
extern "C" void ka_launch_bpm(const uint8_t* codes, const int* off, const int* lens, int numseq, unsigned long long* peq,
                              const int* ia, const int* ib, int npairs, int* dist, hipStream_t stream);
extern "C" long long ka_ctl_bytes_host(void);
extern "C" void ka_launch_pairs(const KaPairDev* P, hipStream_t stream);
extern "C" long long ka_scratch_bytes_host(long long la, long long lb, long long cons_maxlen);

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return KA_FAIL; }
// ka_guide.cpp reports through the same message (library-internal, not part of the ABI)
__attribute__((visibility("hidden"))) int ka_fail_message(const char* m) { return fail(m); }

#define HIPCHK(x)                                                                         \
        do {                                                                              \
                hipError_t e_ = (x);                                                      \
                if (e_ != hipSuccess) {                                                   \
                        return fail(std::string(#x) + ": " + hipGetErrorString(e_));      \
                }                                                                         \
        } while (0)

template <typename T>
struct DevBuf {
        T* p = nullptr;
        size_t n = 0;
        int alloc(size_t count)
        {
                if (count <= n && p) return 0;
                release();
                if (hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) { p = nullptr; n = 0; return 1; }
                n = count;
                return 0;
        }
        void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// The KA_* environment switches (experiments, measurements and tests; none is needed in production), read ONCE when the
// context is created -- ka_debug_reload_env re-reads them for tools and tests that flip a switch on a live context.
struct KaEnv {
        bool trace = false, no_chain = false, no_queue = false, no_half = false, no_lean = false, chain_g1 = false, no_crit = false;
        bool no_staging = false, no_wdfs = false, no_ls0 = false, no_inc = false, no_ldfs = false, refine_serial = false;
        int chain_tasks = 0;           // KA_CHAIN_TASKS: the chained launch starts at the first level with at most this many tasks (0: CUs - 8)
        int max_cluster = 0;           // KA_MAX_CLUSTER: workgroups one task may use (0: the default, 16)
        int crit_greedy = 1;           // KA_CRIT_GREEDY: spare chain workgroups by a simulated schedule first (0: by the ranking alone)
        int crit_top = 0;              // KA_CRIT_TOP: workgroups of the chain entry with the longest way to the root (0: default)
        int prof_task = -1;            // KA_PROF_TASK: the task whose per-level times KA_FLAG_TIMING keeps (-1: the root)
        int q1 = -1;                   // KA_Q1 (-1: the default -- 4 for protein jobs: 64-row strips per recursion level where every strip still gets a helper wave, 0 for nucleotides): 64-row strips (KaTreeDev::q1_mode); measured no faster with 64-column hand-over batches (round 3)
        int lean4 = 1;                 // KA_LEAN4: leaf levels on 4-wave workgroups, four per CU (1.60 -> 1.28 ms on the 4096 x 400 leaf level)
        int mw = 1;                    // KA_MW: multi-wave scan of the top-level meetups
        int per = 0;                   // KA_PER: strips per workgroup (KaTreeDev::per_target; experiments)
        int ho = -1;                   // KA_HO: hand-over between neighbouring strips through LDS (KaTreeDev::ho_mode); -1: on (1)
        int hw = 1;                    // KA_HW: profile-profile strips with helper waves (ka_wstrip.h; KaTreeDev::hw_mode)
        int hw_prio = 3;               // KA_HW_PRIO: s_setprio of a strip wave that has a helper (experiments)
        int subtree = 1;               // KA_SUBTREE: small Hirschberg subtrees run wave-locally in LDS
        int reuse = 1;                 // KA_REUSE: Hirschberg prefix reuse in the 4-wave kernels (queued levels, seq-seq leaves, pair batch)
        int qw = 4, lw = 4, pw = 2;    // KA_QW / KA_LW / KA_PW: waves per workgroup of the queued launch, the seq-seq leaf levels, the pair batch (4, 2, 1)
        bool launch_ev = false;        // KA_LAUNCH_EV: an event behind every launch of a run (ka_tree_launch_ms)
        bool upgma_launches = false;   // KA_UPGMA_LAUNCHES: ka_aln_guide_tree's UPGMA as one launch per merge (the path for > 6144 sequences) at any size
};
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static void read_env(KaEnv& v)
{
        v = KaEnv();
        v.trace = getenv("KA_TRACE") != nullptr; v.no_chain = getenv("KA_NO_CHAIN") != nullptr; v.no_queue = getenv("KA_NO_QUEUE") != nullptr;
        v.no_half = getenv("KA_NO_HALF") != nullptr; v.no_lean = getenv("KA_NO_LEAN") != nullptr; v.chain_g1 = getenv("KA_CHAIN_G1") != nullptr;
        v.no_crit = getenv("KA_NO_CRIT") != nullptr; v.no_staging = getenv("KA_NO_STAGING") != nullptr;
        v.no_wdfs = getenv("KA_NO_WDFS") != nullptr; v.no_ls0 = getenv("KA_NO_LS0") != nullptr; v.no_inc = getenv("KA_NO_INC") != nullptr; v.no_ldfs = getenv("KA_NO_LDFS") != nullptr; v.refine_serial = getenv("KA_REFINE_SERIAL") != nullptr;
        v.chain_tasks = env_int("KA_CHAIN_TASKS", 0); v.max_cluster = env_int("KA_MAX_CLUSTER", 0); v.crit_top = env_int("KA_CRIT_TOP", 0); v.crit_greedy = env_int("KA_CRIT_GREEDY", 1);
        v.prof_task = env_int("KA_PROF_TASK", -1); v.q1 = env_int("KA_Q1", -1); v.lean4 = env_int("KA_LEAN4", 1);
        v.launch_ev = getenv("KA_LAUNCH_EV") != nullptr;
        v.subtree = env_int("KA_SUBTREE", 1);
        v.reuse = env_int("KA_REUSE", 1);
        v.qw = env_int("KA_QW", 4); v.lw = env_int("KA_LW", 4); v.pw = env_int("KA_PW", 2);
        for (int* w : { &v.qw, &v.lw, &v.pw }) if (*w != 1 && *w != 2) *w = 4;
        v.mw = env_int("KA_MW", 1);
        v.ho = env_int("KA_HO", -1);
        v.per = env_int("KA_PER", 0);
        v.hw = env_int("KA_HW", 1);
        v.hw_prio = std::max(0, std::min(3, env_int("KA_HW_PRIO", 3)));
        v.upgma_launches = getenv("KA_UPGMA_LAUNCHES") != nullptr;
}

struct ka_ctx {
        KaEnv env;
        int device = 0;
        hipStream_t stream = nullptr;
        // ---- tree job ----
        bool have_job = false;
        int numseq = 0, n_tasks = 0, flags = 0;
        std::vector<int> lens, off;
        std::vector<int> abc;
        std::vector<KaTaskDesc> descs;
        std::vector<std::vector<int>> levels;        // task ids per dependency level
        std::vector<std::vector<int>> plan_levels;   // ... of the tasks the current launch plan covers (plan_launches)
        std::vector<char> plan_active;               // the tasks it covers (empty: the whole job) -- ka_tree_plan_tasks
        std::vector<int> level_ids_flat, level_off;
        std::vector<int2> blocks_flat;               // per level: (task, member | cluster size << 8) per workgroup
        std::vector<int> blocks_off;
        std::vector<int> level_lean;                 // level consists of seq-seq tasks only -> lean kernel
        int max_cluster = 16;                        // KA_MAX_CLUSTER env: workgroups (CUs) one task may use
        int refine_mode = 0;                         // the run in flight is a refinement pass (ka_tree_refine): 1 all, 2 confident
        DevBuf<int2> d_refine_blocks;                   // its workgroup table, level after level (refine_blocks)
        std::vector<int> refine_off;                    // [levels + 1] first block of every level in it
        int n_cus = 256;                             // compute units of the device (hipDeviceProp)
        bool shared_gpu = false;                     // ka_ctx_set_shared: no multi-workgroup tasks, no chained launch
        bool shared_by_fallback = false;             // shared_gpu was forced by a join watchdog (ka_tree_sync), not by the caller
        int fallback_runs = 0;                       // how often that happened (ka_ctx_fallback_runs)
        int test_hooks = 0;                          // ka_debug_set_hooks (tests only)
        std::vector<long long> leaf_prof_off;
        long long leaf_prof_total = 0;
        long long sum_len = 0;
        int max_len = 0;
        float subm[23 * 23];
        float scal[6];
        int nres = 23;
        DevBuf<uint8_t> d_codes;
        DevBuf<int> d_seq_off, d_node_len, d_level_ids, d_path_arena, d_error;
        DevBuf<long long> d_node_prof, d_dbg_off, d_timing;
        DevBuf<float> d_prof_arena, d_subm, d_dbg_arena;
        DevBuf<unsigned long long> d_counters;
        DevBuf<char> d_scratch, d_ctl;
        DevBuf<KaJoin> d_join;
        int n_trees = 1;               // guide trees in the job (a forest when > 1)
        int chain_level = -1;          // first level of the chained launch (-1: every level is its own launch)
        int queue_first = -1;          // queued launch: levels queue_first .. chain_level-1 run as ONE launch of the half kernel (-1: none)
        int queue_off = 0, queue_n = 0; // its task list in blocks_flat
        std::vector<int2> chain_blocks;
        int chain_blocks_off = 0;
        DevBuf<int2> d_blocks;
        DevBuf<KaTaskDesc> d_tasks;
        DevBuf<ka_task_rec> d_recs;
        long long prof_cap = 0, path_cap = 0, scratch_cap = 0, dbg_cap = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        std::vector<hipEvent_t> launch_ev;           // KA_LAUNCH_EV: one event behind every launch of the last run
        // two pinned bounce buffers for large downloads into the caller's (pageable) memory
        char* pin[2] = { nullptr, nullptr };
        hipEvent_t pin_ev[2] = { nullptr, nullptr };
        int* h_trace = nullptr;       // pinned, device-visible breadcrumbs (KA_TRACE=1)
        bool ran = false, synced = false;
        bool state_valid = false;      // device state reset and consistent with task_done
        bool partial = false;          // last launch was ka_tree_run_tasks (no automatic grow + re-run)
        std::vector<char> task_done;
        std::vector<int> injected;       // nodes whose profile came from ka_tree_set_profile
        std::vector<int> task_level;
        DevBuf<int2> d_blocks_tmp;
        int n_launches = 0;
        double cells = 0.0;
        float pair_ms = 0.0f;                        // kernel time of the last ka_pairwise_batch
        // grow-only device buffers of ka_pairwise_batch (no hipMalloc/hipFree per call)
        DevBuf<uint8_t> p_codes; DevBuf<int> p_off, p_len, p_ia, p_ib, p_paths, p_err; DevBuf<float> p_subm, p_scores;
        DevBuf<long long> p_poff; DevBuf<char> p_scr;
        DevBuf<unsigned long long> b_peq; DevBuf<int> b_dist;   // ka_bpm_batch
        std::vector<ka_task_rec> h_recs;
        unsigned long long h_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // ---- anchor consistency (ka_tree_build_consistency) ----
        std::vector<uint8_t> h_codes;                // host copy of the uploaded sequences
        std::vector<float> seq_dist;                 // msa->seq_distances (empty: none)
        std::vector<int> sip_flat;                   // member lists of every node, reference order
        std::vector<long long> sip_off;
        int cons_K = 0;
        size_t colof_n = 0;
        bool have_colof = false;       // residue->column tables + member lists are on the device
        float cons_weight = 0.0f;
        std::vector<int> cons_anchor_ids, cons_maps;  // cons_maps: host copy of d_cons_maps, filled on demand
        long long cons_maps_total = 0;
        std::vector<long long> cons_map_off;
        DevBuf<int> d_cons_maps, d_colof, d_colof_init, d_sip, d_alnlen, d_pair_of;
        DevBuf<uint8_t> d_letters, d_rows;
        long long rows_stride = 0; int rows_n = 0, rows_alnlen = 0; uint8_t rows_gap = 0;   // what d_rows holds (0 rows: nothing)
        DevBuf<float> d_adm, d_amean; DevBuf<int> d_uactive; DevBuf<unsigned long long> d_ucand; DevBuf<int2> d_umerges;
        DevBuf<long long> d_cons_map_off, d_sip_off;
};

static void build_blocks(const ka_ctx* c, const std::vector<int>& L, std::vector<int2>& tbl, int* lean_out);
static int plan_launches(ka_ctx* c);
static int upload_plan(ka_ctx* c);
static int setup_colof(ka_ctx* c);
static int refine_blocks(ka_ctx* c, int mode);
static int pairwise_on_device(ka_ctx* c, const uint8_t* codes, const int* off, const int* lens, int numseq,
                              const int* ia, const int* ib, int npairs,
                              const float* subm, float gpo, float gpe, float tgpe, const long long* poff, long long* ptotal_out);
static void node_members(const ka_ctx* c, int node, long long* lo, long long* hi);

extern "C" const char* ka_last_error(void) { return g_err.c_str(); }
struct ka_ctx;
int ka_ctx_device_stream(ka_ctx* c, int* device, hipStream_t* stream);    // (library-internal: ka_guide.cpp)
extern "C" int ka_abi_version(void) { return 8; }

extern "C" int ka_ctx_create(int device, ka_ctx** out)
{
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail("no HIP device visible (the HIP path has no CPU fallback)");
        if (device < 0 || device >= n) return fail("bad device index");
        HIPCHK(hipSetDevice(device));
        ka_ctx* c = new ka_ctx();
        c->device = device;
        {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
        }
        hipError_t e = hipEventCreate(&c->ev0);
        if (e == hipSuccess) e = hipEventCreate(&c->ev1);
        read_env(c->env);
        if (e == hipSuccess && c->env.trace) {
                e = hipHostMalloc((void**)&c->h_trace, 64 * sizeof(int), hipHostMallocMapped);
                if (e == hipSuccess) memset(c->h_trace, 0xff, 64 * sizeof(int));
        }
        if (e != hipSuccess) {
                ka_ctx_destroy(c);
                return fail(std::string("ka_ctx_create: ") + hipGetErrorString(e));
        }
        *out = c;
        return KA_OK;
}

// Tests only: fault injection that used to hide behind environment variables.
__attribute__((visibility("hidden"))) int ka_ctx_device_stream(ka_ctx* c, int* device, hipStream_t* stream)
{
        if (!c) return 1;
        *device = c->device; *stream = c->stream;
        return 0;
}

extern "C" int ka_debug_set_hooks(ka_ctx* c, int hooks)
{
        if (!c) return fail("null ctx");
        c->test_hooks = hooks;
        return KA_OK;
}

// Tools and tests that flip a KA_* switch on a live context: the environment is otherwise read once, at ka_ctx_create.
// The launch plan of an uploaded job is rebuilt.
extern "C" int ka_debug_reload_env(ka_ctx* c)
{
        if (!c) return fail("null ctx");
        read_env(c->env);
        if (c->have_job) {
                if (c->ran && !c->synced && ka_tree_sync(c)) return KA_FAIL;
                for (auto& d : c->descs) d.refine = 0;
                c->refine_mode = 0;
                if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
        }
        return KA_OK;
}

// How often a run of this context had to fall back to the no-cluster / no-chain plan because workgroups that wait for
// each other were not all resident (somebody else was using the GPU).
extern "C" int ka_ctx_fallback_runs(ka_ctx* c) { return c ? c->fallback_runs : -1; }

extern "C" void ka_ctx_destroy(ka_ctx* c)
{
        if (!c) return;
        (void)hipSetDevice(c->device);
        c->d_codes.release(); c->d_seq_off.release(); c->d_node_len.release(); c->d_level_ids.release();
        c->d_path_arena.release(); c->d_error.release(); c->d_node_prof.release(); c->d_dbg_off.release();
        c->d_prof_arena.release(); c->d_subm.release(); c->d_dbg_arena.release(); c->d_counters.release();
        c->d_scratch.release(); c->d_tasks.release(); c->d_recs.release(); c->d_timing.release();
        c->d_ctl.release(); c->d_blocks.release(); c->d_blocks_tmp.release(); c->d_join.release();
        c->p_codes.release(); c->p_off.release(); c->p_len.release(); c->p_ia.release(); c->p_ib.release(); c->p_paths.release();
        c->p_err.release(); c->p_subm.release(); c->p_scores.release(); c->p_poff.release(); c->p_scr.release();
        c->b_peq.release(); c->b_dist.release();
        c->d_letters.release(); c->d_rows.release(); c->d_alnlen.release(); c->d_pair_of.release();
        c->d_adm.release(); c->d_amean.release(); c->d_uactive.release(); c->d_ucand.release(); c->d_umerges.release();
        c->d_cons_maps.release(); c->d_colof.release(); c->d_colof_init.release(); c->d_sip.release();
        c->d_cons_map_off.release(); c->d_sip_off.release();
        for (int k = 0; k < 2; k++) { if (c->pin[k]) (void)hipHostFree(c->pin[k]); if (c->pin_ev[k]) (void)hipEventDestroy(c->pin_ev[k]); }
        if (c->h_trace) (void)hipHostFree(c->h_trace);
        for (hipEvent_t e : c->launch_ev) (void)hipEventDestroy(e);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        delete c;
}

// Clusters of workgroups and the chained launch let workgroups wait for each other, which is only safe while
// all of them are resident -- true when this context has the GPU to itself.  A context that shares the GPU
// with other streams or processes (several alignments in flight at once) must say so: every task then runs on
// one workgroup and every guide-tree level is its own launch.  Takes effect at the next ka_tree_upload.
extern "C" int ka_ctx_set_shared(ka_ctx* c, int shared)
{
        if (!c) return fail("null ctx");
        c->shared_gpu = shared != 0; c->shared_by_fallback = false;
        return KA_OK;
}

extern "C" int ka_ctx_set_stream(ka_ctx* c, void* s)
{
        if (!c) return fail("null ctx");
        c->stream = (hipStream_t)s;
        return KA_OK;
}

// mean seq_distance over both clusters in sip order (aln_run.c:126-203)
static float mean_distance(const float* dist, const std::vector<int>& ma, const std::vector<int>& mb, int numseq, int* count)
{
        float sum = 0.0f;
        int n = 0;
        for (int x : ma) if (x < numseq) { sum += dist[x]; n++; }
        for (int x : mb) if (x < numseq) { sum += dist[x]; n++; }
        *count = n;
        return n ? sum / (float)n : 0.0f;
}

// Launch plan of the uploaded job: parents and join counts of the chained launch, workgroup tables per level.
// Depends on c->shared_gpu (no clusters, no chain), so ka_tree_sync can re-plan after a residency failure.
static int plan_launches(ka_ctx* c)
{
        const int numseq = c->numseq, n_tasks = c->n_tasks;
        const int* abc = c->abc.data();
        const int max_level = (int)c->levels.size();
        // the tasks this plan covers: all of them, or the subset of ka_tree_plan_tasks (a rank's subtrees of a sharded
        // tree: closed under descendants).  A task outside the plan is neither a parent nor a producer in it.
        const bool subset = !c->plan_active.empty();
        auto act = [&](int t) { return !subset || c->plan_active[t] != 0; };
        c->plan_levels.assign(max_level, std::vector<int>());
        for (int L = 0; L < max_level; L++) for (int t : c->levels[L]) if (act(t)) c->plan_levels[L].push_back(t);
        const std::vector<std::vector<int>>& levels = c->plan_levels;
        // ---- parents, and the level from which the rest of the tree runs as ONE chained launch: the first
        // non-leaf level with at most one task per CU (all its workgroups resident at once; levels only get
        // narrower above it).  KA_NO_CHAIN=1 keeps one launch per level.
        {
                std::vector<int> task_of((2 * numseq - 1), -1);
                for (int t = 0; t < n_tasks; t++) task_of[abc[3 * t + 2]] = t;
                for (int t = 0; t < n_tasks; t++) { c->descs[t].parent = -1; c->descs[t].chain_need = 0; }
                for (int t = 0; t < n_tasks; t++) c->descs[t].is_root = 1;
                for (int t = 0; t < n_tasks; t++) {
                        const int a = abc[3 * t], b = abc[3 * t + 1];
                        // (is_root is a property of the tree: the root's task builds no profile.  parent is one of the plan.)
                        if (a >= numseq) { c->descs[task_of[a]].is_root = 0; if (act(t) && act(task_of[a])) c->descs[task_of[a]].parent = t; }
                        if (b >= numseq) { c->descs[task_of[b]].is_root = 0; if (act(t) && act(task_of[b])) c->descs[task_of[b]].parent = t; }
                }
                {
                        // join watchdog of the chained launch: ~2 s per 4e9 estimated DP cells below the task (a healthy
                        // sibling subtree of a huge job may legitimately take longer than the base bound)
                        std::vector<double> len(2 * numseq - 1, 0.0), cells(2 * numseq - 1, 0.0);
                        for (int i = 0; i < numseq; i++) len[i] = c->lens[i];
                        for (int t = 0; t < n_tasks; t++) {
                                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                                len[cc] = 1.1 * std::max(len[a], len[b]);
                                cells[cc] = cells[a] + cells[b] + len[a] * len[b];
                                c->descs[t].wait_mult = 1 + (int)std::min(63.0, cells[cc] / 4e9);
                                // (descs[t].refine -- the edges a KALIGN_REFINE_CONFIDENT pass refines -- is not part of the plan: it is
                                // set by ka_tree_refine and must survive the re-plan of a watchdog fallback, ka_tree_sync)
                        }
                }
                c->n_trees = numseq - n_tasks;
                c->chain_level = -1;
                if (!c->env.no_chain && !c->shared_gpu) {
                        for (int L = 0; L + 1 < max_level; L++) {
                                bool all_ss = true;
                                for (int t : levels[L]) if (c->descs[t].nsip_a != 1 || c->descs[t].nsip_b != 1) all_ss = false;
                                int chain_tasks = c->n_cus - 8;
                                if (c->env.chain_tasks > 0) chain_tasks = std::min(chain_tasks, c->env.chain_tasks);   // experiments
                                if (!all_ss && (int)levels[L].size() <= chain_tasks) { c->chain_level = L; break; }   // one workgroup per CU, all resident
                        }
                }
                if (c->chain_level >= 0) {
                        for (int t = 0; t < n_tasks; t++) {
                                if (c->task_level[t] <= c->chain_level || !act(t)) continue;
                                int need = 0;
                                for (int k = 0; k < 2; k++) {
                                        const int ch = abc[3 * t + k];
                                        if (ch >= numseq && act(task_of[ch]) && c->task_level[task_of[ch]] >= c->chain_level) need++;
                                }
                                c->descs[t].chain_need = need;
                        }
                        // tests: make the last join wait for a workgroup that never comes (a residency failure as seen
                        // from the device) -- the bounded wait must report it and ka_tree_sync must re-plan and re-run
                        if (c->test_hooks & KA_DEBUG_STARVE_ROOT_JOIN) c->descs[n_tasks - 1].chain_need += 1;
                }
                // ---- the queued launch: every level between the seq-seq leaves and the chained launch (each holds more
                // tasks than the GPU has workgroup slots) as ONE launch of the half kernel; see ka_task_queue_entry.
                // KA_NO_QUEUE=1 keeps one launch per level.
                for (int t = 0; t < n_tasks; t++) { c->descs[t].qa = -1; c->descs[t].qb = -1; }
                c->queue_first = -1;
                if (c->chain_level >= 1 && !c->env.no_queue && !c->env.no_half) {
                        int L0 = 0;
                        while (L0 < c->chain_level) {                       // skip the leading seq-seq levels (lean kernel)
                                bool all_ss = true;
                                for (int t : levels[L0]) if (c->descs[t].nsip_a != 1 || c->descs[t].nsip_b != 1) all_ss = false;
                                if (!all_ss) break;
                                L0++;
                        }
                        bool ok = c->chain_level - L0 >= 2;                 // one level alone gains nothing
                        if ((int)levels[L0].size() <= c->n_cus) ok = false;   // (the queue's first level must fill the GPU; later ones need not)
                        if (ok) {
                                c->queue_first = L0;
                                for (int t = 0; t < n_tasks; t++) {
                                        if (c->task_level[t] < L0 || c->task_level[t] >= c->chain_level || !act(t)) continue;
                                        const int a = abc[3 * t], b = abc[3 * t + 1];
                                        if (a >= numseq && act(task_of[a]) && c->task_level[task_of[a]] >= L0) c->descs[t].qa = task_of[a];
                                        if (b >= numseq && act(task_of[b]) && c->task_level[task_of[b]] >= L0) c->descs[t].qb = task_of[b];
                                }
                        }
                }
        }

        // ---- workgroup tables, one per dependency level (build_blocks) ----
        // Workgroups one task may use: 16, or 32 for jobs whose top tasks are big enough to be work-bound at 16 (round 4: a
        // 9000 x 9700 task of C3 takes 5.8 ms on 16 workgroups, of which ~1.6 ms are the wavefront's dependent steps) -- by the
        // estimated root (longest sequence x (1 + 0.1 sqrt(sequences)), squared): >= 6e7 cells.  Measured, limit 16 -> 32
        // (profiles/r04_max_cluster.log): C3 81.9 -> 74.8 ms, 1024 x 2000 nt 34.9 -> 33.0, 512 x 3000 nt 43.6 -> 41.9; 16384 x 500 aa
        // and 2048 x 1000 aa unchanged; 4096 x 400 aa and 8192 x 300 aa 1-2 % slower (surplus members waiting at the joins).
        {
                double lmax = 0.0;
                for (int i = 0; i < numseq; i++) lmax = std::max(lmax, (double)c->lens[i]);
                const double root = lmax * (1.0 + 0.1 * std::sqrt((double)numseq));
                // (... and for big jobs with a consistency table: the votes of their top tasks share by member ranges from 20 workgroups on)
                const bool big_cons = c->cons_K > 0 && numseq >= 2048;
                c->max_cluster = c->env.max_cluster > 0 ? std::min(32, c->env.max_cluster) : ((root * root >= 6e7 || big_cons) ? 32 : 16);
        }
        if (c->shared_gpu) c->max_cluster = 1;
        c->blocks_flat.clear(); c->blocks_off.assign(1, 0); c->level_lean.clear();
        for (auto& L : levels) {
                std::vector<int2> tbl;
                int lean = 0;
                build_blocks(c, L, tbl, &lean);
                c->level_lean.push_back(lean);
                c->blocks_flat.insert(c->blocks_flat.end(), tbl.begin(), tbl.end());
                c->blocks_off.push_back((int)c->blocks_flat.size());
        }

        c->queue_off = (int)c->blocks_flat.size(); c->queue_n = 0;
        if (c->queue_first >= 0) {
                for (int L = c->queue_first; L < c->chain_level; L++)
                        for (int t : levels[L]) { c->blocks_flat.push_back(make_int2(t, 1 << 8)); c->queue_n++; }
        }
        if (c->chain_level >= 0) {
                // Every task of the chain's first level starts on a single workgroup; clusters form on the way up.
                // Entries are laid out in depth-first order of the upper tree, one contiguous run per XCD
                // (block b runs on XCD b % 8 -- observed, not contractual): subtrees that merge early share an
                // L2, only the top three levels cross XCDs.
                std::vector<int> task_of((2 * numseq - 1), -1), order;
                for (int t = 0; t < n_tasks; t++) task_of[abc[3 * t + 2]] = t;
                std::vector<int> stack;
                for (int t = n_tasks - 1; t >= 0; t--) if (act(t) && c->descs[t].parent < 0 && c->task_level[t] >= c->chain_level) stack.push_back(t);   // every root above the cut
                while (!stack.empty()) {
                        const int t = stack.back(); stack.pop_back();
                        // an entry of the chain: no child of it runs inside the launch (the chain's first level; in a plan over a
                        // subset also a task whose children were all run before)
                        if (c->task_level[t] == c->chain_level || c->descs[t].chain_need == 0) { order.push_back(t); continue; }
                        for (int k = 1; k >= 0; k--) {
                                const int ch = abc[3 * t + k];
                                if (ch >= numseq && act(task_of[ch]) && c->task_level[task_of[ch]] >= c->chain_level) stack.push_back(task_of[ch]);
                        }
                }
                const int m = ((int)order.size() + 7) / 8;
                // A narrow upper tree (the chain-like UPGMA trees of a realignment pass) never merges clusters: its
                // tasks would all run on the one workgroup they started with.  Start with as many workgroups per
                // task as a separate launch of this level would get (build_blocks); members of one cluster sit in
                // one column = one XCD.
                int G0 = 1;
                while (G0 * 2 <= c->max_cluster && 8 * m * G0 * 2 <= c->n_cus) G0 *= 2;
                if (c->env.chain_g1) G0 = 1;
                // The CUs this leaves idle go to the entries whose way to the root is the longest (estimated wavefront steps
                // of the tasks above them): clusters only grow where subtrees of the SAME launch meet, and the critical path
                // of a k-means tree is a caterpillar that absorbs small subtrees finished by earlier launches -- its tasks
                // would run on the one workgroup their entry started with while most of the GPU waits at join points.  A
                // cluster keeps its workgroups all the way up (surplus members climb with it), so a workgroup given to an
                // entry serves every task on that entry's path.  Extra members sit behind the regular table, in the
                // entry's XCD column.
                std::vector<int> extra(order.size(), 0);
                int spare = (c->n_cus - 8 * m * G0) / 8 * 8;
                if (!c->env.no_crit && spare > 0 && !order.empty()) {
                        std::vector<double> len(2 * numseq - 1, 0.0), up(n_tasks, 0.0);
                        for (int i = 0; i < numseq; i++) len[i] = c->lens[i];
                        if (c->env.crit_greedy) {
                                // profile lengths are only known on the device; the estimate: the longest member sequence times
                                // (1 + 0.1 sqrt(members)) -- the growth of the DSSim sets with their indel-rich tails (13151 columns for
                                // 4096 x 2000 nt, 2965 for 4096 x 400 aa), harmless where alignments stay shorter
                                std::vector<double> lmax(2 * numseq - 1, 0.0), nmem(2 * numseq - 1, 1.0);
                                for (int i = 0; i < numseq; i++) lmax[i] = c->lens[i];
                                for (int t = 0; t < n_tasks; t++) {
                                        const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                                        lmax[cc] = std::max(lmax[a], lmax[b]); nmem[cc] = nmem[a] + nmem[b];
                                        len[cc] = lmax[cc] * (1.0 + 0.1 * std::sqrt(nmem[cc]));
                                }
                        } else
                        for (int t = 0; t < n_tasks; t++) len[abc[3 * t + 2]] = 1.1 * std::max(len[abc[3 * t]], len[abc[3 * t + 1]]);
                        for (int t = n_tasks - 1; t >= 0; t--) {               // parents come after their children in the task list
                                const double la = len[abc[3 * t]], lb = len[abc[3 * t + 1]];
                                up[t] = 2.0 * std::max(la, lb) + std::min(la, lb) + (c->descs[t].parent >= 0 ? up[c->descs[t].parent] : 0.0);
                        }
                        // Round 4: first a GREEDY pass on a simulated schedule.  The ranking below only knows how LONG an entry's way
                        // to the root is, not how it will be staffed: a caterpillar spine that absorbs siblings finished by earlier
                        // launches stays on the one workgroup of its entry through level after level of 2400 x 2300 tasks (C3: six
                        // of them at 3.8 ms, a third of the launch, next to ~200 idle CUs) while a spine fed by subtrees of THIS
                        // launch collects their workgroups at every join.  Model: a task on G workgroups takes
                        // a * (2 max + min) + b * la * lb / G (fitted on C3's and the headline's task times: the first term the
                        // wavefront's dependent steps, the second the cells shared by the cluster; b / a = 0.02 from the fit, 0.01 in use), a parent has the
                        // workgroups of its children in this launch (up to the limit) and starts when the later one ends.  One spare
                        // workgroup at a time goes to the entry under the simulated critical path, until it stops paying; what is
                        // left goes out by the ranking.  KA_CRIT_GREEDY=0: the ranking alone (round 3).
                        if (c->env.crit_greedy) {
                                std::vector<int> entry_of(n_tasks, -1);
                                for (size_t r = 0; r < order.size(); r++) entry_of[order[r]] = (int)r;
                                auto in_chain = [&](int t) { return t >= 0 && act(t) && c->task_level[t] >= c->chain_level; };
                                std::vector<double> fin(n_tasks, 0.0);
                                std::vector<int> Gt(n_tasks, 0), crit_child(n_tasks, -1);
                                const double ba = 1e-3 * (double)env_int("KA_CRIT_BA", 10);   // (b / a of the model, per mille; 10 from a sweep over five job shapes, profiles/r04_crit_ba.log)
                                auto simulate = [&]() -> int {
                                        int last = -1;
                                        for (int t = 0; t < n_tasks; t++) {                  // children come before their parents
                                                if (!in_chain(t)) continue;
                                                double start = 0.0; int cc = -1, G = 0;
                                                if (entry_of[t] >= 0) G = G0 + extra[entry_of[t]];
                                                else {
                                                        for (int k = 0; k < 2; k++) {
                                                                const int ch = abc[3 * t + k];
                                                                const int tc = ch >= numseq ? task_of[ch] : -1;
                                                                if (!in_chain(tc)) continue;
                                                                G += Gt[tc];
                                                                if (fin[tc] >= start) { start = fin[tc]; cc = tc; }
                                                        }
                                                        G = std::max(1, std::min(G, c->max_cluster));
                                                }
                                                const double la = len[abc[3 * t]], lb = len[abc[3 * t + 1]];
                                                fin[t] = start + 2.0 * std::max(la, lb) + std::min(la, lb) + ba * la * lb / G;
                                                Gt[t] = G; crit_child[t] = cc;
                                                if (last < 0 || fin[t] > fin[last]) last = t;
                                        }
                                        return last;                                       // the task that ends last (a root)
                                };
                                // (several paths can be critical at once: a workgroup that shortens ONE of them leaves the end where it was.
                                // Keep going -- the next round takes the next path -- and fall back to the best state seen when a
                                // stretch of eight additions has not moved the end.)
                                int given = 0, since_best = 0;
                                std::vector<int> best_extra = extra;
                                int best_spare = spare;
                                double best_end = -1.0;
                                { const int t = simulate(); if (t >= 0) best_end = fin[t]; }
                                while (spare > 0 && best_end > 0.0 && since_best < 8) {
                                        int t = simulate();
                                        if (t < 0) break;
                                        while (crit_child[t] >= 0) t = crit_child[t];        // down the critical path to its entry
                                        const int r = entry_of[t];
                                        if (r < 0 || G0 + extra[r] >= c->max_cluster) break;
                                        extra[r] += 1; spare -= 1;
                                        const int t2 = simulate();
                                        if (fin[t2] < best_end * (1.0 - 1e-4)) { best_end = fin[t2]; best_extra = extra; best_spare = spare; since_best = 0; }
                                        else since_best += 1;
                                }
                                extra = best_extra; spare = best_spare;
                                for (size_t r = 0; r < order.size(); r++) given += extra[r];
                                if (getenv("KA_PLAN_VERBOSE")) {
                                        const int t = simulate();
                                        fprintf(stderr, "chain plan: greedy pass gave %d workgroups, %d left for the ranking; simulated end %.0f\n", given, spare, t >= 0 ? fin[t] : 0.0);
                                        for (size_t r = 0; r < order.size(); r++) if (extra[r] > 0)
                                                fprintf(stderr, "  entry task %d (node %d) level %d: +%d\n", order[r], abc[3 * order[r] + 2], c->task_level[order[r]], extra[r]);
                                }
                        }
                        std::vector<int> by_up(order.size());
                        for (size_t r = 0; r < order.size(); r++) by_up[r] = (int)r;
                        std::stable_sort(by_up.begin(), by_up.end(), [&](int x, int y) { return up[order[x]] > up[order[y]]; });
                        int top_g = 4;
                        if (c->env.crit_top > 0) top_g = std::min(c->max_cluster, c->env.crit_top);   // experiments
                        for (size_t i = 0; i < by_up.size() && spare > 0; i++) {
                                // (never beyond the cluster limit: surplus workgroups would only spin at a join and leave)
                                const int want = std::min(spare, std::max(0, std::min(c->max_cluster, i == 0 ? top_g : 2 * G0) - G0 - extra[by_up[i]]));
                                extra[by_up[i]] += want; spare -= want;
                        }
                        if (getenv("KA_PLAN_VERBOSE")) {
                                fprintf(stderr, "chain plan: level %d, %zu entries, G0 %d, spare after extras %d, top_g %d\n", c->chain_level, order.size(), G0, spare, top_g);
                                for (size_t i = 0; i < by_up.size() && i < 12; i++) {
                                        const int t = order[by_up[i]];
                                        fprintf(stderr, "  rank %zu: task %d (node %d) level %d lens %.0f x %.0f up %.0f extra %d\n", i, t, abc[3 * t + 2], c->task_level[t],
                                                len[abc[3 * t]], len[abc[3 * t + 1]], up[t], extra[by_up[i]]);
                                }
                        }
                }
                int n_extra = 0;
                std::vector<int> col_need(8, 0);
                for (size_t r = 0; r < order.size(); r++) { n_extra += extra[r]; col_need[r / m] += extra[r]; }
                int extra_rows = *std::max_element(col_need.begin(), col_need.end());
                const bool by_column = 8 * m * G0 + 8 * extra_rows <= c->n_cus;       // else: packed densely, any XCD
                if (!by_column) extra_rows = (n_extra + 7) / 8;
                c->chain_blocks.assign((size_t)8 * m * G0 + (size_t)8 * extra_rows, make_int2(-1, 0));
                std::vector<int> col_fill(8, 0);
                int dense = 0;
                for (int r = 0; r < (int)order.size(); r++) {
                        const int Gr = G0 + extra[r];
                        for (int g = 0; g < G0; g++)
                                c->chain_blocks[((size_t)(r % m) * G0 + g) * 8 + (r / m)] = make_int2(order[r], g | (Gr << 8));
                        for (int g = G0; g < Gr; g++) {
                                const size_t pos = (size_t)8 * m * G0 + (by_column ? (size_t)8 * col_fill[r / m]++ + (r / m) : (size_t)dense++);
                                c->chain_blocks[pos] = make_int2(order[r], g | (Gr << 8));
                        }
                }
                c->chain_blocks_off = (int)c->blocks_flat.size();
                c->blocks_flat.insert(c->blocks_flat.end(), c->chain_blocks.begin(), c->chain_blocks.end());
        }

        return KA_OK;
}

extern "C" int ka_tree_upload(ka_ctx* c, int numseq, const uint8_t* codes, const int* off, const int* lens,
                              const float* seq_distances, int n_tasks, const int* abc,
                              const float* subm, const float* scal, int flags)
{
        if (!c) return fail("null ctx");
        // n_tasks == numseq-1: one guide tree.  Fewer tasks: a FOREST -- several independent alignments (a batch of
        // families, ensemble members) scheduled together; every task with no consumer is the root of its tree.
        if (numseq < 2 || n_tasks < 1 || n_tasks > numseq - 1) return fail("need numseq >= 2 and 1 <= n_tasks <= numseq-1");
        HIPCHK(hipSetDevice(c->device));
        const int nprof = 2 * numseq - 1;
        // kalign_run_realign aligns a second time on a new tree with the consistency table of the first pass
        // (aln_wrap.c:424-431,497-502): same sequences, new task list.  Anything else starts without a table.
        bool keep_cons = false;
        if ((flags & KA_FLAG_KEEP_CONSISTENCY) && c->have_job && c->cons_K > 0) {
                bool same = numseq == c->numseq;
                for (int i = 0; same && i < numseq; i++)
                        same = lens[i] == c->lens[i] && off[i] == c->off[i] && memcmp(codes + off[i], c->h_codes.data() + off[i], lens[i]) == 0;
                if (!same) return fail("KA_FLAG_KEEP_CONSISTENCY: the sequences differ from those the consistency table was built on");
                keep_cons = true;
        }
        c->have_job = false; c->ran = false; c->synced = false; c->state_valid = false;
        // a join watchdog of an earlier job forced the no-cluster plan: a new job gets the fast plan again (the
        // fallback is counted, ka_ctx_fallback_runs); a caller's own ka_ctx_set_shared stays
        if (c->shared_by_fallback) { c->shared_gpu = false; c->shared_by_fallback = false; }
        if (!keep_cons) c->cons_K = 0;           // a new job starts without a consistency table
        c->have_colof = false;
        c->rows_n = 0;
        c->numseq = numseq; c->n_tasks = n_tasks; c->flags = flags;
        c->lens.assign(lens, lens + numseq);
        c->off.assign(off, off + numseq);
        c->abc.assign(abc, abc + 3 * n_tasks);
        memcpy(c->subm, subm, sizeof(c->subm));
        memcpy(c->scal, scal, sizeof(c->scal));
        c->sum_len = 0; c->max_len = 0;
        long long codes_bytes = 0;
        int max_code = 0;
        for (int i = 0; i < numseq; i++)
                for (int j = 0; j < lens[i]; j++) max_code = std::max<int>(max_code, codes[off[i] + j]);
        if (max_code > 22) return fail("sequence code out of range (alphabet is 0..22)");
        // nucleotide alphabets use codes 0..4 (alphabet.c:206-245); proteins without B / Z / X only codes 0..19
        c->nres = (max_code < 5) ? 5 : (max_code < 20 ? 20 : 23);
        for (int i = 0; i < numseq; i++) {
                if (lens[i] < 1) return fail("zero-length sequence (the reference removes them before the dispatcher, msa_check.c:66)");
                c->sum_len += lens[i];
                c->max_len = std::max(c->max_len, lens[i]);
                codes_bytes = std::max<long long>(codes_bytes, (long long)off[i] + lens[i]);
        }

        c->h_codes.assign(codes, codes + codes_bytes);
        if (seq_distances) c->seq_dist.assign(seq_distances, seq_distances + numseq); else c->seq_dist.clear();
        c->sip_flat.clear(); c->sip_off.assign(nprof, 0);
        for (int i = 0; i < numseq; i++) { c->sip_off[i] = (long long)c->sip_flat.size(); c->sip_flat.push_back(i); }

        // ---- host-side task preparation: nsip, sip order, gap_scale / subm_offset, levels ----
        std::vector<int> nsip(nprof, 0), level(nprof, 0);
        std::vector<std::vector<int>> sip(nprof);
        std::vector<char> made(nprof, 0);
        for (int i = 0; i < numseq; i++) { nsip[i] = 1; sip[i] = {i}; made[i] = 1; }
        c->descs.assign(n_tasks, KaTaskDesc());
        const float gpo0 = scal[0], gpe0 = scal[1], tgpe0 = scal[2], dist_scale = scal[3], vsm_amax = scal[4];
        int max_level = 0;
        for (int t = 0; t < n_tasks; t++) {
                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                if (a < 0 || b < 0 || cc < numseq || a >= nprof || b >= nprof || cc >= nprof || !made[a] || !made[b] || made[cc])
                        return fail("task list is not in TASK_ORDER_TREE order (children before parents)");
                // a node is the operand of at most one task, and never both operands of it (its member list is
                // handed to the parent below)
                if (a == b || made[a] == 2 || made[b] == 2)
                        return fail("task list is not in TASK_ORDER_TREE order (a node is consumed twice)");
                made[a] = 2; made[b] = 2;
                KaTaskDesc& d = c->descs[t];
                float gap_scale = 1.0f, soff = 0.0f;
                int cnt = 0;
                if (dist_scale > 0.0f && seq_distances) {
                        const float avg = mean_distance(seq_distances, sip[a], sip[b], numseq, &cnt);
                        if (cnt) {
                                gap_scale = 1.0f - dist_scale * avg;
                                if (gap_scale < 0.3f) gap_scale = 0.3f;
                                if (gap_scale > 1.0f) gap_scale = 1.0f;
                        }
                }
                if (vsm_amax > 0.0f && seq_distances) {
                        const float avg = mean_distance(seq_distances, sip[a], sip[b], numseq, &cnt);
                        if (cnt) {
                                soff = vsm_amax - avg;
                                if (soff < 0.0f) soff = 0.0f;
                        }
                }
                d.a = a; d.b = b; d.c = cc;
                d.nsip_a = nsip[a]; d.nsip_b = nsip[b];
                d.is_root = 0;                                   // set below: tasks nobody consumes
                d.gpo = gpo0; d.gpe = gpe0; d.tgpe = tgpe0;
                if (gap_scale < 1.0f || soff > 0.0f) { d.gpo *= gap_scale; d.gpe *= gap_scale; d.tgpe *= gap_scale; }
                else soff = 0.0f;
                d.soff = soff; d.gap_scale = gap_scale; d.parent = -1; d.chain_need = 0;
                nsip[cc] = nsip[a] + nsip[b];
                sip[cc].reserve(nsip[cc]);
                for (int j = nsip[a]; j--;) sip[cc].push_back(sip[a][j]);        // aln_run.c:428-436
                for (int j = nsip[b]; j--;) sip[cc].push_back(sip[b][j]);
                c->sip_off[cc] = (long long)c->sip_flat.size();
                c->sip_flat.insert(c->sip_flat.end(), sip[cc].begin(), sip[cc].end());
                std::vector<int>().swap(sip[a]);
                std::vector<int>().swap(sip[b]);
                made[cc] = 1;
                level[cc] = 1 + std::max(level[a], level[b]);
                max_level = std::max(max_level, level[cc]);
        }
        c->levels.assign(max_level, std::vector<int>());
        for (int t = 0; t < n_tasks; t++) c->levels[level[abc[3 * t + 2]] - 1].push_back(t);
        c->level_ids_flat.clear(); c->level_off.assign(1, 0);
        for (auto& L : c->levels) {
                c->level_ids_flat.insert(c->level_ids_flat.end(), L.begin(), L.end());
                c->level_off.push_back((int)c->level_ids_flat.size());
        }

        c->task_level.assign(n_tasks, 0);
        for (int t = 0; t < n_tasks; t++) c->task_level[t] = level[abc[3 * t + 2]] - 1;
        c->plan_active.clear();
        if (plan_launches(c)) return KA_FAIL;

        // ---- arenas ----
        c->leaf_prof_off.assign(numseq, 0);
        long long top = 0;
        for (int i = 0; i < numseq; i++) { c->leaf_prof_off[i] = top; top += (long long)(lens[i] + 2) * KA_REC; }
        c->leaf_prof_total = top;
        // merged profiles: alignment lengths are only known on the device; start with a generous
        // estimate and let ka_tree_sync grow + re-run on overflow.
        const long long worst_cols = c->sum_len * (long long)std::max(1, max_level) + 2LL * n_tasks;
        const long long est_cols = 3LL * (long long)n_tasks * (c->max_len + 2) + 1024;
        const long long cols = std::min(worst_cols, est_cols);
        c->prof_cap = std::max(c->prof_cap, top + cols * KA_REC);
        c->path_cap = std::max(c->path_cap, cols + c->sum_len + 2LL * numseq + 1024);
        long long scr = 0;
        // per level every sequence is a member of at most one task; profile lengths never exceed
        // the sum of their members' lengths
        const long long scr_level = ka_scratch_bytes_host(c->sum_len, c->sum_len, c->max_len) / 2 + (long long)numseq * (2048 + 12LL * c->max_len) + 65536;
        scr = scr_level;
        // the chained launch and the queued launch never reset the scratch counter: several levels' worth; grows on demand
        if (c->chain_level >= 0) scr = scr_level * (long long)std::min(max_level - c->chain_level, 8);
        if (c->max_cluster > 1 && !c->shared_gpu) scr *= 2;          // clusters: every member's private queues and row buffers
        if (c->queue_first >= 0) scr = std::max(scr, scr_level * (long long)(c->chain_level - c->queue_first));
        c->scratch_cap = std::max(c->scratch_cap, scr);
        if (c->test_hooks & KA_DEBUG_SMALL_ARENAS) {
                // tests: start with arenas that are certainly too small, so that the overflow -> grow -> re-run
                // path of ka_tree_sync is exercised (also across the join points of the chained launch)
                c->prof_cap = top + 64LL * KA_REC; c->path_cap = 64; c->scratch_cap = 1 << 16;
                c->d_prof_arena.release(); c->d_path_arena.release(); c->d_scratch.release();
        }
        c->dbg_cap = (flags & KA_FLAG_DEBUG_ROWS) ? std::max<long long>(c->dbg_cap, 6LL * (cols + 2LL * n_tasks + c->sum_len)) : c->dbg_cap;

        if (c->d_codes.alloc((size_t)codes_bytes) || c->d_seq_off.alloc(numseq) || c->d_node_len.alloc(nprof) ||
            c->d_node_prof.alloc(nprof) || c->d_level_ids.alloc(c->level_ids_flat.size()) ||
            c->d_tasks.alloc(n_tasks) || c->d_recs.alloc(n_tasks) || c->d_subm.alloc(23 * 23) ||
            c->d_counters.alloc(8) || c->d_timing.alloc(8 * (size_t)n_tasks + 48 + 512) ||
            c->d_ctl.alloc((size_t)ka_ctl_bytes_host() * n_tasks) || c->d_join.alloc(n_tasks) || c->d_blocks.alloc(c->blocks_flat.size()) || c->d_error.alloc(1) || c->d_dbg_off.alloc(n_tasks) ||
            c->d_prof_arena.alloc((size_t)c->prof_cap) || c->d_path_arena.alloc((size_t)c->path_cap) ||
            c->d_scratch.alloc((size_t)c->scratch_cap) || c->d_dbg_arena.alloc((size_t)std::max<long long>(c->dbg_cap, 1)))
                return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->d_codes.p, codes, (size_t)codes_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_seq_off.p, off, sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_level_ids.p, c->level_ids_flat.data(), sizeof(int) * c->level_ids_flat.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_tasks.p, c->descs.data(), sizeof(KaTaskDesc) * n_tasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_blocks.p, c->blocks_flat.data(), sizeof(int2) * c->blocks_flat.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_subm.p, subm, sizeof(float) * 23 * 23, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->have_job = true;
        if (((flags & KA_FLAG_DEVICE_GAPS) || keep_cons) && setup_colof(c)) { c->have_job = false; return KA_FAIL; }
        return KA_OK;
}

// (re)upload the launch plan: task descriptors (parents, join counts) and workgroup tables
static int upload_plan(ka_ctx* c)
{
        if (c->d_blocks.alloc(c->blocks_flat.size())) return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->d_tasks.p, c->descs.data(), sizeof(KaTaskDesc) * c->n_tasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_blocks.p, c->blocks_flat.data(), sizeof(int2) * c->blocks_flat.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return KA_OK;
}

// Device -> caller's memory.  A plain hipMemcpy into pageable memory runs at 2-3 GB/s here; large copies go through
// two pinned bounce buffers instead: the DMA of chunk k+1 overlaps the host-side copy of chunk k.  Ordered after
// everything queued on the context's stream; complete on return.
static int copy_to_host(ka_ctx* c, void* dst, const void* src, size_t bytes)
{
        const size_t CH = (size_t)1 << 20;
        bool staged = bytes >= 2 * CH && !c->env.no_staging;
        for (int k = 0; staged && k < 2; k++) {
                if (!c->pin[k] && hipHostMalloc((void**)&c->pin[k], CH, hipHostMallocDefault) != hipSuccess) { c->pin[k] = nullptr; staged = false; }
                if (staged && !c->pin_ev[k] && hipEventCreateWithFlags(&c->pin_ev[k], hipEventDisableTiming) != hipSuccess) { c->pin_ev[k] = nullptr; staged = false; }
        }
        if (!staged) {
                HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                return KA_OK;
        }
        const size_t n = (bytes + CH - 1) / CH;
        for (size_t k = 0; k <= n; k++) {
                if (k < n) {
                        const size_t len = std::min(CH, bytes - k * CH);
                        HIPCHK(hipMemcpyAsync(c->pin[k & 1], (const char*)src + k * CH, len, hipMemcpyDeviceToHost, c->stream));
                        HIPCHK(hipEventRecord(c->pin_ev[k & 1], c->stream));
                }
                if (k > 0) {                                          // chunk k-1 has landed: hand it to the caller while chunk k moves
                        const size_t j = k - 1, len = std::min(CH, bytes - j * CH);
                        HIPCHK(hipEventSynchronize(c->pin_ev[j & 1]));
                        memcpy((char*)dst + j * CH, c->pin[j & 1], len);
                }
        }
        return KA_OK;
}

// reset the device state so that a run is repeatable
static int tree_reset(ka_ctx* c)
{
        const int numseq = c->numseq, nprof = 2 * numseq - 1;
        std::vector<int> node_len(nprof, 0);
        std::vector<long long> node_prof(nprof, -1);
        for (int i = 0; i < numseq; i++) { node_len[i] = c->lens[i]; node_prof[i] = c->leaf_prof_off[i]; }
        unsigned long long counters[8] = { (unsigned long long)c->leaf_prof_total, 0, 0, 0, 0, 0, 0, 0 };
        int zero = 0;
        HIPCHK(hipMemcpyAsync(c->d_node_len.p, node_len.data(), sizeof(int) * nprof, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_node_prof.p, node_prof.data(), sizeof(long long) * nprof, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_counters.p, counters, sizeof(counters), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_error.p, &zero, sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(c->d_ctl.p, 0, (size_t)ka_ctl_bytes_host() * c->n_tasks, c->stream));
        HIPCHK(hipMemsetAsync(c->d_recs.p, 0, sizeof(ka_task_rec) * c->n_tasks, c->stream));
        HIPCHK(hipMemsetAsync(c->d_join.p, 0, sizeof(KaJoin) * c->n_tasks, c->stream));
        if (c->have_colof)                           // every leaf starts with residue p in column p
                HIPCHK(hipMemcpyAsync(c->d_colof.p, c->d_colof_init.p, sizeof(int) * c->colof_n, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));          // the staging vectors above are stack/heap temporaries
        c->state_valid = true;
        c->task_done.assign(c->n_tasks, 0);
        c->injected.clear();
        return KA_OK;
}

static KaTreeDev tree_dev(ka_ctx* c)
{
        KaTreeDev D;
        D.codes = c->d_codes.p; D.seq_off = c->d_seq_off.p;
        D.node_len = c->d_node_len.p; D.node_prof = c->d_node_prof.p;
        D.prof_arena = c->d_prof_arena.p; D.counters = c->d_counters.p;
        D.prof_cap = c->prof_cap; D.scratch_cap = c->scratch_cap; D.path_cap = c->path_cap; D.dbg_cap = c->dbg_cap;
        D.scratch = c->d_scratch.p; D.path_arena = c->d_path_arena.p;
        D.dbg_arena = c->d_dbg_arena.p; D.dbg_off = c->d_dbg_off.p;
        D.tasks = c->d_tasks.p; D.recs = c->d_recs.p; D.subm = c->d_subm.p;
        D.ctl = (KaCtl*)c->d_ctl.p;
        D.join = c->d_join.p;
        D.gpo0 = c->scal[0]; D.gpe0 = c->scal[1]; D.tgpe0 = c->scal[2]; D.usw = c->scal[5];
        D.numseq = c->numseq; D.flags = c->flags; D.error = c->d_error.p;
        D.nres = c->nres;
        D.trace = c->h_trace;
        D.refine_mode = 0;
        D.refine_adaptive = 0;
        D.refine_trials = 3;
        D.prof_task = -1;
        D.wdfs = (c->env.no_wdfs ? 0 : 1) | (c->env.no_ls0 ? 0 : 2) | (c->env.no_inc ? 0 : 4) | (c->env.no_ldfs ? 0 : 8);   // measurements / tests
        D.prof_task = c->env.prof_task;                                 // measurements only (tools/levels_real.py)
        D.timing = (c->flags & KA_FLAG_TIMING) ? c->d_timing.p : nullptr;
        D.max_g = std::max(1, std::min(c->max_cluster, ka_max_g_host()));
        D.q1_mode = c->env.q1 >= 0 ? c->env.q1 : (c->nres > 5 ? 4 : 0);     // (nucleotides: five residues -- a one-row step is 0.85 of a two-row one: not worth twice the strips)
        D.ho_mode = c->env.ho >= 0 ? c->env.ho : 1;
        D.per_target = c->env.per;
        D.qw = c->env.qw; D.lw = c->env.lw; D.reuse = c->env.reuse;
        D.hw_mode = c->env.hw ? (1 | (c->env.hw_prio << 4)) : 0;
        D.lean4 = c->env.lean4;
        D.sub_mode = c->env.subtree;
        D.mw_mode = c->env.mw;
        D.cons_K = c->cons_K; D.cons_maxlen = c->max_len;
        D.cons_paw = c->cons_K > 0 ? c->cons_weight / (float)c->cons_K : 0.0f;
        D.cons_maps = c->d_cons_maps.p; D.cons_map_off = c->d_cons_map_off.p;
        D.colof = c->d_colof.p; D.sip = c->d_sip.p; D.sip_off = c->d_sip_off.p;
        return D;
}

// Workgroup table of one launch: near the top of the tree there are fewer tasks than CUs, so a task
// gets a cluster of up to max_cluster workgroups (the kernel decides from the actual operand
// lengths how many of them it uses).  Workgroups of one cluster are spaced 8 blocks apart:
// block b runs on XCD b % 8 (observed, not contractual -- used for L2 locality only).
static void build_blocks(const ka_ctx* c, const std::vector<int>& L, std::vector<int2>& tbl, int* lean_out)
{
        const int nt = (int)L.size();
        int lean = 1;                                    // launch kind: 0 = 8 waves, 1 = lean, 2 = half
        for (int t : L) if (c->descs[t].nsip_a != 1 || c->descs[t].nsip_b != 1) lean = 0;
        if (c->env.no_lean) lean = 0;
        if (!lean && nt > c->n_cus && !c->env.no_half) lean = 2;   // more tasks than CUs: two 4-wave workgroups per CU
        int G = 1;
        while (lean == 0 && G * 2 <= c->max_cluster && nt * G * 2 <= c->n_cus) G *= 2;
        const int groups = (nt + 7) / 8;
        tbl.assign((size_t)groups * 8 * G, make_int2(-1, 0));
        for (int j = 0; j < nt; j++)
                for (int m = 0; m < G; m++)
                        tbl[(size_t)(j % 8) + 8 * ((size_t)m + (size_t)G * (j / 8))] = make_int2(L[j], m | (G << 8));
        *lean_out = lean;
}

// KA_LAUNCH_EV: an event behind launch number c->n_launches of the run (measurements; ka_tree_launch_ms)
static int mark_launch(ka_ctx* c)
{
        if (!c->env.launch_ev) return KA_OK;
        while ((int)c->launch_ev.size() < c->n_launches) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->launch_ev.push_back(e); }
        HIPCHK(hipEventRecord(c->launch_ev[c->n_launches - 1], c->stream));
        return KA_OK;
}

// reset: start from the leaves (a whole-tree run); false: a planned subset on top of what the context already holds
static int tree_launch(ka_ctx* c, bool reset = true)
{
        if (reset ? tree_reset(c) : (!c->state_valid && tree_reset(c))) return KA_FAIL;
        KaTreeDev D = tree_dev(c);
        D.refine_mode = c->refine_mode & 255;
        D.refine_adaptive = (c->refine_mode >> 8) & 1;
        D.refine_trials = (c->refine_mode >> 16) & 255 ? (c->refine_mode >> 16) & 255 : 3;
        c->partial = !reset;
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        c->n_launches = 0;
        for (size_t L = 0; L < c->plan_levels.size(); L++) {
                const int n = (int)c->plan_levels[L].size();
                if (!n) continue;
                if (L || !reset) HIPCHK(hipMemsetAsync(c->d_counters.p + 1, 0, sizeof(unsigned long long), c->stream));
                if (!reset && (int)L == c->queue_first) HIPCHK(hipMemsetAsync(c->d_counters.p + 4, 0, sizeof(unsigned long long), c->stream));   // (the queue's head)
                if (c->refine_mode) {
                        // refinement pass: one launch per tree level (see refine_blocks)
                        if (ka_cons_big(&D)) ka_unit9_launch(&D, c->d_refine_blocks.p + c->refine_off[L], c->refine_off[L + 1] - c->refine_off[L], c->stream);
                        else ka_unit4_launch(&D, c->d_refine_blocks.p + c->refine_off[L], c->refine_off[L + 1] - c->refine_off[L], D.cons_K > 0, c->stream);
                        c->n_launches++; if (mark_launch(c)) return KA_FAIL;
                        continue;
                }
                if ((int)L == c->queue_first) {
                        // levels queue_first .. chain_level-1: one launch, two workgroups per CU pulling from the ordered list
                        // (workgroups per CU: two of four waves; of narrower ones as many as the registers (eight waves) and the LDS (160 KB) hold)
                        // (more than fit is harmless: a workgroup that starts late finds the rest of the list, or nothing)
                        const int per_cu = c->env.qw == 4 ? 2 : (c->env.qw == 2 ? 4 : 8);
                        const int nwg = std::min(c->queue_n, per_cu * c->n_cus);
                        if (ka_cons_big(&D)) ka_unit7_launch(&D, c->d_blocks.p + c->queue_off, nwg, c->queue_n, c->stream);
                        else ka_unit2_launch(&D, c->d_blocks.p + c->queue_off, nwg, D.cons_K > 0, c->queue_n, c->stream);
                        c->n_launches++; if (mark_launch(c)) return KA_FAIL;
                        L = (size_t)c->chain_level - 1;
                        continue;
                }
                if ((int)L == c->chain_level) {
                        // this level and everything above it: one launch, tasks chained through their join points
                        ka_launch_task_level(&D, c->d_blocks.p + c->chain_blocks_off, (int)c->chain_blocks.size(), 0, 1, c->stream);
                        c->n_launches++; if (mark_launch(c)) return KA_FAIL;
                        break;
                }
                ka_launch_task_level(&D, c->d_blocks.p + c->blocks_off[L], c->blocks_off[L + 1] - c->blocks_off[L], c->level_lean[L], 0, c->stream);
                c->n_launches++; if (mark_launch(c)) return KA_FAIL;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        for (auto& L : c->plan_levels) for (int t : L) c->task_done[t] = 1;
        return KA_OK;
}

extern "C" int ka_tree_run(ka_ctx* c)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        c->ran = false; c->synced = false;
        if (!c->plan_active.empty()) {                   // the last plan covered a subset (ka_tree_plan_tasks): plan the whole tree again
                c->plan_active.clear();
                if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
        }
        if (c->refine_mode) {                            // the plan on the device carries the refine marks of the last ka_tree_refine
                c->refine_mode = 0;
                for (auto& d : c->descs) d.refine = 0;
                if (upload_plan(c)) return KA_FAIL;
        }
        if (tree_launch(c)) return KA_FAIL;
        c->ran = true;
        return KA_OK;
}

// refine_alignment (aln_refine.c:199-325): a second pass over every edge of the tree with the flip trials of
// refine_edge; mode 1 = KALIGN_REFINE_ALL, 2 = KALIGN_REFINE_CONFIDENT (edges whose first-pass confidence is at or
// below the median), 3 = KALIGN_REFINE_INLINE (create_msa_tree_inline_refine, aln_run.c:448-790: three trials per edge
// in one pass), 4 = the first pass again with the depth-first engine (task confidences are then the reference's exact
// float sums).  conf_in: the first-pass confidence of every task (the reference reads task->confidence); only
// read for mode 2, NULL = computed here by a mode-4 pass.  The job keeps its tree, parameters and consistency table.
// Workgroup table of a refinement pass.  Within one trial the meetups of an edge are serial by construction (the flip
// counter walks them in recursion order), so an edge gets ONE workgroup per trial in flight: on a level that leaves CUs
// idle the flip trials of a refined edge run side by side on 2 or 4 workgroups (ka_task_body_refine), otherwise one
// workgroup runs them one after the other.
#define KA_REFINE_MAX_G 4
static int refine_blocks(ka_ctx* c, int mode)
{
        std::vector<int2> tbl;
        c->refine_off.assign(1, 0);
        const int base_mode = mode & 255;
        const int trials3 = (mode >> 16) & 255 ? (mode >> 16) & 255 : 3;
        const int flips = base_mode == 3 ? trials3 - 1 : (base_mode == 4 ? 0 : 4);      // (adaptive budget: up to 7, shared by at most 4 members)
        bool starved = false;
        for (auto& L : c->levels) {
                int nref = 0;
                for (int t : L) nref += (flips > 0 && (base_mode != 2 || c->descs[t].refine)) ? 1 : 0;
                int G = 1;
                if (!c->shared_gpu && !c->env.refine_serial)
                        // (at most KA_REFINE_MAX_G members: the member report of ka_task_body_refine has that many slots in the
                        // task's control block -- n_trials beyond 9 would otherwise ask for 8 or more and overrun it)
                        while (G * 2 <= std::min(flips, KA_REFINE_MAX_G) && (long long)nref * G * 2 + ((long long)L.size() - nref) <= c->n_cus) G *= 2;
                for (int t : L) {
                        const int g = (flips > 0 && (base_mode != 2 || c->descs[t].refine)) ? G : 1;
                        // tests: one member of the first multi-workgroup edge never starts -- its member barrier starves, the
                        // watchdog reports it and ka_tree_sync re-plans with one workgroup per edge (the refine marks must survive)
                        const bool starve = (c->test_hooks & KA_DEBUG_STARVE_REFINE_MEMBER) && g > 1 && !starved;
                        if (starve) starved = true;
                        for (int m = 0; m < g; m++) tbl.push_back(starve && m == g - 1 ? make_int2(-1, 0) : make_int2(t, m | (g << 8)));
                }
                c->refine_off.push_back((int)tbl.size());
        }
        if (c->d_refine_blocks.alloc(tbl.size())) return fail("hipMalloc failed");
        HIPCHK(hipMemcpy(c->d_refine_blocks.p, tbl.data(), sizeof(int2) * tbl.size(), hipMemcpyHostToDevice));
        return KA_OK;
}

static int refine_launch(ka_ctx* c, int mode)
{
        if (!c->plan_active.empty()) { c->plan_active.clear(); if (plan_launches(c)) return KA_FAIL; }
        c->refine_mode = mode;
        if (refine_blocks(c, mode)) return KA_FAIL;
        if (upload_plan(c)) return KA_FAIL;
        c->ran = false; c->synced = false;
        if (tree_launch(c)) return KA_FAIL;
        c->ran = true;
        return KA_OK;
}

extern "C" int ka_tree_refine(ka_ctx* c, int mode_in, const float* conf_in)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        const int mode = mode_in & 255, adaptive = mode_in & KA_REFINE_ADAPTIVE, trials = mode_in & KA_REFINE_TRIALS(255);
        if (mode < 1 || mode > 4 || (mode_in & ~(255 | KA_REFINE_ADAPTIVE | KA_REFINE_TRIALS(255))))
                return fail("ka_tree_refine: mode must be 1 (all), 2 (confident), 3 (inline) or 4 (first pass, exact confidences), optionally | KA_REFINE_ADAPTIVE");
        if (trials && mode != 3) return fail("ka_tree_refine: KA_REFINE_TRIALS goes with mode 3 (create_msa_tree_inline_refine)");
        if (adaptive && mode != 1 && mode != 2) return fail("ka_tree_refine: KA_REFINE_ADAPTIVE goes with modes 1 and 2 (refine_edge)");
        if (c->n_tasks < 1) return fail("ka_tree_refine: no tasks");
        HIPCHK(hipSetDevice(c->device));
        if (c->ran && !c->synced && ka_tree_sync(c)) return KA_FAIL;
        if (!c->have_colof) {                            // the sum-of-pairs score reads every member's column
                if (setup_colof(c)) return KA_FAIL;
        }
        c->flags |= KA_FLAG_DEVICE_GAPS;
        std::vector<float> conf;
        if (mode == 2 && !conf_in) {
                // task->confidence of the first pass is a float sum in depth-first order (aln_controller.c:194-436): the
                // level-synchronous first pass adds the same margins in another order, so its value can differ in the last
                // bits -- and the median rule below compares them.  Run the first pass again depth first and read its sums.
                for (auto& d : c->descs) d.refine = 0;
                if (refine_launch(c, 4) || ka_tree_sync(c)) return KA_FAIL;
                std::vector<ka_task_rec> r(c->n_tasks);
                HIPCHK(hipMemcpy(r.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
                conf.resize(c->n_tasks);
                for (int t = 0; t < c->n_tasks; t++) conf[t] = r[t].confidence;
                conf_in = conf.data();
        }
        float thr = 0.0f;
        if (mode == 2) {                                 // compute_confidence_threshold (aln_refine.c:674-712): the median
                std::vector<float> v(conf_in, conf_in + c->n_tasks);
                std::sort(v.begin(), v.end());
                const int n = c->n_tasks;
                thr = (n % 2 == 0) ? (v[n / 2 - 1] + v[n / 2]) / 2.0F : v[n / 2];
        }
        for (int t = 0; t < c->n_tasks; t++) c->descs[t].refine = mode == 1 ? 1 : (mode == 2 && conf_in[t] <= thr ? 1 : 0);
        return refine_launch(c, mode | adaptive | trials);
}

extern "C" int ka_tree_sync(ka_ctx* c)
{
        if (!c || !c->ran) return fail("nothing running");
        HIPCHK(hipSetDevice(c->device));
        for (int attempt = 0; attempt < 24; attempt++) {
                int err = 0;
                HIPCHK(hipStreamSynchronize(c->stream));
                HIPCHK(hipMemcpy(&err, c->d_error.p, sizeof(int), hipMemcpyDeviceToHost));
                if (!err) {
                        HIPCHK(hipMemcpy(c->h_counters, c->d_counters.p, sizeof(c->h_counters), hipMemcpyDeviceToHost));
                        c->synced = true;
                        return KA_OK;
                }
                if (err == 5) return fail("device watchdog: a strip pipeline stopped making progress");
                if (err == 6) {
                        // workgroups that wait for each other were not all resident: somebody else is using the GPU.
                        // Fall back to the plan that needs no co-residency (ka_ctx_set_shared) and run again.
                        if (c->shared_gpu || c->partial) return fail("device watchdog: a wait between workgroups never completed");
                        c->shared_gpu = true; c->shared_by_fallback = true; c->fallback_runs++;
                        if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
                        if (c->refine_mode && refine_blocks(c, c->refine_mode)) return KA_FAIL;       // one workgroup per edge from here on
                        if (tree_launch(c)) return KA_FAIL;
                        continue;
                }
                if (c->partial) return fail("a device arena overflowed during a partial run (ka_tree_run_tasks does not re-run)");
                // an arena overflowed: grow it and run again (results are only trusted from a clean run)
                if (err == 1) { c->prof_cap *= 2; c->path_cap *= 2; c->d_prof_arena.release(); c->d_path_arena.release(); }
                else if (err == 2) { c->scratch_cap *= 2; c->d_scratch.release(); }
                else if (err == 3) { c->path_cap *= 2; c->d_path_arena.release(); }
                else { c->dbg_cap *= 2; c->d_dbg_arena.release(); }
                if (c->d_prof_arena.alloc((size_t)c->prof_cap) || c->d_path_arena.alloc((size_t)c->path_cap) ||
                    c->d_scratch.alloc((size_t)c->scratch_cap) || c->d_dbg_arena.alloc((size_t)std::max<long long>(c->dbg_cap, 1)))
                        return fail("hipMalloc failed while growing an arena");
                if (tree_launch(c)) return KA_FAIL;
        }
        return fail("device arenas kept overflowing");
}

extern "C" long long ka_tree_paths_size(ka_ctx* c)
{
        if (!c || !c->synced) return -1;
        return (long long)c->h_counters[2];
}

// make_seq + update_gaps (weave_alignment.c:41-112): fold one task's gap columns into the
// gaps[] arrays of every member sequence.
static void fold_gaps(int len, int* gis, const int* newgaps)
{
        int rel = 0;
        for (int i = 0; i <= len; i++) {
                int add = 0;
                for (int j = rel; j <= rel + gis[i]; j++) add += newgaps[j];
                rel += gis[i] + 1;
                gis[i] += add;
        }
}

// make_seq + update_gaps for every task in tree order (weave_alignment.c:41-112): host-only, needs
// only (a, b, c, path_off) of every task and the coded paths.
extern "C" int ka_weave_gaps(int numseq, const int* lens, int n_tasks, const ka_task_rec* recs, const int* paths, int* gaps_out)
{
        if (numseq < 1 || n_tasks < 0 || n_tasks > numseq - 1 || !lens || !recs || !paths || !gaps_out) return fail("ka_weave_gaps: bad arguments");
        const int nprof = 2 * numseq - 1;
        std::vector<int> goff(numseq);
        long long g = 0;
        for (int i = 0; i < numseq; i++) { goff[i] = (int)g; g += lens[i] + 1; }
        memset(gaps_out, 0, sizeof(int) * (size_t)g);
        std::vector<std::vector<int>> sip(nprof);
        for (int i = 0; i < numseq; i++) sip[i] = {i};
        std::vector<int> ga, gb;
        for (int t = 0; t < n_tasks; t++) {
                const ka_task_rec& r = recs[t];
                if (r.a < 0 || r.b < 0 || r.c < numseq || r.a >= nprof || r.b >= nprof || r.c >= nprof) return fail("ka_weave_gaps: bad task record");
                const int* p = paths + r.path_off;
                ga.assign(p[0] + 1, 0); gb.assign(p[0] + 1, 0);
                int posa = 0, posb = 0;
                for (int k = 1; p[k] != 3; k++) {
                        if (!p[k]) { posa++; posb++; }
                        else if (p[k] & 1) { ga[posa] += 1; posb++; }
                        else if (p[k] & 2) { gb[posb] += 1; posa++; }
                }
                for (int x : sip[r.a]) fold_gaps(lens[x], gaps_out + goff[x], ga.data());
                for (int x : sip[r.b]) fold_gaps(lens[x], gaps_out + goff[x], gb.data());
                sip[r.c].reserve(sip[r.a].size() + sip[r.b].size());
                sip[r.c].insert(sip[r.c].end(), sip[r.a].begin(), sip[r.a].end());
                sip[r.c].insert(sip[r.c].end(), sip[r.b].begin(), sip[r.b].end());
                std::vector<int>().swap(sip[r.a]);
                std::vector<int>().swap(sip[r.b]);
        }
        return KA_OK;
}

// alignment length of the tree each sequence belongs to (a sequence in no task aligns to itself); needs h_recs
static void tree_alnlens(ka_ctx* c, std::vector<int>& alen)
{
        alen.assign(c->lens.begin(), c->lens.end());
        for (int t = 0; t < c->n_tasks; t++) {
                if (!c->descs[t].is_root) continue;
                long long lo, hi;
                node_members(c, c->descs[t].c, &lo, &hi);
                for (long long k = lo; k < hi; k++) alen[c->sip_flat[k]] = c->h_recs[t].plen;
        }
}

extern "C" int ka_tree_download(ka_ctx* c, ka_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out)
{
        if (!c || !c->ran) return fail("nothing to download");
        if (!c->synced && ka_tree_sync(c)) return KA_FAIL;
        HIPCHK(hipSetDevice(c->device));
        const long long used = (long long)c->h_counters[2];
        if (used > paths_cap) { g_err = "paths_out too small"; return KA_ERR_PATHS_CAP; }
        c->h_recs.resize(c->n_tasks);
        HIPCHK(hipMemcpy(c->h_recs.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
        std::vector<int> arena((size_t)used);
        if (copy_to_host(c, arena.data(), c->d_path_arena.p, sizeof(int) * (size_t)used)) return KA_FAIL;
        // repack the paths in task order (arena order depends on workgroup scheduling)
        long long o = 0;
        double cells = 0.0;
        for (int t = 0; t < c->n_tasks; t++) {
                ka_task_rec& r = c->h_recs[t];
                const int n = r.plen + 2;
                memcpy(paths_out + o, arena.data() + r.path_off, sizeof(int) * n);
                r.path_off = (int)o;
                o += n;
                cells += (double)r.len_a * (double)r.len_b;
        }
        c->cells = cells;
        if (c->flags & KA_FLAG_DEBUG_ROWS) {
                // FNV-1a of the top-level rows, to compare with the reference harness
                std::vector<long long> dbg_off(c->n_tasks);
                HIPCHK(hipMemcpy(dbg_off.data(), c->d_dbg_off.p, sizeof(long long) * c->n_tasks, hipMemcpyDeviceToHost));
                std::vector<float> rows;
                for (int t = 0; t < c->n_tasks; t++) {
                        ka_task_rec& r = c->h_recs[t];
                        const int lb = r.swapped ? r.len_a : r.len_b;
                        const size_t n = 3 * (size_t)(lb + 1);
                        if (dbg_off[t] < 0) continue;
                        rows.resize(2 * n);
                        HIPCHK(hipMemcpy(rows.data(), c->d_dbg_arena.p + dbg_off[t], sizeof(float) * 2 * n, hipMemcpyDeviceToHost));
                        auto fnv = [](const void* p, size_t bytes) {
                                const unsigned char* b = (const unsigned char*)p;
                                uint64_t h = 1469598103934665603ULL;
                                for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ULL; }
                                return h;
                        };
                        if (const char* dump = getenv("KA_DUMP_ROWS")) {   // debugging aid: the rows of every task appended to a file (task, n, 2n floats)
                                if (FILE* fh = fopen(dump, "ab")) { const long long hd[2] = { t, (long long)n }; fwrite(hd, sizeof(hd), 1, fh); fwrite(rows.data(), sizeof(float), 2 * n, fh); fclose(fh); }
                        }
                        r.fhash = fnv(rows.data(), sizeof(float) * n);
                        r.bhash = fnv(rows.data() + n, sizeof(float) * n);
                }
        }
        if (recs) memcpy(recs, c->h_recs.data(), sizeof(ka_task_rec) * c->n_tasks);

        if (gaps_out && (c->flags & KA_FLAG_DEVICE_GAPS) && c->have_colof && !c->partial) {
                // the kernels kept every residue's column (make_seq / update_gaps in the device's form): the gap
                // arrays are its first differences, O(sum of lengths) instead of O(N L log N) folding on the host
                std::vector<int> col(c->colof_n);
                if (copy_to_host(c, col.data(), c->d_colof.p, sizeof(int) * c->colof_n)) return KA_FAIL;
                std::vector<int> alen;
                tree_alnlens(c, alen);
                long long g = 0;
                for (int i = 0; i < c->numseq; i++) {
                        const int* cc = col.data() + c->off[i];
                        const int len = c->lens[i];
                        const int alnlen = alen[i];
                        gaps_out[g] = cc[0];
                        for (int p = 1; p < len; p++) gaps_out[g + p] = cc[p] - cc[p - 1] - 1;
                        gaps_out[g + len] = alnlen - 1 - cc[len - 1];
                        g += len + 1;
                }
                return KA_OK;
        }
        if (gaps_out && ka_weave_gaps(c->numseq, c->lens.data(), c->n_tasks, c->h_recs.data(), paths_out, gaps_out)) return KA_FAIL;
        return KA_OK;
}


// After a run made ELSEWHERE -- the sharded tree of ka_dist_* / ka_multi_*: records and coded paths gathered, the gap arrays
// woven on the host -- this context, which holds the same uploaded job, becomes the holder of the finished alignment: the
// records go to HBM, every residue's column follows from gaps[] (make_linear_sequence, msa_op.c:578-598: column of residue p =
// p + gaps[0] + .. + gaps[p]).  ka_tree_aligned_rows, ka_aln_guide_tree and ka_tree_refine then carry on here as after ka_tree_run.
extern "C" int ka_tree_adopt_alignment(ka_ctx* c, const ka_task_rec* recs, const int* gaps)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (!recs || !gaps) return fail("ka_tree_adopt_alignment: null argument");
        HIPCHK(hipSetDevice(c->device));
        if (c->ran && !c->synced) HIPCHK(hipStreamSynchronize(c->stream));
        if (!c->have_colof && setup_colof(c)) return KA_FAIL;
        std::vector<int> col(c->colof_n, 0);
        long long g = 0;
        for (int i = 0; i < c->numseq; i++) {
                int at = 0;
                for (int p = 0; p < c->lens[i]; p++) { at += gaps[g + p]; col[(size_t)c->off[i] + p] = at + p; }
                g += c->lens[i] + 1;
        }
        for (int t = 0; t < c->n_tasks; t++)
                if (recs[t].c != c->abc[3 * t + 2]) return fail("ka_tree_adopt_alignment: the records are not this job's (task order)");
        HIPCHK(hipMemcpy(c->d_colof.p, col.data(), sizeof(int) * col.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_recs.p, recs, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyHostToDevice));
        c->flags |= KA_FLAG_DEVICE_GAPS;
        c->h_counters[2] = 0;                                         // (no coded paths of its own)
        c->ran = true; c->synced = true; c->partial = false; c->state_valid = false;
        c->rows_n = 0;
        return KA_OK;
}

// ---- finalise_alignment (msa_op.c:546-598): the aligned rows, built on the device from the residue->column tables ----
// alignment length per sequence of the finished run (checks included)
static int rows_prepare(ka_ctx* c, const uint8_t* letters, std::vector<int>& alen, int* widest)
{
        if (!c || !c->have_job || !c->ran) return fail("no finished run");
        if (!letters) return fail("null argument");
        if (!(c->flags & KA_FLAG_DEVICE_GAPS) || !c->have_colof) return fail("the job was uploaded without KA_FLAG_DEVICE_GAPS");
        if (c->partial) return fail("aligned rows need a complete run (ka_tree_run), not a partial one");
        HIPCHK(hipSetDevice(c->device));
        if (!c->synced && ka_tree_sync(c)) return KA_FAIL;
        // the records hold the alignment length of every tree
        c->h_recs.resize(c->n_tasks);
        HIPCHK(hipMemcpy(c->h_recs.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
        tree_alnlens(c, alen);
        *widest = 0;
        for (int i = 0; i < c->numseq; i++) *widest = std::max(*widest, alen[i]);
        return KA_OK;
}

// the rows in HBM (c->d_rows, row_stride apart); they stay there for ka_aln_guide_tree
static int rows_build(ka_ctx* c, const uint8_t* letters, uint8_t gap_char, const std::vector<int>& alen, int widest, long long row_stride)
{
        const size_t bytes = (size_t)c->numseq * (size_t)row_stride;
        if (c->d_letters.alloc(c->h_codes.size()) || c->d_alnlen.alloc(c->numseq) || c->d_rows.alloc(bytes)) return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->d_letters.p, letters, c->h_codes.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_alnlen.p, alen.data(), sizeof(int) * c->numseq, hipMemcpyHostToDevice, c->stream));
        ka_launch_rows(c->d_letters.p, c->d_seq_off.p, c->d_node_len.p, c->d_colof.p, c->d_alnlen.p, c->numseq, gap_char,
                       c->d_rows.p, row_stride, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));                      // `alen` and `letters` are the caller's
        // (one alignment only: a forest has no common row length)
        c->rows_n = (c->n_tasks == c->numseq - 1) ? c->numseq : 0;
        c->rows_stride = row_stride; c->rows_alnlen = widest; c->rows_gap = gap_char;
        return KA_OK;
}

extern "C" int ka_tree_aligned_rows(ka_ctx* c, const uint8_t* letters, uint8_t gap_char, uint8_t* rows_out,
                                    long long row_stride, int* alnlen_out)
{
        if (!rows_out && !alnlen_out) return fail("null argument");
        std::vector<int> alen;
        int widest = 0;
        if (rows_prepare(c, letters, alen, &widest)) return KA_FAIL;
        if (!rows_out) { memcpy(alnlen_out, alen.data(), sizeof(int) * c->numseq); return KA_OK; }       // size query
        if (row_stride < (long long)widest + 1) return fail("row_stride is smaller than the longest alignment + terminator");
        if (rows_build(c, letters, gap_char, alen, widest, row_stride)) return KA_FAIL;
        if (copy_to_host(c, rows_out, c->d_rows.p, (size_t)c->numseq * (size_t)row_stride)) return KA_FAIL;
        if (alnlen_out) memcpy(alnlen_out, alen.data(), sizeof(int) * c->numseq);
        return KA_OK;
}

// ---- realignment (kalign_run_realign, aln_wrap.c:449-495): compute_aln_pairwise_dist + build_tree_from_pairwise ----
extern "C" int ka_aln_guide_tree(ka_ctx* c, int numseq, const uint8_t* rows, long long row_stride, int alnlen, uint8_t gap_char,
                                 int* tasks_abc, float* seq_distances, float* dm_out)
{
        if (!c) return fail("null ctx");
        if (!tasks_abc) return fail("null argument");
        HIPCHK(hipSetDevice(c->device));
        const uint8_t* d_rows = nullptr;
        if (rows) {
                if (numseq < 2 || alnlen < 1 || row_stride < alnlen) return fail("ka_aln_guide_tree: bad arguments");
                const size_t bytes = (size_t)numseq * (size_t)row_stride;
                if (c->d_rows.alloc(bytes)) return fail("hipMalloc failed");
                HIPCHK(hipMemcpyAsync(c->d_rows.p, rows, bytes, hipMemcpyHostToDevice, c->stream));
                c->rows_n = 0;                                        // no longer the rows of the last run
        } else {
                if (c->rows_n < 2) return fail("ka_aln_guide_tree: no rows on the device (call ka_tree_aligned_rows on a finished single-tree run first)");
                if (numseq != c->rows_n) return fail("ka_aln_guide_tree: numseq does not match the rows on the device");
                row_stride = c->rows_stride; alnlen = c->rows_alnlen; gap_char = c->rows_gap;
        }
        d_rows = c->d_rows.p;
        if (numseq > 46340) return fail("ka_aln_guide_tree: more than 46340 sequences (pair indices are 32-bit)");
        const size_t nn = (size_t)numseq * (size_t)numseq;
        if (c->d_adm.alloc(nn) || c->d_amean.alloc(numseq) || c->d_uactive.alloc(numseq) || c->d_ucand.alloc(2 * (size_t)numseq) ||
            c->d_umerges.alloc(numseq))
                return fail("hipMalloc failed");
        ka_launch_aln_dist(d_rows, row_stride, alnlen, numseq, gap_char, c->d_adm.p, c->d_amean.p, c->stream);
        HIPCHK(hipGetLastError());
        if (dm_out) HIPCHK(hipMemcpyAsync(dm_out, c->d_adm.p, sizeof(float) * nn, hipMemcpyDeviceToHost, c->stream));
        if (seq_distances) HIPCHK(hipMemcpyAsync(seq_distances, c->d_amean.p, sizeof(float) * numseq, hipMemcpyDeviceToHost, c->stream));
        std::vector<int> ones(numseq, 1);
        HIPCHK(hipMemcpyAsync(c->d_uactive.p, ones.data(), sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        ka_launch_upgma(c->d_adm.p, c->d_uactive.p, c->d_ucand.p, c->d_umerges.p, numseq, c->env.upgma_launches ? 1 : 0, c->stream);
        HIPCHK(hipGetLastError());
        std::vector<int> merges(2 * (size_t)numseq);
        HIPCHK(hipMemcpyAsync(merges.data(), c->d_umerges.p, sizeof(int2) * (numseq - 1), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return ka_tasks_from_merges(numseq, merges.data(), tasks_abc);
}

// ---- kalign_run_seeded / kalign_run_realign between "sequences encoded" and "rows finalised" (aln_wrap.c:144-251,361-527)
//      as one call: the composition of the entry points above, with the intermediate rows of realignment passes
//      never leaving HBM ----
// refine_mode: 0 none; 1 / 2 (| KA_REFINE_ADAPTIVE) refine_alignment after the last alignment (aln_wrap.c:229-232, :506-509);
// 3 KALIGN_REFINE_INLINE: every alignment is create_msa_tree_inline_refine instead of create_msa_tree (:222-226, :498-502)
static int run_encoded(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                       const int* off, const int* lens, const float* subm, const float* scal,
                       int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads, int refine_mode,
                       uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out);

extern "C" int ka_run_encoded(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                              const int* off, const int* lens, const float* subm, const float* scal,
                              int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads,
                              uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out)
{
        return run_encoded(c, numseq, tree_codes, codes, letters, off, lens, subm, scal, n_anchors, weight, realign_iterations, dm_scale,
                           n_threads, 0, gap_char, rows_out, row_stride, alnlen_out);
}

extern "C" int ka_run_encoded_refine(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                                     const int* off, const int* lens, const float* subm, const float* scal,
                                     int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads,
                                     int refine_mode, uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out)
{
        const int base = refine_mode & 255;
        if (refine_mode < 0 || base > 3 || (refine_mode & ~(255 | KA_REFINE_ADAPTIVE)) || ((refine_mode & KA_REFINE_ADAPTIVE) && base != 1 && base != 2))
                return fail("ka_run_encoded_refine: refine_mode must be 0, 1, 2 (optionally | KA_REFINE_ADAPTIVE) or 3");
        return run_encoded(c, numseq, tree_codes, codes, letters, off, lens, subm, scal, n_anchors, weight, realign_iterations, dm_scale,
                           n_threads, refine_mode, gap_char, rows_out, row_stride, alnlen_out);
}

static int run_encoded(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                       const int* off, const int* lens, const float* subm, const float* scal,
                       int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads, int refine_mode,
                       uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out)
{
        const bool inline_refine = (refine_mode & 255) == 3;
        auto align = [&]() -> int {
                if (inline_refine ? ka_tree_refine(c, 3, nullptr) : ka_tree_run(c)) return KA_FAIL;
                return ka_tree_sync(c);
        };
        if (!c) return fail("null ctx");
        if (numseq < 2 || !tree_codes || !codes || !letters || !off || !lens || !subm || !scal || (!rows_out && !alnlen_out))
                return fail("ka_run_encoded: bad arguments");
        std::vector<int> tasks(3 * (size_t)(numseq - 1));
        std::vector<float> sd(numseq);
        if (ka_guide_tree(c, numseq, tree_codes, off, lens, n_threads, dm_scale, tasks.data(), sd.data())) return KA_FAIL;
        if (ka_tree_upload(c, numseq, codes, off, lens, sd.data(), numseq - 1, tasks.data(), subm, scal, KA_FLAG_DEVICE_GAPS)) return KA_FAIL;
        if (n_anchors > 0 && ka_tree_build_consistency(c, n_anchors, weight)) return KA_FAIL;
        if (align()) return KA_FAIL;
        std::vector<int> alen;
        int widest = 0;
        for (int it = 0; it < realign_iterations; it++) {
                if (rows_prepare(c, letters, alen, &widest) || rows_build(c, letters, gap_char, alen, widest, (long long)widest + 1)) return KA_FAIL;
                if (ka_aln_guide_tree(c, numseq, nullptr, 0, 0, 0, tasks.data(), sd.data(), nullptr)) return KA_FAIL;
                if (ka_tree_upload(c, numseq, codes, off, lens, sd.data(), numseq - 1, tasks.data(), subm, scal,
                                   KA_FLAG_DEVICE_GAPS | KA_FLAG_KEEP_CONSISTENCY)) return KA_FAIL;
                if (align()) return KA_FAIL;
        }
        if ((refine_mode & 255) == 1 || (refine_mode & 255) == 2) {
                if (ka_tree_refine(c, refine_mode, nullptr) || ka_tree_sync(c)) return KA_FAIL;
        }
        if (rows_prepare(c, letters, alen, &widest)) return KA_FAIL;
        if (alnlen_out) memcpy(alnlen_out, alen.data(), sizeof(int) * numseq);
        if (!rows_out) return KA_OK;                                  // the alignment stays on the device: ka_tree_aligned_rows fetches it
        if (row_stride < (long long)widest + 1) { g_err = "ka_run_encoded: row_stride is smaller than the alignment + terminator (alnlen_out says how long; ka_tree_aligned_rows fetches the rows)"; return KA_ERR_ROWS_STRIDE; }
        if (rows_build(c, letters, gap_char, alen, widest, row_stride)) return KA_FAIL;
        return copy_to_host(c, rows_out, c->d_rows.p, (size_t)numseq * (size_t)row_stride);
}


// ---- partial runs: the pieces single-tree multi-GPU sharding is made of ----
// Run the listed tasks only, level by level.  Their children must already be available on this context:
// leaves, tasks run earlier (ka_tree_run_tasks does not reset the device state), or injected profiles.
extern "C" int ka_tree_run_tasks(ka_ctx* c, const int* task_ids, int n)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        c->synced = false;
        std::vector<std::vector<int>> by_level(c->levels.size());
        std::vector<char> have(2 * c->numseq - 1, 0);
        for (int i = 0; i < c->numseq; i++) have[i] = 1;
        for (int t = 0; t < c->n_tasks; t++) if (c->task_done[t]) have[c->descs[t].c] = 1;
        for (int node : c->injected) have[node] = 1;
        for (int i = 0; i < n; i++) {
                const int t = task_ids[i];
                if (t < 0 || t >= c->n_tasks) return fail("task id out of range");
                if (c->task_done[t]) return fail("task already run");
                by_level[c->task_level[t]].push_back(t);
        }
        for (auto& L : by_level)                                   // dependency check in level order
                for (int t : L) {
                        if (!have[c->descs[t].a] || !have[c->descs[t].b]) return fail("a task's operand is neither computed nor injected on this context");
                        have[c->descs[t].c] = 1;
                }
        const KaTreeDev D = tree_dev(c);
        c->partial = true;
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        c->n_launches = 0;
        for (auto& L : by_level) {
                if (L.empty()) continue;
                std::vector<int2> tbl;
                int lean = 0;
                build_blocks(c, L, tbl, &lean);
                if (c->d_blocks_tmp.alloc(tbl.size())) return fail("hipMalloc failed");
                // the table is consumed by the launch below; stream order makes the reuse of the buffer safe
                HIPCHK(hipMemsetAsync(c->d_counters.p + 1, 0, sizeof(unsigned long long), c->stream));
                HIPCHK(hipMemcpyAsync(c->d_blocks_tmp.p, tbl.data(), sizeof(int2) * tbl.size(), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                ka_launch_task_level(&D, c->d_blocks_tmp.p, (int)tbl.size(), lean, 0, c->stream);
                HIPCHK(hipStreamSynchronize(c->stream));
                c->n_launches++;
                for (int t : L) c->task_done[t] = 1;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        c->ran = true;
        return KA_OK;
}

// The listed tasks as ONE planned run: queued and chained launches where they apply, as for a whole tree -- what a rank
// of a sharded tree does with its subtrees.  The set must be closed under descendants among the tasks not run yet (a
// task's internal children are in the set, already run, or injected).  task_ids == NULL: plan the whole tree again.
extern "C" int ka_tree_plan_tasks(ka_ctx* c, const int* task_ids, int n)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        if (c->ran && !c->synced && ka_tree_sync(c)) return KA_FAIL;
        if (!task_ids) c->plan_active.clear();
        else {
                c->plan_active.assign(c->n_tasks, 0);
                for (int i = 0; i < n; i++) {
                        if (task_ids[i] < 0 || task_ids[i] >= c->n_tasks) return fail("task id out of range");
                        c->plan_active[task_ids[i]] = 1;
                }
        }
        for (auto& d : c->descs) d.refine = 0;
        c->refine_mode = 0;
        return (plan_launches(c) || upload_plan(c)) ? KA_FAIL : KA_OK;
}

// Run the planned tasks on top of what the context holds (leaves, tasks run earlier, injected profiles); no reset.
extern "C" int ka_tree_run_planned(ka_ctx* c)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (c->plan_active.empty()) return fail("ka_tree_run_planned: no task subset planned (ka_tree_plan_tasks)");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        std::vector<char> have(2 * c->numseq - 1, 0);
        for (int i = 0; i < c->numseq; i++) have[i] = 1;
        for (int t = 0; t < c->n_tasks; t++) if (c->task_done[t]) have[c->descs[t].c] = 1;
        for (int node : c->injected) have[node] = 1;
        for (auto& L : c->plan_levels)
                for (int t : L) {
                        if (c->task_done[t]) return fail("task already run");
                        if (!have[c->descs[t].a] || !have[c->descs[t].b]) return fail("a task's operand is neither computed, planned nor injected on this context");
                        have[c->descs[t].c] = 1;
                }
        c->ran = false; c->synced = false;
        if (tree_launch(c, false)) return KA_FAIL;
        c->ran = true;
        return KA_OK;
}

// Forget every computed / injected node: the next ka_tree_run_tasks starts from the leaves again.
extern "C" int ka_tree_reset(ka_ctx* c)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        c->ran = false; c->synced = false;
        return tree_reset(c);
}

extern "C" int ka_tree_node_len(ka_ctx* c, int node)
{
        if (!c || !c->have_job || !c->state_valid) return -1;
        if (node < 0 || node >= 2 * c->numseq - 1) return -1;
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
        int len = 0;
        if (hipMemcpy(&len, c->d_node_len.p + node, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return len;
}

// Make the merged profile of an internal node available on this context without running its task
// (it was computed on another GPU): (plen+2)*64 floats as ka_tree_get_profile returns them.
extern "C" int ka_tree_set_profile(ka_ctx* c, int node, const float* prof, int plen)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (node < c->numseq || node >= 2 * c->numseq - 1 || plen < 1 || !prof) return fail("bad node / profile");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        unsigned long long top = 0;
        HIPCHK(hipMemcpy(&top, c->d_counters.p, sizeof(top), hipMemcpyDeviceToHost));
        const unsigned long long need = (unsigned long long)(plen + 2) * KA_REC;
        if ((long long)(top + need) > c->prof_cap) return fail("profile arena too small for the injected profile");
        const long long po = (long long)top;
        top += need;
        HIPCHK(hipMemcpy(c->d_prof_arena.p + po, prof, sizeof(float) * (size_t)need, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_counters.p, &top, sizeof(top), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_len.p + node, &plen, sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_prof.p + node, &po, sizeof(long long), hipMemcpyHostToDevice));
        c->injected.push_back(node);
        return KA_OK;
}

// Device-to-device hand-over of a subtree root's profile between the GPUs of a sharded tree: the source exposes
// where the profile lies in its arena, the destination reserves arena space for it; the caller moves the bytes HBM to
// HBM (RCCL send / recv over xGMI, hipMemcpyPeer) -- no host bounce.
extern "C" int ka_tree_profile_dev(ka_ctx* c, int node, void** dev_ptr, int* plen_out)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (node < 0 || node >= 2 * c->numseq - 1 || !dev_ptr || !plen_out) return fail("bad node");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        int len = 0;
        long long po = -1;
        HIPCHK(hipMemcpy(&len, c->d_node_len.p + node, sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&po, c->d_node_prof.p + node, sizeof(long long), hipMemcpyDeviceToHost));
        if (po < 0) return fail("node has no profile (root, or not computed)");
        *dev_ptr = c->d_prof_arena.p + po;
        *plen_out = len;
        return KA_OK;
}

extern "C" int ka_tree_reserve_profile_dev(ka_ctx* c, int node, int plen, void** dev_ptr)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (node < c->numseq || node >= 2 * c->numseq - 1 || plen < 1 || !dev_ptr) return fail("bad node / profile");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        unsigned long long top = 0;
        HIPCHK(hipMemcpy(&top, c->d_counters.p, sizeof(top), hipMemcpyDeviceToHost));
        const unsigned long long need = (unsigned long long)(plen + 2) * KA_REC;
        if ((long long)(top + need) > c->prof_cap) return fail("profile arena too small for the incoming profile");
        const long long po = (long long)top;
        top += need;
        HIPCHK(hipMemcpy(c->d_counters.p, &top, sizeof(top), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_len.p + node, &plen, sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_prof.p + node, &po, sizeof(long long), hipMemcpyHostToDevice));
        c->injected.push_back(node);
        *dev_ptr = c->d_prof_arena.p + po;
        return KA_OK;
}

// Consistency state of a node for partial runs: the residue -> column table of its member sequences, concatenated
// in the node's member order (sum of their lengths ints).  Moves with the profile when a node changes GPUs.
static void node_members(const ka_ctx* c, int node, long long* lo, long long* hi)
{
        // sip_flat holds the leaves first (one entry each), then every internal node in task order
        *lo = c->sip_off[node];
        if (node < c->numseq) { *hi = *lo + 1; return; }
        long long best = (long long)c->sip_flat.size();
        for (size_t k = 0; k < c->sip_off.size(); k++) if (c->sip_off[k] > *lo && c->sip_off[k] < best) best = c->sip_off[k];
        *hi = best;
}

extern "C" long long ka_tree_node_cols_size(ka_ctx* c, int node)
{
        if (!c || !c->have_job || node < 0 || node >= 2 * c->numseq - 1) return -1;
        long long lo, hi, n = 0;
        node_members(c, node, &lo, &hi);
        for (long long k = lo; k < hi; k++) n += c->lens[c->sip_flat[k]];
        return n;
}

extern "C" int ka_tree_get_node_cols(ka_ctx* c, int node, int* out)
{
        if (!c || !c->have_job || c->cons_K <= 0) return fail("no consistency table on this context");
        if (node < 0 || node >= 2 * c->numseq - 1) return fail("bad node");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        long long lo, hi, o = 0;
        node_members(c, node, &lo, &hi);
        for (long long k = lo; k < hi; k++) {
                const int si = c->sip_flat[k];
                HIPCHK(hipMemcpy(out + o, c->d_colof.p + c->off[si], sizeof(int) * c->lens[si], hipMemcpyDeviceToHost));
                o += c->lens[si];
        }
        return KA_OK;
}

extern "C" int ka_tree_set_node_cols(ka_ctx* c, int node, const int* cols)
{
        if (!c || !c->have_job || c->cons_K <= 0) return fail("no consistency table on this context");
        if (node < 0 || node >= 2 * c->numseq - 1) return fail("bad node");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        long long lo, hi, o = 0;
        node_members(c, node, &lo, &hi);
        for (long long k = lo; k < hi; k++) {
                const int si = c->sip_flat[k];
                HIPCHK(hipMemcpy(c->d_colof.p + c->off[si], cols + o, sizeof(int) * c->lens[si], hipMemcpyHostToDevice));
                o += c->lens[si];
        }
        return KA_OK;
}

// Records and coded paths of the listed tasks (they must have been run on this context), paths packed
// in the order of the list; *used receives the number of ints written.
extern "C" int ka_tree_download_tasks(ka_ctx* c, const int* task_ids, int n, ka_task_rec* recs, int* paths_out, long long paths_cap, long long* used_out)
{
        if (!c || !c->ran) return fail("nothing to download");
        if (!c->synced && ka_tree_sync(c)) return KA_FAIL;
        HIPCHK(hipSetDevice(c->device));
        const long long used = (long long)c->h_counters[2];
        std::vector<ka_task_rec> all(c->n_tasks);
        HIPCHK(hipMemcpy(all.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
        std::vector<int> arena((size_t)std::max<long long>(used, 1));
        if (copy_to_host(c, arena.data(), c->d_path_arena.p, sizeof(int) * (size_t)used)) return KA_FAIL;
        long long o = 0;
        for (int i = 0; i < n; i++) {
                const int t = task_ids[i];
                if (t < 0 || t >= c->n_tasks || !c->task_done[t]) return fail("task was not run on this context");
                ka_task_rec r = all[t];
                const int cnt = r.plen + 2;
                if (o + cnt > paths_cap) { g_err = "paths_out too small"; return KA_ERR_PATHS_CAP; }
                memcpy(paths_out + o, arena.data() + r.path_off, sizeof(int) * cnt);
                r.path_off = (int)o;
                o += cnt;
                recs[i] = r;
        }
        if (used_out) *used_out = o;
        return KA_OK;
}

extern "C" int ka_tree_get_profile(ka_ctx* c, int node, float* out, long long cap_floats)
{
        if (!c || !c->synced) return fail("run + sync first");
        HIPCHK(hipSetDevice(c->device));
        const int nprof = 2 * c->numseq - 1;
        if (node < 0 || node >= nprof) return fail("bad node");
        int len = 0;
        long long po = -1;
        HIPCHK(hipMemcpy(&len, c->d_node_len.p + node, sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&po, c->d_node_prof.p + node, sizeof(long long), hipMemcpyDeviceToHost));
        if (po < 0) return fail("node has no profile (root, or not computed)");
        const long long n = (long long)(len + 2) * KA_REC;
        if (n > cap_floats) return fail("profile buffer too small");
        HIPCHK(hipMemcpy(out, c->d_prof_arena.p + po, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
        return KA_OK;
}

extern "C" int ka_tree_get_timing(ka_ctx* c, long long* out)
{
        if (!c || !c->synced) return fail("run + sync first");
        if (!(c->flags & KA_FLAG_TIMING)) return fail("KA_FLAG_TIMING was not set");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemcpy(out, c->d_timing.p, sizeof(long long) * (8 * c->n_tasks + 48 + 512), hipMemcpyDeviceToHost));
        return KA_OK;
}

extern "C" float ka_pairwise_kernel_ms(ka_ctx* c) { return c ? c->pair_ms : 0.0f; }

extern "C" int ka_debug_trace(ka_ctx* c, int* out64)
{
        if (!c || !c->h_trace) return fail("KA_TRACE was not set when the context was created");
        memcpy(out64, c->h_trace, 64 * sizeof(int));
        return KA_OK;
}

extern "C" double ka_tree_cells(ka_ctx* c) { return c ? c->cells : 0.0; }

extern "C" int ka_tree_kernel_ms(ka_ctx* c, float* ms, int* n_launches)
{
        if (!c || !c->synced) return fail("run + sync first");
        HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
        if (n_launches) *n_launches = c->n_launches;
        return KA_OK;
}

// Measurements: the duration of every launch of the last run (context created with KA_LAUNCH_EV=1 in the environment);
// returns the number of launches written, -1 on error.
extern "C" int ka_tree_launch_ms(ka_ctx* c, float* ms, int cap)
{
        if (!c || !c->synced) { fail("run + sync first"); return -1; }
        if (!c->env.launch_ev || (int)c->launch_ev.size() < c->n_launches) { fail("ka_tree_launch_ms: no launch events (KA_LAUNCH_EV=1)"); return -1; }
        const int n = std::min(cap, c->n_launches);
        for (int i = 0; i < n; i++)
                if (hipEventElapsedTime(ms + i, i ? c->launch_ev[i - 1] : c->ev0, c->launch_ev[i]) != hipSuccess) { fail("hipEventElapsedTime"); return -1; }
        return n;
}


// residue -> column tables and member lists on the device (consistency votes, device-side gap arrays)
static int setup_colof(ka_ctx* c)
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
        if (n_anchors > KA_CONS_MAX_ANCHORS) return fail("this build takes at most 32 consistency anchors (KA_CONS_MAX_ANCHORS)");
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
static int pairwise_on_device(ka_ctx* c, const uint8_t* codes, const int* off, const int* lens, int numseq,
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

// =================================================================================================================
// One alignment over the GPUs of a node: one process per GPU, RCCL over xGMI, driven from C.
//
//   * anchor_consistency_build: every rank aligns its share of the N x K seq-seq batch and fills their position maps
//     in its copy of the table; every rank's range is then broadcast IN PLACE, HBM to HBM (ncclBroadcast, all ranges in
//     one group);
//   * the guide tree is cut ONCE per job into one subtree per rank (balanced by estimated DP cells); a rank's subtrees
//     run as ONE planned run (queued / chained launches, like a whole tree: ka_tree_plan_tasks); above the cut the
//     profile of the smaller child moves device to device (ncclSend / ncclRecv: a two-int header, the records from
//     where they lie in the source's arena into room reserved in the destination's, and -- default mode -- the
//     residue -> column table of the moved subtree, packed and unpacked on the device) to the rank that holds the other
//     child, which runs the parent on up to 16 CUs;
//   * records and coded paths: every rank scatters its own into the job-wide layout on the device and ONE all-reduce
//     each (disjoint ranges, zeros elsewhere: the sum of integers words is exact) leaves every rank with everything.
// Results do not depend on the number of ranks: tasks are position-addressed and a task's DP is the same code wherever
// it runs -- the reference's thread-count invariance (lib/src/aln_run.c:95-109, independent subtrees).
// RCCL is loaded at run time (dlopen): the single-GPU library has no link-time dependency on it.
// =================================================================================================================
#include <dlfcn.h>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#if __has_include(<rccl/rccl.h>) && !defined(KA_NO_RCCL_HEADERS)
#include <rccl/rccl.h>
#else
// (hosts without the RCCL development headers: the few names of NCCL's public, stable ABI this file uses -- the library
// itself is only ever looked for at run time, rccl_load)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclInt32 = 2 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
}
#endif

extern "C" void ka_launch_cols_pack(int* colof, const int* seq_off, const int* seq_len, const int* members, const long long* moff, int nmem,
                                    int* buf, int unpack, hipStream_t stream);
extern "C" void ka_launch_path_counts(const ka_task_rec* recs, const char* mine, int n_tasks, int* counts, hipStream_t stream);
extern "C" void ka_launch_path_scatter(const ka_task_rec* recs, const char* mine, int n_tasks, const int* arena, const long long* goff, int* out, hipStream_t stream);

namespace {
struct Rccl {
        void* lib = nullptr;
        ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
        ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
        ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
        ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*GroupStart)() = nullptr;
        ncclResult_t (*GroupEnd)() = nullptr;
        const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load()
{
        if (g_rccl.lib) return KA_OK;
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        void* h = nullptr;
        for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) return fail(std::string("ka_dist: cannot load RCCL (librccl.so.1): ") + dlerror());
#define KA_SYM(field_, name_) *(void**)(&g_rccl.field_) = dlsym(h, name_); if (!g_rccl.field_) return fail(std::string("ka_dist: RCCL lacks ") + name_)
        KA_SYM(GetUniqueId, "ncclGetUniqueId"); KA_SYM(CommInitRank, "ncclCommInitRank"); KA_SYM(CommDestroy, "ncclCommDestroy");
        KA_SYM(Broadcast, "ncclBroadcast"); KA_SYM(AllReduce, "ncclAllReduce"); KA_SYM(Send, "ncclSend"); KA_SYM(Recv, "ncclRecv");
        KA_SYM(GroupStart, "ncclGroupStart"); KA_SYM(GroupEnd, "ncclGroupEnd"); KA_SYM(GetErrorString, "ncclGetErrorString");
#undef KA_SYM
        g_rccl.lib = h;
        return KA_OK;
}
#define NCCLCHK(x)                                                                                         \
        do {                                                                                               \
                ncclResult_t r_ = (x);                                                                     \
                if (r_ != ncclSuccess) return fail(std::string(#x) + ": " + g_rccl.GetErrorString(r_));     \
        } while (0)

// An in-process stand-in for the communicator (tests): the ranks are threads of ONE process, each with its own context on
// the SAME GPU -- RCCL refuses two ranks on one device, and the pool's GPU boxes have one.  Host-synchronous, FIFO per
// (source, destination) pair; collectives meet at a barrier and reduce through the host.  Same call sequence as RCCL.
struct KaLoopback {
        int world = 1;
        std::mutex m;
        std::condition_variable cv;
        struct Msg { const void* ptr; size_t bytes; bool taken; };
        std::map<std::pair<int, int>, std::deque<Msg*>> box;
        std::vector<void*> bufs;
        int arrived = 0;
        long long gen = 0;
        void barrier(std::unique_lock<std::mutex>& lk)
        {
                const long long g = gen;
                if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
                else cv.wait(lk, [&] { return gen != g; });
        }
};
}  // namespace

// A child profile that changes GPUs above the cut
struct KaMove {
        int task, child, src, dst;
        int nmem = 0;                  // default mode: the child's member sequences ...
        long long ncols = 0;           // ... and the ints of their residue -> column tables
        DevBuf<int> d_members; DevBuf<long long> d_moff;
};
struct ka_dist {
        ka_ctx* c = nullptr;
        int rank = 0, world = 1;
        ncclComm_t comm = nullptr;
        KaLoopback* loop = nullptr;                      // tests: threads of one process instead of RCCL
        bool planned = false;
        // ---- the five transport operations (RCCL on the context's stream, or the loopback) ----
        int all_reduce_i32(int* buf, size_t count, bool take_max);
        int broadcast_i32(int* buf, size_t count, int root);
        int send(const void* buf, size_t bytes, int peer);
        int recv(void* buf, size_t bytes, int peer);
        int group_start() { if (loop || world == 1) return KA_OK; ncclResult_t r = g_rccl.GroupStart(); return r == ncclSuccess ? KA_OK : fail(std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r)); }
        int group_end() { if (loop || world == 1) return KA_OK; ncclResult_t r = g_rccl.GroupEnd(); return r == ncclSuccess ? KA_OK : fail(std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r)); }
        std::vector<int> run_rank, top, mine_sub;        // rank of every task; the tasks above the cut (tree order); this rank's subtree tasks
        std::vector<KaMove> moves;                       // in the order the top tasks need them
        std::vector<std::vector<int>> top_moves;         // per top task: indices into moves
        std::vector<DevBuf<int2>> top_blocks;            // per top task this rank runs: its workgroup table
        DevBuf<char> d_mine; DevBuf<int> d_counts, d_gpaths, d_colbuf, d_hdr, d_status; DevBuf<long long> d_goff;
        DevBuf<float> d_sink;                            // an incoming profile the arena has no room for (the step is then repeated)
        int retries = 0;                                 // steps repeated after an arena overflow on some rank
        std::vector<char> mine;
        std::vector<ka_task_rec> h_recs;
        std::vector<int> h_paths;
        std::vector<long long> goff;
        int* h_head = nullptr;                           // pinned: the two-int header of an incoming profile
        double last_ms = 0.0, last_kernel_wait_ms = 0.0;
};

int ka_dist::all_reduce_i32(int* buf, size_t count, bool take_max)
{
        if (world == 1 && !comm) return KA_OK;
        if (!loop) { NCCLCHK(g_rccl.AllReduce(buf, buf, count, ncclInt32, take_max ? ncclMax : ncclSum, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        std::unique_lock<std::mutex> lk(loop->m);
        loop->bufs[rank] = buf;
        loop->barrier(lk);
        std::vector<int> acc(count, take_max ? INT_MIN : 0), tmp(count);
        for (int r = 0; r < world; r++) {
                HIPCHK(hipMemcpy(tmp.data(), loop->bufs[r], sizeof(int) * count, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < count; i++) acc[i] = take_max ? std::max(acc[i], tmp[i]) : acc[i] + tmp[i];
        }
        loop->barrier(lk);                                          // everybody has read every buffer
        HIPCHK(hipMemcpy(buf, acc.data(), sizeof(int) * count, hipMemcpyHostToDevice));
        return KA_OK;
}
int ka_dist::broadcast_i32(int* buf, size_t count, int root)
{
        if (world == 1 && !comm) return KA_OK;
        if (!loop) { NCCLCHK(g_rccl.Broadcast(buf, buf, count, ncclInt32, root, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        std::unique_lock<std::mutex> lk(loop->m);
        loop->bufs[rank] = buf;
        loop->barrier(lk);
        if (rank != root) HIPCHK(hipMemcpy(buf, loop->bufs[root], sizeof(int) * count, hipMemcpyDeviceToDevice));
        loop->barrier(lk);
        return KA_OK;
}
int ka_dist::send(const void* buf, size_t bytes, int peer)
{
        if (!loop) { NCCLCHK(g_rccl.Send(buf, bytes, ncclInt8, peer, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        KaLoopback::Msg msg = { buf, bytes, false };
        std::unique_lock<std::mutex> lk(loop->m);
        loop->box[std::make_pair(rank, peer)].push_back(&msg);
        loop->cv.notify_all();
        loop->cv.wait(lk, [&] { return msg.taken; });
        return KA_OK;
}
int ka_dist::recv(void* buf, size_t bytes, int peer)
{
        if (!loop) { NCCLCHK(g_rccl.Recv(buf, bytes, ncclInt8, peer, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        std::unique_lock<std::mutex> lk(loop->m);
        auto& q = loop->box[std::make_pair(peer, rank)];
        loop->cv.wait(lk, [&] { return !q.empty(); });
        KaLoopback::Msg* msg = q.front();
        q.pop_front();
        if (msg->bytes != bytes) { msg->taken = true; loop->cv.notify_all(); return fail("ka_dist loopback: message size mismatch"); }
        const hipError_t e = hipMemcpy(buf, msg->ptr, bytes, hipMemcpyDeviceToDevice);
        msg->taken = true;
        loop->cv.notify_all();
        if (e != hipSuccess) return fail(std::string("ka_dist loopback: ") + hipGetErrorString(e));
        return KA_OK;
}

// Tests: a loopback "communicator" for `world` ranks living in one process (threads), each with its own context.
extern "C" void* ka_dist_loopback_new(int world)
{
        if (world < 1) return nullptr;
        KaLoopback* l = new KaLoopback();
        l->world = world;
        l->bufs.assign(world, nullptr);
        return l;
}
extern "C" void ka_dist_loopback_free(void* l) { delete (KaLoopback*)l; }
extern "C" int ka_dist_create_loopback(ka_ctx* c, int rank, int world, void* loopback, ka_dist** out)
{
        if (!c || !out || !loopback || world < 1 || rank < 0 || rank >= world || ((KaLoopback*)loopback)->world != world) return fail("ka_dist_create_loopback: bad arguments");
        HIPCHK(hipSetDevice(c->device));
        ka_dist* d = new ka_dist();
        d->c = c; d->rank = rank; d->world = world; d->loop = (KaLoopback*)loopback;
        if (hipHostMalloc((void**)&d->h_head, 64, hipHostMallocDefault) != hipSuccess) { delete d; return fail("hipHostMalloc failed"); }
        *out = d;
        return KA_OK;
}

// Pure planning (no device, no communicator): cut the tree into at most `world` subtrees balanced by estimated DP cells;
// run_rank[t] = the rank that runs task t, top[0 .. *n_top) = the tasks above the cut in tree order.  Every rank derives
// the same plan from the same inputs.
extern "C" int ka_dist_plan_subtrees(int numseq, const int* lens, int n_tasks, const int* abc, int world, int* run_rank, int* top, int* n_top)
{
        if (numseq < 2 || n_tasks < 1 || world < 1 || !lens || !abc || !run_rank || !top || !n_top) return fail("ka_dist_plan_subtrees: bad arguments");
        const int nprof = 2 * numseq - 1;
        std::vector<int> task_of(nprof, -1), members(nprof, 0);
        std::vector<double> est(nprof, 0.0), work(nprof, 0.0);
        for (int i = 0; i < numseq; i++) { est[i] = lens[i]; members[i] = 1; }
        for (int t = 0; t < n_tasks; t++) {
                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                if (a < 0 || b < 0 || cc < numseq || a >= nprof || b >= nprof || cc >= nprof) return fail("ka_dist_plan_subtrees: bad task list");
                task_of[cc] = t;
                est[cc] = 1.05 * std::max(est[a], est[b]);
                work[cc] = work[a] + work[b] + est[a] * est[b];
                members[cc] = members[a] + members[b];
        }
        const int root = abc[3 * (n_tasks - 1) + 2];
        std::vector<int> frontier(1, root), tops;
        auto internal = [&](const std::vector<int>& f) { int n = 0; for (int x : f) n += x >= numseq; return n; };
        while (internal(frontier) < world) {
                int best = -1;
                for (int x : frontier) if (x >= numseq && (best < 0 || work[x] > work[best] || (work[x] == work[best] && x < best))) best = x;
                if (best < 0) break;
                const int t = task_of[best];
                const int kids = (abc[3 * t] >= numseq) + (abc[3 * t + 1] >= numseq);
                if (kids == 0) break;                            // splitting would not add a subtree (both children are leaves)
                frontier.erase(std::find(frontier.begin(), frontier.end(), best));
                frontier.push_back(abc[3 * t]); frontier.push_back(abc[3 * t + 1]);
                tops.push_back(t);
        }
        std::sort(tops.begin(), tops.end());
        std::vector<int> roots;
        for (int x : frontier) if (x >= numseq) roots.push_back(x);
        std::sort(roots.begin(), roots.end(), [&](int x, int y) { return work[x] > work[y] || (work[x] == work[y] && x < y); });
        for (int t = 0; t < n_tasks; t++) run_rank[t] = -1;
        std::vector<int> holder(nprof, -1);
        for (size_t r = 0; r < roots.size(); r++) {
                std::vector<int> stack(1, roots[r]);
                while (!stack.empty()) {
                        const int v = stack.back(); stack.pop_back();
                        if (v < numseq) continue;
                        const int t = task_of[v];
                        run_rank[t] = (int)(r % world);
                        stack.push_back(abc[3 * t]); stack.push_back(abc[3 * t + 1]);
                }
                holder[roots[r]] = (int)(r % world);
        }
        for (int t : tops) {                                     // tree order: children first
                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                const int ha = holder[a], hb = holder[b];
                int r;
                if (ha < 0 && hb < 0) r = 0;
                else if (ha < 0 || (hb >= 0 && members[b] > members[a])) r = hb;
                else r = ha;
                run_rank[t] = r;
                holder[cc] = r;
        }
        for (size_t i = 0; i < tops.size(); i++) top[i] = tops[i];
        *n_top = (int)tops.size();
        return KA_OK;
}

extern "C" int ka_dist_unique_id(void* id128)
{
        if (!id128) return fail("null id");
        if (rccl_load()) return KA_FAIL;
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        ncclUniqueId id;
        NCCLCHK(g_rccl.GetUniqueId(&id));
        memcpy(id128, &id, sizeof(id));
        return KA_OK;
}

extern "C" void ka_dist_destroy(ka_dist* d)
{
        if (!d) return;
        if (d->c) (void)hipSetDevice(d->c->device);
        if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
        for (auto& m : d->moves) { m.d_members.release(); m.d_moff.release(); }
        for (auto& b : d->top_blocks) b.release();
        d->d_mine.release(); d->d_counts.release(); d->d_gpaths.release(); d->d_colbuf.release(); d->d_goff.release();
        if (d->h_head) (void)hipHostFree(d->h_head);
        delete d;
}

// id128: the 128 bytes rank 0 got from ka_dist_unique_id, handed to every rank by the launcher (a file, MPI, torch...).
// world == 1: no communicator is made (every step degenerates to the local run) -- the code path is the same.
extern "C" int ka_dist_create(ka_ctx* c, int rank, int world, const void* id128, ka_dist** out)
{
        if (!c || !out || world < 1 || rank < 0 || rank >= world) return fail("ka_dist_create: bad arguments");
        HIPCHK(hipSetDevice(c->device));
        ka_dist* d = new ka_dist();
        d->c = c; d->rank = rank; d->world = world;
        if (hipHostMalloc((void**)&d->h_head, 64, hipHostMallocDefault) != hipSuccess) { delete d; return fail("hipHostMalloc failed"); }
        if (world > 1 || id128) {
                if (!id128 || rccl_load()) { ka_dist_destroy(d); return id128 ? KA_FAIL : fail("ka_dist_create: a world of several ranks needs the unique id"); }
                ncclUniqueId id;
                memcpy(&id, id128, sizeof(id));
                ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank);
                if (r != ncclSuccess) { const std::string m = g_rccl.GetErrorString(r); ka_dist_destroy(d); return fail("ncclCommInitRank: " + m); }
        }
        *out = d;
        return KA_OK;
}

// Once per uploaded job: the cut, who runs what, the hand-overs above the cut, this rank's subtrees planned as one run.
extern "C" int ka_dist_plan(ka_dist* d)
{
        if (!d || !d->c || !d->c->have_job) return fail("ka_dist_plan: no uploaded job");
        ka_ctx* c = d->c;
        HIPCHK(hipSetDevice(c->device));
        if (c->n_tasks != c->numseq - 1) return fail("ka_dist_plan: one guide tree per job");
        const int n_tasks = c->n_tasks, numseq = c->numseq;
        d->run_rank.assign(n_tasks, -1);
        d->top.assign(n_tasks, 0);
        int n_top = 0;
        if (ka_dist_plan_subtrees(numseq, c->lens.data(), n_tasks, c->abc.data(), d->world, d->run_rank.data(), d->top.data(), &n_top)) return KA_FAIL;
        d->top.resize(n_top);
        std::vector<char> is_top(n_tasks, 0);
        for (int t : d->top) is_top[t] = 1;
        d->mine_sub.clear();
        d->mine.assign(n_tasks, 0);
        for (int t = 0; t < n_tasks; t++) {
                if (d->run_rank[t] == d->rank) d->mine[t] = 1;
                if (d->run_rank[t] == d->rank && !is_top[t]) d->mine_sub.push_back(t);
        }
        // the hand-overs: a child of a top task that sits on another rank than the one running the parent
        for (auto& m : d->moves) { m.d_members.release(); m.d_moff.release(); }
        for (auto& b : d->top_blocks) b.release();
        d->moves.clear(); d->top_moves.assign(n_top, std::vector<int>()); d->top_blocks.clear(); d->top_blocks.resize(n_top);
        std::vector<int> holder(2 * numseq - 1, -1);
        for (int t = 0; t < n_tasks; t++) if (!is_top[t]) holder[c->abc[3 * t + 2]] = d->run_rank[t];
        long long max_cols = 0;
        for (int i = 0; i < n_top; i++) {
                const int t = d->top[i], dst = d->run_rank[t];
                for (int k = 0; k < 2; k++) {
                        const int child = c->abc[3 * t + k];
                        const int src = child >= numseq ? holder[child] : -1;
                        if (child < numseq || src < 0 || src == dst) continue;
                        d->moves.emplace_back();
                        KaMove& m = d->moves.back();
                        m.task = t; m.child = child; m.src = src; m.dst = dst;
                        if (d->rank == src || d->rank == dst) {
                                long long lo, hi;
                                node_members(c, child, &lo, &hi);
                                std::vector<int> mem(c->sip_flat.begin() + lo, c->sip_flat.begin() + hi);
                                std::vector<long long> off(mem.size());
                                long long o = 0;
                                for (size_t q = 0; q < mem.size(); q++) { off[q] = o; o += c->lens[mem[q]]; }
                                m.nmem = (int)mem.size(); m.ncols = o;
                                max_cols = std::max(max_cols, o);
                                if (m.d_members.alloc(mem.size()) || m.d_moff.alloc(off.size())) return fail("hipMalloc failed");
                                HIPCHK(hipMemcpy(m.d_members.p, mem.data(), sizeof(int) * mem.size(), hipMemcpyHostToDevice));
                                HIPCHK(hipMemcpy(m.d_moff.p, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice));
                        }
                        d->top_moves[i].push_back((int)d->moves.size() - 1);
                }
                holder[c->abc[3 * t + 2]] = dst;
                if (dst == d->rank) {
                        std::vector<int2> tbl;
                        int lean = 0;
                        build_blocks(c, std::vector<int>(1, t), tbl, &lean);
                        if (d->top_blocks[i].alloc(tbl.size())) return fail("hipMalloc failed");
                        HIPCHK(hipMemcpy(d->top_blocks[i].p, tbl.data(), sizeof(int2) * tbl.size(), hipMemcpyHostToDevice));
                }
        }
        if (d->d_mine.alloc(n_tasks) || d->d_counts.alloc(n_tasks) || d->d_goff.alloc(n_tasks) || d->d_colbuf.alloc((size_t)std::max<long long>(max_cols, 1)))
                return fail("hipMalloc failed");
        HIPCHK(hipMemcpy(d->d_mine.p, d->mine.data(), n_tasks, hipMemcpyHostToDevice));
        // this rank's subtrees as ONE planned run (queued / chained launches where they apply)
        c->plan_active.assign(n_tasks, 0);
        for (int t : d->mine_sub) c->plan_active[t] = 1;
        if (d->mine_sub.empty()) c->plan_active.assign(n_tasks, 0);
        // (an all-zero mask is a plan over nothing: every level empty)
        if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
        d->planned = true;
        return KA_OK;
}

extern "C" int ka_dist_get_plan(ka_dist* d, int* run_rank, int* top, int* n_top, int* n_moves)
{
        if (!d || !d->planned) return fail("ka_dist_get_plan: plan first");
        if (run_rank) memcpy(run_rank, d->run_rank.data(), sizeof(int) * d->run_rank.size());
        if (top) memcpy(top, d->top.data(), sizeof(int) * d->top.size());
        if (n_top) *n_top = (int)d->top.size();
        if (n_moves) *n_moves = (int)d->moves.size();
        return KA_OK;
}

// anchor_consistency_build over the ranks: this rank's share of the N x K batch, then every share broadcast in place.
extern "C" int ka_dist_consistency(ka_dist* d, int n_anchors, float weight)
{
        if (!d || !d->c) return fail("ka_dist_consistency: null");
        ka_ctx* c = d->c;
        HIPCHK(hipSetDevice(c->device));
        // a part that cannot be built (e.g. it holds only anchors) must not leave the other ranks waiting in a collective:
        // every rank reduces the outcome first
        int rc = ka_tree_build_consistency_part(c, n_anchors, weight, d->rank, d->world);
        const std::string why = rc ? g_err : std::string();
        if (d->world > 1 || d->comm) {
                int* flag = (int*)d->d_counts.p;
                if (!flag && d->d_counts.alloc(std::max(c->n_tasks, 1))) return fail("hipMalloc failed");
                flag = d->d_counts.p;
                const int mine = rc ? 1 : 0;
                HIPCHK(hipMemcpyAsync(flag, &mine, sizeof(int), hipMemcpyHostToDevice, c->stream));
                if (d->all_reduce_i32(flag, 1, true)) return KA_FAIL;
                int any = 0;
                HIPCHK(hipMemcpyAsync(&any, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if (any) return fail(rc ? why : std::string("ka_dist_consistency: another rank could not build its part"));
                if (c->cons_K <= 0) return KA_OK;                    // the job declined on every rank alike (fewer than 3 sequences ...)
                if (d->group_start()) return KA_FAIL;
                for (int r = 0; r < d->world; r++) {
                        long long lo = 0, hi = 0;
                        if (ka_tree_consistency_part_range(c, r, d->world, &lo, &hi)) { (void)d->group_end(); return KA_FAIL; }
                        if (hi > lo && d->broadcast_i32(c->d_cons_maps.p + lo, (size_t)(hi - lo), r)) { (void)d->group_end(); return KA_FAIL; }
                }
                if (d->group_end()) return KA_FAIL;
        } else if (rc) return KA_FAIL;
        return KA_OK;
}

// One attempt at a step of the sharded tree, up to the point where this rank knows how ITS part went.  Conditions a repeat can
// cure -- a device arena overflowed in one of this rank's kernels, the profile arena has no room for an incoming profile, a
// profile this rank should send was never made because an earlier task failed -- do NOT leave the protocol: the rank keeps
// matching every send / receive of the walk (an unusable profile travels as a header of zero and nothing else; a profile
// without room lands in a sink buffer), stops launching, and reports through *status (0 clean, 1 repeat after growing, 2 fatal).
// Only HIP / RCCL API failures return KA_FAIL from inside (nothing sensible can be agreed on a broken device).
static int dist_tree_attempt(ka_dist* d, int* status, int* grow)
{
        ka_ctx* c = d->c;
        const int n_tasks = c->n_tasks;
        *status = 0; *grow = 0;
        if (c->plan_active.empty()) return fail("ka_dist_tree_run: the context's plan was replaced by a whole-tree run; call ka_dist_plan again");
        c->ran = false; c->synced = false;
        if (tree_reset(c)) return KA_FAIL;                          // (keeps the consistency table; residue -> column tables back to the leaves)
        if (tree_launch(c, false)) return KA_FAIL;                  // this rank's subtrees
        const KaTreeDev D = tree_dev(c);
        if (d->d_hdr.alloc(4)) return fail("hipMalloc failed");
        bool stop = false;                                           // something went wrong on this rank: no more launches
        for (size_t i = 0; i < d->top.size(); i++) {
                const int t = d->top[i], dst = d->run_rank[t];
                for (int mi : d->top_moves[i]) {
                        KaMove& m = d->moves[mi];
                        if (d->rank != m.src && d->rank != m.dst) continue;
                        const bool cols = c->have_colof && m.ncols > 0 && (c->cons_K > 0 || (c->flags & KA_FLAG_DEVICE_GAPS));
                        if (d->rank == m.src) {
                                // header (plen) straight from the node table, the records from where they lie in the arena
                                HIPCHK(hipMemcpyAsync(d->h_head, c->d_node_len.p + m.child, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipMemcpyAsync(d->h_head + 2, c->d_node_prof.p + m.child, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipMemcpyAsync(d->h_head + 8, c->d_error.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipStreamSynchronize(c->stream));
                                int plen = d->h_head[0];
                                long long po; memcpy(&po, d->h_head + 2, sizeof(po));
                                const bool missing = d->h_head[8] == 0 && !stop && (plen < 1 || po < 0);
                                if (stop || d->h_head[8] != 0 || plen < 1 || po < 0) { plen = 0; stop = true; }   // (its status comes from the device error word below)
                                // a profile that was never made although no kernel of this rank reported anything (a bug, not an arena to grow):
                                // this rank says so -- the receiver reports "repeat", and repeating cannot cure it
                                if (missing) { *status = 2; fail("sharded tree: the profile of node " + std::to_string(m.child) + " was never made on the rank that owns it"); }
                                d->h_head[12] = plen;
                                HIPCHK(hipMemcpyAsync(d->d_hdr.p, d->h_head + 12, sizeof(int), hipMemcpyHostToDevice, c->stream));
                                // (plain stream-ordered point-to-point operations, matched in order with the receiver's: every rank
                                // walks the hand-overs in the same order, so no two ranks ever wait for each other crosswise)
                                if (d->send(d->d_hdr.p, sizeof(int), m.dst)) return KA_FAIL;
                                if (plen > 0) {
                                        if (cols) ka_launch_cols_pack(c->d_colof.p, c->d_seq_off.p, c->d_node_len.p, m.d_members.p, m.d_moff.p, m.nmem, d->d_colbuf.p, 0, c->stream);
                                        if (d->send(c->d_prof_arena.p + po, sizeof(float) * (size_t)(plen + 2) * KA_REC, m.dst)) return KA_FAIL;
                                        if (cols && d->send(d->d_colbuf.p, sizeof(int) * (size_t)m.ncols, m.dst)) return KA_FAIL;
                                }
                                HIPCHK(hipStreamSynchronize(c->stream));          // (h_head and d_hdr are reused by the next hand-over)
                        } else {
                                // the header first: it sizes the room the records get in this rank's arena
                                if (d->recv(d->d_hdr.p, sizeof(int), m.src)) return KA_FAIL;
                                HIPCHK(hipMemcpyAsync(d->h_head, d->d_hdr.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipMemcpyAsync(d->h_head + 2, c->d_counters.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipStreamSynchronize(c->stream));
                                const int plen = d->h_head[0];
                                if (plen < 1) { stop = true; if (*status < 1) *status = 1; continue; }   // the sender has nothing to send: its own status says why
                                unsigned long long top_; memcpy(&top_, d->h_head + 2, sizeof(top_));
                                const unsigned long long need = (unsigned long long)(plen + 2) * KA_REC;
                                if ((long long)(top_ + need) > c->prof_cap) {
                                        // no room: take the payload off the wire all the same, then ask for a repeat with a bigger arena
                                        if (d->d_sink.alloc((size_t)need)) return fail("hipMalloc failed");
                                        if (d->recv(d->d_sink.p, sizeof(float) * (size_t)need, m.src)) return KA_FAIL;
                                        if (cols && d->recv(d->d_colbuf.p, sizeof(int) * (size_t)m.ncols, m.src)) return KA_FAIL;
                                        HIPCHK(hipStreamSynchronize(c->stream));
                                        stop = true; *status = std::max(*status, 1); *grow |= 1;
                                        continue;
                                }
                                const long long po = (long long)top_;
                                top_ += need;
                                memcpy(d->h_head + 4, &top_, sizeof(top_)); memcpy(d->h_head + 6, &po, sizeof(po));
                                HIPCHK(hipMemcpyAsync(c->d_counters.p, d->h_head + 4, sizeof(top_), hipMemcpyHostToDevice, c->stream));
                                HIPCHK(hipMemcpyAsync(c->d_node_prof.p + m.child, d->h_head + 6, sizeof(po), hipMemcpyHostToDevice, c->stream));
                                HIPCHK(hipMemcpyAsync(c->d_node_len.p + m.child, d->h_head, sizeof(int), hipMemcpyHostToDevice, c->stream));
                                if (d->recv(c->d_prof_arena.p + po, sizeof(float) * (size_t)need, m.src)) return KA_FAIL;
                                if (cols && d->recv(d->d_colbuf.p, sizeof(int) * (size_t)m.ncols, m.src)) return KA_FAIL;
                                if (cols) ka_launch_cols_pack(c->d_colof.p, c->d_seq_off.p, c->d_node_len.p, m.d_members.p, m.d_moff.p, m.nmem, d->d_colbuf.p, 1, c->stream);
                                HIPCHK(hipStreamSynchronize(c->stream));          // (h_head is reused by the next hand-over)
                                c->injected.push_back(m.child);
                        }
                }
                if (dst == d->rank && !stop) {
                        HIPCHK(hipMemsetAsync(c->d_counters.p + 1, 0, sizeof(unsigned long long), c->stream));
                        ka_launch_task_level(&D, d->top_blocks[i].p, (int)d->top_blocks[i].n, 0, 0, c->stream);
                        c->n_launches++;
                        c->task_done[t] = 1;
                }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        c->ran = true; c->partial = true;
        // how this rank's kernels went (ka_tree_sync would turn an overflow of a partial run into a failure of this rank alone)
        int err = 0;
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(&err, c->d_error.p, sizeof(int), hipMemcpyDeviceToHost));
        if (err >= 1 && err <= 4) { *status = std::max(*status, 1); *grow |= (err == 1) ? 1 : (err == 2 ? 2 : (err == 3 ? 4 : 8)); }
        else if (err != 0) {
                // (the device's error words, ka_device.h: 5 / 6 the two watchdogs, 7 the vote table of a profile that outgrew its LDS slot)
                *status = 2;
                fail(err == 5 ? "device watchdog: a strip pipeline stopped making progress"
                   : err == 6 ? "device watchdog: a wait between workgroups never completed"
                   : err == 7 ? "anchor consistency: a vote table did not fit its LDS slot"
                   : "device error " + std::to_string(err));
        }
        (void)n_tasks;
        return KA_OK;
}

// One step of the sharded tree: from the leaves to every rank holding every record and coded path.  Every rank learns how
// every other rank's part went BEFORE the collectives of the gather (all-reduce of the status: a rank that failed alone would
// leave the others waiting in RCCL); an arena overflow anywhere makes every rank repeat the step, the ranks that overflowed
// with bigger arenas -- what ka_tree_sync does for a single GPU.
extern "C" int ka_dist_tree_run(ka_dist* d)
{
        if (!d || !d->planned) return fail("ka_dist_tree_run: plan first");
        ka_ctx* c = d->c;
        HIPCHK(hipSetDevice(c->device));
        const auto t_begin = std::chrono::steady_clock::now();
        const int n_tasks = c->n_tasks;
        if (d->d_status.alloc(1)) return fail("hipMalloc failed");
        for (int attempt = 0; ; attempt++) {
                int status = 0, grow = 0;
                if (dist_tree_attempt(d, &status, &grow)) return KA_FAIL;
                const std::string why = g_err;
                int agreed = status;
                if (d->world > 1 || d->comm) {
                        HIPCHK(hipMemcpyAsync(d->d_status.p, &status, sizeof(int), hipMemcpyHostToDevice, c->stream));
                        if (d->all_reduce_i32(d->d_status.p, 1, true)) return KA_FAIL;
                        HIPCHK(hipMemcpyAsync(&agreed, d->d_status.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                        HIPCHK(hipStreamSynchronize(c->stream));
                }
                if (agreed == 0) break;
                if (agreed >= 2) return fail(status >= 2 ? why : std::string("ka_dist_tree_run: another rank's part of the step failed"));
                if (attempt >= 24) return fail("ka_dist_tree_run: device arenas kept overflowing");
                d->retries++;
                if (grow & 1) { c->prof_cap *= 2; c->path_cap *= 2; c->d_prof_arena.release(); c->d_path_arena.release(); }
                if (grow & 2) { c->scratch_cap *= 2; c->d_scratch.release(); }
                if (grow & 4) { c->path_cap *= 2; c->d_path_arena.release(); }
                if (grow & 8) { c->dbg_cap *= 2; c->d_dbg_arena.release(); }
                if (c->d_prof_arena.alloc((size_t)c->prof_cap) || c->d_path_arena.alloc((size_t)c->path_cap) ||
                    c->d_scratch.alloc((size_t)c->scratch_cap) || c->d_dbg_arena.alloc((size_t)std::max<long long>(c->dbg_cap, 1)))
                        return fail("hipMalloc failed while growing an arena");
        }
        // ---- every rank ends with every record and every coded path ----
        if (ka_tree_sync(c)) return KA_FAIL;                        // (clean on every rank: reads the counters)
        ka_launch_path_counts(c->d_recs.p, d->d_mine.p, n_tasks, d->d_counts.p, c->stream);
        if (d->all_reduce_i32(d->d_counts.p, (size_t)n_tasks, false)) return KA_FAIL;
        std::vector<int> counts(n_tasks);
        HIPCHK(hipMemcpyAsync(counts.data(), d->d_counts.p, sizeof(int) * n_tasks, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        d->goff.assign(n_tasks, 0);
        long long total = 0;
        for (int t = 0; t < n_tasks; t++) { d->goff[t] = total; total += counts[t]; }
        if (d->d_gpaths.alloc((size_t)std::max<long long>(total, 1))) return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(d->d_goff.p, d->goff.data(), sizeof(long long) * n_tasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(d->d_gpaths.p, 0, sizeof(int) * (size_t)total, c->stream));
        ka_launch_path_scatter(c->d_recs.p, d->d_mine.p, n_tasks, c->d_path_arena.p, d->d_goff.p, d->d_gpaths.p, c->stream);
        static_assert(sizeof(ka_task_rec) % 4 == 0, "records are reduced as 32-bit words");
        if (d->group_start()) return KA_FAIL;
        if (d->all_reduce_i32(d->d_gpaths.p, (size_t)total, false)) return KA_FAIL;
        if (d->all_reduce_i32((int*)c->d_recs.p, (size_t)n_tasks * (sizeof(ka_task_rec) / 4), false)) return KA_FAIL;
        if (d->group_end()) return KA_FAIL;
        d->h_recs.resize(n_tasks);
        d->h_paths.resize((size_t)total);
        HIPCHK(hipMemcpyAsync(d->h_recs.data(), c->d_recs.p, sizeof(ka_task_rec) * n_tasks, hipMemcpyDeviceToHost, c->stream));
        if (copy_to_host(c, d->h_paths.data(), d->d_gpaths.p, sizeof(int) * (size_t)total)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int t = 0; t < n_tasks; t++) d->h_recs[t].path_off = (int)d->goff[t];
        d->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return KA_OK;
}

// How many times ka_dist_tree_run repeated a step on this rank because an arena overflowed somewhere (tests, reports)
extern "C" int ka_dist_retries(ka_dist* d) { return d ? d->retries : -1; }

// Records (task order, path_off into paths) and coded paths of the last ka_dist_tree_run; *used = ints written.
extern "C" int ka_dist_download(ka_dist* d, ka_task_rec* recs, int* paths, long long paths_cap, long long* used)
{
        if (!d || d->h_recs.empty()) return fail("ka_dist_download: run first");
        if ((long long)d->h_paths.size() > paths_cap) { g_err = "paths_out too small"; return KA_ERR_PATHS_CAP; }
        memcpy(recs, d->h_recs.data(), sizeof(ka_task_rec) * d->h_recs.size());
        memcpy(paths, d->h_paths.data(), sizeof(int) * d->h_paths.size());
        if (used) *used = (long long)d->h_paths.size();
        return KA_OK;
}

extern "C" long long ka_dist_paths_size(ka_dist* d) { return d ? (long long)d->h_paths.size() : -1; }
extern "C" double ka_dist_last_ms(ka_dist* d) { return d ? d->last_ms : -1.0; }
