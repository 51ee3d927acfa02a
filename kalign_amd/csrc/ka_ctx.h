// ka_ctx.h -- what the host-side translation units of the library share (round 5: ka_api.cpp was one file of 2700 lines): the
// context, the environment switches, the kernel launchers' prototypes and the functions one unit calls in another.  Library-internal:
// nothing here is part of the C ABI (include/kalign_amd.h).
//   ka_api.cpp   contexts, upload, the runs (ka_tree_run / _refine / _sync / _download), rows, realignment tree, ka_run_encoded, partial runs
//   ka_plan.cpp  the launch planner: levels, leaf / queued / chained launches, clusters, spare workgroups by a simulated schedule
//   ka_cons.cpp  anchor consistency (anchors, the N x K batch, position maps), the seq-seq pair batch, the distance batch
//   ka_dist.cpp  one alignment over the GPUs of a node: RCCL loaded at run time, the sharded consistency batch and tree, the in-process transport
#pragma once
#define KA_INTERNAL __attribute__((visibility("hidden")))
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <thread>
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
static inline bool ka_cons_big(const KaTreeDev* D) { return D->cons_K > KA_NB - 1; }
// round 6: the throughput kernel (unit 10; three four-wave workgroups per CU) in place of the half kernel: fast mode, no B / Z / X
extern "C" void ka_unit10_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int nqueue, hipStream_t stream);
static inline bool ka_tp_ok(const KaTreeDev* D) { return D->tp != 0 && D->cons_K == 0 && D->nres <= 20; }
// kind: 0 = 8-wave kernel, 1 = lean (seq-seq only), 2 = half (4 waves, two workgroups per CU)
static inline void ka_launch_task_level(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int kind, int chain, hipStream_t stream)
{
        const int cons = D->cons_K > 0;
        if (ka_cons_big(D)) {
                if (kind == 1) ka_unit8_launch(D, blocks_dev, nblocks, stream);
                else if (kind == 2) ka_unit7_launch(D, blocks_dev, nblocks, 0, stream);
                else ka_unit6_launch(D, blocks_dev, nblocks, chain, stream);
                return;
        }
        if (kind == 1) ka_unit3_launch(D, blocks_dev, nblocks, cons, stream);
        else if (kind == 2 && ka_tp_ok(D)) ka_unit10_launch(D, blocks_dev, nblocks, 0, stream);
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
extern "C" long long ka_scratch_bytes_host_big(long long la, long long lb, long long cons_maxlen, long long k_anchors);

// the thread's error text (ka_last_error) and the one way to set it: defined in ka_api.cpp
KA_INTERNAL int fail(const std::string& m);

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
        int merge = 15;                // KA_MERGE: ka_update_profile in batches (bit 0: operands with records in HBM, bit 1: sequences too, bit 2: clusters too, bit 3: the seq-seq tasks of the 128-register units)
        int per = 0;                   // KA_PER: strips per workgroup (KaTreeDev::per_target; experiments)
        int ho = -1;                   // KA_HO: hand-over between neighbouring strips through LDS (KaTreeDev::ho_mode); -1: on (1)
        int hw = 1;                    // KA_HW: profile-profile strips with helper waves (ka_wstrip.h; KaTreeDev::hw_mode)
        int hw_prio = 3;               // KA_HW_PRIO: s_setprio of a strip wave that has a helper (experiments)
        int subtree = 1;               // KA_SUBTREE: small Hirschberg subtrees run wave-locally in LDS
        int overlap = 1;               // KA_OVERLAP: the chained launch goes out beside the queued launch (a stream of its own, ordered by the tasks' done flags)
        int overlap_help = 1;          // KA_OVERLAP_HELP: workgroups of the chained launch that arrive before the queue's last round take queue tasks
        int carry = 0;                 // KA_CARRY=1: carried vote tables (ka_votes_merge; measured, off: DESIGN 4i; 3: marks settled by the sweep only) -- 0: every task counts its members' votes
        int reuse = 1;                 // KA_REUSE: Hirschberg prefix reuse in the 4-wave kernels (queued levels, seq-seq leaves, pair batch)
        int tp = 0;                    // KA_TP=1: the queued launch and the levels with more tasks than CUs on the throughput kernel (unit 10) where it applies (ka_tp_ok); measured slower than the 4-wave kernel: off
        int qw = 4, lw = 4, pw = 2;    // KA_QW / KA_LW / KA_PW: waves per workgroup of the queued launch, the seq-seq leaf levels, the pair batch (4, 2, 1)
        bool launch_ev = false;        // KA_LAUNCH_EV: an event behind every launch of a run (ka_tree_launch_ms)
        bool upgma_launches = false;   // KA_UPGMA_LAUNCHES: ka_aln_guide_tree's UPGMA as one launch per merge (the path for > 6144 sequences) at any size
};
static inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline void read_env(KaEnv& v)
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
        v.carry = env_int("KA_CARRY", 0);
        v.overlap_help = env_int("KA_OVERLAP_HELP", 1);
        v.overlap = env_int("KA_OVERLAP", 1);
        v.tp = env_int("KA_TP", 0);
        v.qw = env_int("KA_QW", 4); v.lw = env_int("KA_LW", 4); v.pw = env_int("KA_PW", 2);
        for (int* w : { &v.qw, &v.lw, &v.pw }) if (*w != 1 && *w != 2) *w = 4;
        v.mw = env_int("KA_MW", 1);
        v.merge = env_int("KA_MERGE", 15);
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
        bool own_stream = false;                     // `stream` was created by ka_ctx_set_shared (destroyed with the context)
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
        int fallback_streak = 0;                     // ... on fast-plan jobs in a row (a clean fast-plan run resets it)
        int fallback_hold = 0;                       // jobs that stay on the shared plan before the fast plan is tried again (0 after a first fallback, then 4, 16, 64)
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
        DevBuf<long long> d_node_prof, d_node_vote, d_dbg_off, d_timing;
        DevBuf<float> d_prof_arena, d_subm, d_dbg_arena;
        DevBuf<unsigned long long> d_counters;
        DevBuf<char> d_scratch, d_ctl;
        DevBuf<KaJoin> d_join;
        int n_trees = 1;               // guide trees in the job (a forest when > 1)
        int chain_level = -1;          // first level of the chained launch (-1: every level is its own launch)
        int queue_first = -1;          // queued launch: levels queue_first .. chain_level-1 run as ONE launch of the half kernel (-1: none)
        int queue_off = 0, queue_n = 0; // its task list in blocks_flat
        std::vector<char> spine;       // round 6: tasks below the chain's first level that run in the chained launch all the same (plan_launches)
        int reserve_cus = 0;           // round 6: CUs of XCC 0 the queued launch leaves to the head of the chained launch (plan_launches; 0: none)
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
        // overlapping launches (KA_OVERLAP): the chained launch on a stream of its own (lowest priority) beside the queued launch, events to
        // fork from / join into the context's stream; overlap_plan: the current plan carries the dependencies for it
        hipStream_t s_chain = nullptr;
        hipEvent_t e_fork = nullptr, e_chain = nullptr;
        int overlap_plan = 0;
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

KA_INTERNAL void build_blocks(const ka_ctx* c, const std::vector<int>& L, std::vector<int2>& tbl, int* lean_out);
KA_INTERNAL int plan_launches(ka_ctx* c);
KA_INTERNAL int upload_plan(ka_ctx* c);
KA_INTERNAL int setup_colof(ka_ctx* c);
KA_INTERNAL int refine_blocks(ka_ctx* c, int mode);
KA_INTERNAL int pairwise_on_device(ka_ctx* c, const uint8_t* codes, const int* off, const int* lens, int numseq,
                              const int* ia, const int* ib, int npairs,
                              const float* subm, float gpo, float gpe, float tgpe, const long long* poff, long long* ptotal_out);
KA_INTERNAL void node_members(const ka_ctx* c, int node, long long* lo, long long* hi);

KA_INTERNAL int copy_to_host(ka_ctx* c, void* dst, const void* src, size_t bytes);
KA_INTERNAL int tree_reset(ka_ctx* c);
KA_INTERNAL KaTreeDev tree_dev(ka_ctx* c);
KA_INTERNAL int tree_launch(ka_ctx* c, bool reset = true);

