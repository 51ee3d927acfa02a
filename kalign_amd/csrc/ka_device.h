// ka_device.h -- device-side data layout shared by the kernels and the host dispatcher.
#pragma once
#include <stdint.h>
#include "kalign_amd.h"

#define KA_F 3.402823466e+38f          // FLT_MAX: the reference's "minus infinity" is -FLT_MAX (aln_seqseq.c:45)
#define KA_REC 64                      // floats per profile record (aln_setup.c:40-99)

enum { KA_SS = 0, KA_SP = 1, KA_PP = 2 };
enum { KA_FWD = 0, KA_BWD = 1 };

struct KaState { float a, ga, gb; };   // struct states, aln_struct.h:9-14

// One task (a, b) -> c, prepared on the host: everything do_align derives before the DP
// (aln_run.c:213-237) that depends on the tree only.
struct KaTaskDesc {
        int a, b, c;
        int nsip_a, nsip_b;
        int is_root;
        float gpo, gpe, tgpe;          // scaled by gap_scale when scaling applies
        float soff;                    // subm_offset
        float gap_scale;
        int parent;                    // task that consumes node c (-1: root)
        int chain_need;                // chained launch: how many children of this task run inside the same launch (0: entry task)
        int qa, qb;                    // queued launch: the tasks of the same launch that produce operands a / b (-1: ready before it starts);
                                       // overlapping launches (KaTreeDev::overlap): of ANY launch of the run but the task's own chained launch
        int wait_mult;                 // chained launch: multiplier of the join watchdog (~2 s each), from the estimated DP cells below this task
        int refine;                    // refinement pass, KALIGN_REFINE_CONFIDENT: this edge is refined (confidence at or below the median)
};

// Join point of a task in a chained launch (see ka_task_entry): the clusters that computed its children meet here.
struct KaJoin {
        unsigned int arrive;           // clusters arrived so far
        unsigned int sum_g;            // workgroups they bring
        unsigned int go;               // set by the last arriver once join_base / join_g are valid
        int join_base;                 // member index of the first joining workgroup (= size of the leading cluster)
        int join_g;                    // workgroups offered to this task (the kernel may use fewer)
        int role;                      // written by the child's first workgroup for its own cluster: 1 leads the parent, 2 joins
        int pad[2];
};

// One Hirschberg sub-problem: window + injected boundary states (aln_controller.c:194-436)
struct KaSub {
        int starta, enda, startb, endb;
        KaState fin, bin;
        int roff;                      // offset of its f/b row slices in the task's row buffers
        int pad;
        // Hirschberg prefix reuse (round 5; ka_meetup.h): >= 0 -- this sub-problem's forward / backward row is not computed by a pass
        // of its own: it is the row its parent's pass saved on the way, at this offset of the PARENT level's save buffer
        // (TaskShared::sfbuf / sbbuf); -1: the pass runs
        int fsrc, bsrc;
};

struct KaCtl;

struct KaTreeDev {
        const uint8_t* codes;
        const int* seq_off;
        int* node_len;                 // [2N-1] sequence length / profile length (msa->plen)
        long long* node_prof;          // [2N-1] offset (floats) of the node's profile in prof_arena
        long long* node_vote;          // [2N-1] offset (floats) of the node's carried vote table in prof_arena (ka_votes_merge), -1: none
        float* prof_arena;
        unsigned long long* counters;  // [0] prof_top, [1] scratch_top, [2] path_top, [3] dbg_top (floats), [4] head of the queued launch, [5] queue tasks taken by workgroups of the chained launch
        long long prof_cap, scratch_cap, path_cap, dbg_cap;
        char* scratch;
        int* path_arena;
        float* dbg_arena;
        long long* dbg_off;            // [n_tasks] offset of the task's debug rows, -1 if none
        const KaTaskDesc* tasks;
        KaCtl* ctl;                    // [n_tasks] cluster control blocks, zeroed before every run
        KaJoin* join;                  // [n_tasks] join points of the chained launch, zeroed before every run
        ka_task_rec* recs;
        const float* subm;             // 23*23
        float gpo0, gpe0, tgpe0, usw;  // unscaled penalties for update_n, use_seq_weights
        int numseq;
        int flags;
        int nres;                      // alphabet size: 23 protein, 5 nucleotide (alphabet.c)
        long long* timing;             // [n_tasks][8] phase cycle counts (KA_FLAG_TIMING) or null
        int refine_mode;               // refinement pass (ka_tree_refine): 0 none, 1 KALIGN_REFINE_ALL, 2 KALIGN_REFINE_CONFIDENT
        int refine_adaptive;           // ... with aln_param's adaptive_budget (modes 1, 2)
        int refine_trials;             // ... mode 3 (KALIGN_REFINE_INLINE): trials per edge (create_msa_tree_inline_refine's n_trials; 3 in kalign_run)
        int wdfs;                      // refinement: bit 0 small subtrees of the depth-first recursion run wave-locally (KA_NO_WDFS=1 in the
                                       // environment: off), bit 1 the baseline trial runs level-synchronously (KA_NO_LS0=1: off)
                                       // bit 2 flip trials re-run only the subtrees they flip (ka_trial_incremental; KA_NO_INC=1: off)
                                       // bit 3 the wave-local subtrees of the depth-first recursion keep everything in LDS (ka_subtree_dfs; KA_NO_LDFS=1: off)
        int prof_task;                 // KA_FLAG_TIMING: the task whose per-level times are kept (-1: the root; KA_PROF_TASK in the environment)
        int* trace;                    // host-pinned breadcrumb buffer (KA_TRACE=1) or null
        int* error;                    // 0 ok; 1 prof arena, 2 scratch, 3 path arena, 4 dbg arena overflow, 5/6 watchdogs, 7 LDS vote table
        // ---- anchor consistency (anchor_consistency.c); cons_K == 0: off ----
        int max_g;                     // workgroups one task may use when clusters merge up the chained launch
        int mw_mode;                   // the top-level meetup scans are shared by the waves of the leading workgroup (KA_MW=0: off)
        int sub_mode;                  // wave-local subtrees in LDS (ka_subtree.h); 0: off (KA_SUBTREE=0, experiments)
        int lean4;                     // leaf levels (seq-seq tasks only) on 4-wave workgroups, four per CU (KA_LEAN4)
        int q1_mode;                   // 64-row strips (one DP row per lane): 0 never, 1 for tasks whose cluster has a SIMD per top-level strip,
                                       // 2 also at two strips per SIMD, 3 always (experiments; KA_Q1 in the environment)
        int ho_mode;                   // neighbouring strips of a pass hand over through LDS rings (ka_strip<.., HO>): 0 off, 1 on,
                                       // 2 on with four strips per workgroup (KA_HO in the environment)
        int hw_mode;                   // strips with helper waves (ka_wstrip.h) on levels with at most four items per workgroup: 0 off, 1 on (KA_HW in the environment)
        int reuse;                     // Hirschberg prefix reuse in the 4-wave kernels (KA_REUSE=0: off)
        int overlap;                   // round 5: the launches of a run go out TOGETHER on streams of their own and order themselves by the tasks' done flags
                                       // (KaJoin::go): every task sets its flag, every task waits for the flags of the tasks that make its operands (qa / qb)
        const int2* q_order;           // the chained launch of an overlapping run: the queued launch's list, its length and its workgroup slots --
        int q_n, q_slots;              // a chained workgroup that finds more than a round of the queue still to do helps (ka_task_entry); 0: no
        int qw, lw;                    // waves per workgroup of the queued launch (KA_QW: 4, 2 or 1) / of the seq-seq leaf levels (KA_LW)
        int per_target;                // experiments (KA_PER): strips per workgroup a profile-profile task aims for at its top level (0: the built-in table)
        int carry;                     // round 5: a node's anchor votes are carried up the tree (ka_votes_merge) instead of counted again from every
                                       // member at every task (KA_CARRY=1; off by default: the sweeps that settle its marked cells cost what the votes cost)
        int cons_K;                    // anchors
        int cons_maxlen;               // longest sequence: bounds every anchor position
        float cons_paw;                // weight / (float)K  (per_anchor_weight, anchor_consistency.c:487)
        const int* cons_maps;          // position maps: (sequence i, anchor k) at cons_map_off[i] + k * len_i
        const long long* cons_map_off; // [numseq]
        int* colof;                    // [sum of lengths], indexed like codes: residue -> column of the profile of the
                                       // node that currently contains the sequence (the device's form of gaps[])
        const int* sip;                // member lists of every node in the reference's order (aln_run.c:428-436)
        const long long* sip_off;      // [2N-1]
        int reserve;                   // round 6, the queued launch of an overlapping run: its workgroups leave the first `reserve` CUs of XCC 0 (shader
                                       // engines 0 .. reserve / 8 - 1) to the head of the chained launch (plan_launches; 0: none)
        int tp;                        // round 6: launches of the 4-wave kind go to the throughput kernel (unit 10) when the job allows it (host: ka_tp_ok)
        int merge_batch;               // round 6, ka_update_profile: bit 0 = items in batches when both operands' records are in HBM, bit 1 = also when one is a sequence, bit 2 = also in a cluster of workgroups (KA_MERGE)
};

#define KA_BLK_NOHELP (1 << 30)        // a block of the chained launch (blocks[b].y): this workgroup does not help the queued launch -- it sits on a CU kept for it
#define KA_NB 6                        // bonus entries a DP row carries: <= 5 anchors + the wrap-around entry
// ... of the second set of consistency kernels, `--consistency K` with 5 < K <= KA_CONS_MAX_ANCHORS (128; 32 until round 6).  Round 4 carried eleven entries in
// registers (K <= 10: 1383 VGPR spills in the task kernel); round 5 STREAMS them: a row's entries lie sorted by column in the task's
// scratch ([0] pad, [1] a lower sentinel that also holds the count, the entries, an upper sentinel, a pad) and every lane walks its row's
// list along with its columns (KaBonus<NB>::STREAM, ka_pass.h) -- eight registers per DP row whatever K is.
#define KA_NB_BIG 136                   // (K + the wrap-around entry + two sentinels + two pads, K <= KA_CONS_MAX_ANCHORS = 128)

struct KaPairDev {
        const uint8_t* codes;
        const int* seq_off;
        const int* seq_len;
        const int* ia;
        const int* ib;
        const float* subm;
        float gpo, gpe, tgpe;
        char* scratch;
        long long scratch_stride;      // bytes per pair
        int* paths_out;
        const long long* poff;
        float* scores;
        int* error;
        int npairs;
        int pw;                        // waves per workgroup (KA_PW: 4, 2 or 1)
        int reuse;                     // Hirschberg prefix reuse (KA_REUSE=0: off)
};
