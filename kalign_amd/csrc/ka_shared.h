// ka_shared.h -- the workgroup's shared state (TaskShared, KaCtl), small device helpers and the column-operand terms.
// One of the text sections of the task kernels, included by ka_kernels.hip in this order: ka_shared.h, ka_pass.h, ka_best.h,
// ka_subtree.h, ka_wstrip.h, ka_meetup.h, ka_hirschberg.h, ka_path.h, ka_profile.h, ka_task.h.  Not a stand-alone header.
#pragma once

#define KA_BLOCK 512                     // task kernel: 8 waves, one workgroup per CU (LDS ring per wave)
#define KA_WAVES (KA_BLOCK / 64)
#define KA_PAIR_BLOCK 256                // seq-seq pair kernel: 4 waves, no ring -> several workgroups per CU
#define KA_LEAN_BLOCK 512                // seq-seq levels of the tree: 8 waves (a 400-row task has 8 strips at level 2), two workgroups per CU
#define KA_NT ((int)blockDim.x)          // threads / waves of the running workgroup
#define KA_NW ((int)blockDim.x >> 6)

__device__ __forceinline__ float kmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float kmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// lane l receives lane l-1's value (lane 0 keeps its own): v_mov_b32_dpp wave_shr:1
__device__ __forceinline__ float wave_shr1(float x)
{
        int xi = __float_as_int(x);
        return __int_as_float(__builtin_amdgcn_update_dpp(xi, xi, 0x138, 0xf, 0xf, false));
}

__device__ __forceinline__ float ka_uniform_f(float x)
{
        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

// a wave-uniform pointer that came out of LDS (a VGPR pair as far as the compiler knows) as an SGPR pair: the strip's event code
// kept such pointers in scratch memory and reloaded them -- a memory round trip each -- several times per event step
template <typename T>
__device__ __forceinline__ T* ka_uniform_ptr(T* p)
{
        const unsigned long long x = (unsigned long long)p;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32));
        return (T*)(((unsigned long long)hi << 32) | lo);
}

// Bounded spin: gives up when the limit is reached, reporting `code` unless an error is already set.
// other_tasks: the wait depends on ANOTHER task (a join point of the chained launch): also give up, checked
// every 256 iterations, as soon as any workgroup has reported an error -- a failed task (arena overflow)
// never signals its consumers, and the run is going to be repeated anyway.  Waits inside a task must not do
// that: the task itself is healthy and has to run to its end.
__device__ __forceinline__ bool ka_spin_expired(int* err, int spins, int limit, int code, bool other_tasks = false)
{
        if (spins > limit) { atomicCAS(err, 0, code); return true; }
        if (other_tasks && (spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
        return false;
}

__device__ __forceinline__ float lane_bcast(float x, int src_lane)
{
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src_lane));
}

// The mutable state of one task's recursion: lives in LDS when one workgroup owns the task, in
// HBM (zeroed by the host before the run) when a cluster of workgroups on different CUs shares it.
struct KaCtl {
        // per-level counters, double-buffered by level parity: level L consumes lvl[L & 1] while its
        // meetups fill lvl[(L + 1) & 1] (zeroed at the start of level L) -> two barriers per level
        struct Lvl { int nsub, rowalloc, nitems, next_item, npack[2], next_job, pad; } lvl[2];
        int mcount;
        int top_meet, top_tr;
        float top_score;
        double msum;
        int alnlen;
        int fail;
        unsigned int bar;               // cluster barrier: arrivals so far (monotonic)
        int nrec;                       // (recursion-order key, margin) records appended so far (exact confidences; all members of a cluster)
        long long scratch_off;          // cluster: scratch block allocated by member 0
        long long newp_off;             // merged profile offset in the arena (-1: root / none)
        long long path_off;             // coded path offset in the path arena
        int vote_conf;                  // carried votes (ka_votes_merge): cells of the merged node that need an operand's members counted ...
        int vote_types;                 // ... bit 0: some of them b's members, bit 1: a's
};

// Everything the waves of a workgroup share about the task being aligned.
struct TaskShared {
        int kind, swapped;
        int len_a, len_b;              // operand lengths in (a, b) order
        int La, Lb;                    // DP rows / columns
        const uint8_t* s1;             // row residues (seq-seq)
        const uint8_t* s2;             // column residues (seq-seq, seq-profile)
        const float* p1;               // row profile
        const float* p2;               // column profile
        float* profa;                  // operand profiles in (a, b) order
        float* profb;
        const float* subm;
        float gpo, gpe, tgpe, soff;
        float sp_open, sp_ext, sp_text;
        float p1_mult, p2_mult;        // (float)nsip of the OTHER operand: set_gap_penalties_n folded into the loads
        KaState* fbuf;
        KaState* bbuf;
        KaState* xfbuf;                // hand-over rows between strips of one pass that run in DIFFERENT workgroups with helper waves (ka_whelper):
        KaState* xbbuf;                //   written and read past the caches (agent-scope atomics) -- kept apart from fbuf / bbuf, which plain loads read
        // Hirschberg prefix reuse (ka_meetup.h): the row a pass of recursion level L leaves for the sub-problem's usual child, by level
        // parity (the children read it in level L+1's meetups while that level's own passes fill the other one); sliced by KaSub::roff
        KaState* sfbuf[2];
        KaState* sbbuf[2];
        int carried;                   // this task's anchor positions come from its operands' carried vote tables (ka_cons_from_tables)
        int reuse_ok;                  // ... enabled for this task (a kernel built with it, one workgroup, no recursion-order records)
        KaSub* q[2];
        int* raw;
        int* raw2;
        int* coded;
        int* srcA;
        int* srcB;
        // anchor consistency (only carved when the job has a consistency table)
        int2* ent;                     // [La][KA_NB] bonus entries of every DP row: (column, value bits)
        int* apos_r;                   // per anchor: anchor position / confidence of every DP row and column
        float* conf_r;
        int* apos_c;
        float* conf_c;
        int* invj;                     // anchor position -> DP column
        char* vote;                    // HBM vote tables for profiles too long for LDS (16 B per column)
        KaCtl* ctl;                    // the task's control block: -> ctl_lds (one workgroup) or the task's block in HBM (cluster)
        KaCtl* lctl;                   // level counters + margin sums of the recursion: == ctl until a cluster SPLITS, then -> ctl_lds
        KaCtl ctl_lds;
        int G, member;                 // cluster size / this workgroup's index in it
        int sub_ok, nres_t, sub_stride; // wave-local subtrees (ka_subtree.h): enabled for this task / alphabet class (5, 20, 23) / bytes per wave
        char* sub_base;                //   ... and where the waves' LDS regions start
        float mw_mx[8], mw_mx2[8];     // multi-wave meetup scan: the waves' partial (best, second best, key of the best)
        int mw_key[8];
        int mw_ok;                     //   ... enabled (KaTreeDev::mw_mode)
        int sub_tm;                    // KA_FLAG_TIMING, the profiled task: subtree phase times are accumulated in sub_t
        unsigned long long sub_t[7];   //   subtrees, staging / pass / meetup / total cycles (sums over the workgroup's subtrees), longest one, sum of level*1e6 + R*1e3 + C
        int srows;                     // rows per strip of this task: 128 (two DP rows per lane) or 64 (one; ka_strip<.., Q = 1>)
        int lvl_srows[2];              // ... of the recursion level with this parity (== srows unless q1_lvl)
        int q1_lvl;                    // KaTreeDev::q1_mode 4: every recursion level takes 64-row strips when the cluster has a SIMD for each of them (ka_level_srows)
        int ho_ok;                     // neighbouring strips of this task hand over through LDS rings (ka_strip<.., HO>; KaTreeDev::ho_mode, profile-profile tasks of the 8-wave kernel)
        int hw_ok;                     // levels with at most four items per workgroup run their strips with helper waves (ka_wstrip.h; KaTreeDev::hw_mode, profile-profile tasks of the 8-wave kernel)
        // The recursion of a cluster: levels whose passes need more than one CU run cluster-wide (Gw = G: strips spread
        // over the workgroups, agent-scope hand-over, two cluster barriers per level).  As soon as a level has at least
        // G sub-problems (or single-strip passes) the cluster SPLITS: every workgroup takes its share of the
        // sub-problems -- independent subtrees of the recursion -- into private queues / row buffers and finishes them
        // on its own (Gw = 1: workgroup barriers and workgroup-scope hand-over only); one cluster barrier at the end.
        // refinement trial state (ka_meetup<.., FLIP>; aln_struct.h:32-35): threshold, trial / stride / running counter of the
        // round-robin flips, fp32 margin sum and count in DFS order
        struct Refine { float thr; int trial, stride, counter; float msum; int mcount; } rf;
        int dfs_top, dfs_valid;        // ka_hirschberg_dfs: sub-problems on the stack / a sub-problem was popped
        int* best_coded; int* best_srcA; int* best_srcB;   // refinement: the best trial's coded path and column sources
        int* sp_freq;                  // refinement: residue counts [23] + residues per column [1] of both operands (compute_sp_score)
        float* mlog;                   // refinement, adaptive budget: the margins of the trial in recursion order (first mlog_cap of them), or null
        int mlog_cap, adapt_trials;
        int2* mrec;                    // refinement, level-synchronous baseline trial: (recursion-order key, margin) of every meetup
        char* inc;                     // refinement, incremental flip trials (KaInc): the baseline's meetups with their windows, sorted; or null
        int inc_n, inc_nunc;           //   ... records of the baseline / the uncertain ones among them (margin < threshold)
        int inc_j, inc_p;              //   ... walk state: sorted position of the next flip (-1: none) / first record not yet taken over
        int rec_on;                    // first pass with exact confidences (KA_FLAG_EXACT_CONFIDENCE): every meetup records (key, margin)
        float sp_value;
        int Gw, member_w;              // cluster size / member index the recursion currently works with
        int split;
        struct Priv { KaSub* q[2]; int2* items[2]; int* prog[2]; int2* pack[2][2]; KaState* f; KaState* b; } priv;
        unsigned int bar_phase;        // cluster barriers passed so far
        int2* items[2];                // work items of the current / next recursion level: (sub-problem, dir<<16 | strip)
        int* prog[2];                  // per-item progress words (columns of the strip's last row published)
        int2* pack[2][2];              // [level parity][class]: small passes (sub-problem, dir); class 0: 16-lane slots, 1: 4-lane slots
        float* newp;
        int* path_dst;
        int* trace;
        int dbgskip;
        int* watchdog;                 // device error word: a bounded spin that expired writes 5 here
        long long t_pass, t_meet;      // KA_FLAG_TIMING: shader-clock cycles spent in passes / meetups
        int n_levels;
        int next_member, next_g;       // chained launch: this workgroup's place in the parent task's cluster
        long long* prof;               // KA_PROF builds: per (level, wave) timestamps of the root task
        int lvl_n[16];                 // per recursion level: sub-problems, pass / meetup cycles
        int lvl_pass[16], lvl_meet[16];
};

// ------------------------------------------------------------------------------------------
// column-operand terms for column record `rec`
// ------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void col_terms(const TaskShared& S, int rec, float& copen, float& cext, float& ctext)
{
        if (KIND == KA_SS) { copen = -S.gpo; cext = -S.gpe; ctext = -S.tgpe; }
        else if (KIND == KA_SP) { copen = -S.sp_open; cext = -S.sp_ext; ctext = -S.sp_text; }
        else {
                // set_gap_penalties_n (aln_setup.c:101-119): [27..29] = [55..57] * nsip_other, applied
                // on the fly so that profiles stay immutable in HBM
                const float* c = S.p2 + ((long long)rec << 6);
                copen = c[55] * S.p2_mult; cext = c[56] * S.p2_mult; ctext = c[57] * S.p2_mult;
        }
}
