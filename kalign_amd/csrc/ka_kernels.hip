// ka_kernels.hip -- CDNA4 (gfx950) kernels for Kalign's progressive-alignment hot path.
//
// One workgroup aligns one pairwise task (a, b) -> c end to end:
//   P1  operand preparation   make_profile_n / set_gap_penalties_n   (aln_setup.c:40-119)
//   P2  Hirschberg recursion  aln_runner / aln_continue              (aln_controller.c:21-436)
//         level-synchronous inside the workgroup: every wave pulls (sub-problem, direction)
//         passes of the current recursion level, then the waves run the meetups and emit
//         the next level's sub-problems
//       each pass is an anti-diagonal wavefront: lane l owns DP row u0+l of a 64-row strip
//       and walks the columns one step behind lane l-1; the three cell states move to the
//       next lane with a DPP wave shift (v_mov_b32_dpp wave_shr:1)
//   P3  path post-processing  mirror_path_n / add_gap_info_to_path_n (aln_setup.c:121-228,438-462)
//   P4  profile merge         update_n                               (aln_setup.c:230-436)
//
// Arithmetic is IEEE binary32 in the reference's source order with NO contraction
// (compile with -ffp-contract=off): the traceback is an argmax over float sums and must be
// bit-identical to the CPU reference (hard part 1).
//
// The formulation of a pass over (u, v) = (row counter, column counter) is the same as
// oracle/kalign_oracle.c:ko_pass; see there for the mapping to the reference's six functions.
#include <hip/hip_runtime.h>
#include "ka_device.h"
// #define KA_TRACE_STRIP 1   // per-step breadcrumbs into D.trace (debugging hangs)

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

#include "ka_pass.h"

// ------------------------------------------------------------------------------------------
// Meetup of one sub-problem by one wave (aln_seqseq.c:241-420 and the two profile variants),
// then aln_continue: path writes and the two child sub-problems (aln_controller.c:194-436).
// ------------------------------------------------------------------------------------------
// Incremental flip trials of refinement (ka_trial_incremental): what the level-synchronous baseline trial leaves behind, carved
// from the task's scratch behind TaskShared::inc.  n = len_a + len_b + 8 bounds the number of meetups of a trial.
struct KaInc {
        KaSub* win;        // [record] the sub-problem (pad = its recursion-order key)
        int2* mx;          // [record] (width of its subtree's key range, raw path entry of its first row before its subtree ran)
        int* msort;        // [sorted position] record
        int* skey;         // [sorted position] key
        float* mseq0;      // [sorted position] margin = the baseline's margins in recursion order
        float* mseq;       // the running trial's margins in recursion order (2n)
        int* upos;         // [u] sorted position of the u-th uncertain meetup of the baseline
        int* ucnt;         // [sorted position] uncertain meetups in front of it (n + 1)
        int* raw0;         // the baseline's raw path
};
__device__ __host__ inline long long ka_inc_bytes(long long n) { return 88 * n + 64; }
__device__ __forceinline__ KaInc ka_inc_from(char* base, const long long n)
{
        KaInc I;
        I.win = (KaSub*)base; base += 48 * n;
        I.mx = (int2*)base; base += 8 * n;
        I.msort = (int*)base; base += 4 * n;
        I.skey = (int*)base; base += 4 * n;
        I.mseq0 = (float*)base; base += 4 * n;
        I.mseq = (float*)base; base += 8 * n;
        I.upos = (int*)base; base += 4 * n;
        I.ucnt = (int*)base; base += 4 * n + 16;
        I.raw0 = (int*)base;
        return I;
}
static_assert(sizeof(KaSub) == 48, "KaInc::win stride");

__device__ __forceinline__ KaInc ka_inc_view(const TaskShared& S) { return ka_inc_from(S.inc, (long long)S.len_a + S.len_b + 8); }

struct Best { float mx; float mx2; int key; int key2; };          // key2 (who the runner-up is) only matters to refinement trials

__device__ __forceinline__ void best_consider(Best& b, float s, int key)
{
        if (s > b.mx) { b.mx2 = b.mx; b.key2 = b.key; b.mx = s; b.key = key; }
        else if (s > b.mx2) { b.mx2 = s; b.key2 = key; }
}

// (value, key) pairs in the order the reference's sequential scan ranks them: higher value first, among equal values the
// earlier candidate (a later candidate only displaces on a strictly greater value, aln_seqseq.c:284-291)
__device__ __forceinline__ bool best_before(float v1, int k1, float v2, int k2) { return v1 > v2 || (v1 == v2 && k1 < k2); }

__device__ __forceinline__ void best_merge(Best& x, float omx, float omx2, int okey, int okey2 = 0x7fffffff)
{
        if (best_before(omx, okey, x.mx, x.key)) {
                // the other side's best wins: the runner-up is the better of our best and its runner-up
                const bool mine = best_before(x.mx, x.key, omx2, okey2);
                x.mx2 = mine ? x.mx : omx2; x.key2 = mine ? x.key : okey2;
                x.mx = omx; x.key = okey;
        } else {
                const bool theirs = best_before(omx, okey, x.mx2, x.key2);
                x.mx2 = theirs ? omx : x.mx2; x.key2 = theirs ? okey : x.key2;
        }
}

#include "ka_subtree.h"
#include "ka_wstrip.h"

// Queue the two passes of sub-problem `slot` for the next recursion level.  A pass with more
// than 32 rows becomes strip items (its strips are contiguous and ascending, so strip k-1 is
// always pulled before strip k); smaller passes go to the packed lists (16-lane slots for up
// to 32 rows, 4-lane slots for up to 8 rows).
struct KaLevelOut { int2* items; int* prog; int* nitems; int2* pack16; int2* pack4; int* n16; int* n4; int* nsub; int* rowalloc; int srows;
                    int sub_ok, kind, nres, sub_bytes; };       // wave-local subtrees (ka_subtree.h): allowed / what decides whether a window fits
#define KA_ITEM_SUBTREE 2                                      // `dir` of a work item that is a whole subtree
#define KA_SUB_MARK 0x7fffffff                                 // KaSub::pad of such a sub-problem: its level's meetups skip it (the wave that ran it did them)

__device__ __forceinline__ bool ka_child_is_subtree(const KaLevelOut& o, int rows, int cols)
{
        return o.sub_ok && rows <= KA_SUB_MAXROWS && ka_sub_bytes(o.kind, o.nres, rows, cols) <= o.sub_bytes && cols < 4096;
}

// A thin but long pass (few rows, many columns -- gap-rich regions of deep profiles produce them) also runs
// as a strip: its ncols + nrows/2 dependent steps are the level's critical path, a strip step costs about
// 60 % of a packed step, and the other waves of the cluster are idle at that depth anyway.
#define KA_LONG_COLS 96
__device__ __forceinline__ bool ka_pass_is_strip(int nrows, int ncols) { return nrows > 32 || (nrows > 2 && ncols >= KA_LONG_COLS); }

__device__ __forceinline__ void ka_emit_pass(const KaLevelOut& o, int slot, int dir, int nrows, int ncols)
{
        if (ka_pass_is_strip(nrows, ncols)) {
                const int ns = ka_strips_of(nrows, o.srows);
                const int base = atomicAdd(o.nitems, ns);
                for (int k = 0; k < ns; ++k) { o.items[base + k] = make_int2(slot, (dir << 16) | k); o.prog[base + k] = 0; }
        } else if (nrows > 8) {
                o.pack16[atomicAdd(o.n16, 1)] = make_int2(slot, dir);
        } else {
                o.pack4[atomicAdd(o.n4, 1)] = make_int2(slot, dir);
        }
}

__device__ __forceinline__ void ka_emit_items(const KaLevelOut& o, int slot, int starta, int enda, int ncols, bool allow_sub = true)
{
        if (allow_sub && ka_child_is_subtree(o, enda - starta, ncols)) {
                const int base = atomicAdd(o.nitems, 1);
                o.items[base] = make_int2(slot, KA_ITEM_SUBTREE << 16); o.prog[base] = 0;
                return;
        }
        const int mid = ((enda - starta) / 2) + starta;
        ka_emit_pass(o, slot, KA_FWD, mid - starta, ncols);
        ka_emit_pass(o, slot, KA_BWD, enda - mid, ncols);
}

// Rows per strip of recursion level `level` (q1_mode 4): one DP row per lane costs 0.72 of a two-row step (ka_wstrip<.., Q = 1>:
// 290 against 400 cycles) at twice the strips, so a level takes 64-row strips exactly when all of them still get a strip
// wave with a helper -- four per workgroup of the cluster.  From the task's shape and the level alone (an upper bound on the
// level's strips: 2^(level+1) passes of ceil(La / 2^(level+1)) rows): every workgroup and every emitting wave derives the
// same answer without talking.  Once the cluster has split, workgroups work alone on small sub-problems: 128.
__device__ __forceinline__ int ka_level_srows(const TaskShared& S, int level)
{
        if (!S.q1_lvl) return S.srows;
        if (S.split || level > 12) return KA_STRIP_ROWS;
        const int pr = (S.La + (2 << level) - 1) >> (level + 1);
        const long long strips = (long long)(2 << level) * ((pr + KA_STRIP1_ROWS - 1) / KA_STRIP1_ROWS);
        return strips <= 4ll * S.G ? KA_STRIP1_ROWS : KA_STRIP_ROWS;
}

__device__ __forceinline__ KaLevelOut ka_level_out(TaskShared& S, int parity, bool next)
{
        KaLevelOut o;
        o.items = S.items[parity]; o.prog = S.prog[parity];
        o.pack16 = S.pack[parity][0]; o.pack4 = S.pack[parity][1];
        (void)next;
        o.nitems = &S.lctl->lvl[parity].nitems;
        o.n16 = &S.lctl->lvl[parity].npack[0];
        o.n4 = &S.lctl->lvl[parity].npack[1];
        o.nsub = &S.lctl->lvl[parity].nsub;
        o.rowalloc = &S.lctl->lvl[parity].rowalloc;
        o.srows = S.lvl_srows[parity];
        o.sub_ok = S.sub_ok; o.kind = S.kind; o.nres = S.nres_t; o.sub_bytes = S.sub_stride;
        return o;
}

// GL lanes per sub-problem (64 / GL sub-problems per wave): deep recursion levels have hundreds of
// sub-problems with a handful of columns each.
// FLIP: a refinement trial (one sub-problem per call, in DFS order): the margins are summed in fp32 in that order and an
// uncertain meetup may take its runner-up (aln_seqseq.c:376-414, round-robin mode); state in S.rf.
// REC (refinement's baseline trial run level-synchronously): every sub-problem carries its place in the reference's
// depth-first order as a base-3 key in KaSub::pad -- digit 1 / 2 at its depth for the child the recursion enters first /
// second, zeros below: numeric order of the keys = preorder of the recursion tree -- and every meetup appends (key, margin)
// to S.mrec; sorted by key afterwards, the margins add up in the reference's order.  kdig: weight of the children's digit.
// MW (GL = 64 only; all waves of the workgroup call it together): the top recursion levels have one or two sub-problems with
// thousands of candidate columns -- every wave scans every NW-th block of 64 columns, the partial (best, second best)
// pairs meet in TaskShared::mw_* behind a workgroup barrier, and wave 0 merges them (the merge ranks by value and scan
// position, so it does not depend on who found what) and carries on alone: decision, path entries, children.
template <int KIND, int GL, bool FLIP = false, bool REC = false, bool MW = false>
__device__ __forceinline__ void ka_meetup(TaskShared& S, const KaSub* qc, const int k0, const int ncur, KaSub* qnext,
                                          const KaLevelOut& lout, const int wlane, const bool top_level, const int kdig = 0)
{
        static_assert(!MW || GL == 64, "the multi-wave scan works on 64-lane groups");
        const int lane = wlane % GL;                                 // lane within the sub-problem's group
        const int ksub = k0 + wlane / GL;
        const bool in_range = ksub < ncur;
        const KaSub sb = qc[in_range ? ksub : k0];
        // (a wave-local subtree is already complete: path entries written, margins added, no children left)
        const bool rec = REC || S.rec_on;                             // (REC: refinement's baseline trial; rec_on: the first pass with exact confidences)
        const bool valid = in_range && (FLIP || rec || sb.pad != KA_SUB_MARK);
        const bool is_top = top_level && ksub == 0;
        const int startb = sb.startb, endb = sb.endb;
        const int mid = ((sb.enda - sb.starta) / 2) + sb.starta;
        const KaState* f = S.fbuf + sb.roff;
        const KaState* b = S.bbuf + sb.roff;
        const float middle = (float)(endb - startb) / 2.0f + (float)startb;
        const int rrec = mid + 1;
        float g3, g7, g6n, g6f;
        if (KIND == KA_SS) {
                g3 = -S.gpo; g7 = -S.gpo;
                g6n = (startb == 0) ? -S.tgpe : -S.gpe;
                g6f = (endb == S.Lb) ? -S.tgpe : -S.gpe;
        } else {
                const float* R = S.p1 + ((long long)rrec << 6);
                g3 = R[55] * S.p1_mult; g7 = R[55 - 64] * S.p1_mult;
                g6n = (startb == 0) ? R[57] * S.p1_mult : R[56] * S.p1_mult;
                g6f = (endb == S.Lb) ? R[57] * S.p1_mult : R[56] * S.p1_mult;
        }
        Best B = { -KA_F, -KA_F, 0x7fffffff, 0x7fffffff };
        const int mw_wave = MW ? (int)(threadIdx.x >> 6) : 0, mw_nw = MW ? KA_NW : 1;
        for (int i = startb + lane + GL * mw_wave; valid && i <= endb; i += GL * mw_nw) {
                const KaState fi = f[i - startb], bi = b[i - startb];
                float sub = fabsf(middle - (float)i);
                sub = sub / 1000.0f;
                const int kb = (i - startb) * 8;
                if (i < endb) {
                        float c2, c5, dummy1, dummy2;
                        col_terms<KIND>(S, i + 1, c2, dummy1, dummy2);
                        col_terms<KIND>(S, i, c5, dummy1, dummy2);
                        best_consider(B, fi.a + bi.a - sub, kb + 0);
                        best_consider(B, fi.a + bi.ga + c2 - sub, kb + 1);
                        best_consider(B, fi.a + bi.gb + g3 - sub, kb + 2);
                        best_consider(B, fi.ga + bi.a + c5 - sub, kb + 3);
                        best_consider(B, fi.gb + bi.gb + g6n - sub, kb + 4);
                        best_consider(B, fi.gb + bi.a + g7 - sub, kb + 5);
                } else {
                        best_consider(B, fi.a + bi.gb + g3 - sub, kb + 2);
                        best_consider(B, fi.gb + bi.gb + g6f - sub, kb + 4);
                }
        }
        // group reduction (butterfly); every lane of the group ends with the same answer
#pragma unroll
        for (int off = GL / 2; off >= 1; off >>= 1) {
                const float omx = __shfl_xor(B.mx, off, 64);
                const float omx2 = __shfl_xor(B.mx2, off, 64);
                const int okey = __shfl_xor(B.key, off, 64);
                const int okey2 = FLIP ? __shfl_xor(B.key2, off, 64) : 0x7fffffff;
                best_merge(B, omx, omx2, okey, okey2);
        }
        if (MW) {
                // (the caller's barrier in front of this call separates the previous use of mw_* from these stores)
                if (lane == 0) { S.mw_mx[mw_wave] = B.mx; S.mw_mx2[mw_wave] = B.mx2; S.mw_key[mw_wave] = B.key; }
                __syncthreads();
                if (mw_wave != 0) return;
                B.mx = -KA_F; B.mx2 = -KA_F; B.key = 0x7fffffff; B.key2 = 0x7fffffff;
                for (int w = 0; w < mw_nw; ++w) best_merge(B, S.mw_mx[w], S.mw_mx2[w], S.mw_key[w]);
        }
        // ---- aln_continue for the group's sub-problem (its lane 0 = "leader"), wave-cooperatively: the level's
        // counters live in HBM when a cluster shares the task, and per-sub-problem atomics on five addresses
        // serialise in L2 (a level with 250 sub-problems spent 30 us there).  Leaders only compute what they
        // need; the wave adds it up and makes ONE atomic per counter.
        const bool leader = (lane == 0) && valid;
        int meet = -1, tr = -1;
        if (leader && B.key != 0x7fffffff) {
                const int ord = B.key & 7;                           // candidate order 0..5 -> codes 1,2,3,5,6,7
                meet = startb + (B.key >> 3);
                tr = ord + 1 + (ord >= 3 ? 1 : 0);
        }
        if (leader && is_top) { S.ctl->top_meet = meet; S.ctl->top_tr = tr; S.ctl->top_score = B.mx; }
        if (rec && leader && B.mx2 > -KA_F) {
                const int idx = atomicAdd(&S.ctl->nrec, 1);
                S.mrec[idx] = make_int2(sb.pad, __float_as_int(B.mx - B.mx2));
                // incremental flip trials: the sub-problem itself, the width of its subtree's key range, and what its first row
                // holds before anything below it writes (only an ancestor can have written there; the windows of other nodes are disjoint)
                if (REC && S.inc) { const KaInc I = ka_inc_view(S); I.win[idx] = sb; I.mx[idx] = make_int2(3 * kdig, S.raw[sb.starta]); }
        }
        if (FLIP && leader) {
                // the reference's meetups run one after the other in DFS order: fp32 margin sum in that order, and the
                // running number of uncertain meetups decides which of them a trial flips (round-robin)
                if (B.mx2 > -KA_F) { S.rf.msum += B.mx - B.mx2; S.rf.mcount += 1; }
                if (S.rf.thr > 0.0f && B.key2 != 0x7fffffff && B.mx2 > -KA_F) {
                        const float margin = B.mx - B.mx2;
                        if (margin < S.rf.thr) {
                                if (S.rf.trial > 0 && S.rf.counter % S.rf.stride == S.rf.trial - 1) {
                                        const int ord2 = B.key2 & 7;
                                        meet = startb + (B.key2 >> 3);
                                        tr = ord2 + 1 + (ord2 >= 3 ? 1 : 0);
                                }
                                S.rf.counter += 1;
                        }
                }
        }

        const KaState Z = { 0.0f, -KA_F, -KA_F };
        const KaState GA = { -KA_F, 0.0f, -KA_F };
        const KaState GB = { -KA_F, -KA_F, 0.0f };
        KaSub c1, c2;
        c1.starta = sb.starta; c1.startb = startb; c1.fin = sb.fin;
        c2.enda = sb.enda; c2.endb = endb; c2.bin = sb.bin;
        c1.enda = c1.starta; c1.endb = c1.startb; c1.bin = Z;          // empty unless a transition fills them in
        c2.starta = c2.enda; c2.startb = c2.endb; c2.fin = Z;
        c1.pad = rec ? sb.pad + kdig : 0; c2.pad = rec ? sb.pad + 2 * kdig : 0; c1.roff = 0; c2.roff = 0;
        if (tr > 0) {
                int* path = S.raw;
                switch (tr) {
                case 1:
                        path[mid] = meet; path[mid + 1] = meet + 1;
                        c1.enda = mid - 1; c1.endb = meet - 1; c1.bin = Z;
                        c2.starta = mid + 1; c2.startb = meet + 1; c2.fin = Z;
                        break;
                case 2:
                        path[mid] = meet;
                        c1.enda = mid - 1; c1.endb = meet - 1; c1.bin = Z;
                        c2.starta = mid; c2.startb = meet + 1; c2.fin = GA;
                        break;
                case 3:
                        path[mid] = meet;
                        c1.enda = mid - 1; c1.endb = meet - 1; c1.bin = Z;
                        c2.starta = mid + 1; c2.startb = meet; c2.fin = GB;
                        break;
                case 5:
                        path[mid + 1] = meet + 1;
                        c1.enda = mid; c1.endb = meet - 1; c1.bin = GA;
                        c2.starta = mid + 1; c2.startb = meet + 1; c2.fin = Z;
                        break;
                case 6:
                        c1.enda = mid - 1; c1.endb = meet; c1.bin = GB;
                        c2.starta = mid + 1; c2.startb = meet; c2.fin = GB;
                        break;
                default: /* 7 */
                        path[mid + 1] = meet + 1;
                        c1.enda = mid - 1; c1.endb = meet; c1.bin = GB;
                        c2.starta = mid + 1; c2.startb = meet + 1; c2.fin = Z;
                        break;
                }
        }
        const bool v1 = (tr > 0) && c1.starta < c1.enda && c1.startb < c1.endb;
        const bool v2 = (tr > 0) && c2.starta < c2.enda && c2.startb < c2.endb;
        // what this leader needs: sub-problem slots, row-buffer cells, strip items, 16-lane and 4-lane packed entries
        int need[5] = {0, 0, 0, 0, 0};
        int pr[4], pc[4];                                            // rows / columns of the (up to) four passes
        {
                const int m1 = ((c1.enda - c1.starta) / 2) + c1.starta, m2 = ((c2.enda - c2.starta) / 2) + c2.starta;
                pr[0] = m1 - c1.starta; pr[1] = c1.enda - m1; pr[2] = m2 - c2.starta; pr[3] = c2.enda - m2;
                pc[0] = pc[1] = c1.endb - c1.startb; pc[2] = pc[3] = c2.endb - c2.startb;
        }
        if (v1) { need[0] += 1; need[1] += c1.endb - c1.startb + 1; }
        if (v2) { need[0] += 1; need[1] += c2.endb - c2.startb + 1; }
        // a child small enough for one wave's LDS is ONE work item: the whole subtree below it (ka_subtree.h)
        const bool st1 = !FLIP && !rec && v1 && ka_child_is_subtree(lout, c1.enda - c1.starta, c1.endb - c1.startb);
        const bool st2 = !FLIP && !rec && v2 && ka_child_is_subtree(lout, c2.enda - c2.starta, c2.endb - c2.startb);
        if (st1) need[2] += 1;
        if (st2) need[2] += 1;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
                if (!((x < 2) ? v1 : v2) || ((x < 2) ? st1 : st2)) continue;
                if (ka_pass_is_strip(pr[x], pc[x])) need[2] += ka_strips_of(pr[x], lout.srows);
                else if (pr[x] > 8) need[3] += 1;
                else need[4] += 1;
        }
        float marg = 0.0f;
        int mc = 0;
        if (leader && B.mx2 > -KA_F) { marg = B.mx - B.mx2; mc = 1; }
        // exclusive scans over the wave (non-leaders contribute nothing)
        int off[5], tot[5];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
                int sc = need[x];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(sc, d, 64); if (wlane >= d) sc += y; }
                tot[x] = __shfl(sc, 63, 64);
                off[x] = sc - need[x];
        }
        double msum_w = (double)marg;
        int mcnt_w = mc;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { msum_w += __shfl_xor(msum_w, d, 64); mcnt_w += __shfl_xor(mcnt_w, d, 64); }
        int base[5] = {0, 0, 0, 0, 0};
        if (wlane == 0) {
                if (tot[0]) base[0] = atomicAdd(lout.nsub, tot[0]);
                if (tot[1]) base[1] = atomicAdd(lout.rowalloc, tot[1]);
                if (tot[2]) base[2] = atomicAdd(lout.nitems, tot[2]);
                if (tot[3]) base[3] = atomicAdd(lout.n16, tot[3]);
                if (tot[4]) base[4] = atomicAdd(lout.n4, tot[4]);
                if (mcnt_w) { atomicAdd(&S.lctl->msum, msum_w); atomicAdd(&S.lctl->mcount, mcnt_w); }
        }
#pragma unroll
        for (int x = 0; x < 5; ++x) base[x] = __shfl(base[x], 0, 64) + off[x];
        if (!leader || tr < 0) return;
        // the leader's own ranges, filled in the order children / passes are numbered
        int slot = base[0], row = base[1], ip = base[2], p16 = base[3], p4 = base[4];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
                KaSub& cs = ch ? c2 : c1;
                if (!(ch ? v2 : v1)) continue;
                cs.roff = row; row += cs.endb - cs.startb + 1;
                if (ch ? st2 : st1) cs.pad = KA_SUB_MARK;
                qnext[slot] = cs;
                if (ch ? st2 : st1) {
                        lout.items[ip] = make_int2(slot, KA_ITEM_SUBTREE << 16); lout.prog[ip] = 0; ++ip; ++slot;
                        continue;
                }
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                        const int nrows = pr[2 * ch + x], ncols = pc[2 * ch + x], dir = x ? KA_BWD : KA_FWD;
                        if (ka_pass_is_strip(nrows, ncols)) {
                                const int ns = ka_strips_of(nrows, lout.srows);
                                for (int k = 0; k < ns; ++k) { lout.items[ip + k] = make_int2(slot, (dir << 16) | k); lout.prog[ip + k] = 0; }
                                ip += ns;
                        } else if (nrows > 8) {
                                lout.pack16[p16++] = make_int2(slot, dir);
                        } else {
                                lout.pack4[p4++] = make_int2(slot, dir);
                        }
                }
                ++slot;
        }
}

// The whole recursion for the task described by S (all threads of the workgroup).
// Barrier over all workgroups of the task's cluster (plain __syncthreads for a single workgroup).
// Monotonic arrival counter in HBM; lane 0 releases at agent scope before arriving and acquires
// after the last arrival, the surrounding __syncthreads extend both to the whole workgroup
// (guide section 6 G16).  Bounded spin -> device watchdog.
__device__ void ka_cluster_sync(TaskShared& S)
{
        __syncthreads();
        if (S.G == 1 || S.split) return;
        if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                S.bar_phase += 1;
                const unsigned int target = S.bar_phase * (unsigned int)S.G;
                __hip_atomic_fetch_add(&S.ctl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&S.ctl->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        __builtin_amdgcn_s_sleep(4);
                        if (ka_spin_expired(S.watchdog, ++spins, 1 << 24, 6)) break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
}

// debug breadcrumbs into a host-pinned buffer (KA_TRACE=1): survives a hung kernel
#define KA_CRUMB(D_trace, slot, val) do { if (D_trace) { ((volatile int*)(D_trace))[slot] = (val); __threadfence_system(); } } while (0)

// The passes of one recursion level: its work items (strips, packed jobs) dealt to / pulled by the waves of the team.
// Q1: the kernel also carries the one-row-per-lane strip (TaskShared::srows == 64 selects it per task)
// HO: strips dealt to neighbouring waves of a workgroup hand over through LDS rings (ka_strip<.., HO>; TaskShared::ho_ok)
template <int KIND, int NRES, int NB, bool Q1 = false, bool HO = false, bool HW = false>
__device__ __forceinline__ void ka_run_items(TaskShared& S, KaCtl::Lvl* const cur, const int level, const KaSub* qc, char* lds_waves,
                                             const float* tss, long long* pslot)
{
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        {
                        const int2* items = S.items[level & 1];
                        int* prog = S.prog[level & 1];
                        const int nitems = cur->nitems;
                        const int n16 = cur->npack[0], n4 = cur->npack[1];
                        const int njobs16 = (n16 + 3) / 4, njobs4 = (n4 + 15) / 16;
                        const int2* pack16 = S.pack[level & 1][0];
                        const int2* pack4 = S.pack[level & 1][1];
                        // A wave64 VALU instruction occupies its SIMD for 4 cycles and both strips and packed jobs are
                        // almost pure VALU: two of them on one SIMD run at half speed each.  The first 8*G work items
                        // are therefore dealt out statically, spread first over the workgroups of the cluster and
                        // over waves 0..3 of each (one per SIMD), then over waves 4..7; whatever is left is pulled
                        // dynamically.  (Item i only ever waits for items < i, and every wave takes its items in
                        // increasing order, so the dealing cannot deadlock the strip pipelines.)
                        const int ntotal = nitems + njobs16 + njobs4;
                        const int Gw = __builtin_amdgcn_readfirstlane(S.Gw), member_w = __builtin_amdgcn_readfirstlane(S.member_w);
                        const int nslots = __builtin_amdgcn_readfirstlane(KA_NW * Gw);
                        // a level that keeps only waves 0 .. NW/2-1 (or NW/4-1) of every workgroup busy: its packed jobs may
                        // stage their columns in the idle waves' LDS regions too (ka_packed)
                        int nreg = 1;
                        if (ntotal <= nslots) {
                                const int per_wg = (ntotal + Gw - 1) / Gw;
                                if (per_wg <= KA_NW / 4) nreg = 4; else if (per_wg <= KA_NW / 2) nreg = 2;
                        }
                        nreg = __builtin_amdgcn_readfirstlane(nreg);
                        const int reg_stride = (KA_NW / nreg) * KA_WAVE_LDS;
                        // static dealing in contiguous blocks: workgroup m of the cluster takes items m*per .. m*per+per-1, one per
                        // wave -- the strips of one pass are consecutive items, so a strip and the strip it hands its last row to
                        // mostly sit in the same workgroup (workgroup-scope hand-over; the agent-scope one costs an L2
                        // write-back per 64 columns, and that gets slower the more the other CUs of the XCD have written)
                        const int nstatic = min(ntotal, nslots);
                        const int per = max((nstatic + Gw - 1) / Gw, 1);
                        int it = __builtin_amdgcn_readfirstlane((wave < per && member_w * per + wave < nstatic) ? member_w * per + wave : ntotal);
                        bool dealt = true;
                        // Helper mode (ka_wstrip.h): every item of the level is dealt statically, at most four per workgroup
                        // (waves 0..3, one per SIMD) -- wave w + 4 serves the strip of wave w.  The same for every workgroup of the
                        // cluster (ntotal, Gw and per are), so both ends of a hand-over between workgroups speak the same protocol.
                        const bool wmode = HW && KIND == KA_PP && KA_NW == 8 && ntotal <= nslots && per <= KA_NW / 2
                                           && __builtin_amdgcn_readfirstlane(S.hw_ok) != 0
                                           && (__builtin_amdgcn_readfirstlane(S.lvl_srows[level & 1]) == KA_STRIP_ROWS || __builtin_amdgcn_readfirstlane(S.q1_lvl) != 0);
                        // (64-row strips with helper waves only in the per-level experiment, KaTreeDev::q1_mode 4)
                        const int wsrows = __builtin_amdgcn_readfirstlane(S.lvl_srows[level & 1]);
                        if (HW && KIND == KA_PP && wmode && wave >= KA_NW / 2) {
                                const int sw = wave - KA_NW / 2;                       // the strip wave this one helps
                                const int hit = __builtin_amdgcn_readfirstlane((sw < per && member_w * per + sw < nstatic) ? member_w * per + sw : ntotal);
                                if (hit < nitems) {
                                        const int2 item = items[hit];
                                        const int subi = __builtin_amdgcn_readfirstlane(item.x);
                                        const int dk = __builtin_amdgcn_readfirstlane(item.y);
                                        const KaSub* sp = qc + subi;
                                        const int dir = dk >> 16, k = dk & 0xffff;
                                        if (dir != KA_ITEM_SUBTREE) {
                                                const int sa = __builtin_amdgcn_readfirstlane(sp->starta);
                                                const int ea = __builtin_amdgcn_readfirstlane(sp->enda);
                                                const int sbb = __builtin_amdgcn_readfirstlane(sp->startb);
                                                const int eb = __builtin_amdgcn_readfirstlane(sp->endb);
                                                const int roff = __builtin_amdgcn_readfirstlane(sp->roff);
                                                const float ja = ka_uniform_f(dir == KA_FWD ? sp->fin.a : sp->bin.a);
                                                const float jga = ka_uniform_f(dir == KA_FWD ? sp->fin.ga : sp->bin.ga);
                                                const float jgb = ka_uniform_f(dir == KA_FWD ? sp->fin.gb : sp->bin.gb);
                                                const int mid_ = ((ea - sa) / 2) + sa;
                                                const int nrows_ = (dir == KA_FWD) ? mid_ - sa : ea - mid_;
                                                const int ns = ka_strips_of(nrows_, wsrows);
                                                const bool prod_local = k > 0 && (hit - 1) / per == member_w;
                                                const bool cons_local = k + 1 < ns && hit + 1 < nstatic && (hit + 1) / per == member_w;
                                                if (nrows_ > 0) {
                                                        KaWHelperArgs ha;
                                                        ha.p2 = S.p2; ha.rows = (dir == KA_FWD ? S.fbuf : S.bbuf) + roff; ha.xrows = (dir == KA_FWD ? S.xfbuf : S.xbbuf) + roff;
                                                        ha.prog = prog + (hit - k); ha.watchdog = S.watchdog;
                                                        ha.m2 = S.p2_mult; ha.inj_a = ja; ha.inj_ga = jga; ha.inj_gb = jgb; ha.Lb = S.Lb;
                                                        ha.starta = sa; ha.enda = ea; ha.startb = sbb; ha.endb = eb; ha.dir = dir; ha.k = k; ha.ns = ns;
                                                        ha.slds_u = (unsigned)(unsigned long long)(lds_waves + sw * KA_WAVE_LDS);
                                                        ha.hlds_u = (unsigned)(unsigned long long)(lds_waves + wave * KA_WAVE_LDS);
                                                        ha.ctl_u = (unsigned)(unsigned long long)(lds_waves - KA_LDS_HO_BACK);
                                                        ha.w = sw; ha.in_mode = k == 0 ? 0 : (prod_local ? 1 : 2); ha.out_local = cons_local ? 1 : 0;
                                                        if (Q1 && wsrows == KA_STRIP1_ROWS) ka_whelper<NRES, 1>(ha); else ka_whelper<NRES, 2>(ha);
                                                }
                                        }
                                }
                                return;
                        }
                        while (true) {
                                // One lane takes the next item, then it is broadcast.  The puller lane is
                                // compared through an opaque copy: with a plain `lane == 0` the optimiser
                                // threads this test with the `lane == 0` regions inside ka_strip, splits the
                                // loop per lane set and runs readfirstlane without lane 0 (observed: lanes
                                // 1..63 spinning on item 0 forever).
                                if (!dealt) {
                                        if (ntotal <= nslots) break;
                                        int puller = lane;
                                        asm volatile("" : "+v"(puller));
                                        int x = 0;
                                        if (puller == 0) x = atomicAdd(&cur->next_item, 1);
                                        it = nslots + __builtin_amdgcn_readfirstlane(x);
                                }
                                dealt = false;
                                if (it >= ntotal) break;
        #ifdef KA_PROF
                                if (pslot && lane == 0) { if (pslot[1] == 0) pslot[1] = __builtin_amdgcn_s_memtime(); pslot[5] += 1; }
        #endif
                                if (it >= nitems + njobs16) {
                                        ka_packed<KIND, NRES, 4, NB>(S, qc, pack4, n4, it - nitems - njobs16, lane, tss, KIND != KA_SS ? lds_waves + wave * KA_WAVE_LDS : nullptr, nreg, reg_stride);
                                        continue;
                                }
                                if (it >= nitems) {
                                        ka_packed<KIND, NRES, 16, NB>(S, qc, pack16, n16, it - nitems, lane, tss, KIND != KA_SS ? lds_waves + wave * KA_WAVE_LDS : nullptr, nreg, reg_stride);
                                        continue;
                                }
                                // everything about the item is wave-uniform: keep it in SGPRs
                                const int2 item = items[it];
                                const int subi = __builtin_amdgcn_readfirstlane(item.x);
                                const int dk = __builtin_amdgcn_readfirstlane(item.y);
                                const KaSub* sp = qc + subi;
                                const int dir = dk >> 16, k = dk & 0xffff;
                                if (dir == KA_ITEM_SUBTREE) {
                                        ka_subtree<KIND, NRES>(S, *sp, lane, S.sub_base + wave * S.sub_stride, tss);
                                        continue;
                                }
                                const int sa = __builtin_amdgcn_readfirstlane(sp->starta);
                                const int ea = __builtin_amdgcn_readfirstlane(sp->enda);
                                const int sbb = __builtin_amdgcn_readfirstlane(sp->startb);
                                const int eb = __builtin_amdgcn_readfirstlane(sp->endb);
                                const int roff = __builtin_amdgcn_readfirstlane(sp->roff);
                                const float ja = ka_uniform_f(dir == KA_FWD ? sp->fin.a : sp->bin.a);
                                const float jga = ka_uniform_f(dir == KA_FWD ? sp->fin.ga : sp->bin.ga);
                                const float jgb = ka_uniform_f(dir == KA_FWD ? sp->fin.gb : sp->bin.gb);
                                const int mid_ = ((ea - sa) / 2) + sa;
                                const int srows = wsrows;
                                const int ns = ka_strips_of(dir == KA_FWD ? mid_ - sa : ea - mid_, srows);
                                const bool st_me = it < nstatic;
                                const bool prod_local = k > 0 && st_me && (it - 1) / per == member_w;
                                const bool cons_local = k + 1 < ns && st_me && it + 1 < nstatic && (it + 1) / per == member_w;
                                // LDS hand-over: only on levels where every wave has at most ONE item (nothing is pulled after a strip, so
                                // a producer's LDS region stays as it is until the level's barrier); the neighbour strip runs on the
                                // neighbour wave by the static dealing above (item it +- 1 <-> wave +- 1 of this workgroup)
                                const bool ho_lvl = HO && ntotal <= nslots && __builtin_amdgcn_readfirstlane(S.ho_ok) != 0;
                                const bool in_lds = ho_lvl && prod_local && wave > 0;
                                const bool out_lds = ho_lvl && cons_local && wave + 1 < KA_NW;
                                int* const ho_ctl_w = (int*)(lds_waves - KA_LDS_HO_BACK) + wave;
                                if constexpr (HW && KIND == KA_PP) {
                                        if (wmode && (dir == KA_FWD ? mid_ - sa : ea - mid_) > 0) {
                                                const unsigned ctl_u = (unsigned)(unsigned long long)(lds_waves - KA_LDS_HO_BACK);
                                                // the row above: the out ring and step count of the wave before this one, or the in ring my helper fills
                                                const unsigned in_ring_u = prod_local ? (unsigned)(unsigned long long)(lds_waves + (wave - 1) * KA_WAVE_LDS + KA_HO_RING)
                                                                                      : (unsigned)(unsigned long long)(lds_waves + (wave + KA_NW / 2) * KA_WAVE_LDS + KA_W_INRING);
                                                const unsigned in_word_u = ctl_u + 4 * (prod_local ? KA_W_TPUB(wave - 1) : KA_W_IN(wave));
                                                KaWStripArgs wa;
                                                wa.p1 = S.p1; wa.ent = S.ent; wa.watchdog = S.watchdog; wa.pslot = pslot; wa.m1 = S.p1_mult; wa.Lb = S.Lb; wa.prio = (S.hw_ok >> 4) & 3;
                                                wa.starta = sa; wa.enda = ea; wa.startb = sbb; wa.endb = eb; wa.dir = dir; wa.k = k;
                                                wa.wlds_u = (unsigned)(unsigned long long)(lds_waves + wave * KA_WAVE_LDS);
                                                wa.in_ring_u = in_ring_u; wa.in_word_u = in_word_u; wa.in_bias = prod_local ? 63 : 0; wa.ctl_u = ctl_u; wa.w = wave;
                                                if (Q1 && srows == KA_STRIP1_ROWS) ka_wstrip<NRES, NB, 1>(wa); else ka_wstrip<NRES, NB, 2>(wa);
                                                continue;
                                        }
                                }
                                if (Q1 && srows == KA_STRIP1_ROWS)
                                        ka_strip<KIND, NRES, NB, 1, HO>(S, sa, ea, sbb, eb, ja, jga, jgb, dir, k,
                                                             ka_uniform_ptr((dir == KA_FWD ? S.fbuf : S.bbuf) + roff), ka_uniform_ptr(prog + (it - k)), lane,
                                                             lds_waves + wave * KA_WAVE_LDS, tss, Gw > 1 && !prod_local, Gw > 1 && !(cons_local || k + 1 == ns), pslot,
                                                             in_lds, out_lds, ho_ctl_w);
                                else
                                        ka_strip<KIND, NRES, NB, 2, HO>(S, sa, ea, sbb, eb, ja, jga, jgb, dir, k,
                                                             ka_uniform_ptr((dir == KA_FWD ? S.fbuf : S.bbuf) + roff), ka_uniform_ptr(prog + (it - k)), lane,
                                                             lds_waves + wave * KA_WAVE_LDS, tss, Gw > 1 && !prod_local, Gw > 1 && !(cons_local || k + 1 == ns), pslot,
                                                             in_lds, out_lds, ho_ctl_w);
                        }
        }
}

__device__ const int ka_pow3[20] = { 1, 3, 9, 27, 81, 243, 729, 2187, 6561, 19683, 59049, 177147, 531441, 1594323, 4782969, 14348907,
                                     43046721, 129140163, 387420489, 1162261467 };
#define KA_REC_DEPTH 19                                              // recursion levels the keys of ka_meetup<.., REC> can tell apart

template <int KIND, int NRES, int NB, bool REC = false, bool Q1 = false, bool HO = false, bool HW = false>
__device__ __forceinline__ void ka_hirschberg(TaskShared& S, float* dbg_rows, char* lds_waves, const float* tss, int* trace)
{
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const int g = max(S.La, S.Lb) + 2;
        const bool lead = (S.member == 0);
        // HO: the waves' hand-over control words (columns written / columns read, ka_strip) go back to zero while no strip runs:
        // before the first level and at the start of every meetup phase; the barrier that follows orders it before the next strips
        auto ho_clear = [&]() {
                if (((HO && S.ho_ok) || (HW && S.hw_ok)) && tid < 16) ((int*)(lds_waves - KA_LDS_HO_BACK))[tid] = 0;
        };
        ho_clear();
        if (lead) for (int i = tid; i < g; i += KA_NT) S.raw[i] = -1;  // init_alnmem, aln_setup.c:33-36
        if (tid == 0) { S.lctl = S.ctl; S.Gw = S.G; S.member_w = S.member; S.split = 0; S.lvl_srows[0] = ka_level_srows(S, 0); S.lvl_srows[1] = ka_level_srows(S, 1); }
        if (lead && tid == 0) {
                KaSub root;
                const KaState Z = { 0.0f, -KA_F, -KA_F };
                root.starta = 0; root.enda = S.La; root.startb = 0; root.endb = S.Lb;
                root.fin = Z; root.bin = Z; root.roff = 0; root.pad = 0;
                S.q[0][0] = root;
                for (int par = 0; par < 2; ++par) {
                        S.ctl->lvl[par].nsub = 0; S.ctl->lvl[par].rowalloc = 0; S.ctl->lvl[par].nitems = 0;
                        S.ctl->lvl[par].next_item = 0; S.ctl->lvl[par].next_job = 0; S.ctl->lvl[par].npack[0] = 0; S.ctl->lvl[par].npack[1] = 0;
                }
                S.ctl->lvl[0].nsub = (S.La > 0 && S.Lb > 0) ? 1 : 0;
                S.ctl->lvl[0].rowalloc = S.Lb + 1;
                S.lctl = S.ctl;
                if (S.ctl->lvl[0].nsub) ka_emit_items(ka_level_out(S, 0, false), 0, 0, S.La, S.Lb, false);   // (the top level keeps its rows in HBM: records, tests)
                S.ctl->msum = 0.0; S.ctl->mcount = 0;
                S.ctl->top_meet = -1; S.ctl->top_tr = -1; S.ctl->top_score = 0.0f;
                S.t_pass = 0; S.t_meet = 0; S.n_levels = 0;
        }
        ka_cluster_sync(S);
        int level = 0;
        bool did_split = false;                                       // (a register copy of S.split: uniform over the workgroup)
        while (true) {
                // ---- split the cluster (see TaskShared::Gw): from here on every workgroup recurses on its own ----
                if (S.G > 1 && !did_split && level >= 1) {
                        const int nshared = S.ctl->lvl[level & 1].nsub;
                        // every member sees the same numbers here (the barrier that ended the previous level published them)
                        if (nshared >= S.G || (S.La >> (level + 1)) <= S.srows / 2) {
                                did_split = true;
                                __syncthreads();
                                if (tid == 0) {
                                        const KaSub* shared_q = S.q[level & 1];
                                        S.q[0] = S.priv.q[0]; S.q[1] = S.priv.q[1];
                                        S.items[0] = S.priv.items[0]; S.items[1] = S.priv.items[1];
                                        S.prog[0] = S.priv.prog[0]; S.prog[1] = S.priv.prog[1];
                                        S.pack[0][0] = S.priv.pack[0][0]; S.pack[0][1] = S.priv.pack[0][1];
                                        S.pack[1][0] = S.priv.pack[1][0]; S.pack[1][1] = S.priv.pack[1][1];
                                        S.fbuf = S.priv.f; S.bbuf = S.priv.b;
                                        S.lctl = &S.ctl_lds;
                                        for (int par = 0; par < 2; ++par) {
                                                KaCtl::Lvl& L = S.ctl_lds.lvl[par];
                                                L.nsub = 0; L.rowalloc = 0; L.nitems = 0; L.next_item = 0; L.next_job = 0; L.npack[0] = 0; L.npack[1] = 0;
                                        }
                                        S.ctl_lds.msum = 0.0; S.ctl_lds.mcount = 0;
                                        KaCtl::Lvl& L = S.ctl_lds.lvl[level & 1];
                                        KaLevelOut lo = ka_level_out(S, level & 1, false);
                                        lo.srows = S.srows;                      // (what ka_level_srows says once S.split is set, below)
                                        // this member's share: every G-th sub-problem of the level (they are independent
                                        // subtrees of the recursion; their order in the queue is arbitrary)
                                        for (int k = S.member; k < nshared; k += S.G) {
                                                KaSub sb = shared_q[k];
                                                sb.roff = L.rowalloc;
                                                L.rowalloc += sb.endb - sb.startb + 1;
                                                S.q[level & 1][L.nsub] = sb;
                                                ka_emit_items(lo, L.nsub, sb.starta, sb.enda, sb.endb - sb.startb, sb.pad == KA_SUB_MARK);
                                                L.nsub += 1;
                                        }
                                        S.Gw = 1; S.member_w = 0; S.split = 1;
                                        S.lvl_srows[0] = ka_level_srows(S, level); S.lvl_srows[1] = S.lvl_srows[0];     // (split: the task's own strip shape from here on)
                                }
                                __syncthreads();
                        }
                }
                const bool lead_w = (S.member_w == 0);
                if (tid == 0) S.lvl_srows[(level + 1) & 1] = ka_level_srows(S, level + 1);    // (read by this level's meetups, behind the barrier that ends its passes)
                KaCtl::Lvl* const cur = &S.lctl->lvl[level & 1];
                const int ncur = cur->nsub;
                if (ncur == 0) break;
                KaSub* qc = S.q[level & 1];
                KaSub* qn = S.q[(level + 1) & 1];
                if (lead_w && tid == 0 && level > 0) {
                        // the other parity was consumed by level-1 and is idle until this level's meetups
                        // (which start after the barrier below): reset it now
                        KaCtl::Lvl* const nxt = &S.lctl->lvl[(level + 1) & 1];
                        nxt->nsub = 0; nxt->rowalloc = 0; nxt->nitems = 0; nxt->next_item = 0; nxt->next_job = 0; nxt->npack[0] = 0; nxt->npack[1] = 0;
                }
                const long long tp0 = __builtin_amdgcn_s_memtime();
                long long* pslot = nullptr;
#ifdef KA_PROF
                if (S.prof && lead && level < 4) { pslot = S.prof + (level * 8 + wave) * 8; if (lane == 0) { pslot[0] = tp0; pslot[1] = 0; pslot[2] = 0; pslot[3] = 0; pslot[4] = 0; pslot[5] = 0; pslot[6] = 0; pslot[7] = 0; if (level < 4) for (int x = 0; x < 8; ++x) pslot[256 + x] = 0; } }
#endif
                ka_run_items<KIND, NRES, NB, Q1, HO, HW>(S, cur, level, qc, lds_waves, tss, pslot);
#ifdef KA_PROF
                if (pslot && lane == 0) pslot[3] = __builtin_amdgcn_s_memtime();
#endif
                ka_cluster_sync(S);
                ho_clear();
#ifdef KA_PROF
                if (pslot && lane == 0) pslot[4] = __builtin_amdgcn_s_memtime();
#endif
                if (tid == 0 && blockIdx.x == 0) KA_CRUMB(trace, 3, 1000 * level + 1);
                const long long tp1 = __builtin_amdgcn_s_memtime();
                if (level == 0 && dbg_rows && lead) {
                        // tests only: keep the top-level rows f[0..Lb], b[0..Lb]
                        const int n = 3 * (S.Lb + 1);
                        const float* f = (const float*)S.fbuf;
                        const float* b = (const float*)S.bbuf;
#ifdef KA_DBG_SC1
                        for (int i = tid; i < n; i += KA_NT) { dbg_rows[i] = __hip_atomic_load((float*)f + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dbg_rows[n + i] = __hip_atomic_load((float*)b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
                        for (int i = tid; i < n; i += KA_NT) { dbg_rows[i] = f[i]; dbg_rows[n + i] = b[i]; }
#endif
                }
                {
                        const KaLevelOut lout = ka_level_out(S, (level + 1) & 1, true);
                        const int est_cols = S.Lb >> level;          // typical columns per sub-problem at this depth
                        const int kdig = (REC || S.rec_on) ? ka_pow3[max(KA_REC_DEPTH - 2 - level, 0)] : 0;
                        if (est_cols > 128 && ncur <= 2 && S.mw_ok) {
                                // one or two sub-problems with thousands of columns: the leading workgroup's waves share each scan
                                // (every wave of it takes part in both barriers of a round; the other members have nothing to do)
                                if (S.member_w == 0)
                                        for (int k = 0; k < ncur; ++k) {
                                                __syncthreads();
                                                ka_meetup<KIND, 64, false, REC, true>(S, qc, k, ncur, qn, lout, lane, level == 0, kdig);
                                        }
                        } else if (est_cols > 48) {
                                for (int k = S.member_w * KA_NW + wave; k < ncur; k += KA_NW * S.Gw)
                                        ka_meetup<KIND, 64, false, REC>(S, qc, k, ncur, qn, lout, lane, level == 0, kdig);
                        } else if (est_cols > 6) {
                                for (int k = (S.member_w * KA_NW + wave) * 4; k < ncur; k += KA_NW * S.Gw * 4)
                                        ka_meetup<KIND, 16, false, REC>(S, qc, k, ncur, qn, lout, lane, level == 0, kdig);
                        } else {
                                for (int k = (S.member_w * KA_NW + wave) * 16; k < ncur; k += KA_NW * S.Gw * 16)
                                        ka_meetup<KIND, 4, false, REC>(S, qc, k, ncur, qn, lout, lane, level == 0, kdig);
                        }
                }
                ka_cluster_sync(S);
                if (lead && tid == 0) {
                        const long long tp2 = __builtin_amdgcn_s_memtime();
                        S.t_pass += tp1 - tp0; S.t_meet += tp2 - tp1; S.n_levels = level + 1;
                        if (level < 16) { S.lvl_n[level] = ncur; S.lvl_pass[level] = (int)(tp1 - tp0); S.lvl_meet[level] = (int)(tp2 - tp1); }
                }
                ++level;
        }
        if (S.split) {
                // the members of a split cluster meet again: margins into the task's block, then ONE cluster barrier
                // (agent-scope release / acquire: the raw path rows every member wrote become visible to the first one)
                __syncthreads();
                if (tid == 0) {
                        if (S.ctl_lds.mcount) { atomicAdd(&S.ctl->msum, S.ctl_lds.msum); atomicAdd(&S.ctl->mcount, S.ctl_lds.mcount); }
                        S.split = 0; S.lctl = S.ctl;
                }
                ka_cluster_sync(S);
        }
}

// ------------------------------------------------------------------------------------------
// Depth-first recursion, small subtrees: ONE wave, no workgroup barriers, queues in LDS (ka_wave_dfs).
//
// The passes of a sub-problem only need its window, i.e. its parent's decision; only its own decision (which of the two
// best candidates it takes) needs the flip counter, i.e. everything before it in recursion order.  So when a node is
// decided, the passes AND the candidate scans of both its children run at once (one packed job of four 16-lane slots,
// one scan with 32 lanes per child); the left child is decided next, the right child's candidates wait on the stack
// until the left subtree is through.  A round of passes per decided node instead of per node, no __syncthreads, no
// work lists in HBM.
// ------------------------------------------------------------------------------------------
#define KA_WDFS_ROWS 64                                              // subtrees of at most this many rows run wave-locally
#define KA_LDFS_ROWS 128                                             // ... when they run in LDS (ka_subtree_dfs)
struct KaWdfsEntry { KaSub sub; float mx, mx2; int key, key2; };

// the meetup candidates of one sub-problem, scanned by GL lanes: same candidates, same order, same arithmetic as ka_meetup
template <int KIND, int GL>
__device__ __forceinline__ Best ka_meet_scan(const TaskShared& S, const KaSub& sb, const int lane, const bool valid)
{
        const int startb = sb.startb, endb = sb.endb;
        const int mid = ((sb.enda - sb.starta) / 2) + sb.starta;
        const KaState* f = S.fbuf + sb.roff;
        const KaState* b = S.bbuf + sb.roff;
        const float middle = (float)(endb - startb) / 2.0f + (float)startb;
        const int rrec = mid + 1;
        float g3, g7, g6n, g6f;
        if (KIND == KA_SS) {
                g3 = -S.gpo; g7 = -S.gpo;
                g6n = (startb == 0) ? -S.tgpe : -S.gpe;
                g6f = (endb == S.Lb) ? -S.tgpe : -S.gpe;
        } else {
                const float* R = S.p1 + ((long long)rrec << 6);
                g3 = R[55] * S.p1_mult; g7 = R[55 - 64] * S.p1_mult;
                g6n = (startb == 0) ? R[57] * S.p1_mult : R[56] * S.p1_mult;
                g6f = (endb == S.Lb) ? R[57] * S.p1_mult : R[56] * S.p1_mult;
        }
        Best B = { -KA_F, -KA_F, 0x7fffffff, 0x7fffffff };
        for (int i = startb + lane; valid && i <= endb; i += GL) {
                const KaState fi = f[i - startb], bi = b[i - startb];
                float sub = fabsf(middle - (float)i);
                sub = sub / 1000.0f;
                const int kb = (i - startb) * 8;
                if (i < endb) {
                        float c2, c5, dummy1, dummy2;
                        col_terms<KIND>(S, i + 1, c2, dummy1, dummy2);
                        col_terms<KIND>(S, i, c5, dummy1, dummy2);
                        best_consider(B, fi.a + bi.a - sub, kb + 0);
                        best_consider(B, fi.a + bi.ga + c2 - sub, kb + 1);
                        best_consider(B, fi.a + bi.gb + g3 - sub, kb + 2);
                        best_consider(B, fi.ga + bi.a + c5 - sub, kb + 3);
                        best_consider(B, fi.gb + bi.gb + g6n - sub, kb + 4);
                        best_consider(B, fi.gb + bi.a + g7 - sub, kb + 5);
                } else {
                        best_consider(B, fi.a + bi.gb + g3 - sub, kb + 2);
                        best_consider(B, fi.gb + bi.gb + g6f - sub, kb + 4);
                }
        }
#pragma unroll
        for (int off = GL / 2; off >= 1; off >>= 1) {
                const float omx = __shfl_xor(B.mx, off, 64);
                const float omx2 = __shfl_xor(B.mx2, off, 64);
                const int okey = __shfl_xor(B.key, off, 64);
                const int okey2 = __shfl_xor(B.key2, off, 64);
                best_merge(B, omx, omx2, okey, okey2);
        }
        return B;
}

// The decision of one sub-problem (one lane): margin into the trial's running sum, the flip rule (aln_seqseq.c:376-414),
// the raw path entries and the two child windows (aln_controller.c:194-436).  Returns the number of non-empty children
// (c[0] is the one the recursion enters first).
__device__ __forceinline__ int ka_dfs_decide(TaskShared& S, const KaSub& sb, const Best& B, const bool is_top, KaSub* c)
{
        const int startb = sb.startb, endb = sb.endb;
        const int mid = ((sb.enda - sb.starta) / 2) + sb.starta;
        int meet = -1, tr = -1;
        if (B.key != 0x7fffffff) {
                const int ord = B.key & 7;
                meet = startb + (B.key >> 3);
                tr = ord + 1 + (ord >= 3 ? 1 : 0);
        }
        if (is_top) { S.ctl->top_meet = meet; S.ctl->top_tr = tr; S.ctl->top_score = B.mx; }
        if (B.mx2 > -KA_F) {
                if (S.mlog && S.rf.mcount < S.mlog_cap) S.mlog[S.rf.mcount] = B.mx - B.mx2;      // aln_seqseq.c:378-380
                S.rf.msum += B.mx - B.mx2; S.rf.mcount += 1;
        }
        if (S.rf.thr > 0.0f && B.key2 != 0x7fffffff && B.mx2 > -KA_F) {
                const float margin = B.mx - B.mx2;
                if (margin < S.rf.thr) {
                        if (S.rf.trial > 0 && S.rf.counter % S.rf.stride == S.rf.trial - 1) {
                                const int ord2 = B.key2 & 7;
                                meet = startb + (B.key2 >> 3);
                                tr = ord2 + 1 + (ord2 >= 3 ? 1 : 0);
                        }
                        S.rf.counter += 1;
                }
        }
        if (tr <= 0) return 0;
        const KaState Z = { 0.0f, -KA_F, -KA_F };
        const KaState GA = { -KA_F, 0.0f, -KA_F };
        const KaState GB = { -KA_F, -KA_F, 0.0f };
        KaSub c1, c2;
        c1.starta = sb.starta; c1.startb = startb; c1.fin = sb.fin;
        c2.enda = sb.enda; c2.endb = endb; c2.bin = sb.bin;
        c1.enda = c1.starta; c1.endb = c1.startb; c1.bin = Z;
        c2.starta = c2.enda; c2.startb = c2.endb; c2.fin = Z;
        c1.pad = 0; c2.pad = 0; c1.roff = 0; c2.roff = 0;
        int* path = S.raw;
        switch (tr) {
        case 1:
                path[mid] = meet; path[mid + 1] = meet + 1;
                c1.enda = mid - 1; c1.endb = meet - 1; c1.bin = Z;
                c2.starta = mid + 1; c2.startb = meet + 1; c2.fin = Z;
                break;
        case 2:
                path[mid] = meet;
                c1.enda = mid - 1; c1.endb = meet - 1; c1.bin = Z;
                c2.starta = mid; c2.startb = meet + 1; c2.fin = GA;
                break;
        case 3:
                path[mid] = meet;
                c1.enda = mid - 1; c1.endb = meet - 1; c1.bin = Z;
                c2.starta = mid + 1; c2.startb = meet; c2.fin = GB;
                break;
        case 5:
                path[mid + 1] = meet + 1;
                c1.enda = mid; c1.endb = meet - 1; c1.bin = GA;
                c2.starta = mid + 1; c2.startb = meet + 1; c2.fin = Z;
                break;
        case 6:
                c1.enda = mid - 1; c1.endb = meet; c1.bin = GB;
                c2.starta = mid + 1; c2.startb = meet; c2.fin = GB;
                break;
        default: /* 7 */
                path[mid + 1] = meet + 1;
                c1.enda = mid - 1; c1.endb = meet; c1.bin = GB;
                c2.starta = mid + 1; c2.startb = meet + 1; c2.fin = Z;
                break;
        }
        int n = 0;
        if (c1.starta < c1.enda && c1.startb < c1.endb) c[n++] = c1;
        if (c2.starta < c2.enda && c2.startb < c2.endb) c[n++] = c2;
        return n;
}

// The whole subtree below `root` (at most KA_WDFS_ROWS rows), depth first, by the calling wave.  `area`: LDS of an idle
// wave (stack of KaWdfsEntry, the sub-problems in flight, their pack list); wlds: staging regions for ka_packed.
template <int KIND, int NRES, int NB>
__device__ __forceinline__ void ka_wave_dfs(TaskShared& S, const KaSub root, const Best rootB, const bool root_is_top, const int lane,
                                            char* wlds, const int nreg, const int reg_stride, char* area, const float* tss)
{
        KaWdfsEntry* stack = (KaWdfsEntry*)area;                     // <= 2 * log2(rows) + 2 entries
        KaSub* fly = (KaSub*)(area + 32 * sizeof(KaWdfsEntry));       // the (up to two) sub-problems whose passes run
        int2* pack = (int2*)(fly + 2);
        int* ctl = (int*)(pack + 4);                                  // [0] stack height, [1] children of the last decision
        // the root arrives with its candidates (its passes ran with its sibling's)
        if (lane == 0) {
                KaWdfsEntry e; e.sub = root; e.mx = rootB.mx; e.mx2 = rootB.mx2; e.key = rootB.key; e.key2 = rootB.key2;
                stack[0] = e; ctl[0] = 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        bool top = root_is_top;
        while (true) {
                const int h = ((volatile int*)ctl)[0];
                if (h <= 0) break;
                // decide the node on top of the stack
                if (lane == 0) {
                        const KaWdfsEntry e = stack[h - 1];
                        const Best B = { e.mx, e.mx2, e.key, e.key2 };
                        KaSub c[2];
                        const int n = ka_dfs_decide(S, e.sub, B, top, c);
                        int row = 0;
                        for (int k = 0; k < n; ++k) {
                                c[k].roff = row; row += c[k].endb - c[k].startb + 1;
                                fly[k] = c[k];
                                pack[2 * k] = make_int2(k, KA_FWD); pack[2 * k + 1] = make_int2(k, KA_BWD);
                        }
                        ctl[0] = h - 1; ctl[1] = n;
                }
                top = false;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int n = ((volatile int*)ctl)[1];
                if (n == 0) continue;
                // passes of the children (all of them in one job), then their candidates, 32 lanes per child
                ka_packed<KIND, NRES, 16, NB>(S, fly, pack, 2 * n, 0, lane, tss, KIND != KA_SS ? wlds : nullptr, nreg, reg_stride);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                {
                        const int g = lane >> 5;
                        const bool valid = g < n;
                        const KaSub cs = fly[valid ? g : 0];
                        const Best B = ka_meet_scan<KIND, 32>(S, cs, lane & 31, valid);
                        // the child entered first (index 0) must end on top: push the second one first
                        if ((lane & 31) == 0 && valid) {
                                const int hh = ((volatile int*)ctl)[0];
                                KaWdfsEntry e; e.sub = cs; e.mx = B.mx; e.mx2 = B.mx2; e.key = B.key; e.key2 = B.key2;
                                stack[hh + (n - 1 - g)] = e;
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) ctl[0] = ((volatile int*)ctl)[0] + n;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
}

// ------------------------------------------------------------------------------------------
// Depth-first Hirschberg recursion for refinement trials (aln_refine.c:93-346).  A trial flips the n-th uncertain
// meetup in DFS order, and a flip changes the sub-problems below it: the number of uncertain meetups in the whole left
// subtree decides what happens in the right one, so the sub-problems of a trial are inherently sequential (as in the
// reference, aln_controller.c: child 1 completely before child 2).  But only the DECISIONS are: the passes of a
// sub-problem need nothing but its window.  One decision per iteration (thread 0: the flip rule, the fp32 margin sum
// of S.rf, the children's windows), then the passes of BOTH children as the usual work items (strips pipelined over
// the waves, packed jobs) and their candidate scans (one wave each); both go on the stack with their candidates, the
// one the recursion enters first on top.  Subtrees of at most KA_WDFS_ROWS rows are handed to one wave (ka_wave_dfs).
// The stack is S.q[0] (+ candidates), the sub-problems in flight are S.q[1][0..1].
// ------------------------------------------------------------------------------------------
// seed != nullptr: not a whole trial but the subtree below *seed (ka_trial_incremental) -- the raw path, the trial's counters
// and its margin log are the caller's; the seed's passes run alone like the root's.
template <int KIND, int NRES, int NB>
__device__ __forceinline__ void ka_hirschberg_dfs(TaskShared& S, char* lds_waves, const float* tss, const bool first_trial, const KaSub* seed = nullptr)
{
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const int g = max(S.La, S.Lb) + 2;
        // candidates of the sub-problems on the stack (S.q[0]): four words each, in a work list the depth-first order never fills
        int4* const cand = (int4*)S.pack[1][0];
        if (!seed) for (int i = tid; i < g; i += KA_NT) S.raw[i] = -1;    // init_alnmem / the re-initialisation of refine_edge (:206-215)
        if (tid == 0) {
                KaSub root;
                const KaState Z = { 0.0f, -KA_F, -KA_F };
                root.starta = 0; root.enda = S.La; root.startb = 0; root.endb = S.Lb;
                root.fin = Z; root.bin = Z; root.roff = 0; root.pad = 0;
                if (seed) { root = *seed; root.roff = 0; root.pad = 0; }
                S.lctl = S.ctl; S.Gw = 1; S.member_w = 0; S.split = 0;
                S.dfs_top = 0;
                if (!seed) {
                        S.rf.msum = 0.0f; S.rf.mcount = 0; S.rf.counter = 0;
                        S.ctl->msum = 0.0; S.ctl->mcount = 0;
                }
                if (first_trial) { S.ctl->top_meet = -1; S.ctl->top_tr = -1; S.ctl->top_score = 0.0f; }
                // the root's passes run alone
                S.dfs_valid = 0;
                if (root.starta < root.enda && root.startb < root.endb) {
                        S.q[1][0] = root;
                        for (int par = 0; par < 2; ++par) {
                                KaCtl::Lvl& L = S.ctl->lvl[par];
                                L.nsub = 0; L.rowalloc = 0; L.nitems = 0; L.next_item = 0; L.next_job = 0; L.npack[0] = 0; L.npack[1] = 0;
                        }
                        S.ctl->lvl[0].nsub = 1;
                        S.ctl->lvl[0].rowalloc = root.endb - root.startb + 1;
                        ka_emit_items(ka_level_out(S, 0, false), 0, root.starta, root.enda, root.endb - root.startb);
                        S.dfs_valid = 1;
                }
        }
        __syncthreads();
        bool at_root = (seed == nullptr);
        while (true) {
                // ---- the passes and candidate scans of the sub-problems in flight (S.q[1][0 .. n-1]: a decided node's children) ----
                const int n = S.dfs_valid;
                if (n > 0) {
                        const KaSub* qc = S.q[1];
                        ka_run_items<KIND, NRES, NB>(S, &S.ctl->lvl[0], 0, qc, lds_waves, tss, nullptr);
                        __syncthreads();
                        if (wave < n) {
                                const KaSub cs = qc[wave];
                                const Best B = ka_meet_scan<KIND, 64>(S, cs, lane, true);
                                // the child the recursion enters first (index 0) ends on top of the stack
                                if (lane == 0) {
                                        const int pos = S.dfs_top + (n - 1 - wave);
                                        S.q[0][pos] = cs;
                                        cand[pos] = make_int4(__float_as_int(B.mx), __float_as_int(B.mx2), B.key, B.key2);
                                }
                        }
                        __syncthreads();
                }
                // ---- the decision of the node on top of the stack ----
                if (tid == 0) {
                        S.dfs_top += n;
                        S.dfs_valid = -1;                             // stack empty: the trial is complete
                        if (S.dfs_top > 0) {
                                const int pos = --S.dfs_top;
                                const KaSub cur = S.q[0][pos];
                                const int4 cb = cand[pos];
                                // (the LDS walk takes windows of up to 128 rows: the passes of the children have at most 64)
                                const int cr = cur.enda - cur.starta, cc = cur.endb - cur.startb;
                                const bool lds_walk = NB == 0 && (S.dbgskip & 2) == 0 && cr >= 1 && cr <= KA_LDFS_ROWS && cc >= 1 && cc < 4096 &&
                                                      ka_sub_bytes(KIND, NRES, cr, cc) <= 7 * KA_WAVE_LDS;
                                if ((cr <= KA_WDFS_ROWS || lds_walk) && !(S.dbgskip & 1)) {
                                        S.q[1][0] = cur; cand[pos] = cb;      // (the wave below reads them from here)
                                        S.q[1][1].pad = pos;
                                        S.dfs_valid = -2;
                                } else {
                                        const Best B = { __int_as_float(cb.x), __int_as_float(cb.y), cb.z, cb.w };
                                        KaSub c[2];
                                        const int nc = ka_dfs_decide(S, cur, B, first_trial && at_root, c);
                                        for (int par = 0; par < 2; ++par) {
                                                KaCtl::Lvl& L = S.ctl->lvl[par];
                                                L.nsub = 0; L.rowalloc = 0; L.nitems = 0; L.next_item = 0; L.next_job = 0; L.npack[0] = 0; L.npack[1] = 0;
                                        }
                                        const KaLevelOut lo = ka_level_out(S, 0, false);
                                        int row = 0;
                                        for (int k = 0; k < nc; ++k) {
                                                c[k].roff = row; row += c[k].endb - c[k].startb + 1;
                                                S.q[1][k] = c[k];
                                                ka_emit_items(lo, k, c[k].starta, c[k].enda, c[k].endb - c[k].startb);
                                        }
                                        S.ctl->lvl[0].nsub = nc;
                                        S.ctl->lvl[0].rowalloc = row;
                                        S.dfs_valid = nc;
                                }
                        }
                }
                __syncthreads();
                const int st = S.dfs_valid;
                if (st == -1) break;
                if (st == -2) {
                        // a small subtree: wave 0 takes all of it (in a depth-first order the other waves have nothing to do anyway)
                        if (wave == 0) {
                                const int4 cb = cand[S.q[1][1].pad];
                                const Best B = { __int_as_float(cb.x), __int_as_float(cb.y), cb.z, cb.w };
                                const KaSub cur = S.q[1][0];
                                // operands, row buffers and stack in LDS when the window fits what the idle waves leave free
                                // (no consistency bonus there: those rows come from the task's tables in HBM)
                                const int wr = cur.enda - cur.starta, wc = cur.endb - cur.startb;
                                if (NB == 0 && (S.dbgskip & 2) == 0 && wr >= 1 && wr <= KA_LDFS_ROWS && wc >= 1 && wc < 4096 &&
                                    ka_sub_bytes(KIND, NRES, wr, wc) <= 7 * KA_WAVE_LDS)
                                        ka_subtree_dfs<KIND, NRES>(S, cur, B, first_trial && at_root, lane, lds_waves, tss);
                                else
                                        ka_wave_dfs<KIND, NRES, NB>(S, cur, B, first_trial && at_root, lane, lds_waves, 4, KA_WAVE_LDS,
                                                                    lds_waves + 7 * KA_WAVE_LDS, tss);
                        }
                        __syncthreads();
                        if (tid == 0) S.dfs_valid = 0;
                        __syncthreads();
                }
                at_root = false;
        }
}

// ------------------------------------------------------------------------------------------
// Incremental flip trials.  A flip trial differs from the baseline trial only below the meetups it flips: a node that is not
// flipped and has no flipped ancestor has the baseline's window, hence the baseline's candidates, margin and decision; only
// WHETHER an uncertain node flips depends on what came before it (the running count of uncertain meetups in recursion order,
// aln_seqseq.c:376-414).  So the trial walks the baseline's uncertain meetups in recursion order (sorted keys), counts them,
// and where the rule says "flip" it re-runs just that node's subtree depth first (ka_hirschberg_dfs with a seed: passes and
// candidates of the node again, this time with the runner-up, the flip, and everything below it in recursion order -- further
// flips included, the counter runs on); the baseline's meetups inside the old subtree are skipped (a contiguous key range),
// the raw path rows of the node's window are put back to what they held before its subtree ran.  Margins in recursion order =
// baseline segments and re-run subtrees concatenated, added in fp32 at the end.  Bit-identical with the depth-first trial
// (tests/test_gpu_refine.py), at the cost of the re-run subtrees instead of the whole recursion.
// ------------------------------------------------------------------------------------------
// after ka_margins_in_order (lds still holds the sorted (key, margin) pairs): sorted tables + the baseline's raw path
__device__ void ka_inc_build(TaskShared& S, const char* lds)
{
        const int tid = threadIdx.x;
        const int n = S.ctl->nrec;
        const int2* buf = (const int2*)lds;
        const KaInc I = ka_inc_view(S);
        for (int idx = tid; idx < n; idx += KA_NT) {
                const int key = S.mrec[idx].x;                        // keys are unique: one node, one key
                int lo = 0, hi = n - 1;
                while (lo < hi) { const int md = (lo + hi) >> 1; if (buf[md].x < key) lo = md + 1; else hi = md; }
                I.msort[lo] = idx;
        }
        for (int pos = tid; pos < n; pos += KA_NT) { I.skey[pos] = buf[pos].x; I.mseq0[pos] = __int_as_float(buf[pos].y); }
        const int g = max(S.La, S.Lb) + 2;
        for (int i = tid; i < g; i += KA_NT) I.raw0[i] = S.raw[i];
        if (tid == 0) S.inc_n = n;
        __syncthreads();
}

// the uncertain meetups of the baseline (margin below the trials' threshold), in recursion order; wave 0
__device__ void ka_inc_uncertain(TaskShared& S, const float thr)
{
        if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                const int n = S.inc_n;
                const KaInc I = ka_inc_view(S);
                int running = 0;
                for (int base = 0; base < n; base += 64) {
                        const int i = base + lane;
                        const bool flag = i < n && thr > 0.0f && I.mseq0[i] < thr;
                        const unsigned long long mask = __ballot(flag);
                        const int before = __popcll(mask & ((1ull << lane) - 1ull));
                        if (i < n) I.ucnt[i] = running + before;
                        if (flag) I.upos[running + before] = i;
                        running += __popcll(mask);
                }
                if (lane == 0) { I.ucnt[n] = running; S.inc_nunc = running; }
        }
        __syncthreads();
}

template <int KIND, int NRES, int NB>
__device__ __forceinline__ void ka_trial_incremental(TaskShared& S, char* lds_waves, const float* tss)
{
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const KaInc I = ka_inc_view(S);
        const int n = S.inc_n, nunc = S.inc_nunc;
        const int g = max(S.La, S.Lb) + 2;
        for (int i = tid; i < g; i += KA_NT) S.raw[i] = I.raw0[i];
        if (tid == 0) {
                S.rf.msum = 0.0f; S.rf.mcount = 0; S.rf.counter = 0; S.inc_p = 0;
                S.mlog = I.mseq; S.mlog_cap = 2 * (S.len_a + S.len_b + 8);
        }
        while (true) {
                __syncthreads();
                if (tid == 0) {
                        // the next flip: the uncertain meetup at which the running counter hits the trial's residue
                        const int c = S.rf.counter, st = S.rf.stride;
                        const int u = I.ucnt[S.inc_p];
                        const int skip = (((S.rf.trial - 1 - c) % st) + st) % st;
                        if (u + skip >= nunc) { S.inc_j = -1; S.rf.counter = c + (nunc - u); }
                        else { S.inc_j = I.upos[u + skip]; S.rf.counter = c + skip; }
                }
                __syncthreads();
                const int j = S.inc_j, p = S.inc_p, end = (j < 0) ? n : j, mc = S.rf.mcount;
                for (int i = p + tid; i < end; i += KA_NT) I.mseq[mc + (i - p)] = I.mseq0[i];
                __syncthreads();
                if (tid == 0) S.rf.mcount = mc + (end - p);
                if (j < 0) break;
                const int idx = I.msort[j];
                const KaSub X = I.win[idx];
                const int2 xm = I.mx[idx];
                for (int i = X.starta + tid; i <= X.enda; i += KA_NT) S.raw[i] = (i == X.starta) ? xm.y : -1;
                __syncthreads();
                ka_hirschberg_dfs<KIND, NRES, NB>(S, lds_waves, tss, false, &X);
                __syncthreads();
                // the baseline's next meetup behind the old subtree: first sorted key >= key + range (wave 0)
                if (wave == 0) {
                        const int bound = X.pad + xm.x;
                        int lo = j + 1, hi = n;
                        while (hi - lo > 64) {
                                const int step = (hi - lo + 63) / 64;
                                const int pos = lo + lane * step;
                                const bool less = pos < hi && I.skey[pos] < bound;
                                const int c = __popcll(__ballot(less));
                                const int nlo = c > 0 ? lo + (c - 1) * step + 1 : lo;
                                const int nhi = min(hi, lo + c * step);
                                lo = nlo; hi = max(nhi, nlo);
                        }
                        const int pos = lo + lane;
                        const bool less = pos < hi && I.skey[pos] < bound;
                        const int c = __popcll(__ballot(less));
                        if (lane == 0) S.inc_p = lo + c;
                }
        }
        __syncthreads();
        // the margins of the trial, added in recursion order in fp32 (the reference's running sum)
        if (wave == 0) {
                const int mcount = S.rf.mcount;
                float sum = 0.0f;
                for (int base = 0; base < mcount; base += 64) {
                        const float v = (base + lane < mcount) ? I.mseq[base + lane] : 0.0f;
                        const int cnt = min(64, mcount - base);
                        for (int i = 0; i < cnt; ++i) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
                }
                if (lane == 0) S.rf.msum = sum;
        }
        __syncthreads();
}

// ------------------------------------------------------------------------------------------
// P3: mirror + coding of the raw path by the whole workgroup (see
// oracle/kalign_oracle.c:ko_code_path for the as-executed semantics of add_gap_info_to_path_n).
// Row i of the (a-indexed) raw path emits g_i gap-in-a ops followed by one op (match or
// gap-in-b); two block-wide prefix sums (ops emitted, b positions consumed) place every row's
// ops independently.  Also records, per output column, which record of profile a / b feeds
// it (srcA/srcB, -1 = none) for the parallel update_n.  `lds` = 2*blockDim.x+4 ints of scratch.
// ------------------------------------------------------------------------------------------
__device__ void ka_code_path(TaskShared& S, int* lds)
{
        const int tid = threadIdx.x;
        const int len_a = S.len_a, len_b = S.len_b;
        const int* raw = S.raw;
        if (S.swapped) {
                int* r2 = S.raw2;
                for (int i = tid; i < len_a + 2; i += KA_NT) r2[i] = -1;
                __syncthreads();
                for (int i = 1 + tid; i <= len_b; i += KA_NT) { const int c = S.raw[i]; if (c != -1) r2[c] = i; }
                __syncthreads();
                raw = r2;
        }
        int* o = S.coded;
        int* tot_ops = lds;
        int* tot_b = lds + KA_NT;
        int* zmin = lds + 2 * KA_NT;
        int* zmax = zmin + 1;
        // rows [lo, hi) of this thread (1-based rows 1..len_a)
        const int per = (len_a + KA_NT - 1) / KA_NT;
        const int lo = 1 + tid * per, hi = min(len_a + 1, lo + per);
        auto row_gaps = [&](int i, int cur, int prev) -> int {
                // gap-in-a ops emitted before row i's own op (aln_setup.c:145-188)
                if (cur == -1) return 0;
                if (i == 1) return cur - 1;
                return (cur - 1 != prev && prev != -1) ? (cur - prev - 1) : 0;
        };
        int nops = 0, nb = 0;
        for (int i = lo; i < hi; ++i) {
                const int cur = raw[i], prev = (i > 1) ? raw[i - 1] : -1;
                const int g = row_gaps(i, cur, prev);
                nops += g + 1;
                nb += g + (cur != -1 ? 1 : 0);
        }
        // exclusive prefix sums of (ops, b positions) over the threads: wave scan + one pass over the wave totals
        const int lane_ = tid & 63, wave_ = tid >> 6;
        int sc_ops = nops, sc_b = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
                const int y1 = __shfl_up(sc_ops, d, 64), y2 = __shfl_up(sc_b, d, 64);
                if (lane_ >= d) { sc_ops += y1; sc_b += y2; }
        }
        if (lane_ == 63) { tot_ops[wave_] = sc_ops; tot_b[wave_] = sc_b; }
        if (tid == 0) { *zmin = 0x7fffffff; *zmax = 0; }
        __syncthreads();
        int off = sc_ops - nops, offb = sc_b - nb, all_ops = 0, total_b = 0;
        for (int k = 0; k < KA_NW; ++k) {
                const int a = tot_ops[k], b2 = tot_b[k];
                if (k < wave_) { off += a; offb += b2; }
                all_ops += a; total_b += b2;
        }
        // trailing gap-in-a run (aln_setup.c:180-186)
        const int last = raw[len_a];
        const int tail = (last != -1 && last < len_b) ? (len_b - last) : 0;
        const int alnlen = all_ops + tail;
        int my_zmin = 0x7fffffff, my_zmax = 0;
        int j = 1 + off, rb = 1 + offb;
        for (int i = lo; i < hi; ++i) {
                const int cur = raw[i], prev = (i > 1) ? raw[i - 1] : -1;
                const int g = row_gaps(i, cur, prev);
                for (int k = 0; k < g; ++k) { o[j] = 1; S.srcA[j] = -1; S.srcB[j] = rb++; ++j; }
                if (cur == -1) { o[j] = 2; S.srcA[j] = i; S.srcB[j] = -1; }
                else { o[j] = 0; S.srcA[j] = i; S.srcB[j] = rb++; my_zmin = min(my_zmin, j); my_zmax = max(my_zmax, j); }
                ++j;
        }
        if (my_zmax > 0) { atomicMin(zmin, my_zmin); atomicMax(zmax, my_zmax); }
        for (int k = tid; k < tail; k += KA_NT) { o[1 + all_ops + k] = 1; S.srcA[1 + all_ops + k] = -1; S.srcB[1 + all_ops + k] = 1 + total_b + k; }
        if (tid == 0) { o[0] = alnlen; o[alnlen + 1] = 3; S.ctl->alnlen = alnlen; }
        __syncthreads();
        // terminal-run flag (aln_setup.c:209-219): everything before the first and after the last
        // match column; the 4/8/16 flag loop never executes in the reference
        const int z1 = *zmin, z2 = *zmax;
        for (int c = 1 + tid; c <= alnlen; c += KA_NT) if (c < z1 || c > z2) o[c] |= 32;
        __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Path coding of the refinement pass: convert_raw_path (aln_refine.c:591-672) by the whole workgroup.  Differences to
// add_gap_info_to_path_n (ka_code_path): the gap-in-a run in front of a match is counted from the last MATCHED column
// (a prefix maximum over the rows), and the open / extend / close flags are real: 4 = first op of a gap run that
// follows a match, 8 = continuation of a run of the same kind, 16 = last op before a match (an op carrying 8 gets +8,
// which is 16 as well), 32 = runs before the first / after the last match.  Every flag depends on the op kinds of the
// two neighbours only.  `lds` = 3*blockDim.x+4 ints of scratch.
// ------------------------------------------------------------------------------------------
__device__ void ka_code_path_refine(TaskShared& S, int* lds)
{
        const int tid = threadIdx.x;
        const int len_a = S.len_a, len_b = S.len_b;
        const int* raw = S.raw;
        if (S.swapped) {
                int* r2 = S.raw2;
                for (int i = tid; i < len_a + 2; i += KA_NT) r2[i] = -1;
                __syncthreads();
                for (int i = 1 + tid; i <= len_b; i += KA_NT) { const int c = S.raw[i]; if (c != -1) r2[c] = i; }
                __syncthreads();
                raw = r2;
        }
        int* o = S.coded;
        int* tot_ops = lds;
        int* tot_b = lds + KA_NT;
        int* tot_m = lds + 2 * KA_NT;
        int* zmin = lds + 3 * KA_NT;
        int* zmax = zmin + 1;
        const int per = (len_a + KA_NT - 1) / KA_NT;
        const int lo = 1 + tid * per, hi = min(len_a + 1, lo + per);
        // last matched column before this thread's rows: exclusive prefix maximum over the threads
        int mymax = 0;
        for (int i = lo; i < hi; ++i) mymax = max(mymax, raw[i]);
        const int lane_ = tid & 63, wave_ = tid >> 6;
        int sc_m = mymax;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(sc_m, d, 64); if (lane_ >= d) sc_m = max(sc_m, y); }
        if (lane_ == 63) tot_m[wave_] = sc_m;
        if (tid == 0) { *zmin = 0x7fffffff; *zmax = 0; }
        __syncthreads();
        int blast = __shfl_up(sc_m, 1, 64);
        if (lane_ == 0) blast = 0;
        for (int k = 0; k < wave_; ++k) blast = max(blast, tot_m[k]);
        blast = max(blast, 0);
        // ops and b positions of this thread's rows
        int nops = 0, nb = 0;
        {
                int bl = blast;
                for (int i = lo; i < hi; ++i) {
                        const int cur = raw[i];
                        if (cur == -1) { nops += 1; }
                        else { const int gpre = max(cur - bl - 1, 0); nops += gpre + 1; nb += gpre + 1; bl = cur; }
                }
        }
        int sc_ops = nops, sc_b = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
                const int y1 = __shfl_up(sc_ops, d, 64), y2 = __shfl_up(sc_b, d, 64);
                if (lane_ >= d) { sc_ops += y1; sc_b += y2; }
        }
        if (lane_ == 63) { tot_ops[wave_] = sc_ops; tot_b[wave_] = sc_b; }
        __syncthreads();
        int off = sc_ops - nops, offb = sc_b - nb, all_ops = 0, total_b = 0, all_max = 0;
        for (int k = 0; k < KA_NW; ++k) {
                const int a = tot_ops[k], b2 = tot_b[k];
                if (k < wave_) { off += a; offb += b2; }
                all_ops += a; total_b += b2; all_max = max(all_max, tot_m[k]);
        }
        all_max = max(all_max, 0);
        const int tail = len_b - all_max;                                 // trailing gap-in-a run (:630-634)
        const int alnlen = all_ops + tail;
        int my_zmin = 0x7fffffff, my_zmax = 0;
        {
                int j = 1 + off, rb = 1 + offb, bl = blast;
                for (int i = lo; i < hi; ++i) {
                        const int cur = raw[i];
                        if (cur == -1) { o[j] = 2; S.srcA[j] = i; S.srcB[j] = -1; ++j; }
                        else {
                                const int gpre = max(cur - bl - 1, 0);
                                for (int k = 0; k < gpre; ++k) { o[j] = 1; S.srcA[j] = -1; S.srcB[j] = rb++; ++j; }
                                o[j] = 0; S.srcA[j] = i; S.srcB[j] = rb++;
                                my_zmin = min(my_zmin, j); my_zmax = max(my_zmax, j);
                                ++j; bl = cur;
                        }
                }
        }
        if (my_zmax > 0) { atomicMin(zmin, my_zmin); atomicMax(zmax, my_zmax); }
        for (int k = tid; k < tail; k += KA_NT) { o[1 + all_ops + k] = 1; S.srcA[1 + all_ops + k] = -1; S.srcB[1 + all_ops + k] = 1 + total_b + k; }
        if (tid == 0) { o[0] = alnlen; o[alnlen + 1] = 3; S.ctl->alnlen = alnlen; }
        __syncthreads();
        const int z1 = *zmin, z2 = *zmax;
        // flags: the op kinds are final; every position reads its neighbours' kinds (low two bits) and writes itself
        for (int c = 1 + tid; c <= alnlen; c += KA_NT) {
                const int t = o[c] & 3;
                int v = t;
                if (t != 0) {
                        if (c >= 2) {
                                const int tp = o[c - 1] & 3;
                                if (tp == 0) v |= 4; else if (tp == t) v |= 8;
                        }
                        if (c <= alnlen - 1 && (o[c + 1] & 3) == 0) { if (v & 8) v += 8; else v |= 16; }
                }
                if (c < z1 || c > z2) v |= 32;
                S.raw2[c] = v;                                            // (raw2 is free once the path is mirrored / coded)
        }
        __syncthreads();
        for (int c = 1 + tid; c <= alnlen; c += KA_NT) o[c] = S.raw2[c];
        __syncthreads();
}

// ------------------------------------------------------------------------------------------
// compute_sp_score (sp_score.c:22-201): residue counts per column of both groups from the members' residue -> column
// tables (build_profile expands every member through its gaps[]; D.colof is the same information), then ONE sequential
// fp32 walk along the coded path -- substitution terms in (i, j) order, then the gap term, exactly as the reference
// accumulates them (the total decides which trial wins; it is not reassociated).
// S.sp_freq: [col][24] for operand a (23 counts + residues in the column), then the same for b.
// ------------------------------------------------------------------------------------------
__device__ void ka_sp_build(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T)
{
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const long long total = 24ll * (S.len_a + S.len_b);
        for (long long x = tid; x < total; x += KA_NT) S.sp_freq[x] = 0;
        __syncthreads();
        const int na = T.nsip_a, nb = T.nsip_b;
        const int* ma = D.sip + D.sip_off[T.a];
        const int* mb = D.sip + D.sip_off[T.b];
        for (int m = wave; m < na + nb; m += KA_NW) {
                const bool in_a = m < na;
                const int si = in_a ? ma[m] : mb[m - na];
                int* fr = S.sp_freq + (in_a ? 0 : 24 * S.len_a);
                const int* col = D.colof + D.seq_off[si];
                const uint8_t* res = D.codes + D.seq_off[si];
                const int len = D.node_len[si];
                for (int p = lane; p < len; p += 64) {
                        const int c = col[p], r = res[p];
                        if (r < 23) { atomicAdd(&fr[24 * c + r], 1); atomicAdd(&fr[24 * c + 23], 1); }
                }
        }
        __syncthreads();
}

// The walk adds ONE term at a time to ONE float (sp_score.c:134-189): the order of the additions is part of the result
// and the chain cannot be split.  What can be parallel is everything around the additions: one thread per path column
// works out its column's terms -- the (i, j) products in the reference's order, then the gap term(s), with the sign
// folded in (x - y == x + (-y)) -- into an LDS buffer, and one thread adds the buffer up in order (loads run ahead of
// the dependent adds: the chain costs an add per term instead of a trip to L2 per term).
// lds: KA_SP_TB floats + 2 * blockDim.x ints.
#define KA_SP_TB 24576
__device__ void ka_sp_score(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, char* lds)
{
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        float* const buf = (float*)lds;
        int* const wtot = (int*)(buf + KA_SP_TB);                     // per-wave totals of the scan
        const int* path = S.coded;
        const int* fa0 = S.sp_freq;
        const int* fb0 = S.sp_freq + 24 * S.len_a;
        const int nsa = T.nsip_a, nsb = T.nsip_b;
        const float gpo = T.gpo, gpe = T.gpe, tgpe = T.tgpe;
        const float* subm = D.subm;
        const int plen = path[0];
        float total = 0.0f;                                            // (thread 0's)
        for (int c0 = 1; c0 <= plen; c0 += KA_NT) {
                const int c = c0 + tid;
                const bool in = c <= plen;
                const int code = in ? path[c] : 0;
                const int step = code & 3;
                const float pen = (code & 32) ? tgpe : gpe;
                const int prev = (in && c > 1) ? (path[c - 1] & 3) : 0;
                const int* fa = fa0 + 24 * ((in && step != 1) ? S.srcA[c] - 1 : 0);
                const int* fb = fb0 + 24 * ((in && step != 2) ? S.srcB[c] - 1 : 0);
                // the column's term count
                int nza = 0, nzb = 0, cnt = 0;
                if (in) {
                        if (step == 0) {
                                for (int i = 0; i < 23; ++i) nza += fa[i] != 0;
                                for (int j = 0; j < 23; ++j) nzb += fb[j] != 0;
                                cnt = nza * nzb + 1;
                        } else if (step == 1) cnt = (prev == 1) ? 1 : 2;
                        else cnt = (prev == 2) ? 1 : 2;
                }
                // exclusive scan over the block
                int sc = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(sc, d, 64); if (lane >= d) sc += y; }
                __syncthreads();                                       // (the previous block's sum is done with buf / wtot)
                if (lane == 63) wtot[wave] = sc;
                __syncthreads();
                int off = sc - cnt, all = 0;
                for (int w = 0; w < KA_NW; ++w) { const int x = wtot[w]; if (w < wave) off += x; all += x; }
                for (int base = 0; base < all; base += KA_SP_TB) {
                        // this thread's terms with a buffer index in [base, base + KA_SP_TB)
                        if (in && off < base + KA_SP_TB && off + cnt > base) {
                                int k = off - base;
                                auto put = [&](float v) { if (k >= 0 && k < KA_SP_TB) buf[k] = v; ++k; };
                                if (step == 0) {
                                        for (int i = 0; i < 23; ++i) {
                                                const int ai = fa[i];
                                                if (ai == 0) continue;
                                                for (int j = 0; j < 23; ++j) {
                                                        const int bj = fb[j];
                                                        if (bj == 0) continue;
                                                        put((float)(ai * bj) * subm[i * 23 + j]);
                                                }
                                        }
                                        const int n_res_a = fa[23], n_res_b = fb[23];
                                        const int n_gap_a = nsa - n_res_a, n_gap_b = nsb - n_res_b;
                                        put(-((float)(n_res_a * n_gap_b + n_gap_a * n_res_b) * pen));
                                } else if (step == 1) {
                                        const int n_pairs = nsa * fb[23];
                                        if (prev != 1) put(-((float)n_pairs * gpo));
                                        put(-((float)n_pairs * pen));
                                } else {
                                        const int n_pairs = fa[23] * nsb;
                                        if (prev != 2) put(-((float)n_pairs * gpo));
                                        put(-((float)n_pairs * pen));
                                }
                        }
                        __syncthreads();
                        if (tid == 0) {
                                const int n = min(KA_SP_TB, all - base);
                                int k = 0;
                                for (; k + 8 <= n; k += 8) {
                                        const float4v x = *(const float4v*)(buf + k), y = *(const float4v*)(buf + k + 4);
                                        total += x.x; total += x.y; total += x.z; total += x.w;
                                        total += y.x; total += y.y; total += y.z; total += y.w;
                                }
                                for (; k < n; ++k) total += buf[k];
                        }
                        if (base + KA_SP_TB < all) __syncthreads();        // the buffer is refilled
                }
        }
        if (tid == 0) S.sp_value = total;
        __syncthreads();
}

// ------------------------------------------------------------------------------------------
// P4: update_n (aln_setup.c:230-436), one thread per (output column, field).
// ------------------------------------------------------------------------------------------
__device__ void ka_update_profile(const TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, const int alnlen)
{
        const float* pa = S.profa;
        const float* pb = S.profb;
        float* np = S.newp;
        const float sipa = (float)T.nsip_a, sipb = (float)T.nsip_b;
        float sA = 1.0f, sB = 1.0f;
        bool rebalance = false;
        if (D.usw > 0.0f && T.nsip_a > 0 && T.nsip_b > 0) {
                const float pseudo = D.usw;
                const float total = (float)(T.nsip_a + T.nsip_b);
                const float denom = total + 2.0f * pseudo;
                sA = total * (sipa + pseudo) / (denom * sipa);
                sB = total * (sipb + pseudo) / (denom * sipb);
                rebalance = true;
        }
        // fields 27..29 of an operand as the reference would see them at this point: zero for a
        // leaf (make_profile_n), [55..57] * nsip_other for a profile (set_gap_penalties_n ran on it
        // for this merge, aln_run.c:239-253).  They are dead values (always rewritten before
        // use) but part of the merged record, so they are reproduced for bit-identical profiles.
        const bool leaf_a = (T.nsip_a == 1), leaf_b = (T.nsip_b == 1);
        auto fa = [&](const float* rec, int k) -> float {
                if (k >= 27 && k <= 29) return leaf_a ? 0.0f : rec[k + 28] * sipb;
                return rec[k];
        };
        auto fb = [&](const float* rec, int k) -> float {
                if (k >= 27 && k <= 29) return leaf_b ? 0.0f : rec[k + 28] * sipa;
                return rec[k];
        };
        // one thread per (output column, 4 consecutive fields): the column's op code and source
        // records are looked up once, the four field values are independent
        const float* __restrict__ pa_r = pa;
        const float* __restrict__ pb_r = pb;
        float* __restrict__ np_r = np;
        const int* __restrict__ coded = S.coded;
        const int* __restrict__ srcA = S.srcA;
        const int* __restrict__ srcB = S.srcB;
        auto elem = [&](const int c, const int k, const int code, const float* __restrict__ ra, const float* __restrict__ rb) -> float {
                float val;
                if (c == 0 || c == alnlen + 1) {
                        const float va = fa(ra, k), vb = fb(rb, k);
                        val = (rebalance && k < 23) ? (va * sA + vb * sB) : (va + vb);
                } else if (!code) {
                        if (rebalance && k < 23) {
                                val = ra[k] * sA + rb[k] * sB;
                        } else {
                                val = fa(ra, k) + fb(rb, k);
                                if (rebalance && k >= 32 && k < 55) {
                                        const float dA = sA - 1.0f, dB = sB - 1.0f;
                                        const int jj = k - 32;
                                        float delta = 0.0f;
                                        for (int aa = 0; aa < 23; ++aa) {
                                                delta += (ra[aa] * dA + rb[aa] * dB) * D.subm[23 * aa + jj];
                                        }
                                        val += delta;
                                }
                        }
                } else {
                        const bool gap_in_a = (code & 1) != 0;
                        const float sip = gap_in_a ? sipa : sipb;
                        val = gap_in_a ? fb(rb, k) : fa(ra, k);
                        // as the reference: up to two successive adjustments (close, then open)
                        if (!(code & 20)) {
                                if (code & 32) {
                                        if (k == 25) val += sip;
                                        if (k >= 32 && k < 55) val -= D.tgpe0 * sip;
                                } else {
                                        if (k == 24) val += sip;
                                        if (k >= 32 && k < 55) val -= D.gpe0 * sip;
                                }
                        } else {
                                for (int pass = 0; pass < 2; ++pass) {
                                        const int bit = pass == 0 ? 16 : 4;
                                        if (!(code & bit)) continue;
                                        float gp;
                                        if (code & 32) {
                                                if (k == 25) val += sip;
                                                gp = D.tgpe0 * sip;
                                                if (k == 23) val += sip;
                                                gp += D.gpo0 * sip;
                                        } else {
                                                if (k == 23) val += sip;
                                                gp = D.gpo0 * sip;
                                        }
                                        if (k >= 32 && k < 55) val -= gp;
                                }
                        }
                }
                return val;
        };
        const long long total4 = (long long)(alnlen + 2) * 16;
        const float gpe_a = D.gpe0 * sipa, gpe_b = D.gpe0 * sipb, tgpe_a = D.tgpe0 * sipa, tgpe_b = D.tgpe0 * sipb;
        for (long long x4 = (long long)S.member * KA_NT + threadIdx.x; x4 < total4; x4 += (long long)S.G * KA_NT) {
                const int c = (int)(x4 >> 4);
                const int k4 = (int)(x4 & 15) << 2;
                int code = 0;
                const float* ra;
                const float* rb;
                if (c == 0) { ra = pa_r; rb = pb_r; }
                else if (c == alnlen + 1) { ra = pa_r + ((long long)(S.len_a + 1) << 6); rb = pb_r + ((long long)(S.len_b + 1) << 6); }
                else {
                        code = coded[c];
                        const int ia = srcA[c], ib = srcB[c];
                        ra = pa_r + ((long long)(ia < 0 ? 0 : ia) << 6);
                        rb = pb_r + ((long long)(ib < 0 ? 0 : ib) << 6);
                }
                float4v out;
                if (!rebalance && !(code & 20)) {
                        // The common case (no sequence weights; the coded path carries only the flags the reference
                        // really sets), four fields at a time -- same operations as elem() below, without the per-field
                        // branching: a match / boundary column is the sum of the two records, a gap column the present
                        // side with its gap counter bumped and the scores lowered by (t)gpe * members of the absent side.
                        float4v A = *(const float4v*)(ra + k4), B = *(const float4v*)(rb + k4);
                        if (k4 == 24) {                                  // field 27 (see fa / fb)
                                A.w = leaf_a ? 0.0f : ra[55] * sipb; B.w = leaf_b ? 0.0f : rb[55] * sipa;
                        } else if (k4 == 28) {                           // fields 28, 29
                                A.x = leaf_a ? 0.0f : ra[56] * sipb; A.y = leaf_a ? 0.0f : ra[57] * sipb;
                                B.x = leaf_b ? 0.0f : rb[56] * sipa; B.y = leaf_b ? 0.0f : rb[57] * sipa;
                        }
                        if (c == 0 || c == alnlen + 1 || !code) {
                                out = A + B;
                        } else {
                                const bool gap_in_a = (code & 1) != 0, term = (code & 32) != 0;
                                const float sip = gap_in_a ? sipa : sipb;
                                const float g = term ? (gap_in_a ? tgpe_a : tgpe_b) : (gap_in_a ? gpe_a : gpe_b);
                                out = gap_in_a ? B : A;
                                if (k4 == 24) { if (term) out.y += sip; else out.x += sip; }        // [25] / [24]
                                else if (k4 >= 32 && k4 < 52) { out.x -= g; out.y -= g; out.z -= g; out.w -= g; }
                                else if (k4 == 52) { out.x -= g; out.y -= g; out.z -= g; }           // [55] is not a score
                        }
                } else {
                        out.x = elem(c, k4 + 0, code, ra, rb);
                        out.y = elem(c, k4 + 1, code, ra, rb);
                        out.z = elem(c, k4 + 2, code, ra, rb);
                        out.w = elem(c, k4 + 3, code, ra, rb);
                }
                *(float4v*)(np_r + (x4 << 2)) = out;
        }
}

// Leaf profile (make_profile_n, aln_setup.c:40-99), one float4 per thread.  The pre-summed
// substitution scores subm[c][j] - soff come from the seq-seq table in LDS (same expression, same
// bits).  Non-leaf operands need nothing here: set_gap_penalties_n is folded into the loads.
__device__ void ka_make_leaf_profile(float* __restrict__ prof, int len, const uint8_t* __restrict__ seq,
                                     float gpo, float gpe, float tgpe, const float* tss)
{
        const long long total4 = (long long)(len + 2) * 16;
        for (long long x4 = threadIdx.x; x4 < total4; x4 += KA_NT) {
                const int r = (int)(x4 >> 4);
                const int k4 = (int)(x4 & 15) << 2;
                const bool inner = (r >= 1 && r <= len);
                const int c = inner ? seq[r - 1] : 0;
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                        const int k = k4 + u;
                        float val = 0.0f;
                        if (k == 55) val = -gpo;
                        else if (k == 56) val = -gpe;
                        else if (k == 57) val = -tgpe;
                        else if (inner) {
                                if (k == c) val = 1.0f;
                                else if (k >= 32 && k < 55) val = tss[c * KA_T_STRIDE + (k - 32)];
                        }
                        v[u] = val;
                }
                float4v out; out.x = v[0]; out.y = v[1]; out.z = v[2]; out.w = v[3];
                *(float4v*)(prof + (x4 << 2)) = out;
        }
}

// ------------------------------------------------------------------------------------------
// Anchor consistency, per task (anchor_consistency.c:352-561 + do_align's bonus block,
// aln_run.c:262-295).  The reference materialises a dense La x Lb bonus matrix on the host for
// every task; per anchor every row has at most ONE non-zero entry, so the device keeps <= K
// (column, value) entries per DP row instead and the passes carry them in registers.
//
// ka_cons_votes = get_node_anchor_positions for both operands and all anchors: for a leaf the
// position map itself; for a profile a vote over its member sequences, where "best" is the anchor
// position of the FIRST member (in the reference's sip order) that has one in the column.  Being
// first in a fixed order is a min-reduction over the member index, and `agree` / `total` are
// counts, so the vote runs in parallel over (member, residue) with LDS atomics:
// key = member_index << 32 | position, counts = total | agree << 16.  Which column a residue sits
// in comes from D.colof (kept up to date by ka_update_colof).  The workgroups of a cluster share
// the work by operand and by anchor; tables that do not fit into LDS live in the task's HBM scratch.
// ------------------------------------------------------------------------------------------
template <bool LEAN>
__device__ void ka_cons_votes(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, char* lds, const long long lds_bytes)
{
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int K = D.cons_K;
        const long long n = (long long)S.len_a + S.len_b + 8;           // stride of the per-anchor arrays
        const int half = (S.G >= 2) ? (S.G >> 1) : 1;
        for (int side = 0; side < 2; ++side) {
                if (S.G >= 2 && side != S.member / half) continue;
                const int sub = (S.G >= 2) ? (S.member % half) : 0;
                const bool is_rows = (side == 0);
                const int node = (is_rows != (S.swapped != 0)) ? T.a : T.b;  // rows: a unless swapped
                const int nmem = (node == T.a) ? T.nsip_a : T.nsip_b;
                const int dp_len = is_rows ? S.La : S.Lb;
                int* apos = is_rows ? S.apos_r : S.apos_c;
                float* conf = is_rows ? S.conf_r : S.conf_c;
                // this workgroup's anchors: sub, sub + half, ...  (closed form: an indexed array would live in scratch
                // memory and put a scratch load in front of every gather)
                const int nk = (K - sub + half - 1) / half;
#define KS(b_) (sub + (b_) * half)
                if (nmem == 1) {
                        // leaf: direct lookup (a leaf's dp_len is its length)
                        for (int b = 0; b < nk; ++b) {
                                const int* map = D.cons_maps + D.cons_map_off[node] + (long long)KS(b) * dp_len;
                                for (int i = tid; i < dp_len; i += KA_NT) {
                                        const int a = map[i];
                                        apos[KS(b) * n + i] = a; conf[KS(b) * n + i] = (a >= 0) ? 1.0f : 0.0f;
                                }
                        }
                        continue;
                }
                if (LEAN) continue;                                      // lean levels hold leaf-leaf tasks only
                // a cell's `total` and `agree` counts share one 32-bit word (16 bits each) -- below 65536 members; from there on
                // `agree` has a word of its own (anchor_consistency.c:352-470 counts in ints)
                const bool wide = nmem >= 65536;
                const long long cell = wide ? 16 : 12;
                const int* members = D.sip + D.sip_off[node];
                int fit = (int)(lds_bytes / (cell * dp_len));             // anchors whose tables fit into LDS together
                const bool in_lds = fit >= 1;
                if (!in_lds) fit = nk;
                for (int b0 = 0; b0 < nk; b0 += fit) {
                        const int nb = min(fit, nk - b0);
                        unsigned long long* key = in_lds ? (unsigned long long*)lds : (unsigned long long*)S.vote;
                        unsigned int* cnt = (unsigned int*)(key + (long long)nb * dp_len);
                        unsigned int* agr = cnt + (long long)nb * dp_len;        // (wide only)
                        for (int x = tid; x < nb * dp_len; x += KA_NT) { key[x] = ~0ull; cnt[x] = 0u; if (wide) agr[x] = 0u; }
                        __syncthreads();
                        // Two sweeps over (member, residue): [0] first-member key + total, [1] agreement with the
                        // winner.  Latency-bound gathers, so each wave pre-loads the metadata of 64 of its
                        // members lane-parallel and keeps 4 x 64 residues of loads in flight before the atomics.
                        for (int sweep = 0; sweep < 2; ++sweep) {
                                const int mine = (nmem - wave + KA_NW - 1) / KA_NW;          // members of this wave
                                for (int base = 0; base < mine; base += 64) {
                                        const int ml = min(base + lane, mine - 1);
                                        const int mi_l = wave + KA_NW * ml;
                                        const int si_l = members[mi_l];
                                        const int len_l = D.node_len[si_l];
                                        const long long mo_l = D.cons_map_off[si_l];
                                        const int so_l = D.seq_off[si_l];
                                        const int cntm = min(64, mine - base);
                                        for (int jm = 0; jm < cntm; ++jm) {
                                                const int mi = wave + KA_NW * (base + jm);
                                                const int len = __shfl(len_l, jm, 64);
                                                const long long mo = __shfl(mo_l, jm, 64);
                                                const int* map = D.cons_maps + mo;
                                                const int* col = D.colof + __shfl(so_l, jm, 64);
                                                for (int p0 = lane; p0 < len; p0 += 256) {
                                                        int cc[4], aa[4][KA_NB - 1];
#pragma unroll
                                                        for (int u = 0; u < 4; ++u) {
                                                                const int pp = p0 + 64 * u;
                                                                const bool ok = pp < len;
                                                                cc[u] = ok ? col[pp] : 0;
#pragma unroll
                                                                for (int b = 0; b < KA_NB - 1; ++b)
                                                                        aa[u][b] = (ok && b < nb) ? map[(long long)KS(b0 + b) * len + pp] : -1;
                                                        }
#pragma unroll
                                                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                                                                for (int b = 0; b < KA_NB - 1; ++b) {
                                                                        const int a = aa[u][b];
                                                                        if (a < 0) continue;
                                                                        const int x = b * dp_len + cc[u];
                                                                        if (sweep == 0) {
                                                                                atomicMin(&key[x], ((unsigned long long)(unsigned int)mi << 32) | (unsigned int)a);
                                                                                atomicAdd(&cnt[x], 1u);
                                                                        } else {
                                                                                // (HBM tables: the atomics were performed at L2; read them back past the L1)
                                                                                const unsigned long long kk = in_lds ? key[x]
                                                                                        : __hip_atomic_load(&key[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                                                if ((unsigned int)a == (unsigned int)(kk & 0xffffffffull)) { if (wide) atomicAdd(&agr[x], 1u); else atomicAdd(&cnt[x], 0x10000u); }
                                                                        }
                                                                }
                                                        }
                                                }
                                        }
                                }
                                __syncthreads();
                        }
                        for (int x = tid; x < nb * dp_len; x += KA_NT) {
                                const int b = x / dp_len, c = x - b * dp_len;
                                unsigned long long kk;
                                unsigned int cc, ca = 0u;
                                if (in_lds) { kk = key[x]; cc = cnt[x]; if (wide) ca = agr[x]; }
                                else {
                                        kk = __hip_atomic_load(&key[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        cc = __hip_atomic_load(&cnt[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        if (wide) ca = __hip_atomic_load(&agr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                const int tot = wide ? (int)cc : (int)(cc & 0xffffu), ag = wide ? (int)ca : (int)(cc >> 16);
                                const long long o = KS(b0 + b) * n + c;
                                if (tot > 0 && ag > 0) { apos[o] = (int)(unsigned int)(kk & 0xffffffffull); conf[o] = (float)ag / (float)tot; }
                                else { apos[o] = -1; conf[o] = 0.0f; }
                        }
                        __syncthreads();
                }
        }
}

#undef KS

// anchor_consistency_get_bonus_profile in sparse form (first workgroup of the cluster, after the votes).
// After it S.ent[row][0..KA_NB) holds the row's non-zero bonus cells with distinct columns: entries of
// different anchors that hit the same cell are summed in anchor order (the dense matrix accumulates
// k = 0..K-1 into a zeroed cell), and slot KA_NB-1 carries the cell the reference reaches when a forward
// pass indexes column Lb of row i -- flat index i*Lb + Lb is cell (i+1, 0) (aln_seqseq.c:83-85 uses the
// 1-based column).
__device__ void ka_cons_entries(TaskShared& S, const KaTreeDev& D)
{
        const int tid = threadIdx.x;
        const int K = D.cons_K;
        const int rows = S.La, cols = S.Lb;
        const long long n = (long long)S.len_a + S.len_b + 8;
        const int ml = D.cons_maxlen + 8;
        const float paw = D.cons_paw;
        // inverse maps anchor position -> column; of several columns the last one wins (:521-526)
        for (int x = tid; x < K * ml; x += KA_NT) S.invj[x] = -1;
        __syncthreads();
        for (int x = tid; x < K * cols; x += KA_NT) {
                const int k = x / cols, j = x - k * cols;
                const int a = S.apos_c[k * n + j];
                if (a >= 0) atomicMax(&S.invj[k * ml + a], j);
        }
        __syncthreads();
        for (int i = tid; i < rows; i += KA_NT) {
                int mc[KA_NB];
                float mv[KA_NB];
                int cnt = 0;
                for (int k = 0; k < K; ++k) {
                        const int a = S.apos_r[k * n + i];
                        if (a < 0) continue;
                        const int bj = __hip_atomic_load(&S.invj[k * ml + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (bj < 0) continue;
                        const float val = paw * S.conf_r[k * n + i] * S.conf_c[k * n + bj];                // :534-535
                        int hit = -1;
                        for (int m = 0; m < cnt; ++m) if (mc[m] == bj) hit = m;
                        if (hit >= 0) mv[hit] += val;
                        else { mc[cnt] = bj; mv[cnt] = 0.0f + val; ++cnt; }
                }
                int2* e = S.ent + (long long)i * KA_NB;
                for (int m = 0; m < KA_NB - 1; ++m) e[m] = (m < cnt) ? make_int2(mc[m], __float_as_int(mv[m])) : make_int2(-1, 0);
        }
        __syncthreads();
        for (int i = tid; i < rows; i += KA_NT) {
                int2 w = make_int2(-1, 0);
                if (i + 1 < rows) {
                        const int2* nx = S.ent + (long long)(i + 1) * KA_NB;
                        for (int m = 0; m < KA_NB - 1; ++m) if (nx[m].x == 0) w = make_int2(cols, nx[m].y);
                }
                S.ent[(long long)i * KA_NB + KA_NB - 1] = w;
        }
        __syncthreads();
}

// make_seq / update_gaps (weave_alignment.c:41-112) in the device's form: after the merge of a and
// b, residue p of a member of a moves from column col to amap[col].  S.raw / S.raw2 (dead after the
// path coding) receive amap / bmap.  All workgroups of the cluster share the members.
__device__ void ka_update_colof(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, const int alnlen)
{
        const int tid = threadIdx.x;
        if (S.member == 0) {
                for (int j = 1 + tid; j <= alnlen; j += KA_NT) {
                        const int ia = S.srcA[j], ib = S.srcB[j];
                        if (ia >= 1) S.raw[ia - 1] = j - 1;
                        if (ib >= 1) S.raw2[ib - 1] = j - 1;
                }
        }
        ka_cluster_sync(S);
        const int lane = tid & 63, wave = tid >> 6;
        const int na = T.nsip_a, nb = T.nsip_b;
        const int* ma = D.sip + D.sip_off[T.a];
        const int* mb = D.sip + D.sip_off[T.b];
        for (int m = S.member * KA_NW + wave; m < na + nb; m += KA_NW * S.G) {
                const int si = (m < na) ? ma[m] : mb[m - na];
                const int* mp = (m < na) ? S.raw : S.raw2;
                int* col = D.colof + D.seq_off[si];
                const int len = D.node_len[si];
                for (int p = lane; p < len; p += 64) col[p] = mp[col[p]];
        }
}

// dynamic-LDS layout of a workgroup
#define KA_LDS_DBG 1400
#define KA_LDS_TSS 1408
#define KA_LDS_WAVES 4096                                           // per-wave regions: 2048-B aligned (ring addressing ORs the column offset in)
static_assert(KA_LDS_TSS + 23 * KA_T_STRIDE * 4 <= KA_LDS_WAVES - KA_LDS_HO_BACK, "score table overlaps the hand-over control words");
static_assert(KA_LDS_TSS + 23 * KA_T_STRIDE * 4 <= KA_LDS_WAVES, "score table overlaps the wave regions");
#define KA_LDS_TOTAL (KA_LDS_WAVES + KA_WAVES * KA_WAVE_LDS)
#define KA_HALF_BLOCK 256
#define KA_LDS_HALF (KA_LDS_WAVES + (KA_HALF_BLOCK / 64) * KA_WAVE_LDS)   // 4 rings: two workgroups per CU
// seq-seq kernels: the path-coding scratch, then a small region per wave for wave-local subtrees (ka_subtree.h)
#define KA_LEAN_SCRATCH(nt_) ((((2 * (nt_) + 16) * 4) + 15) & ~15)
#define KA_LDS_PAIR (KA_LDS_WAVES + KA_LEAN_SCRATCH(KA_PAIR_BLOCK) + (KA_PAIR_BLOCK / 64) * KA_WAVE_LDS_LEAN)
#define KA_LDS_LEAN (KA_LDS_WAVES + KA_LEAN_SCRATCH(KA_LEAN_BLOCK) + (KA_LEAN_BLOCK / 64) * KA_WAVE_LDS_LEAN)
static_assert(sizeof(TaskShared) <= KA_LDS_DBG, "TaskShared outgrew its LDS slot");
static_assert(KA_LDS_WAVES % 16 == 0, "wave regions must be 16-B aligned");

// seq-seq score table T[a][b] = subm[a][b] - soff (one rounding, as aln_seqseq.c:82 evaluates it)
__device__ void ka_build_tss(float* tss, const float* subm, float soff)
{
        for (int x = threadIdx.x; x < 23 * KA_T_STRIDE; x += KA_NT) {
                const int a = x / KA_T_STRIDE, b = x % KA_T_STRIDE;
                tss[x] = (b < 23) ? (subm[23 * a + b] - soff) : 0.0f;
        }
}

__device__ __forceinline__ long long ka_align_up(long long x, long long a) { return (x + a - 1) / a * a; }

// bytes of one member's private recursion state (queues, work lists, row buffers) in a cluster that splits
__device__ __host__ inline long long ka_private_bytes(long long la, long long lb)
{
        const long long n = la + lb + 8;
        const long long nq = (la < lb ? la : lb) + 20;
        const long long ni = 2 * nq + 2 * (n / KA_STRIP1_ROWS + 2);
        return 2 * ((nq * (long long)sizeof(KaSub) + 15) / 16 * 16) + 2 * ((ni * 8 + 15) / 16 * 16) + 2 * ((ni * 4 + 15) / 16 * 16)
             + 4 * ((2 * nq * 8 + 15) / 16 * 16) + 2 * ((n * 12 + 15) / 16 * 16);
}

// carve the per-task scratch region (cons_maxlen > 0: the job has a consistency table)
__device__ long long ka_carve(TaskShared& S, char* base, int la, int lb, int cons_maxlen, bool refine = false, bool rec = false)
{
        const long long n = (long long)la + lb + 8;
        long long o = 0;
        S.raw = (int*)(base + o);   o += ka_align_up(n * 4, 16);
        S.raw2 = (int*)(base + o);  o += ka_align_up(n * 4, 16);
        S.coded = (int*)(base + o); o += ka_align_up(n * 4, 16);
        S.srcA = (int*)(base + o);  o += ka_align_up(n * 4, 16);
        S.srcB = (int*)(base + o);  o += ka_align_up(n * 4, 16);
        S.fbuf = (KaState*)(base + o); o += ka_align_up(n * 12, 16);
        S.bbuf = (KaState*)(base + o); o += ka_align_up(n * 12, 16);
        S.xfbuf = (KaState*)(base + o); o += ka_align_up(n * 12, 16);
        S.xbbuf = (KaState*)(base + o); o += ka_align_up(n * 12, 16);
        const long long nq = (long long)(la < lb ? la : lb) + 20;
        S.q[0] = (KaSub*)(base + o); o += ka_align_up(nq * (long long)sizeof(KaSub), 16);
        S.q[1] = (KaSub*)(base + o); o += ka_align_up(nq * (long long)sizeof(KaSub), 16);
        const long long ni = 2 * nq + 2 * (n / KA_STRIP1_ROWS + 2);
        S.items[0] = (int2*)(base + o); o += ka_align_up(ni * 8, 16);
        S.items[1] = (int2*)(base + o); o += ka_align_up(ni * 8, 16);
        S.prog[0] = (int*)(base + o); o += ka_align_up(ni * 4, 16);
        S.prog[1] = (int*)(base + o); o += ka_align_up(ni * 4, 16);
        for (int par = 0; par < 2; ++par)
                for (int cls = 0; cls < 2; ++cls) { S.pack[par][cls] = (int2*)(base + o); o += ka_align_up(2 * nq * 8, 16); }
        // a cluster that splits (TaskShared::Gw): every member's private queues, work lists and row buffers
        if (S.G > 1) {
                const long long pb = ka_private_bytes(la, lb);
                char* pr = base + o + (long long)S.member * pb;
                long long x = 0;
                S.priv.q[0] = (KaSub*)(pr + x); x += ka_align_up(nq * (long long)sizeof(KaSub), 16);
                S.priv.q[1] = (KaSub*)(pr + x); x += ka_align_up(nq * (long long)sizeof(KaSub), 16);
                S.priv.items[0] = (int2*)(pr + x); x += ka_align_up(ni * 8, 16);
                S.priv.items[1] = (int2*)(pr + x); x += ka_align_up(ni * 8, 16);
                S.priv.prog[0] = (int*)(pr + x); x += ka_align_up(ni * 4, 16);
                S.priv.prog[1] = (int*)(pr + x); x += ka_align_up(ni * 4, 16);
                for (int par = 0; par < 2; ++par)
                        for (int cls = 0; cls < 2; ++cls) { S.priv.pack[par][cls] = (int2*)(pr + x); x += ka_align_up(2 * nq * 8, 16); }
                S.priv.f = (KaState*)(pr + x); x += ka_align_up(n * 12, 16);
                S.priv.b = (KaState*)(pr + x); x += ka_align_up(n * 12, 16);
                o += (long long)S.G * pb;
        }
        S.best_coded = nullptr; S.best_srcA = nullptr; S.best_srcB = nullptr; S.sp_freq = nullptr; S.mrec = nullptr; S.inc = nullptr;
        if (refine) {
                S.best_coded = (int*)(base + o); o += ka_align_up(n * 4, 16);
                S.best_srcA = (int*)(base + o);  o += ka_align_up(n * 4, 16);
                S.best_srcB = (int*)(base + o);  o += ka_align_up(n * 4, 16);
                S.sp_freq = (int*)(base + o);    o += ka_align_up(n * 24 * 4, 16);
                S.mrec = (int2*)(base + o);      o += ka_align_up(n * 8, 16);
                S.inc = base + o;                o += ka_align_up(ka_inc_bytes(n), 16);
        }
        if (rec && !refine) { S.mrec = (int2*)(base + o); o += ka_align_up(n * 8, 16); }
        S.ent = nullptr; S.apos_r = nullptr; S.conf_r = nullptr; S.apos_c = nullptr; S.conf_c = nullptr; S.invj = nullptr; S.vote = nullptr;
        if (cons_maxlen > 0) {
                S.ent = (int2*)(base + o);    o += ka_align_up(n * 8 * KA_NB, 16);
                const long long KM = KA_NB - 1;                      // anchors
                S.apos_r = (int*)(base + o);  o += ka_align_up(KM * n * 4, 16);
                S.conf_r = (float*)(base + o); o += ka_align_up(KM * n * 4, 16);
                S.apos_c = (int*)(base + o);  o += ka_align_up(KM * n * 4, 16);
                S.conf_c = (float*)(base + o); o += ka_align_up(KM * n * 4, 16);
                S.invj = (int*)(base + o);    o += ka_align_up(KM * ((long long)cons_maxlen + 8) * 4, 16);
                S.vote = base + o;            o += ka_align_up(KM * n * 16, 16);
        }
        return o;
}

__device__ __host__ inline long long ka_scratch_bytes(long long la, long long lb, long long cons_maxlen, long long g = 1, bool refine = false, bool rec = false)
{
        const long long n = la + lb + 8;
        const long long nq = (la < lb ? la : lb) + 20;
        const long long ni = 2 * nq + 2 * (n / KA_STRIP1_ROWS + 2);
        long long b = 5 * ((n * 4 + 15) / 16 * 16) + 4 * ((n * 12 + 15) / 16 * 16)
             + 2 * ((nq * (long long)sizeof(KaSub) + 15) / 16 * 16)
             + 2 * ((ni * 8 + 15) / 16 * 16) + 2 * ((ni * 4 + 15) / 16 * 16)
             + 4 * ((2 * nq * 8 + 15) / 16 * 16) + 64;
        if (g > 1) b += g * ka_private_bytes(la, lb);
        if (refine) b += 3 * ((n * 4 + 15) / 16 * 16) + (n * 24 * 4 + 15) / 16 * 16 + (n * 8 + 15) / 16 * 16 + (ka_inc_bytes(n) + 15) / 16 * 16;
        if (rec && !refine) b += (n * 8 + 15) / 16 * 16;
        if (cons_maxlen > 0) b += (n * 8 * KA_NB + 15) / 16 * 16 + 4 * (((KA_NB - 1) * n * 4 + 15) / 16 * 16) + ((KA_NB - 1) * (cons_maxlen + 8) * 4 + 15) / 16 * 16 + ((KA_NB - 1) * n * 16 + 15) / 16 * 16;
        return b;
}

// The margins of a level-synchronous baseline trial (ka_meetup<.., REC>) in the reference's recursion order: sort the
// (key, margin) records by key in LDS (bitonic, padded to a power of two), then one thread adds them up in fp32 -- and
// keeps the first mlog_cap of them for the adaptive budget.  Returns false when there are more records than the buffer
// holds (the caller repeats the trial depth first).
#define KA_REC_SORT_CAP 8192
__device__ bool ka_margins_in_order(TaskShared& S, char* lds, const int cap = KA_REC_SORT_CAP)
{
        const int tid = threadIdx.x;
        const int n = S.ctl->nrec;
        int2* buf = (int2*)lds;
        int m = 1;
        while (m < n) m <<= 1;
        if (m > cap) return false;                                  // (uniform: n comes from the control block)
        for (int i = tid; i < m; i += KA_NT) buf[i] = (i < n) ? S.mrec[i] : make_int2(0x7fffffff, 0);
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = tid; i < m; i += KA_NT) {
                                const int l = i ^ j;
                                if (l > i) {
                                        const int2 a = buf[i], b = buf[l];
                                        const bool up = (i & k) == 0;
                                        if ((a.x > b.x) == up) { buf[i] = b; buf[l] = a; }
                                }
                        }
                        __syncthreads();
                }
        }
        if (S.mlog) for (int i = tid; i < min(n, S.mlog_cap); i += KA_NT) S.mlog[i] = __int_as_float(buf[i].y);
        if (tid == 0) {
                float sum = 0.0f;
                for (int i = 0; i < n; ++i) sum += __int_as_float(buf[i].y);
                S.rf.msum = sum; S.rf.mcount = n; S.rf.counter = 0;
        }
        __syncthreads();
        return true;
}

// ------------------------------------------------------------------------------------------
// The task kernel: one workgroup per task of the current guide-tree level.
// ------------------------------------------------------------------------------------------
// LEAN = true: a level whose tasks are all seq-seq (the guide tree's leaf level): 4 waves, no LDS
// ring, <=128 VGPRs -> four workgroups per CU instead of one.
// Returns 0 when this workgroup took part in the task to its end, 1 when the task failed (arena overflow),
// 2 when the workgroup was surplus to the task's cluster (the task is too small for all of them).
// Q1: the 8-wave kernels also carry the one-row-per-lane strip (ka_strip<.., Q = 1>) for tasks that own idle SIMDs.
template <bool LEAN, int NB, bool Q1 = false>
__device__ __forceinline__ int ka_task_body(const KaTreeDev& D, const int task, const int member, const int g_launch)
{
        // all LDS lives in the dynamic region (16-B aligned carve-outs, guide section 6 G17)
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        TaskShared& S = *(TaskShared*)ka_smem;
        float** s_dbg_p = (float**)(ka_smem + KA_LDS_DBG);
        float* tss = (float*)(ka_smem + KA_LDS_TSS);
        char* lds_waves = ka_smem + KA_LDS_WAVES;
#define s_dbg (*s_dbg_p)
        const KaTaskDesc T = D.tasks[task];
        const int tid = threadIdx.x;
        const long long tk0 = __builtin_amdgcn_s_memtime();
        long long tk1 = 0, tk2 = 0, tk3 = 0;

        if (tid == 0 && blockIdx.x == 0) KA_CRUMB(D.trace, 4, 1);
        if (tid == 0) {
                // (a chained launch reads what other workgroups of the SAME launch wrote: go past L1 / scalar cache)
                const int len_a = __hip_atomic_load(&D.node_len[T.a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int len_b = __hip_atomic_load(&D.node_len[T.b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S.watchdog = D.error; S.trace = D.trace; S.dbgskip = D.flags >> 16;
                S.prof = (D.timing && T.is_root) ? (D.timing + 8ll * (D.numseq - 1) + 48) : nullptr;
                S.len_a = len_a; S.len_b = len_b;
                S.profa = D.prof_arena + __hip_atomic_load(&D.node_prof[T.a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S.profb = D.prof_arena + __hip_atomic_load(&D.node_prof[T.b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S.subm = D.subm;
                S.gpo = T.gpo; S.gpe = T.gpe; S.tgpe = T.tgpe; S.soff = T.soff;
                S.s1 = nullptr; S.s2 = nullptr; S.p1 = nullptr; S.p2 = nullptr;
                S.sp_open = 0.0f; S.sp_ext = 0.0f; S.sp_text = 0.0f;
                int swapped = 0, kind;
                // operand selection and swap rules, aln_run.c:297-388
                if (T.nsip_a == 1 && T.nsip_b == 1) {
                        kind = KA_SS;
                        if (len_a < len_b) { S.s1 = D.codes + D.seq_off[T.a]; S.s2 = D.codes + D.seq_off[T.b]; }
                        else { swapped = 1; S.s1 = D.codes + D.seq_off[T.b]; S.s2 = D.codes + D.seq_off[T.a]; }
                } else if (T.nsip_a == 1) {
                        kind = KA_SP; swapped = 1;
                        S.s2 = D.codes + D.seq_off[T.a]; S.p1 = S.profb;
                        S.sp_open = T.gpo * (float)T.nsip_b; S.sp_ext = T.gpe * (float)T.nsip_b; S.sp_text = T.tgpe * (float)T.nsip_b;
                } else if (T.nsip_b == 1) {
                        kind = KA_SP;
                        S.s2 = D.codes + D.seq_off[T.b]; S.p1 = S.profa;
                        S.sp_open = T.gpo * (float)T.nsip_a; S.sp_ext = T.gpe * (float)T.nsip_a; S.sp_text = T.tgpe * (float)T.nsip_a;
                } else {
                        kind = KA_PP;
                        if (len_a < len_b) { S.p1 = S.profa; S.p2 = S.profb; }
                        else { swapped = 1; S.p1 = S.profb; S.p2 = S.profa; }
                }
                S.kind = kind; S.swapped = swapped;
                // p1 is profile b when swapped, else profile a; its gap terms scale with the other side's nsip
                S.p1_mult = swapped ? (float)T.nsip_a : (float)T.nsip_b;
                S.p2_mult = swapped ? (float)T.nsip_b : (float)T.nsip_a;
                S.La = swapped ? len_b : len_a;
                S.Lb = swapped ? len_a : len_b;
                // how many of the launched workgroups this task really uses (every member derives the
                // same number from the operand lengths): one CU saturates at about 8 strips in flight
                int g_eff = (S.La >= 1536) ? 8 : ((S.La >= 1152) ? 6 : ((S.La >= 768) ? 4 : ((S.La >= 320) ? 2 : 1)));
                // The strip shape: 64-row strips (one DP row per lane, about half the instructions per step) when the
                // cluster has a SIMD for every strip of the two top-level passes -- the number of strips in flight stays
                // about the same down the recursion (rows halve, passes double) -- else 128-row strips.
                int srows = KA_STRIP_ROWS;
                int q1_lvl = 0;
                if (Q1 && D.q1_mode == 4) {
                        // per level (ka_level_srows): profile-profile tasks with helper waves; the cluster as wide as the top level's
                        // 64-row strips want it, if the launch gave that many workgroups
                        if (D.hw_mode && kind == KA_PP) {
                                const int s1 = ka_strips_of(S.La / 2, KA_STRIP1_ROWS) + ka_strips_of(S.La - S.La / 2, KA_STRIP1_ROWS);
                                q1_lvl = 1;
                                g_eff = max(g_eff, min(g_launch, (s1 + 3) / 4));
                        }
                } else if (Q1 && D.q1_mode) {
                        const int s1 = ka_strips_of(S.La / 2, KA_STRIP1_ROWS) + ka_strips_of(S.La - S.La / 2, KA_STRIP1_ROWS);
                        const int g1 = (s1 + 3) / 4;
                        if (g1 <= g_launch || (D.q1_mode >= 2 && (s1 + 7) / 8 <= g_launch) || D.q1_mode >= 3) { srows = KA_STRIP1_ROWS; g_eff = g1; }
                }
                S.q1_lvl = q1_lvl;
                S.lvl_srows[0] = srows; S.lvl_srows[1] = srows;
                // LDS hand-over between neighbouring strips (ka_strip<.., HO>): profile-profile tasks of the 8-wave kernel, fast mode.
                // ho_mode >= 2: four strips per workgroup (one per SIMD) instead of three -- fewer hand-overs cross workgroups.
                S.ho_ok = (Q1 && NB == 0 && D.ho_mode && kind == KA_PP) ? 1 : 0;
                S.hw_ok = (Q1 && D.hw_mode && kind == KA_PP) ? D.hw_mode : 0;
                if (Q1 && kind == KA_PP) {
                        // Tasks with more top-level strips than the table's workgroups have SIMDs (rows beyond ~4000: nucleotide
                        // jobs) take a workgroup per four strips, up to what the launch gave them: 4096 x 2000 nt 104 -> 93 ms.
                        // (ho_mode 2, experiments: four strips per workgroup whatever the table says -- costs protein 7 %.)
                        const int s2 = ka_strips_of(S.La / 2, srows) + ka_strips_of(S.La - S.La / 2, srows);
                        g_eff = (S.ho_ok && D.ho_mode >= 2) ? max((s2 + 3) / 4, 1) : max(g_eff, (s2 + 3) / 4);
                }
                // experiments (KA_PER): strips per workgroup at the task's top level -> workgroups used
                if (Q1 && D.per_target > 0 && kind == KA_PP) {
                        const int s2 = ka_strips_of(S.La / 2, srows) + ka_strips_of(S.La - S.La / 2, srows);
                        g_eff = max((s2 + D.per_target - 1) / D.per_target, 1);
                }
                if (g_eff > g_launch) g_eff = g_launch;
                S.srows = srows;
                // wave-local subtrees (ka_subtree.h): every kernel shape has a region per wave behind the workgroup's scratch
                // exact task confidences (aln_run.c:391-395 adds the margins in recursion order): every meetup records its margin
                // with its place in that order; sorted and added up in fp32 after the recursion (ka_margins_in_order).  The
                // wave-local subtrees do not keep those records: off.
                S.rec_on = (D.flags & KA_FLAG_EXACT_CONFIDENCE) ? 1 : 0;
                S.sub_ok = (NB == 0 && D.sub_mode && !S.rec_on) ? 1 : 0;
                S.nres_t = (D.nres <= 5) ? 5 : ((D.nres <= 20) ? 20 : 23);
                S.sub_stride = LEAN ? KA_WAVE_LDS_LEAN : KA_WAVE_LDS;
                S.sub_base = LEAN ? (lds_waves + KA_LEAN_SCRATCH(KA_NT)) : lds_waves;
                S.mw_ok = D.mw_mode;
                S.sub_tm = (D.timing && (D.prof_task >= 0 ? task == D.prof_task : T.is_root)) ? 1 : 0;
                for (int x = 0; x < 7; ++x) S.sub_t[x] = 0;
                S.G = g_eff; S.member = member; S.bar_phase = 0;
                S.Gw = g_eff; S.member_w = member; S.split = 0;
                S.ctl = (g_eff == 1) ? &S.ctl_lds : (D.ctl + task);
                S.lctl = S.ctl;
                if (g_eff == 1) { S.ctl_lds.fail = 0; S.ctl_lds.bar = 0; S.ctl_lds.nrec = 0; }
                s_dbg = nullptr;
                if (member == 0) {
                        const long long need = ka_scratch_bytes(len_a, len_b, NB ? D.cons_maxlen : 0, g_eff, false, (D.flags & KA_FLAG_EXACT_CONFIDENCE) != 0);
                        const unsigned long long so = atomicAdd(&D.counters[1], (unsigned long long)need);
                        if ((long long)so + need > D.scratch_cap) { S.ctl->fail = 1; atomicExch(D.error, 2); }
                        // an earlier task of this run already failed (arena overflow): its outputs -- possibly this
                        // task's operands -- do not exist, and the host is going to repeat the run anyway
                        if (__hip_atomic_load(D.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) S.ctl->fail = 1;
                        S.ctl->scratch_off = (long long)so;
                        if (D.flags & KA_FLAG_DEBUG_ROWS) {
                                const unsigned long long nd = 6ull * (unsigned long long)(S.Lb + 1);
                                const unsigned long long d0 = atomicAdd(&D.counters[3], nd);
                                if ((long long)(d0 + nd) <= D.dbg_cap) { s_dbg = D.dbg_arena + d0; D.dbg_off[task] = (long long)d0; }
                                else { D.dbg_off[task] = -1; atomicExch(D.error, 4); }
                        }
                }
        }
        __syncthreads();
        if (S.member >= S.G) return 2;                       // surplus workgroup of an over-provisioned cluster
        ka_cluster_sync(S);
        if (S.ctl->fail) return 1;
        if (tid == 0) ka_carve(S, D.scratch + S.ctl->scratch_off, S.len_a, S.len_b, NB ? D.cons_maxlen : 0, false, S.rec_on != 0);

        // P1
        ka_build_tss(tss, D.subm, T.soff);
        __syncthreads();
        if (S.member == 0) {
                if (T.nsip_a == 1) ka_make_leaf_profile(S.profa, S.len_a, D.codes + D.seq_off[T.a], T.gpo, T.gpe, T.tgpe, tss);
                if (T.nsip_b == 1) ka_make_leaf_profile(S.profb, S.len_b, D.codes + D.seq_off[T.b], T.gpo, T.gpe, T.tgpe, tss);
        }
        // P1b: anchor positions of both operands (all workgroups of the cluster), then the bonus entries
        // of every DP row (the first one); the barrier below publishes them
        if (NB) {
                __syncthreads();
                ka_cons_votes<LEAN>(S, D, T, lds_waves, LEAN ? 0 : (long long)KA_NW * KA_WAVE_LDS);
                ka_cluster_sync(S);
                if (S.member == 0) ka_cons_entries(S, D);
        }
        ka_cluster_sync(S);
        tk1 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && blockIdx.x == 0) KA_CRUMB(D.trace, 4, 2);

        // P2
        if (LEAN || S.kind == KA_SS) ka_hirschberg<KA_SS, 23, NB, false, Q1>(S, s_dbg, lds_waves, tss, D.trace);
        else if (S.kind == KA_SP) ka_hirschberg<KA_SP, 23, NB, false, Q1>(S, s_dbg, lds_waves, tss, D.trace);
        else if (D.nres <= 5) ka_hirschberg<KA_PP, 5, NB, false, Q1, Q1 && NB == 0, Q1>(S, s_dbg, lds_waves, tss, D.trace);
        // no B / Z / X in the job (the usual case): every profile's counts [20..22] are zero and the reference skips
        // zero counts (aln_profileprofile.c:70-77) -- 20 terms per cell instead of 23
        else if (D.nres <= 20) ka_hirschberg<KA_PP, 20, NB, false, Q1, Q1 && NB == 0, Q1>(S, s_dbg, lds_waves, tss, D.trace);
        else ka_hirschberg<KA_PP, 23, NB, false, Q1, Q1 && NB == 0, Q1>(S, s_dbg, lds_waves, tss, D.trace);
        __syncthreads();
        // exact confidence: the cluster's last barrier (inside ka_hirschberg) has published every member's records
        bool conf_exact = false;
        if (S.rec_on && S.member == 0 && S.n_levels < KA_REC_DEPTH) {   // (deeper: the keys no longer tell the levels apart)
                if (tid == 0) { S.mlog = nullptr; S.mlog_cap = 0; }
                __syncthreads();
                const int cap = (int)min((long long)KA_REC_SORT_CAP, (LEAN ? (long long)KA_NW * KA_WAVE_LDS_LEAN + KA_LEAN_SCRATCH(KA_NT) : (long long)KA_NW * KA_WAVE_LDS) / 8);
                conf_exact = ka_margins_in_order(S, lds_waves, cap);
        }
        tk2 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && blockIdx.x == 0) KA_CRUMB(D.trace, 4, 3);
#undef s_dbg

        // P3 (the cluster's first workgroup)
        if (S.member == 0) {
                ka_code_path(S, (int*)lds_waves);
                if (tid == 0) {
                        const int alnlen = S.ctl->alnlen;
                        const unsigned long long pn = (unsigned long long)alnlen + 2;
                        const unsigned long long po = atomicAdd(&D.counters[2], pn);
                        if ((long long)(po + pn) > D.path_cap) { S.ctl->fail = 1; atomicExch(D.error, 3); }
                        S.ctl->path_off = (long long)po;
                        S.ctl->newp_off = -1;
                        D.node_len[T.c] = alnlen;
                        if (!T.is_root) {
                                const unsigned long long fn = pn * 64ull;
                                const unsigned long long fo = atomicAdd(&D.counters[0], fn);
                                if ((long long)(fo + fn) > D.prof_cap) { S.ctl->fail = 1; atomicExch(D.error, 1); }
                                else { S.ctl->newp_off = (long long)fo; D.node_prof[T.c] = (long long)fo; }
                        }
                        ka_task_rec r;
                        r.a = T.a; r.b = T.b; r.c = T.c;
                        r.len_a = S.len_a; r.len_b = S.len_b; r.nsip_a = T.nsip_a; r.nsip_b = T.nsip_b;
                        r.plen = alnlen; r.kind = S.kind; r.swapped = S.swapped;
                        r.meet = S.ctl->top_meet; r.transition = S.ctl->top_tr;
                        r.path_off = (int)po;
                        r.gap_scale = T.gap_scale; r.subm_off = T.soff;
                        r.score = S.ctl->top_score;
                        r.confidence = (S.ctl->mcount > 0) ? (float)S.ctl->msum / (float)S.ctl->mcount : 0.0f;
                        // (the reference: m->margin_sum / (float)m->margin_count, both summed in recursion order, aln_run.c:391-395)
                        if (conf_exact) r.confidence = (S.rf.mcount > 0) ? S.rf.msum / (float)S.rf.mcount : 0.0f;
                        r.prof_hash = 0; r.fhash = 0; r.bhash = 0;
                        D.recs[task] = r;
                }
        }
        ka_cluster_sync(S);
        tk3 = __builtin_amdgcn_s_memtime();
        if (S.ctl->fail) return 1;

        // P4 (all workgroups of the cluster)
        if (tid == 0) {
                S.path_dst = D.path_arena + S.ctl->path_off;
                S.newp = (S.ctl->newp_off >= 0) ? (D.prof_arena + S.ctl->newp_off) : nullptr;
        }
        __syncthreads();
        const int alnlen = S.ctl->alnlen;
        if (S.member == 0) for (int i = tid; i < alnlen + 2; i += KA_NT) S.path_dst[i] = S.coded[i];
        if (S.newp) ka_update_profile(S, D, T, alnlen);
        if ((NB && !T.is_root) || (D.flags & KA_FLAG_DEVICE_GAPS)) ka_update_colof(S, D, T, alnlen);
        if (D.timing && S.member == 0) {
                __syncthreads();
                if (tid == 0) {
                        long long* tm = D.timing + 8ll * task;
                        tm[0] = tk1 - tk0; tm[1] = tk2 - tk1; tm[2] = tk3 - tk2; tm[3] = __builtin_amdgcn_s_memtime() - tk3;
                        tm[4] = S.t_pass; tm[5] = S.t_meet; tm[6] = S.n_levels | (S.G << 8) | (g_launch << 16); tm[7] = (long long)S.La * S.Lb;
                        if (D.prof_task >= 0 ? task == D.prof_task : T.is_root) {
                                long long* lv = D.timing + 8ll * (D.numseq - 1);
                                for (int l = 0; l < 16; ++l) {
                                        lv[3 * l] = l < S.n_levels ? S.lvl_n[l] : 0;
                                        lv[3 * l + 1] = l < S.n_levels ? S.lvl_pass[l] : 0;
                                        lv[3 * l + 2] = l < S.n_levels ? S.lvl_meet[l] : 0;
                                }
                                // (the leading workgroup's wave-local subtrees: the last seven of the 48 level slots)
                                if (S.n_levels <= 13) for (int x = 0; x < 7; ++x) lv[41 + x] = (long long)S.sub_t[x];
                        }
                }
        }
        return 0;
}

// ------------------------------------------------------------------------------------------
// Refinement pass (refine_alignment, aln_refine.c:36-346): one workgroup per edge.  Operand preparation as in
// ka_task_body; then refine_edge's trials -- trial 0 without flips, trials 1..4 with the baseline's mean margin as the
// flip threshold (replay_edge: trial 0 only) -- each one a depth-first recursion (ka_hirschberg_dfs), coded with
// convert_raw_path and, on a refined edge, scored with compute_sp_score; the first best trial's path makes the merged
// profile (update_n honours its open / extend / close flags) and moves the members' columns.
// ------------------------------------------------------------------------------------------
//
// Trials in parallel (G = 2 or 4 workgroups per refined edge, when the level leaves CUs idle): the flip trials only
// depend on the baseline's mean margin, so every member runs the baseline itself (same operands, same result -- nothing
// to exchange), then member m the flip trials m+1, m+1+G, ...; each reports its best (score, trial, margin sum /
// count) in the task's control block, one barrier, and everybody picks the winner the way the serial loop does (highest
// score, the earliest trial among equals; aln_refine.c:247-253).  The member that ran the winning trial finishes the
// task (record, path, merged profile, columns); the others leave.
template <int NB>
__device__ __forceinline__ void ka_task_body_refine(const KaTreeDev& D, const int task, const int member, const int G)
{
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        TaskShared& S = *(TaskShared*)ka_smem;
        float* tss = (float*)(ka_smem + KA_LDS_TSS);
        char* lds_waves = ka_smem + KA_LDS_WAVES;
        const KaTaskDesc T = D.tasks[task];
        const int tid = threadIdx.x;
        // KA_FLAG_TIMING: cycles of this member in [0] preparation, [1] SP tables, [2] recursions, [3] path coding, [4] SP scoring,
        // [5] waiting for the other members, [6] record / path / merged profile / columns, [7] DP cells
        long long tq[7] = {0, 0, 0, 0, 0, 0, 0};
        long long tlast = __builtin_amdgcn_s_memtime();
        auto lap = [&](int k) { const long long now = __builtin_amdgcn_s_memtime(); tq[k] += now - tlast; tlast = now; };
        if (tid == 0) {
                const int len_a = D.node_len[T.a], len_b = D.node_len[T.b];
                S.watchdog = D.error; S.trace = D.trace; S.dbgskip = ((D.wdfs & 1) ? 0 : 1) | ((D.wdfs & 8) ? 0 : 2); S.prof = nullptr;   // (dbgskip here: bit 0 "no wave-local subtrees", bit 1 "... not in LDS")
                S.len_a = len_a; S.len_b = len_b;
                S.profa = D.prof_arena + D.node_prof[T.a];
                S.profb = D.prof_arena + D.node_prof[T.b];
                S.subm = D.subm;
                S.gpo = T.gpo; S.gpe = T.gpe; S.tgpe = T.tgpe; S.soff = T.soff;
                S.s1 = nullptr; S.s2 = nullptr; S.p1 = nullptr; S.p2 = nullptr;
                S.sp_open = 0.0f; S.sp_ext = 0.0f; S.sp_text = 0.0f;
                int swapped = 0, kind;
                if (T.nsip_a == 1 && T.nsip_b == 1) {                // operand selection and swap rules, aln_refine.c:476-560
                        kind = KA_SS;
                        if (len_a < len_b) { S.s1 = D.codes + D.seq_off[T.a]; S.s2 = D.codes + D.seq_off[T.b]; }
                        else { swapped = 1; S.s1 = D.codes + D.seq_off[T.b]; S.s2 = D.codes + D.seq_off[T.a]; }
                } else if (T.nsip_a == 1) {
                        kind = KA_SP; swapped = 1;
                        S.s2 = D.codes + D.seq_off[T.a]; S.p1 = S.profb;
                        S.sp_open = T.gpo * (float)T.nsip_b; S.sp_ext = T.gpe * (float)T.nsip_b; S.sp_text = T.tgpe * (float)T.nsip_b;
                } else if (T.nsip_b == 1) {
                        kind = KA_SP;
                        S.s2 = D.codes + D.seq_off[T.b]; S.p1 = S.profa;
                        S.sp_open = T.gpo * (float)T.nsip_a; S.sp_ext = T.gpe * (float)T.nsip_a; S.sp_text = T.tgpe * (float)T.nsip_a;
                } else {
                        kind = KA_PP;
                        if (len_a < len_b) { S.p1 = S.profa; S.p2 = S.profb; }
                        else { swapped = 1; S.p1 = S.profb; S.p2 = S.profa; }
                }
                S.kind = kind; S.swapped = swapped;
                S.p1_mult = swapped ? (float)T.nsip_a : (float)T.nsip_b;
                S.p2_mult = swapped ? (float)T.nsip_b : (float)T.nsip_a;
                S.La = swapped ? len_b : len_a;
                S.Lb = swapped ? len_a : len_b;
                S.G = 1; S.member = 0; S.bar_phase = 0; S.Gw = 1; S.member_w = 0; S.split = 0; S.srows = KA_STRIP_ROWS; S.q1_lvl = 0; S.lvl_srows[0] = KA_STRIP_ROWS; S.lvl_srows[1] = KA_STRIP_ROWS;
                S.sub_ok = 0; S.rec_on = 0; S.nres_t = 23; S.sub_stride = 0; S.sub_base = nullptr; S.sub_tm = 0; S.mw_ok = 0;   // (flip trials decide in recursion order: no wave-local subtrees)
                S.ctl = &S.ctl_lds; S.lctl = S.ctl;
                S.ctl_lds.fail = 0; S.ctl_lds.bar = 0;
                const long long need = ka_scratch_bytes(len_a, len_b, NB ? D.cons_maxlen : 0, 1, true);
                const unsigned long long so = atomicAdd(&D.counters[1], (unsigned long long)need);
                if ((long long)so + need > D.scratch_cap) { S.ctl->fail = 1; atomicExch(D.error, 2); }
                if (__hip_atomic_load(D.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) S.ctl->fail = 1;
                S.ctl->scratch_off = (long long)so;
                if (!S.ctl->fail) ka_carve(S, D.scratch + so, len_a, len_b, NB ? D.cons_maxlen : 0, true);
        }
        __syncthreads();
        if (S.ctl->fail) return;

        // P1
        ka_build_tss(tss, D.subm, T.soff);
        __syncthreads();
        if (T.nsip_a == 1) ka_make_leaf_profile(S.profa, S.len_a, D.codes + D.seq_off[T.a], T.gpo, T.gpe, T.tgpe, tss);
        if (T.nsip_b == 1) ka_make_leaf_profile(S.profb, S.len_b, D.codes + D.seq_off[T.b], T.gpo, T.gpe, T.tgpe, tss);
        if (NB) {
                __syncthreads();
                ka_cons_votes<false>(S, D, T, lds_waves, (long long)KA_NW * KA_WAVE_LDS);
                __syncthreads();
                ka_cons_entries(S, D);
        }
        // modes: 1 KALIGN_REFINE_ALL, 2 _CONFIDENT (T.refine marks the edges), 3 _INLINE (do_align_inline_refine,
        // aln_run.c:515-790: three trials on every edge, first-pass path coding, confidence = the best SP score),
        // 4 = one depth-first trial with first-pass coding (the first pass with the reference's exact confidence sums)
        const bool inline_mode = D.refine_mode == 3;
        const bool refine_it = D.refine_mode == 1 || inline_mode || (D.refine_mode == 2 && T.refine != 0);
        int n_trials = inline_mode ? max(D.refine_trials, 1) : refine_it ? 5 : 1;      // (create_msa_tree_inline_refine takes any number of trials, aln_run.c:448-475)
        // --adaptive-budget (aln_refine.c:187-193, 255-282; refine_edge only): the baseline's margins are kept (the first
        // max(64, min(len_a, len_b) + 1) of them) and the number of trials, 1 .. 8, follows from the share of meetups
        // whose margin is below a quarter of the mean
        const bool adaptive_it = D.refine_adaptive && refine_it && !inline_mode;
        lap(0);
        if (refine_it) ka_sp_build(S, D, T);
        __syncthreads();
        lap(1);

        // P2: the trials
        float best_sp = -KA_F, avg_margin = 0.0f, best_msum = 0.0f;
        int best_mcount = 0, best_k = 0;
        int top_meet0 = -1, top_tr0 = -1;                            // the record carries the baseline's top-level meetup
        bool inc_ok = false, inc_listed = false;                     // incremental flip trials: tables built / uncertain meetups listed
        float top_score0 = 0.0f;
        const int Gt = (refine_it && G > 1) ? G : 1;                 // members that share this edge's flip trials
        if (member >= Gt) return;
        for (int k = 0; k < n_trials; ++k) {
                if (k > 0 && (k - 1) % Gt != member) continue;       // another member's trial
                if (tid == 0) {
                        S.rf.thr = (k == 0) ? 0.0f : avg_margin; S.rf.trial = k; S.rf.stride = max(n_trials - 1, 1);
                        S.mlog = (adaptive_it && k == 0) ? (float*)S.raw2 : nullptr;       // (raw2 is idle until the path is coded)
                        S.mlog_cap = min(max(min(S.len_a, S.len_b) + 1, 64), S.len_a + S.len_b + 8);
                        S.adapt_trials = 0;
                }
                __syncthreads();
                // The baseline trial has no flips: its sub-problems are independent and run level-synchronously (all waves busy,
                // a fifth of the depth-first time); the margins are put back into recursion order afterwards.
                bool done = false;
                if (k == 0 && (D.wdfs & 2) && S.La < (1 << 17)) {
                        inc_ok = false;
                        if (tid == 0) S.ctl->nrec = 0;
                        __syncthreads();
                        if (S.kind == KA_SS) ka_hirschberg<KA_SS, 23, NB, true>(S, nullptr, lds_waves, tss, D.trace);
                        else if (S.kind == KA_SP) ka_hirschberg<KA_SP, 23, NB, true>(S, nullptr, lds_waves, tss, D.trace);
                        else if (D.nres <= 5) ka_hirschberg<KA_PP, 5, NB, true>(S, nullptr, lds_waves, tss, D.trace);
                        else ka_hirschberg<KA_PP, 23, NB, true>(S, nullptr, lds_waves, tss, D.trace);
                        __syncthreads();
                        done = ka_margins_in_order(S, lds_waves);
                        // the flip trials re-run only the subtrees they flip (ka_trial_incremental)
                        inc_ok = done && (n_trials > 1 || adaptive_it) && (D.wdfs & 4) && S.inc != nullptr;
                        if (inc_ok) ka_inc_build(S, lds_waves);
                }
                if (k > 0 && inc_ok) {
                        if (!inc_listed) { ka_inc_uncertain(S, avg_margin); inc_listed = true; }
                        if (S.kind == KA_SS) ka_trial_incremental<KA_SS, 23, NB>(S, lds_waves, tss);
                        else if (S.kind == KA_SP) ka_trial_incremental<KA_SP, 23, NB>(S, lds_waves, tss);
                        else if (D.nres <= 5) ka_trial_incremental<KA_PP, 5, NB>(S, lds_waves, tss);
                        else ka_trial_incremental<KA_PP, 23, NB>(S, lds_waves, tss);
                        done = true;
                }
                if (!done) {
                        if (S.kind == KA_SS) ka_hirschberg_dfs<KA_SS, 23, NB>(S, lds_waves, tss, k == 0);
                        else if (S.kind == KA_SP) ka_hirschberg_dfs<KA_SP, 23, NB>(S, lds_waves, tss, k == 0);
                        else if (D.nres <= 5) ka_hirschberg_dfs<KA_PP, 5, NB>(S, lds_waves, tss, k == 0);
                        else ka_hirschberg_dfs<KA_PP, 23, NB>(S, lds_waves, tss, k == 0);
                }
                __syncthreads();
                lap(2);
                if (k == 0 && adaptive_it) {
                        const int mc = S.rf.mcount;
                        if (mc > 0) {
                                const float vu = (S.rf.msum / (float)mc) * 0.25F;
                                const int seen = min(mc, S.mlog_cap);
                                int mine = 0;
                                for (int i = tid; i < seen; i += KA_NT) mine += (S.mlog[i] < vu) ? 1 : 0;
                                if (mine) atomicAdd(&S.adapt_trials, mine);
                                __syncthreads();
                                const float frac = (float)S.adapt_trials / (float)mc;
                                n_trials = 1 + (int)(7.0F * frac + 0.5F);
                        }
                        __syncthreads();
                }
                if (k == 0) { top_meet0 = S.ctl->top_meet; top_tr0 = S.ctl->top_tr; top_score0 = S.ctl->top_score; }
                if (D.refine_mode >= 3) ka_code_path(S, (int*)lds_waves);     // add_gap_info_to_path_n (aln_run.c:713)
                else ka_code_path_refine(S, (int*)lds_waves);                 // convert_raw_path (aln_refine.c:243)
                lap(3);
                const float tr_msum = S.rf.msum;
                const int tr_mcount = S.rf.mcount;
                bool take = true;
                if (refine_it) {
                        ka_sp_score(S, D, T, lds_waves);
                        take = S.sp_value > best_sp;
                        if (take) best_sp = S.sp_value;
                        lap(4);
                }
                if (take) {
                        best_msum = tr_msum; best_mcount = tr_mcount; best_k = k;
                        const int n = S.coded[0] + 2;
                        for (int i = tid; i < n; i += KA_NT) { S.best_coded[i] = S.coded[i]; S.best_srcA[i] = S.srcA[i]; S.best_srcB[i] = S.srcB[i]; }
                }
                if (k == 0 && tr_mcount > 0) avg_margin = tr_msum / (float)tr_mcount;
                __syncthreads();
        }
        if (Gt > 1) {
                // report, meet, pick the winner (every member computes the same answer)
                KaCtl* C = D.ctl + task;
                int* slot = (int*)&C->lvl[0];                         // 4 ints per member: score, trial, margin sum, margin count
                if (tid == 0) {
                        slot[4 * member + 0] = __float_as_int(best_sp); slot[4 * member + 1] = best_k;
                        slot[4 * member + 2] = __float_as_int(best_msum); slot[4 * member + 3] = best_mcount;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_fetch_add(&C->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        int spins = 0;
                        while (__hip_atomic_load(&C->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)Gt) {
                                __builtin_amdgcn_s_sleep(8);
                                if (ka_spin_expired(D.error, ++spins, 1 << 24, 6, true)) break;    // (a member that failed never arrives)
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        float wsp = -KA_F; int wk = 0x7fffffff, wm = 0;
                        for (int m = 0; m < Gt; ++m) {
                                const float sp = __int_as_float(__hip_atomic_load(&slot[4 * m + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                const int kk = __hip_atomic_load(&slot[4 * m + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (sp > wsp || (sp == wsp && kk < wk)) { wsp = sp; wk = kk; wm = m; }
                        }
                        // the baseline (trial 0) is every member's own: when it wins, member 0 finishes the task
                        S.dfs_valid = (wk == 0) ? (member == 0) : (wm == member);
                }
                __syncthreads();
                lap(5);
                if (!S.dfs_valid) return;
        }
        {
                const int n = S.best_coded[0] + 2;
                for (int i = tid; i < n; i += KA_NT) { S.coded[i] = S.best_coded[i]; S.srcA[i] = S.best_srcA[i]; S.srcB[i] = S.best_srcB[i]; }
                if (tid == 0) S.ctl->alnlen = S.best_coded[0];
        }
        __syncthreads();

        // P3: output slots and the task record
        if (tid == 0) {
                const int alnlen = S.ctl->alnlen;
                const unsigned long long pn = (unsigned long long)alnlen + 2;
                const unsigned long long po = atomicAdd(&D.counters[2], pn);
                if ((long long)(po + pn) > D.path_cap) { S.ctl->fail = 1; atomicExch(D.error, 3); }
                S.ctl->path_off = (long long)po;
                S.ctl->newp_off = -1;
                D.node_len[T.c] = alnlen;
                if (!T.is_root) {
                        const unsigned long long fn = pn * 64ull;
                        const unsigned long long fo = atomicAdd(&D.counters[0], fn);
                        if ((long long)(fo + fn) > D.prof_cap) { S.ctl->fail = 1; atomicExch(D.error, 1); }
                        else { S.ctl->newp_off = (long long)fo; D.node_prof[T.c] = (long long)fo; }
                }
                ka_task_rec r;
                r.a = T.a; r.b = T.b; r.c = T.c;
                r.len_a = S.len_a; r.len_b = S.len_b; r.nsip_a = T.nsip_a; r.nsip_b = T.nsip_b;
                r.plen = alnlen; r.kind = S.kind; r.swapped = S.swapped;
                r.meet = top_meet0; r.transition = top_tr0;
                r.path_off = (int)po;
                r.gap_scale = T.gap_scale; r.subm_off = T.soff;
                r.score = top_score0;
                r.confidence = (best_mcount > 0) ? best_msum / (float)best_mcount : 0.0f;
                if (inline_mode) r.confidence = best_sp;                      // aln_run.c:742
                r.prof_hash = 0; r.fhash = 0; r.bhash = 0;
                D.recs[task] = r;
        }
        __syncthreads();
        if (S.ctl->fail) return;

        // P4
        if (tid == 0) {
                S.path_dst = D.path_arena + S.ctl->path_off;
                S.newp = (S.ctl->newp_off >= 0) ? (D.prof_arena + S.ctl->newp_off) : nullptr;
        }
        __syncthreads();
        const int alnlen = S.ctl->alnlen;
        for (int i = tid; i < alnlen + 2; i += KA_NT) S.path_dst[i] = S.coded[i];
        if (S.newp) ka_update_profile(S, D, T, alnlen);
        ka_update_colof(S, D, T, alnlen);
        if (D.timing) {
                __syncthreads();
                lap(6);
                if (tid == 0) {
                        long long* tm = D.timing + 8ll * task;
                        for (int x = 0; x < 7; ++x) tm[x] = tq[x];
                        tm[7] = (long long)S.La * S.Lb;
                }
        }
}

// Entry of the task kernels.  blocks[b] = (task, member | launched cluster size << 8); task < 0: padding.
//
// chain != 0: the launch covers the first guide-tree level with at most one task per CU AND everything above
// it.  Every workgroup starts as a one-workgroup cluster on a task of that level; when a task is done its
// cluster moves up the tree: the clusters of the two children meet at the parent's KaJoin -- the first to
// arrive waits, the last one leads, and both together (up to KA_MAX_G workgroups) run the parent.  Tasks start
// as soon as both operands exist instead of at the next launch, clusters grow as the tree narrows, and the
// whole upper tree is one launch.  All workgroups are resident from the start (<= one per CU), so the waits
// cannot starve anybody; they are bounded all the same (device watchdog).
#define KA_MAX_G 16
template <bool LEAN, int NB>
__device__ __forceinline__ void ka_task_entry(const KaTreeDev& D, const int2* __restrict__ blocks, const int chain)
{
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        TaskShared& S = *(TaskShared*)ka_smem;
        const int2 blk = blocks[blockIdx.x];
        int task = blk.x;
        if (task < 0) return;
        int member = blk.y & 0xff, g = blk.y >> 8;
        const int tid = threadIdx.x;
        while (true) {
                const int st = ka_task_body<LEAN, NB, !LEAN>(D, task, member, g);
                if (!chain || st == 1) return;
                // A workgroup the task had no use for stays with its cluster: it skips the task, waits for the cluster's
                // role at the parent and moves up with it -- a bigger task further up may need it (in a chain-like
                // tree clusters never merge, so a workgroup that left would be gone for good).
                const bool surplus = st == 2;
                const int parent = D.tasks[task].parent;
                if (parent < 0) return;
                KaJoin* J = D.join + parent;
                KaJoin* Jc = D.join + task;
                if (!surplus) {
                        // everything this cluster wrote for the task (profile, node_len, colof) is released ...
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        ka_cluster_sync(S);
                        if (S.member == 0 && tid == 0) {
                                const unsigned int need = (unsigned int)D.tasks[parent].chain_need;
                                // clusters are counted at their launched size g: surplus members are still with them
                                __hip_atomic_fetch_add(&J->sum_g, (unsigned int)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const unsigned int slot = __hip_atomic_fetch_add(&J->arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                                if (slot + 1 == need) {
                                        const unsigned int tot = __hip_atomic_load(&J->sum_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        const int gp = (int)(tot < (unsigned int)D.max_g ? tot : (unsigned int)D.max_g);
                                        J->join_base = g; J->join_g = gp;
                                        __hip_atomic_store(&J->go, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                                        __hip_atomic_store(&Jc->role, 1 | (gp << 8), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                                } else {
                                        __hip_atomic_store(&Jc->role, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                                }
                        }
                        // ... and the role of this cluster at the parent is published to all of its workgroups
                        ka_cluster_sync(S);
                }
                if (tid == 0) {
                        int role = __hip_atomic_load(&Jc->role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (surplus) {
                                int spins = 0;
                                while (role == 0) {
                                        __builtin_amdgcn_s_sleep(32);
                                        if (ka_spin_expired(S.watchdog, ++spins, (1 << 21) * max(1, min(D.tasks[parent].wait_mult, 64)), 6, true)) break;
                                        role = __hip_atomic_load(&Jc->role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                        }
                        int nm, ng;
                        if ((role & 0xff) == 1) {
                                nm = S.member; ng = role >> 8;
                        } else {
                                int spins = 0;
                                while (__hip_atomic_load(&J->go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                        __builtin_amdgcn_s_sleep(32);
                                        // ~2 s per unit of wait_mult (the host scales it with the DP cells of the subtrees that
                                        // meet here: a healthy sibling of a huge job may take longer): then the host re-plans without joins
                                        if (ka_spin_expired(S.watchdog, ++spins, (1 << 21) * max(1, min(D.tasks[parent].wait_mult, 64)), 6, true)) break;
                                }
                                nm = __hip_atomic_load(&J->join_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + S.member;
                                ng = __hip_atomic_load(&J->join_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        S.next_member = nm; S.next_g = ng;
                }
                __syncthreads();
                member = S.next_member; g = S.next_g;
                __syncthreads();
                if (member >= g || __hip_atomic_load(S.watchdog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                task = parent;
        }
}

// Queued launch (the guide-tree levels between the seq-seq leaves and the chained launch, which hold more tasks than
// the GPU has workgroup slots): ONE launch over all of those levels.  Every workgroup pulls the next task of a list
// ordered by level; a task whose operands come from the same launch waits for their producers' done flags
// (KaJoin::go).  A producer was pulled before its consumer, so it is already running on a resident workgroup: the
// wait cannot deadlock, whatever the residency (no co-scheduling assumption, unlike the chained launch).  No launch
// boundary between levels: the tail of one level overlaps the head of the next.
template <bool LEAN, int NB>
__device__ __forceinline__ void ka_task_queue_entry(const KaTreeDev& D, const int2* __restrict__ order, const int n)
{
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        TaskShared& S = *(TaskShared*)ka_smem;
        const int tid = threadIdx.x;
        // n == 0: not a queue -- one workgroup per entry of `order` (a per-level launch); one body, one call site
        while (true) {
                int task, member = 0, g = 1;
                if (n > 0) {
                        __syncthreads();
                        if (tid == 0) S.next_member = (int)atomicAdd(&D.counters[4], 1ull);
                        __syncthreads();
                        const int qi = S.next_member;
                        if (qi >= n) return;
                        task = order[qi].x;
                        if (tid == 0) {
                                const int dep[2] = { D.tasks[task].qa, D.tasks[task].qb };
                                bool waited = false;
                                for (int k = 0; k < 2; ++k) {
                                        if (dep[k] < 0) continue;
                                        int spins = 0;
                                        while (__hip_atomic_load(&D.join[dep[k]].go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                                __builtin_amdgcn_s_sleep(16);
                                                // (the bound scales with the DP cells below the producer, like the joins of the chained launch)
                                                if (ka_spin_expired(D.error, ++spins, (1 << 22) * max(1, min(D.tasks[dep[k]].wait_mult, 64)), 6, true)) break;
                                        }
                                        waited = true;
                                }
                                if (waited) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        }
                        __syncthreads();
                } else {
                        const int2 blk = order[blockIdx.x];
                        task = blk.x;
                        if (task < 0) return;
                        member = blk.y & 0xff; g = blk.y >> 8;
                }
                ka_task_body<LEAN, NB>(D, task, member, g);
                if (n == 0) return;
                // everything this workgroup wrote for the task (profile, node_len / node_prof, colof) is released, then the
                // done flag goes up -- also after a failed task: its consumers must not hang, the host repeats the run anyway
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(&D.join[task].go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}

// The kernels are compiled as four translation units from this one file (-DKA_UNIT=0..3, csrc/Makefile): every
// instantiation of ka_task_body takes about a minute of compile time, the units build in parallel.
//   unit 0: ka_task_kernel            unit 1: ka_task_kernel_cons
//   unit 2: the two half kernels      unit 3: the two lean kernels + ka_pair_kernel
#ifndef KA_UNIT
#error "compile with -DKA_UNIT=0..5 (see csrc/Makefile)"
#endif

// more than 64 KiB of dynamic LDS needs an explicit opt-in per kernel
template <typename K>
static hipError_t ka_optin(K kernel, int bytes, bool* done)
{
        if (*done) return hipSuccess;
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) *done = true;
        return e;
}

#if KA_UNIT == 0
__global__ __launch_bounds__(KA_BLOCK) void ka_task_kernel(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<false, 0>(D, blocks, chain);
}
extern "C" void ka_unit0_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel, KA_LDS_TOTAL, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_task_kernel, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, chain);
}
extern "C" long long ka_scratch_bytes_host(long long la, long long lb, long long cons_maxlen) { return ka_scratch_bytes(la, lb, cons_maxlen); }
extern "C" long long ka_ctl_bytes_host(void) { return (long long)sizeof(KaCtl); }
extern "C" int ka_max_g_host(void) { return KA_MAX_G; }
#endif

#if KA_UNIT == 4
// refinement pass (units 4 and 5: one kernel each -- they are the longest compiles of the library, side by side they halve the
// build's critical path): one workgroup per edge, per-level launches
extern "C" void ka_unit5_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream);
__global__ __launch_bounds__(KA_BLOCK) void ka_refine_kernel(const KaTreeDev D, const int2* __restrict__ blocks, const int unused)
{
        const int2 blk = blocks[blockIdx.x];
        if (blk.x >= 0) ka_task_body_refine<0>(D, blk.x, blk.y & 0xff, blk.y >> 8);
}
extern "C" void ka_unit4_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, hipStream_t stream)
{
        static bool done0 = false;
        if (cons) { ka_unit5_launch(D, blocks_dev, nblocks, stream); return; }
        if (ka_optin(ka_refine_kernel, KA_LDS_TOTAL, &done0) != hipSuccess) return;
        hipLaunchKernelGGL(ka_refine_kernel, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, 0);
}
#endif

#if KA_UNIT == 5
// the refinement pass with the anchor-consistency bonus
__global__ __launch_bounds__(KA_BLOCK) void ka_refine_kernel_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int unused)
{
        const int2 blk = blocks[blockIdx.x];
        if (blk.x >= 0) ka_task_body_refine<KA_NB>(D, blk.x, blk.y & 0xff, blk.y >> 8);
}
extern "C" void ka_unit5_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream)
{
        static bool done1 = false;
        if (ka_optin(ka_refine_kernel_cons, KA_LDS_TOTAL, &done1) != hipSuccess) return;
        hipLaunchKernelGGL(ka_refine_kernel_cons, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, 0);
}
#endif

#if KA_UNIT == 1
// the same with the anchor-consistency bonus (default mode of the reference's CLI)
__global__ __launch_bounds__(KA_BLOCK) void ka_task_kernel_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<false, KA_NB>(D, blocks, chain);
}
extern "C" void ka_unit1_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel_cons, KA_LDS_TOTAL, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_task_kernel_cons, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, chain);
}
#endif

#if KA_UNIT == 2
// Throughput variant for levels with more tasks than CUs (big trees, forests): 4 waves, 4 rings -> TWO workgroups per
// CU.  The four strip waves of one task keep a CU's SIMDs busy only part of the time (pipeline fill and drain, deep
// recursion levels, meetups); a second resident task fills the holes.  Latency per task is no better -- levels
// with at most one task per CU use the 8-wave kernel.
// nqueue > 0: a queued launch over the first nqueue entries of `blocks` (ka_task_queue_entry); else one workgroup per entry
__global__ __launch_bounds__(KA_HALF_BLOCK, 2) void ka_task_kernel_half(const KaTreeDev D, const int2* __restrict__ blocks, const int nqueue)
{
        ka_task_queue_entry<false, 0>(D, blocks, nqueue);
}
__global__ __launch_bounds__(KA_HALF_BLOCK, 2) void ka_task_kernel_half_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int nqueue)
{
        ka_task_queue_entry<false, KA_NB>(D, blocks, nqueue);
}
// nqueue > 0: `nblocks` workgroups share the `nqueue` tasks listed in blocks_dev
extern "C" void ka_unit2_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, int nqueue, hipStream_t stream)
{
        static bool done0 = false, done1 = false;
        if (cons) {
                if (ka_optin(ka_task_kernel_half_cons, KA_LDS_HALF, &done1) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_half_cons, dim3(nblocks), dim3(KA_HALF_BLOCK), KA_LDS_HALF, stream, *D, blocks_dev, nqueue);
        } else {
                if (ka_optin(ka_task_kernel_half, KA_LDS_HALF, &done0) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_half, dim3(nblocks), dim3(KA_HALF_BLOCK), KA_LDS_HALF, stream, *D, blocks_dev, nqueue);
        }
}
#endif

#if KA_UNIT == 3
// (the second launch-bound is waves per SIMD: 4 -> <=128 VGPRs -> two 8-wave workgroups per CU)
__global__ __launch_bounds__(KA_LEAN_BLOCK, 4) void ka_task_kernel_lean(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<true, 0>(D, blocks, 0);
}
// the bonus entries cost ~40 VGPRs: 4 waves, 3 waves per SIMD (<=168 VGPRs) -> three workgroups per CU
__global__ __launch_bounds__(KA_PAIR_BLOCK, 3) void ka_task_kernel_lean_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<true, KA_NB>(D, blocks, 0);
}
// the same body with 4 waves: four workgroups per CU, the shape ka_pair_kernel runs the same alignments in (experiment: KA_LEAN4=1)
__global__ __launch_bounds__(KA_PAIR_BLOCK, 4) void ka_task_kernel_lean4(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<true, 0>(D, blocks, 0);
}
extern "C" void ka_unit3_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, hipStream_t stream)
{
        static bool done0 = false, done1 = false, done2 = false;
        if (!cons && D->lean4) {
                if (ka_optin(ka_task_kernel_lean4, KA_LDS_PAIR, &done2) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_lean4, dim3(nblocks), dim3(KA_PAIR_BLOCK), KA_LDS_PAIR, stream, *D, blocks_dev, 0);
                return;
        }
        if (cons) {
                if (ka_optin(ka_task_kernel_lean_cons, KA_LDS_PAIR, &done1) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_lean_cons, dim3(nblocks), dim3(KA_PAIR_BLOCK), KA_LDS_PAIR, stream, *D, blocks_dev, 0);
        } else {
                if (ka_optin(ka_task_kernel_lean, KA_LDS_LEAN, &done0) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_lean, dim3(nblocks), dim3(KA_LEAN_BLOCK), KA_LDS_LEAN, stream, *D, blocks_dev, 0);
        }
}

// ------------------------------------------------------------------------------------------
// Batch of independent seq-seq alignments (pairwise_align_map, anchor_consistency.c:19-120)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KA_PAIR_BLOCK, 4) void ka_pair_kernel(const KaPairDev P)
{
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        TaskShared& S = *(TaskShared*)ka_smem;
        float* tss = (float*)(ka_smem + KA_LDS_TSS);
        char* lds_waves = ka_smem + KA_LDS_WAVES;
        const int k = blockIdx.x;
        const int tid = threadIdx.x;
        if (tid == 0) {
                const int i = P.ia[k], j = P.ib[k];
                const int len_i = P.seq_len[i], len_j = P.seq_len[j];
                const int swapped = !(len_i <= len_j);
                S.ctl = &S.ctl_lds; S.G = 1; S.member = 0; S.bar_phase = 0; S.srows = KA_STRIP_ROWS; S.q1_lvl = 0; S.lvl_srows[0] = KA_STRIP_ROWS; S.lvl_srows[1] = KA_STRIP_ROWS;
                S.sub_ok = 1; S.rec_on = 0; S.nres_t = 23; S.sub_stride = KA_WAVE_LDS_LEAN; S.sub_base = lds_waves + KA_LEAN_SCRATCH(KA_NT); S.sub_tm = 0; S.mw_ok = 1;
                S.lctl = S.ctl; S.Gw = 1; S.member_w = 0; S.split = 0;
                S.ctl_lds.fail = 0; S.ctl_lds.bar = 0;
                S.watchdog = P.error; S.trace = nullptr; S.dbgskip = 0; S.prof = nullptr;
                S.kind = KA_SS; S.swapped = swapped;
                S.len_a = len_i; S.len_b = len_j;
                S.La = swapped ? len_j : len_i;
                S.Lb = swapped ? len_i : len_j;
                S.s1 = P.codes + P.seq_off[swapped ? j : i];
                S.s2 = P.codes + P.seq_off[swapped ? i : j];
                S.p1 = nullptr; S.p2 = nullptr; S.profa = nullptr; S.profb = nullptr;
                S.subm = P.subm;
                S.gpo = P.gpo; S.gpe = P.gpe; S.tgpe = P.tgpe; S.soff = 0.0f;
                S.sp_open = 0.0f; S.sp_ext = 0.0f; S.sp_text = 0.0f;
                S.p1_mult = 1.0f; S.p2_mult = 1.0f;
                ka_carve(S, P.scratch + (long long)k * P.scratch_stride, len_i, len_j, 0);
        }
        ka_build_tss(tss, P.subm, 0.0f);
        __syncthreads();
        ka_hirschberg<KA_SS, 23, 0>(S, nullptr, lds_waves, tss, nullptr);
        __syncthreads();
        ka_code_path(S, (int*)lds_waves);
        if (tid == 0 && P.scores) P.scores[k] = S.ctl->top_score;
        __syncthreads();
        int* dst = P.paths_out + P.poff[k];
        for (int i = tid; i < S.ctl->alnlen + 2; i += KA_NT) dst[i] = S.coded[i];
}

extern "C" void ka_launch_pairs(const KaPairDev* P, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_pair_kernel, KA_LDS_PAIR, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_pair_kernel, dim3(P->npairs), dim3(KA_PAIR_BLOCK), KA_LDS_PAIR, stream, *P);
}
#endif
