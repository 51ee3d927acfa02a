// ka_task.h -- dynamic-LDS layout, scratch carving, the task bodies (first pass, refinement) and the entries that deal tasks to workgroups.
// One of the text sections of the task kernels, included by ka_kernels.hip in this order: ka_shared.h, ka_pass.h, ka_best.h,
// ka_subtree.h, ka_wstrip.h, ka_meetup.h, ka_hirschberg.h, ka_path.h, ka_profile.h, ka_task.h.  Not a stand-alone header.
#pragma once

#ifndef KA_RU_TREE
#define KA_RU_TREE 0                                                // Hirschberg prefix reuse in the 4-wave TREE kernels (see ka_task_body): measured, off
#endif
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
// nb: bonus entries per DP row (the stride of S.ent: the kernels' NB); kanch: anchors the per-anchor tables are made for (-1: nb - 1 -- the
// streamed set, KA_NB_BIG, passes the job's own count)
__device__ long long ka_carve(TaskShared& S, char* base, int la, int lb, int cons_maxlen, bool refine = false, bool rec = false, int nb = KA_NB, int kanch = -1)
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
        for (int par = 0; par < 2; ++par) {
                S.sfbuf[par] = (KaState*)(base + o); o += ka_align_up(n * 12, 16);
                S.sbbuf[par] = (KaState*)(base + o); o += ka_align_up(n * 12, 16);
        }
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
                S.ent = (int2*)(base + o);    o += ka_align_up(n * 8 * nb, 16);
                const long long KM = kanch >= 0 ? kanch : nb - 1;    // anchors
                S.apos_r = (int*)(base + o);  o += ka_align_up(KM * n * 4, 16);
                S.conf_r = (float*)(base + o); o += ka_align_up(KM * n * 4, 16);
                S.apos_c = (int*)(base + o);  o += ka_align_up(KM * n * 4, 16);
                S.conf_c = (float*)(base + o); o += ka_align_up(KM * n * 4, 16);
                S.invj = (int*)(base + o);    o += ka_align_up(KM * ((long long)cons_maxlen + 8) * 4, 16);
                S.vote = base + o;            o += ka_align_up(KM * n * 16, 16);
        }
        return o;
}

__device__ __host__ inline long long ka_scratch_bytes(long long la, long long lb, long long cons_maxlen, long long g = 1, bool refine = false, bool rec = false, long long nb = KA_NB, long long kanch = -1)
{
        const long long km = kanch >= 0 ? kanch : nb - 1;
        const long long n = la + lb + 8;
        const long long nq = (la < lb ? la : lb) + 20;
        const long long ni = 2 * nq + 2 * (n / KA_STRIP1_ROWS + 2);
        long long b = 5 * ((n * 4 + 15) / 16 * 16) + 8 * ((n * 12 + 15) / 16 * 16)
             + 2 * ((nq * (long long)sizeof(KaSub) + 15) / 16 * 16)
             + 2 * ((ni * 8 + 15) / 16 * 16) + 2 * ((ni * 4 + 15) / 16 * 16)
             + 4 * ((2 * nq * 8 + 15) / 16 * 16) + 64;
        if (g > 1) b += g * ka_private_bytes(la, lb);
        if (refine) b += 3 * ((n * 4 + 15) / 16 * 16) + (n * 24 * 4 + 15) / 16 * 16 + (n * 8 + 15) / 16 * 16 + (ka_inc_bytes(n) + 15) / 16 * 16;
        if (rec && !refine) b += (n * 8 + 15) / 16 * 16;
        if (cons_maxlen > 0) b += (n * 8 * nb + 15) / 16 * 16 + 4 * ((km * n * 4 + 15) / 16 * 16) + (km * (cons_maxlen + 8) * 4 + 15) / 16 * 16 + (km * n * 16 + 15) / 16 * 16;
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
        long long tk0 = __builtin_amdgcn_s_memtime();
        long long tk1 = 0, tk2 = 0, tk3 = 0;

        if (tid == 0 && blockIdx.x == 0) KA_CRUMB(D.trace, 4, 1);
        if (tid == 0) {
                if (D.overlap) {
                        // the launches of this run overlap: the tasks that make this task's operands may still be running -- in another
                        // kernel, on another stream -- or not have started; they were launched, so they finish whatever this workgroup does
                        const int dep[2] = { T.qa, T.qb };
                        for (int k = 0; k < 2; ++k) {
                                if (dep[k] < 0) continue;
                                int spins = 0;
                                unsigned long long head = 0;
                                while (__hip_atomic_load(&D.join[dep[k]].go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                        __builtin_amdgcn_s_sleep(32);
                                        // (only a STALLED queue may expire the wait: while the queued launch still hands tasks out -- its head
                                        // moves -- the producer may simply not have been pulled yet, however long the launch is)
                                        if ((spins & 1023) == 1023) {
                                                const unsigned long long h = __hip_atomic_load(&D.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                if (h != head) { head = h; spins = 0; }
                                        }
                                        if (ka_spin_expired(D.error, ++spins, (1 << 21) * max(1, min(D.tasks[dep[k]].wait_mult, 64)), 6, true)) break;
                                }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        tk0 = __builtin_amdgcn_s_memtime();           // (KA_FLAG_TIMING, thread 0's clock: a task's time starts when its operands exist)
                }
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
                        // Helper-wave strips want at most four items per workgroup on EVERY level they serve: rows halve and passes
                        // double, but the last strip of a pass is partial, so the count creeps up with the depth (the root of C3:
                        // 98, 100, 104, 112, 128 strips on levels 0 .. 4; at 25 workgroups level 2 fell back to ka_strip and took
                        // as long as level 1).  The widest of the first levels decides, if the launch gave that many workgroups.
                        if (S.hw_ok && !q1_lvl) {
                                int need = s2;
                                for (int l = 1; l < 5; ++l) {
                                        const int rows = (S.La + (2 << l) - 1) >> (l + 1);
                                        if (rows < 2 * srows) break;
                                        need = max(need, (2 << l) * ka_strips_of(rows, srows));
                                }
                                g_eff = max(g_eff, min(g_launch, (need + 3) / 4));
                        }
                }
                // experiments (KA_PER): strips per workgroup at the task's top level -> workgroups used
                if (Q1 && D.per_target > 0 && kind == KA_PP) {
                        const int s2 = ka_strips_of(S.La / 2, srows) + ka_strips_of(S.La - S.La / 2, srows);
                        g_eff = max((s2 + D.per_target - 1) / D.per_target, 1);
                }
                // Jobs with a consistency table: the votes of a task (ka_cons_votes) are shared by operand and by anchor -- ten units
                // at five anchors -- and take 2 ms at the root of a 4096-sequence tree when one workgroup has two of them: a cluster of
                // two workgroups per anchor where the members are many and the launch gave that many workgroups.
                // (round 5: with carried vote tables the votes themselves are a read of K x (La + Lb) cells -- the wide cluster still shares the
                // members' column updates and the sweep of the marked cells, ka_votes_merge)
                S.carried = (NB && D.cons_K > 0 && ka_votes_carried(D, T)) ? 1 : 0;
                if (NB && D.cons_K > 0 && T.nsip_a + T.nsip_b >= 128) {
                        // (two / three workgroups per (operand, anchor) where one would spend a millisecond and more on the bigger
                        // operand's members: ka_cons_votes_split)
                        const int mem = max(T.nsip_a, T.nsip_b);
                        g_eff = max(g_eff, 2 * D.cons_K * (mem >= 1536 ? 3 : (mem >= 384 ? 2 : 1)));
                }
                if (g_eff > g_launch) g_eff = g_launch;
                S.srows = srows;
                // wave-local subtrees (ka_subtree.h): every kernel shape has a region per wave behind the workgroup's scratch
                // exact task confidences (aln_run.c:391-395 adds the margins in recursion order): every meetup records its margin
                // with its place in that order; sorted and added up in fp32 after the recursion (ka_margins_in_order).  The
                // wave-local subtrees do not keep those records: off.
                S.rec_on = (D.flags & KA_FLAG_EXACT_CONFIDENCE) ? 1 : 0;
                S.sub_ok = (D.sub_mode && !S.rec_on) ? D.sub_mode : 0;     // (KA_SUBTREE: 1 subtrees incl. the wide-subtree levels of round 5, 2 without those)
                S.reuse_ok = (KA_RU_TREE && !Q1 && D.reuse && g_eff == 1 && !S.rec_on) ? 1 : 0;
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
                if (g_eff == 1) { S.ctl_lds.fail = 0; S.ctl_lds.bar = 0; S.ctl_lds.nrec = 0; S.ctl_lds.vote_conf = 0; S.ctl_lds.vote_types = 0; }
                s_dbg = nullptr;
                if (member == 0) {
                        const long long need = ka_scratch_bytes(len_a, len_b, NB ? D.cons_maxlen : 0, g_eff, false, (D.flags & KA_FLAG_EXACT_CONFIDENCE) != 0, NB ? NB : KA_NB, NB > KA_NB ? D.cons_K : -1);
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
        if (tid == 0) ka_carve(S, D.scratch + S.ctl->scratch_off, S.len_a, S.len_b, NB ? D.cons_maxlen : 0, false, S.rec_on != 0, NB ? NB : KA_NB, NB > KA_NB ? D.cons_K : -1);

        // P1
        ka_build_tss(tss, D.subm, T.soff);
        __syncthreads();
        // (round 6: without sequence weights a sequence's record is never written -- the merge makes what it needs of it from the residue,
        // ka_update_profile / ka_leaf_rec4; KA_FLAG_LEAF_PROFILES keeps them for callers that read a leaf's profile back)
        const bool leaf_syn = !(D.usw > 0.0f) && !(D.flags & KA_FLAG_LEAF_PROFILES);
        if (S.member == 0 && !leaf_syn) {
                if (T.nsip_a == 1) ka_make_leaf_profile(S.profa, S.len_a, D.codes + D.seq_off[T.a], T.gpo, T.gpe, T.tgpe, tss);
                if (T.nsip_b == 1) ka_make_leaf_profile(S.profb, S.len_b, D.codes + D.seq_off[T.b], T.gpo, T.gpe, T.tgpe, tss);
        }
        // P1b: anchor positions of both operands (all workgroups of the cluster), then the bonus entries
        // of every DP row (the first one); the barrier below publishes them
        if (NB) {
                __syncthreads();
                if (S.carried) { if (S.member == 0) ka_cons_from_tables(S, D, T); }
                else ka_cons_votes<LEAN, (NB ? NB : KA_NB)>(S, D, T, lds_waves, LEAN ? 0 : (long long)KA_NW * KA_WAVE_LDS);
                ka_cluster_sync(S);
                if (S.member == 0) ka_cons_entries<(NB ? NB : KA_NB)>(S, D);
        }
        ka_cluster_sync(S);
        tk1 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && blockIdx.x == 0) KA_CRUMB(D.trace, 4, 2);

        // P2
        // (last template argument: Hirschberg prefix reuse.  Built for the 4-wave tree kernels too -- -DKA_RU_TREE=1 -- and measured
        // there: 11 % fewer VALU instructions in the queued launch, 8 % in the leaf launch, and NO time gained (16 trees in flight:
        // 77.0 against 75.9 ms; profiles/r05_prefix_reuse.log) -- those launches wait (46 % of the wave cycles parked, VALU busy 32 %),
        // they do not issue, and a task's critical path through its levels is as long with half the passes.  The seq-seq pair batch,
        // whose workgroups are two waves wide, gains 14 %: on there, off here.)
        if (LEAN || S.kind == KA_SS) ka_hirschberg<KA_SS, 23, NB, false, Q1, false, false, KA_RU_TREE && !Q1>(S, s_dbg, lds_waves, tss, D.trace);
        else if (S.kind == KA_SP) ka_hirschberg<KA_SP, 23, NB, false, Q1, false, false, KA_RU_TREE && !Q1>(S, s_dbg, lds_waves, tss, D.trace);
        else if (D.nres <= 5) ka_hirschberg<KA_PP, 5, NB, false, Q1, Q1 && NB == 0, Q1, KA_RU_TREE && !Q1>(S, s_dbg, lds_waves, tss, D.trace);
        // no B / Z / X in the job (the usual case): every profile's counts [20..22] are zero and the reference skips
        // zero counts (aln_profileprofile.c:70-77) -- 20 terms per cell instead of 23
#ifdef KA_FAKE_NRES
        // (measurement only, WRONG results: the profile-profile DP with KA_FAKE_NRES terms per cell -- an upper bound of what a sparse dot
        // product for shallow profiles could gain; tools/build_alt.sh)
        else if (D.nres <= 20) ka_hirschberg<KA_PP, KA_FAKE_NRES, NB, false, Q1, Q1 && NB == 0, Q1, KA_RU_TREE && !Q1>(S, s_dbg, lds_waves, tss, D.trace);
#else
        else if (D.nres <= 20) ka_hirschberg<KA_PP, 20, NB, false, Q1, Q1 && NB == 0, Q1, KA_RU_TREE && !Q1>(S, s_dbg, lds_waves, tss, D.trace);
#endif
        else ka_hirschberg<KA_PP, 23, NB, false, Q1, Q1 && NB == 0, Q1, KA_RU_TREE && !Q1>(S, s_dbg, lds_waves, tss, D.trace);
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
                                // (jobs with a consistency table: the node's carried vote table -- 5 planes of K x alnlen ints -- behind its records)
                                const bool votes = NB && D.cons_K > 0 && S.carried;
                                const unsigned long long fn = pn * 64ull + (votes ? 5ull * (unsigned long long)D.cons_K * pn : 0ull);
                                const unsigned long long fo = atomicAdd(&D.counters[0], fn);
                                if ((long long)(fo + fn) > D.prof_cap) { S.ctl->fail = 1; atomicExch(D.error, 1); }
                                else { S.ctl->newp_off = (long long)fo; D.node_prof[T.c] = (long long)fo; if (votes) D.node_vote[T.c] = (long long)(fo + pn * 64ull); }
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
        if (S.newp) ka_update_profile(S, D, T, alnlen, leaf_syn ? tss : nullptr);
        if ((NB && !T.is_root) || (D.flags & KA_FLAG_DEVICE_GAPS)) ka_update_colof(S, D, T, alnlen);
        if (NB && D.cons_K > 0 && S.carried && S.newp) ka_votes_merge(S, D, T, alnlen, (int*)(S.newp + ((long long)alnlen + 2) * 64));
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
                S.reuse_ok = 0;
                S.sub_ok = 0; S.rec_on = 0; S.nres_t = 23; S.sub_stride = 0; S.sub_base = nullptr; S.sub_tm = 0; S.mw_ok = 0;   // (flip trials decide in recursion order: no wave-local subtrees)
                S.ctl = &S.ctl_lds; S.lctl = S.ctl;
                S.ctl_lds.fail = 0; S.ctl_lds.bar = 0; S.ctl_lds.vote_conf = 0; S.ctl_lds.vote_types = 0;
                S.carried = (NB && D.cons_K > 0 && ka_votes_carried(D, T)) ? 1 : 0;
                const long long need = ka_scratch_bytes(len_a, len_b, NB ? D.cons_maxlen : 0, 1, true, false, NB ? NB : KA_NB, NB > KA_NB ? D.cons_K : -1);
                const unsigned long long so = atomicAdd(&D.counters[1], (unsigned long long)need);
                if ((long long)so + need > D.scratch_cap) { S.ctl->fail = 1; atomicExch(D.error, 2); }
                if (__hip_atomic_load(D.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) S.ctl->fail = 1;
                S.ctl->scratch_off = (long long)so;
                if (!S.ctl->fail) ka_carve(S, D.scratch + so, len_a, len_b, NB ? D.cons_maxlen : 0, true, false, NB ? NB : KA_NB, NB > KA_NB ? D.cons_K : -1);
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
                if (S.carried) ka_cons_from_tables(S, D, T);
                else ka_cons_votes<false, (NB ? NB : KA_NB)>(S, D, T, lds_waves, (long long)KA_NW * KA_WAVE_LDS);
                __syncthreads();
                ka_cons_entries<(NB ? NB : KA_NB)>(S, D);
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
                        const bool votes = NB && D.cons_K > 0 && S.carried;
                        const unsigned long long fn = pn * 64ull + (votes ? 5ull * (unsigned long long)D.cons_K * pn : 0ull);
                        const unsigned long long fo = atomicAdd(&D.counters[0], fn);
                        if ((long long)(fo + fn) > D.prof_cap) { S.ctl->fail = 1; atomicExch(D.error, 1); }
                        else { S.ctl->newp_off = (long long)fo; D.node_prof[T.c] = (long long)fo; if (votes) D.node_vote[T.c] = (long long)(fo + pn * 64ull); }
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
        if (NB && D.cons_K > 0 && S.carried && S.newp) ka_votes_merge(S, D, T, alnlen, (int*)(S.newp + ((long long)alnlen + 2) * 64));
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
#define KA_MAX_G 32
template <bool LEAN, int NB>
__device__ __forceinline__ void ka_task_entry(const KaTreeDev& D, const int2* __restrict__ blocks, const int chain)
{
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        TaskShared& S = *(TaskShared*)ka_smem;
        const int2 blk = blocks[blockIdx.x];
        int task = blk.x;
        if (task < 0) return;
        int member = blk.y & 0xff, g = (blk.y >> 8) & 0xffff;
        const int tid = threadIdx.x;
        // Round 5, overlapping launches: this launch went out BESIDE the queued one.  Normally its workgroups become resident as the
        // queue's leave; when the dispatcher brings them in first (stream priorities are a hint) they would sit on the CUs the queue needs
        // and wait for it -- a slow round for one tree, seconds for a forest.  So a workgroup that arrives while more than a round of the
        // queue is still to be handed out takes tasks from the queue's list like one of the queue's own workgroups (same counter, same
        // done flags, this kernel's task body), and turns to its place in the chain when the list is down to its last round.
        bool helping = chain && D.q_n > 0 && !(blk.y & KA_BLK_NOHELP);      // (the head of the chain sits on CUs kept for it: it waits for its operands, plan_launches)
        const int own_task = task, own_member = member, own_g = g;
        while (true) {
                if (helping) {
                        __syncthreads();
                        if (tid == 0) {
                                const long long head = (long long)__hip_atomic_load(&D.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                S.next_member = ((long long)D.q_n - head > (long long)D.q_slots) ? (int)atomicAdd(&D.counters[4], 1ull) : D.q_n;
                        }
                        __syncthreads();
                        const int qi = S.next_member;
                        __syncthreads();
                        if (qi < D.q_n) { task = D.q_order[qi].x; member = 0; g = 1; if (tid == 0) atomicAdd(&D.counters[5], 1ull); }
                        else { helping = false; task = own_task; member = own_member; g = own_g; }
                }
                const int st = ka_task_body<LEAN, NB, !LEAN>(D, task, member, g);
                if (helping) {
                        // (as the queue does: everything the task wrote is released, then its done flag goes up -- also after a failed task)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        __syncthreads();
                        if (tid == 0) __hip_atomic_store(&D.join[task].go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        continue;
                }
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
        // round 6: CUs kept for the head of the chained launch (plan_launches) -- the queue's workgroups that landed there leave at once
        if (n > 0 && D.reserve > 0) {
                unsigned id, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                // (XCC 0 hands its share of a launch to its four shader engines in turn -- tools/microbench/cu_map.hip: blocks 0, 8, 16 on SE
                // 0, 1, 3 --, so the kept CUs are the first reserve / 4 of EVERY engine: CU ids up to reserve / 4, one id being fused off in
                // some engines)
                if ((xcc & 15u) == 0u && (int)((id >> 8) & 15u) <= (D.reserve >> 2)) return;
        }
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
