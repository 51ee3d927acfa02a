// ka_hirschberg.h -- the Hirschberg recursion: cluster barrier, a level's passes dealt to the waves, the level-synchronous driver, the depth-first drivers of refinement trials and the incremental flip trials.
// One of the text sections of the task kernels, included by ka_kernels.hip in this order: ka_shared.h, ka_pass.h, ka_best.h,
// ka_subtree.h, ka_wstrip.h, ka_meetup.h, ka_hirschberg.h, ka_path.h, ka_profile.h, ka_task.h.  Not a stand-alone header.
#pragma once

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

#if KA_TP
// THE THROUGHPUT KERNEL keeps its passes and meetups out of the task body (round 6): real functions with register allocations of
// their own.  Inlined, everything shared the kernel's budget of 168 VGPRs (three workgroups per CU) and the kernel spilled 1099 of
// them, in the strips' loops too.  What is wave-uniform arrives in vector registers and goes back to scalars first; the workgroup's
// TaskShared is where it always is (the head of the dynamic LDS).
#define KA_TPF __device__ __attribute__((noinline))
__device__ __forceinline__ TaskShared& ka_tp_shared()
{
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        return *(TaskShared*)ka_smem;
}
template <int KIND, int NRES, int SLOT, int NB>
KA_TPF void ka_packed_tp(const KaSub* qc, const int2* pack, int nslots, int job, const float* tss, char* wlds, int nreg, int reg_stride)
{
        ka_packed<KIND, NRES, SLOT, NB>(ka_tp_shared(), ka_uniform_ptr(qc), ka_uniform_ptr(pack), ka_u(nslots), ka_u(job), (int)(threadIdx.x & 63),
                                        ka_uniform_ptr(tss), ka_uniform_ptr(wlds), ka_u(nreg), ka_u(reg_stride));
}
template <int KIND, int NRES, int NB>
KA_TPF void ka_subtree_tp(const KaSub* sp, char* area, const float* tss)
{
        const KaSub* spu = ka_uniform_ptr(sp);
        ka_subtree<KIND, NRES, NB>(ka_tp_shared(), *spu, (int)(threadIdx.x & 63), ka_uniform_ptr(area), ka_uniform_ptr(tss));
}
template <int KIND, int GL, bool MW>
KA_TPF void ka_meetup_tp(const KaSub* qc, int k0, int ncur, KaSub* qn, const KaLevelOut lout, bool top_level, int kdig, int lvl)
{
        ka_meetup<KIND, GL, false, false, MW, false>(ka_tp_shared(), ka_uniform_ptr(qc), ka_u(k0), ka_u(ncur), ka_uniform_ptr(qn), lout, (int)(threadIdx.x & 63),
                                                     ka_u((int)top_level) != 0, ka_u(kdig), ka_u(lvl));
}
struct KaStripArgsTP { int starta, enda, startb, endb, dir, k; float ja, jga, jgb; KaState* rows; int* prog; char* wlds; const float* tss; };
template <int KIND, int NRES, int NB>
KA_TPF void ka_strip_tp(const KaStripArgsTP a)
{
        ka_strip<KIND, NRES, NB, 2, false, false>(ka_tp_shared(), ka_u(a.starta), ka_u(a.enda), ka_u(a.startb), ka_u(a.endb),
                                                  ka_uniform_f(a.ja), ka_uniform_f(a.jga), ka_uniform_f(a.jgb), ka_u(a.dir), ka_u(a.k),
                                                  ka_uniform_ptr(a.rows), ka_uniform_ptr(a.prog), (int)(threadIdx.x & 63), ka_uniform_ptr(a.wlds), ka_uniform_ptr(a.tss),
                                                  false, false);
}
#endif

// The passes of one recursion level: its work items (strips, packed jobs) dealt to / pulled by the waves of the team.
// Q1: the kernel also carries the one-row-per-lane strip (TaskShared::srows == 64 selects it per task)
// HO: strips dealt to neighbouring waves of a workgroup hand over through LDS rings (ka_strip<.., HO>; TaskShared::ho_ok)
// RU: Hirschberg prefix reuse (ka_meetup.h) -- a strip that holds the row its pass leaves for the sub-problem's child writes it out
template <int KIND, int NRES, int NB, bool Q1 = false, bool HO = false, bool HW = false, bool RU = false>
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
                        // WIDE-SUBTREE LEVEL (round 5).  A window of 65 .. 128 rows used to go one more recursion level as strips before its
                        // halves were small enough for a wave's LDS region -- the level with the worst lanes-per-strip of the whole task (38-row
                        // passes: 19 of 64 lanes), and, with eight passes on a workgroup, no helper waves: 88 us for a level that holds 6 % of a
                        // 618 x 451 task's cells, then 55 us more for the subtrees below it.  When EVERY sub-problem of a level has at most 128
                        // rows and fits TWO wave regions (its passes then have at most 64 rows: one per lane), the level's pass items are left
                        // alone and every even wave runs whole subtrees -- its own region and its idle neighbour's; the sub-problems are
                        // marked like emitted subtrees (this level's meetups skip them, nothing is emitted below).  Decided here, by every wave
                        // alike, from the level's sub-problem records: the emitting meetups cannot know what else the level will hold.
                        int wide_n = 0;
                        // (never the task's top level: its rows stay in HBM -- the record's meetup and score, the tests' row hashes)
                        if (KA_WIDE_SUB && level >= 1 && KA_NW >= 2 && Gw == 1 && __builtin_amdgcn_readfirstlane(S.sub_ok) != 0 && __builtin_amdgcn_readfirstlane(S.sub_ok) != 2) {
                                const int ns = __builtin_amdgcn_readfirstlane(cur->nsub);
                                if (ns >= 1 && ns <= 64) {
                                        const int two = 2 * __builtin_amdgcn_readfirstlane(S.sub_stride);
                                        bool fits = true, gains = false;
                                        if (lane < ns) {
                                                const KaSub* sq = qc + lane;
                                                const int rr = sq->enda - sq->starta, cc = sq->endb - sq->startb;
                                                fits = rr >= 1 && rr <= KA_SUB_WIDEROWS && cc >= 1 && cc < 4096 && ka_sub_bytes(KIND, NRES, rr, cc) <= two;
                                                gains = sq->pad != KA_SUB_MARK;                  // (it would run as passes)
                                        }
                                        if (__ballot(!fits) == 0ull && __ballot(gains) != 0ull) wide_n = ns;
                                }
                        }
                        const bool wmode = !wide_n && HW && KIND == KA_PP && KA_NW == 8 && ntotal <= nslots && per <= KA_NW / 2
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
                        int wide_k = wave >> 1;                                 // wide-subtree level: the next sub-problem of this (even) wave
                        while (true) {
                                int2 item;
                                if (wide_n) {
                                        if ((wave & 1) || wide_k >= wide_n) break;
                                        item = make_int2(wide_k, KA_ITEM_SUBTREE << 16);
                                        wide_k += KA_NW >> 1;
                                } else {
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
#if KA_TP
                                        ka_packed_tp<KIND, NRES, 4, NB>(qc, pack4, n4, it - nitems - njobs16, tss, KIND != KA_SS ? lds_waves + wave * KA_WAVE_LDS : nullptr, nreg, reg_stride);
#else
                                        ka_packed<KIND, NRES, 4, NB>(S, qc, pack4, n4, it - nitems - njobs16, lane, tss, KIND != KA_SS ? lds_waves + wave * KA_WAVE_LDS : nullptr, nreg, reg_stride);
#endif
                                        continue;
                                }
                                if (it >= nitems) {
#if KA_TP
                                        ka_packed_tp<KIND, NRES, 16, NB>(qc, pack16, n16, it - nitems, tss, KIND != KA_SS ? lds_waves + wave * KA_WAVE_LDS : nullptr, nreg, reg_stride);
#else
                                        ka_packed<KIND, NRES, 16, NB>(S, qc, pack16, n16, it - nitems, lane, tss, KIND != KA_SS ? lds_waves + wave * KA_WAVE_LDS : nullptr, nreg, reg_stride);
#endif
                                        continue;
                                }
                                item = items[it];
                                }
                                // everything about the item is wave-uniform: keep it in SGPRs
                                const int subi = __builtin_amdgcn_readfirstlane(item.x);
                                const int dk = __builtin_amdgcn_readfirstlane(item.y);
                                const KaSub* sp = qc + subi;
                                const int dir = dk >> 16, k = dk & 0xffff;
                                if (dir == KA_ITEM_SUBTREE) {
                                        // (a wide-subtree level: the wave's region and its idle neighbour's are one area)
#if KA_TP
                                        ka_subtree_tp<KIND, NRES, NB>(sp, S.sub_base + wave * S.sub_stride, tss);
#else
                                        ka_subtree<KIND, NRES, NB>(S, *sp, lane, S.sub_base + wave * S.sub_stride, tss);
#endif
                                        if (wide_n && lane == 0) ((KaSub*)sp)->pad = KA_SUB_MARK;
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
                                // prefix reuse: the row after (n - 1) / 2 rows of a forward pass, after n / 2 rows of a backward one
                                KaState* sv = nullptr;
                                int sv_rows = 0;
                                if (RU && __builtin_amdgcn_readfirstlane(S.reuse_ok) != 0) {
                                        const int n_ = dir == KA_FWD ? mid_ - sa : ea - mid_;
                                        sv_rows = dir == KA_FWD ? (n_ - 1) / 2 : n_ / 2;
                                        if (sv_rows >= 1) sv = ka_uniform_ptr((dir == KA_FWD ? S.sfbuf[level & 1] : S.sbbuf[level & 1]) + roff);
                                }
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
#if KA_TP
                                // the throughput kernel: profile-profile strips in their lean form (ka_lstrip.h) -- ka_strip's column ring
                                // does not fit this kernel's wave regions and is never instantiated for them
                                if constexpr (KIND == KA_PP) {
                                        KaLStripArgs la;
                                        la.p1 = S.p1; la.p2 = S.p2; la.ent = S.ent; la.watchdog = S.watchdog;
                                        la.rows = (dir == KA_FWD ? S.fbuf : S.bbuf) + roff; la.prog = prog + (it - k);
                                        la.wlds = lds_waves + wave * KA_WAVE_LDS;
                                        la.m1 = S.p1_mult; la.m2 = S.p2_mult; la.inj_a = ja; la.inj_ga = jga; la.inj_gb = jgb; la.Lb = S.Lb;
                                        la.starta = sa; la.enda = ea; la.startb = sbb; la.endb = eb; la.dir = dir; la.k = k;
#ifdef KA_L_PROF
                                        la.prof = S.sub_tm ? S.sub_t : nullptr;
#else
                                        la.prof = nullptr;
#endif
                                        ka_lstrip<NRES, NB>(la);
                                        continue;
                                } else {
                                        KaStripArgsTP ta;
                                        ta.starta = sa; ta.enda = ea; ta.startb = sbb; ta.endb = eb; ta.dir = dir; ta.k = k; ta.ja = ja; ta.jga = jga; ta.jgb = jgb;
                                        ta.rows = (dir == KA_FWD ? S.fbuf : S.bbuf) + roff; ta.prog = prog + (it - k); ta.wlds = lds_waves + wave * KA_WAVE_LDS; ta.tss = tss;
                                        ka_strip_tp<KIND, NRES, NB>(ta);
                                        continue;
                                }
#else
                                if (Q1 && srows == KA_STRIP1_ROWS)
                                        ka_strip<KIND, NRES, NB, 1, HO, RU>(S, sa, ea, sbb, eb, ja, jga, jgb, dir, k,
                                                             ka_uniform_ptr((dir == KA_FWD ? S.fbuf : S.bbuf) + roff), ka_uniform_ptr(prog + (it - k)), lane,
                                                             lds_waves + wave * KA_WAVE_LDS, tss, Gw > 1 && !prod_local, Gw > 1 && !(cons_local || k + 1 == ns), pslot,
                                                             in_lds, out_lds, ho_ctl_w, sv, sv_rows);
                                else
                                        ka_strip<KIND, NRES, NB, 2, HO, RU>(S, sa, ea, sbb, eb, ja, jga, jgb, dir, k,
                                                             ka_uniform_ptr((dir == KA_FWD ? S.fbuf : S.bbuf) + roff), ka_uniform_ptr(prog + (it - k)), lane,
                                                             lds_waves + wave * KA_WAVE_LDS, tss, Gw > 1 && !prod_local, Gw > 1 && !(cons_local || k + 1 == ns), pslot,
                                                             in_lds, out_lds, ho_ctl_w, sv, sv_rows);
#endif
                        }
        }
}

__device__ const int ka_pow3[20] = { 1, 3, 9, 27, 81, 243, 729, 2187, 6561, 19683, 59049, 177147, 531441, 1594323, 4782969, 14348907,
                                     43046721, 129140163, 387420489, 1162261467 };
#define KA_REC_DEPTH 19                                              // recursion levels the keys of ka_meetup<.., REC> can tell apart

#if KA_TP
#define KA_MEETUP_CALL(GL_, MW_) ka_meetup_tp<KIND, GL_, MW_>(qc, k, ncur, qn, lout, level == 0, kdig, level)
#else
#define KA_MEETUP_CALL(GL_, MW_) ka_meetup<KIND, GL_, false, REC, MW_, RU && !REC>(S, qc, k, ncur, qn, lout, lane, level == 0, kdig, level)
#endif
template <int KIND, int NRES, int NB, bool REC = false, bool Q1 = false, bool HO = false, bool HW = false, bool RU = false>
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
                root.fin = Z; root.bin = Z; root.roff = 0; root.pad = 0; root.fsrc = -1; root.bsrc = -1;
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
                ka_run_items<KIND, NRES, NB, Q1, HO, HW, RU && !REC>(S, cur, level, qc, lds_waves, tss, pslot);
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
                                                KA_MEETUP_CALL(64, true);
                                        }
                        } else if (est_cols > 48) {
                                for (int k = S.member_w * KA_NW + wave; k < ncur; k += KA_NW * S.Gw)
                                        KA_MEETUP_CALL(64, false);
                        } else if (est_cols > 6) {
                                for (int k = (S.member_w * KA_NW + wave) * 4; k < ncur; k += KA_NW * S.Gw * 4)
                                        KA_MEETUP_CALL(16, false);
                        } else {
                                for (int k = (S.member_w * KA_NW + wave) * 16; k < ncur; k += KA_NW * S.Gw * 16)
                                        KA_MEETUP_CALL(4, false);
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
        c1.fsrc = -1; c1.bsrc = -1; c2.fsrc = -1; c2.bsrc = -1;
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
                root.fin = Z; root.bin = Z; root.roff = 0; root.pad = 0; root.fsrc = -1; root.bsrc = -1;
                if (seed) { root = *seed; root.roff = 0; root.pad = 0; root.fsrc = -1; root.bsrc = -1; }
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
