// ka_meetup.h -- work items of a recursion level (strips, packed jobs, wave-local subtrees) and the meetup of a sub-problem with aln_continue.
// One of the text sections of the task kernels, included by ka_kernels.hip in this order: ka_shared.h, ka_pass.h, ka_best.h,
// ka_subtree.h, ka_wstrip.h, ka_meetup.h, ka_hirschberg.h, ka_path.h, ka_profile.h, ka_task.h.  Not a stand-alone header.
#pragma once

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
#if KA_TP
#define KA_LONG_COLS 40                                        // (the throughput kernel stages 96 column records per wave region: a packed step that streams them from L2 costs eight strip steps)
#else
#define KA_LONG_COLS 96
#endif
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
//
// RU (round 5): HIRSCHBERG PREFIX REUSE.  A child shares one corner with its parent: the top-left child ([starta, mid') x
// [startb, meet']) starts its forward pass from the parent's forward start state, so its forward rows are the parent's forward
// rows restricted to its columns -- except in its LAST column, where a pass writes ga = -FLT_MAX (aln_seqseq.c:108-117,
// aln_profileprofile.c:128-151) while the parent computed an inner ga there (the terminal rule of that column can only differ
// when it is not the parent's last column too, and then it is off on both sides; the meetup never reads the forward ga of its
// last column, and the backward one -- the first column of the window -- is overwritten below).  The bottom-right child shares
// the backward corner the same way.  So a strip pass that RUNS leaves, besides its last row, the row its usual child will want:
// after (n - 1) / 2 of its n rows going forward (child rows [starta, mid - 1)), after n / 2 going backward (child rows
// [mid + 1, enda)) -- ka_strip<.., SAVE> -- and a child whose own pass would have exactly that many rows takes that row
// (KaSub::fsrc / bsrc) and gets no pass item.  One level deep: a pass that was taken over has left nothing.  ~21 % of the DP
// cells are never computed; oracle/kalign_oracle.c:ko_hirschberg_r restates the rule on the CPU and tests/test_oracle_golden.py
// shows it bit-exact on every golden.  lvl: the recursion level of the meetups (the parent's passes ran at lvl - 1).
template <int KIND, int GL, bool FLIP = false, bool REC = false, bool MW = false, bool RU = false>
__device__ __forceinline__ void ka_meetup(TaskShared& S, const KaSub* qc, const int k0, const int ncur, KaSub* qnext,
                                          const KaLevelOut& lout, const int wlane, const bool top_level, const int kdig = 0, const int lvl = 0)
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
        const bool ru = RU && S.reuse_ok != 0;
        const bool f_taken = ru && sb.fsrc >= 0, b_taken = ru && sb.bsrc >= 0;
        const KaState* f = f_taken ? S.sfbuf[(lvl + 1) & 1] + sb.fsrc : S.fbuf + sb.roff;
        const KaState* b = b_taken ? S.sbbuf[(lvl + 1) & 1] + sb.bsrc : S.bbuf + sb.roff;
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
                KaState fi = f[i - startb], bi = b[i - startb];
                // (a row taken over from the parent: the last column of the child's pass has no ga state -- the forward one is never
                // read there, the backward one is)
                if (RU) { if (b_taken && i == startb) bi.ga = -KA_F; if (f_taken && i == endb) fi.ga = -KA_F; }
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
        c1.fsrc = -1; c1.bsrc = -1; c2.fsrc = -1; c2.bsrc = -1;
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
        // prefix reuse: what this sub-problem's own passes left (they ran as strips, ka_run_items), and which child takes it
        bool take_f1 = false, take_b2 = false;
        if (RU && ru && !FLIP && !rec) {
                const int nfp = mid - sb.starta, nbp = sb.enda - mid, pcols = endb - startb;
                const int rs_f = (nfp - 1) / 2, rs_b = nbp / 2;
                take_f1 = v1 && !st1 && !f_taken && rs_f >= 1 && ka_pass_is_strip(nfp, pcols) && pr[0] == rs_f;
                take_b2 = v2 && !st2 && !b_taken && rs_b >= 1 && ka_pass_is_strip(nbp, pcols) && pr[3] == rs_b;
                if (take_f1) c1.fsrc = sb.roff;
                if (take_b2) c2.bsrc = sb.roff + (c2.startb - startb);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
                if (!((x < 2) ? v1 : v2) || ((x < 2) ? st1 : st2)) continue;
                if (RU && ((x == 0 && take_f1) || (x == 3 && take_b2))) continue;
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
                        if (RU && ((ch == 0 && x == 0 && take_f1) || (ch == 1 && x == 1 && take_b2))) continue;   // (taken over: no pass)
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
