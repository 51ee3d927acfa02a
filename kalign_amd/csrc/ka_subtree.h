// ka_subtree.h -- a small Hirschberg SUBTREE, run to its leaves by ONE wave with everything in LDS.
//
// Below the top few recursion levels a task has hundreds of sub-problems of a few rows each.  Run level by level
// across the workgroup (ka_hirschberg), every one of those levels costs a fixed ~10-15 us whatever its size: the
// work lists, the sub-problem records, the operand rows and the f / b row buffers all live in HBM, and a level is a
// chain of 4-5 dependent trips to L2 (list -> record -> operands -> rows -> meetup -> children) between two
// workgroup barriers -- 35 % of the time of a 450 x 450 task for 6 % of its cells.
//
// Here a sub-problem of at most 64 rows whose operand windows fit the wave's LDS region becomes ONE work item
// (dir = 2 in the item list): the wave that takes it copies both operand windows into LDS once (profile rows: the
// counts and the three gap terms, already multiplied by set_gap_penalties_n's factor; profile columns: the scores
// and gap terms; sequences: the residues), and then runs the whole recursion below it level-synchronously on its own:
// sub-problem queue, row buffers and operands in LDS, no workgroup barrier, no trip to L2 except the path entries
// it writes.  Same cells, same arithmetic, same order as ka_packed / ka_meetup (aln_profileprofile.c:17-298,
// aln_seqseq.c:241-420, aln_controller.c:194-436): the sub-problems of a level are independent given their windows.
#pragma once

#define KA_SUB_MAXROWS 64
#define KA_SUB_NQ 72                                            // queue entries per level: a level has at most one sub-problem per row

// compact sub-problem: window relative to the subtree root's (starta, startb); boundary states as codes
// (0 = the root's own injected state, 1 = Z, 2 = GA, 3 = GB: a child inherits one side and gets a constant on the other)
struct KaSubL { int a; int b; int c; };                         // sa | ea << 16;  sb | eb << 16;  roff | fcode << 16 | bcode << 18

// explicit LDS pointers: ds_read / ds_write instead of flat accesses (which count on vmcnt AND lgkmcnt)
typedef __attribute__((address_space(3))) float ka_lf;
typedef __attribute__((address_space(3))) float4v ka_lf4;
typedef __attribute__((address_space(3))) unsigned char ka_lu8;
typedef __attribute__((address_space(3))) int ka_li;

__device__ __forceinline__ KaSubL ka_subl_load(const ka_li* q, int k) { KaSubL e; e.a = q[3 * k]; e.b = q[3 * k + 1]; e.c = q[3 * k + 2]; return e; }
__device__ __forceinline__ void ka_subl_store(ka_li* q, int k, const KaSubL& e) { q[3 * k] = e.a; q[3 * k + 1] = e.b; q[3 * k + 2] = e.c; }

__device__ __forceinline__ int ka_sub_rw(int kind, int nres) { return kind == KA_PP ? 4 * ((nres + 3) / 4) + 4 : (kind == KA_SP ? 28 : 0); }

// LDS bytes a subtree of R rows x C columns needs in the wave's region (0 < R <= 64)
__device__ __forceinline__ int ka_sub_bytes(int kind, int nres, int R, int C)
{
        const int rw = ka_sub_rw(kind, nres);
        int b = 0;
        b += (kind == KA_SS) ? ((R + 2 + 15) & ~15) : (R + 2) * rw * 4;                  // rows: residues or records
        b += (kind == KA_PP) ? (C + 2) * rw * 4 : ((C + 3 + 15) & ~15);                  // columns: records or residues
        b += 2 * KA_SUB_NQ * (int)sizeof(KaSubL);
        b += 2 * (((C + 1 + KA_SUB_MAXROWS) * 12 + 15) & ~15);
        return b;
}

__device__ __forceinline__ bool ka_sub_fits(int kind, int nres, int R, int C)
{
        return R >= 1 && R <= KA_SUB_MAXROWS && C >= 1 && C < 4096 && ka_sub_bytes(kind, nres, R, C) <= KA_WAVE_LDS;
}

__device__ __forceinline__ void ka_wave_lds_sync()
{
        // written and read by different lanes of THIS wave only (LDS executes a wave's operations in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// everything a subtree's passes and meetups need (wave-uniform)
struct KaSubCtx {
        int a0, b0;                    // absolute origin of the window
        int R, C;
        int La, Lb;
        ka_lf* rowsL;                  // PP / SP: (R + 2) records of RW floats, record `starta + k` at k
        ka_lf* colsL;                  // PP: (C + 2) records, record `startb + k` at k
        ka_lu8* rowres;                // SS: residue of DP row i (absolute) at i - a0
        ka_lu8* colres;                // SS / SP: residue of position b0 + j of the column sequence at 1 + j
        ka_li* q[2];
        ka_lf* F;
        ka_lf* B;
        KaState rfin, rbin;            // the root's injected states
        float gpo, gpe, tgpe;          // SS: -gpo ... are the terms
        float kc_open, kc_ext, kc_text; // sequence columns (SS / SP): column gap terms
};

__device__ __forceinline__ KaState ka_sub_state(const int code, const KaState root)
{
        KaState s;
        s.a = (code == 0) ? root.a : (code == 1 ? 0.0f : -KA_F);
        s.ga = (code == 0) ? root.ga : (code == 2 ? 0.0f : -KA_F);
        s.gb = (code == 0) ? root.gb : (code == 3 ? 0.0f : -KA_F);
        return s;
}

// The passes pass0 .. pass0 + 64/SLOT - 1 of the current level (pass p = sub-problem p / 2, direction p & 1), SLOT lanes
// apiece, two DP rows per lane, in lock-step.  Same cell code as ka_packed.
template <int KIND, int NRES, int SLOT>
__device__ __forceinline__ void ka_sub_pass(const KaSubCtx& X, const ka_li* qc, const int npass, const int pass0, const int lane, const float* tss)
{
        constexpr int RW = (KIND == KA_PP) ? 4 * ((NRES + 3) / 4) + 4 : (KIND == KA_SP ? 28 : 0);
        constexpr int G0 = RW - 4;                                     // the gap chunk of a record
        constexpr int NV = (NRES + 3) / 4;
        const int p = pass0 + lane / SLOT;
        const int ls = lane % SLOT;
        const bool live = p < npass;
        const KaSubL e = ka_subl_load(qc, live ? (p >> 1) : 0);
        const int dir = p & 1;
        const int sa = e.a & 0xffff, ea = e.a >> 16, sb = e.b & 0xffff, eb = e.b >> 16;
        const int roff = e.c & 0xffff;
        const int ncols = eb - sb;
        const int mid = ((ea - sa) / 2) + sa;
        const int r0 = (dir == KA_FWD) ? sa : mid;
        const int r1 = (dir == KA_FWD) ? mid : ea;
        const int nrows = r1 - r0;                                    // 0 .. 2 * SLOT
        const int nl = (nrows + 1) >> 1;
        const bool near_t = (dir == KA_FWD) ? (X.b0 + sb == 0) : (X.b0 + eb == X.Lb);
        const bool far_t = (dir == KA_FWD) ? (X.b0 + eb == X.Lb) : (X.b0 + sb == 0);
        const KaState inj = (dir == KA_FWD) ? ka_sub_state((e.c >> 16) & 3, X.rfin) : ka_sub_state((e.c >> 18) & 3, X.rbin);
        ka_lf* const rowbuf = ((dir == KA_FWD) ? X.F : X.B) + 3 * roff;

#define SREC(v_) ((dir == KA_FWD) ? (sb + (v_)) : (eb + 1 - (v_)))     /* column record, relative to b0 */
#define SIDX(v_) ((dir == KA_FWD) ? (v_) : (ncols - (v_)))

        const bool writer = live && (ls == (nl > 0 ? nl - 1 : 0));
        const bool last_is_b = (nrows & 1) == 0;
        const int uA = min(2 * ls, max(nrows - 1, 0));
        const int uB = min(2 * ls + 1, max(nrows - 1, 0));
        const bool actB = live && (2 * ls + 1 < nrows);
        const int iA = (dir == KA_FWD) ? (r0 + uA) : (r1 - 1 - uA);   // DP rows, relative to a0
        const int iB = (dir == KA_FWD) ? (r0 + uB) : (r1 - 1 - uB);
        const int recA = min(max(iA + 1, 0), X.R + 1), recB = min(max(iB + 1, 0), X.R + 1);
        const int prevA = min(max((dir == KA_FWD) ? recA - 1 : recA + 1, 0), X.R + 1);
        const int prevB = min(max((dir == KA_FWD) ? recB - 1 : recB + 1, 0), X.R + 1);

        float oA, eA, tA, oB, eB, tB, orpA, orpB;
        float2v p1v[KIND == KA_PP ? NRES : 1];
        int res1A = 0, res1B = 0;
        const ka_lf* srowA = nullptr;
        const ka_lf* srowB = nullptr;
        if (KIND == KA_SS) {
                oA = oB = -X.gpo; eA = eB = -X.gpe; tA = tB = -X.tgpe; orpA = orpB = -X.gpo;
                res1A = X.rowres[min(max(iA, 0), X.R - 1)] * KA_T_STRIDE; res1B = X.rowres[min(max(iB, 0), X.R - 1)] * KA_T_STRIDE;
        } else {
                const ka_lf* pA = X.rowsL + recA * RW;
                const ka_lf* pB = X.rowsL + recB * RW;
                const float4v ga = *(const ka_lf4*)(pA + G0), gb = *(const ka_lf4*)(pB + G0);
                oA = ga.x; eA = ga.y; tA = ga.z; oB = gb.x; eB = gb.y; tB = gb.z;
                orpA = X.rowsL[prevA * RW + G0]; orpB = X.rowsL[prevB * RW + G0];
                if (KIND == KA_PP) {
                        float4v va[NV], vb[NV];
#pragma unroll
                        for (int i = 0; i < NV; ++i) { va[i] = ((const ka_lf4*)pA)[i]; vb[i] = ((const ka_lf4*)pB)[i]; }
#pragma unroll
                        for (int c = 0; c < NRES; ++c) {
                                p1v[c].x = va[c >> 2][c & 3];
                                p1v[c].y = actB ? vb[c >> 2][c & 3] : 0.0f;
                        }
                } else {
                        srowA = pA; srowB = pB;                       // seq-profile: the score rows stay in LDS, indexed by the column's residue
                }
        }

        float cAa = -KA_F, cAga = -KA_F, cAgb = -KA_F;
        float cBa = -KA_F, cBga = -KA_F, cBgb = -KA_F;
        float dga = -KA_F, dgga = -KA_F, dggb = -KA_F;
        float inia = inj.a, iniga = inj.ga, inigb = inj.gb;
        float copen_prev = 0.0f;
        float4v q[2][KIND == KA_PP ? NV + 1 : 1];
        int resq[2] = {0, 0};

        auto fetch = [&](float4v* dstq, int& dstres, int vcol) {
                const int vv = min(max(vcol, 0), ncols);
                if (KIND == KA_PP) {
                        const ka_lf4* src = (const ka_lf4*)(X.colsL + SREC(vv) * RW);
#pragma unroll
                        for (int ch = 0; ch < NV + 1; ++ch) dstq[ch] = src[ch];
                } else {
                        // residue of column record rec sits at rec - 1 of the sequence (clamped like ka_packed does)
                        dstres = X.colres[min(max(SREC(max(vv, 1)) - 1, 0), X.C - 1) + 1];
                }
        };
        fetch(q[0], resq[0], -ls);

        int nsteps = live ? (ncols + max(nl, 1)) : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, off, 64));

        auto step = [&](const int t, auto par_tag) {
                constexpr int P = decltype(par_tag)::value;
                const int v = t - ls;
                const bool vin = live && (v >= 0) && (v <= ncols);
                fetch(q[1 - P], resq[1 - P], v + 1);

                float copen, cext, ctext;
                if (KIND == KA_PP) { copen = q[P][NV].x; cext = q[P][NV].y; ctext = q[P][NV].z; }
                else { copen = X.kc_open; cext = X.kc_ext; ctext = X.kc_text; }
                {
                        const float gx = near_t ? ctext : cext, gy = near_t ? ctext : copen;
                        const float g = kmax(iniga + gx, inia + gy);
                        const bool v0 = (v == 0), vmid = (v < ncols);
                        inia = v0 ? inj.a : -KA_F;
                        iniga = v0 ? inj.ga : (vmid ? g : -KA_F);
                        inigb = v0 ? inj.gb : -KA_F;
                }
                float upa = wave_shr1(cBa), upga = wave_shr1(cBga), upgb = wave_shr1(cBgb);
                if (ls == 0) { upa = inia; upga = iniga; upgb = inigb; }

                float2v acc;
                acc.x = kmax3(dga, dgga + copen_prev, dggb + orpA);
                acc.y = kmax3(cAa, cAga + copen_prev, cAgb + orpB);
                if (KIND == KA_SS) {
                        acc.x += tss[res1A + resq[P]];
                        acc.y += tss[res1B + resq[P]];
                } else if (KIND == KA_SP) {
                        acc.x += srowA[resq[P]];
                        acc.y += srowB[resq[P]];
                } else {
#pragma unroll
                        for (int c = NRES - 1; c >= 0; --c) {
                                const float sc = q[P][c >> 2][c & 3];
                                float2v w; w.x = sc; w.y = sc;
                                acc = acc + p1v[c] * w;
                        }
                }
                const bool at0 = (v == 0), atN = (v == ncols);
                const bool edge = at0 || atN;
                const bool term = (at0 && near_t) || (atN && far_t);
                const float nAa = at0 ? -KA_F : acc.x;
                const float nAga = edge ? -KA_F : kmax(cAga + cext, cAa + copen);
                const float nAgb = term ? kmax(upgb, upa) + tA : kmax(upgb + eA, upa + oA);
                const float nBa = at0 ? -KA_F : acc.y;
                const float nBga = edge ? -KA_F : kmax(cBga + cext, cBa + copen);
                const float nBgb = term ? kmax(nAgb, nAa) + tB : kmax(nAgb + eB, nAa + oB);
                cAa = nAa; cAga = nAga; cAgb = nAgb;
                cBa = nBa; cBga = nBga; cBgb = nBgb;
                dga = upa; dgga = upga; dggb = upgb;
                copen_prev = copen;
                if (vin && writer) {
                        ka_lf* w = rowbuf + 3 * SIDX(v);
                        if (nrows == 0) { w[0] = inia; w[1] = iniga; w[2] = inigb; }
                        else {
                                w[0] = last_is_b ? cBa : cAa;
                                w[1] = last_is_b ? cBga : cAga;
                                w[2] = last_is_b ? cBgb : cAgb;
                        }
                }
        };
        int t = 0;
        for (; t + 1 < nsteps; t += 2) {
                step(t, std::integral_constant<int, 0>());
                step(t + 1, std::integral_constant<int, 1>());
        }
        if (t < nsteps) step(t, std::integral_constant<int, 0>());
#undef SREC
#undef SIDX
}

// The meetups of sub-problems k0 .. k0 + 64/GL - 1 of the current level (GL lanes apiece) and their children
// (aln_continue) into the next level's queue.  n_next / row_next: running totals of the next level (wave-uniform).
template <int KIND, int NRES, int GL>
__device__ __forceinline__ void ka_sub_meet(TaskShared& S, const KaSubCtx& X, const ka_li* qc, const int ncur, const int k0,
                                            ka_li* qn, int& n_next, int& row_next, double& msum, int& mcount, const int wlane)
{
        constexpr int RW = (KIND == KA_PP) ? 4 * ((NRES + 3) / 4) + 4 : (KIND == KA_SP ? 28 : 0);
        constexpr int G0 = RW - 4;
        const int lane = wlane % GL;
        const int ksub = k0 + wlane / GL;
        const bool valid = ksub < ncur;
        const KaSubL e = ka_subl_load(qc, valid ? ksub : k0);
        const int sa = e.a & 0xffff, ea = e.a >> 16, sb = e.b & 0xffff, eb = e.b >> 16;
        const int roff = e.c & 0xffff, fcode = (e.c >> 16) & 3, bcode = (e.c >> 18) & 3;
        const int startb = X.b0 + sb, endb = X.b0 + eb;              // absolute, as the reference's formulas use them
        const int mid = ((ea - sa) / 2) + sa;                         // relative
        const ka_lf* f = X.F + 3 * roff;
        const ka_lf* b = X.B + 3 * roff;
        const float middle = (float)(endb - startb) / 2.0f + (float)startb;
        float g3, g7, g6n, g6f;
        if (KIND == KA_SS) {
                g3 = -X.gpo; g7 = -X.gpo;
                g6n = (startb == 0) ? -X.tgpe : -X.gpe;
                g6f = (endb == X.Lb) ? -X.tgpe : -X.gpe;
        } else {
                const ka_lf* Rr = X.rowsL + (mid + 1) * RW + G0;         // record mid + 1 (relative to a0): (open, ext, text) * nsip
                g3 = Rr[0]; g7 = Rr[-RW];
                g6n = (startb == 0) ? Rr[2] : Rr[1];
                g6f = (endb == X.Lb) ? Rr[2] : Rr[1];
        }
        Best Bt = { -KA_F, -KA_F, 0x7fffffff, 0x7fffffff };
        for (int i = sb + lane; valid && i <= eb; i += GL) {
                const int x = i - sb;
                const float fa = f[3 * x], fga = f[3 * x + 1], fgb = f[3 * x + 2];
                const float ba = b[3 * x], bga = b[3 * x + 1], bgb = b[3 * x + 2];
                float sub = fabsf(middle - (float)(X.b0 + i));
                sub = sub / 1000.0f;
                const int kb = x * 8;
                if (i < eb) {
                        float c2, c5;
                        if (KIND == KA_PP) { c2 = X.colsL[(i + 1) * RW + G0]; c5 = X.colsL[i * RW + G0]; }
                        else { c2 = X.kc_open; c5 = X.kc_open; }
                        best_consider(Bt, fa + ba - sub, kb + 0);
                        best_consider(Bt, fa + bga + c2 - sub, kb + 1);
                        best_consider(Bt, fa + bgb + g3 - sub, kb + 2);
                        best_consider(Bt, fga + ba + c5 - sub, kb + 3);
                        best_consider(Bt, fgb + bgb + g6n - sub, kb + 4);
                        best_consider(Bt, fgb + ba + g7 - sub, kb + 5);
                } else {
                        best_consider(Bt, fa + bgb + g3 - sub, kb + 2);
                        best_consider(Bt, fgb + bgb + g6f - sub, kb + 4);
                }
        }
#pragma unroll
        for (int off = GL / 2; off >= 1; off >>= 1) {
                const float omx = __shfl_xor(Bt.mx, off, 64);
                const float omx2 = __shfl_xor(Bt.mx2, off, 64);
                const int okey = __shfl_xor(Bt.key, off, 64);
                best_merge(Bt, omx, omx2, okey);
        }
        const bool leader = (lane == 0) && valid;
        int meet = -1, tr = -1;
        if (leader && Bt.key != 0x7fffffff) {
                const int ord = Bt.key & 7;
                meet = sb + (Bt.key >> 3);                            // relative to b0
                tr = ord + 1 + (ord >= 3 ? 1 : 0);
        }
        // aln_continue (aln_controller.c:194-436): path entries and the two child windows
        int c1sa = sa, c1ea = sa, c1sb = sb, c1eb = sb, c1bc = 1;        // empty unless a transition fills them in
        int c2sa = ea, c2ea = ea, c2sb = eb, c2eb = eb, c2fc = 1;
        if (tr > 0) {
                int* path = S.raw;
                const int am = X.a0 + mid, bm = X.b0 + meet;             // absolute
                switch (tr) {
                case 1:
                        path[am] = bm; path[am + 1] = bm + 1;
                        c1ea = mid - 1; c1eb = meet - 1; c1bc = 1;
                        c2sa = mid + 1; c2sb = meet + 1; c2fc = 1;
                        break;
                case 2:
                        path[am] = bm;
                        c1ea = mid - 1; c1eb = meet - 1; c1bc = 1;
                        c2sa = mid; c2sb = meet + 1; c2fc = 2;
                        break;
                case 3:
                        path[am] = bm;
                        c1ea = mid - 1; c1eb = meet - 1; c1bc = 1;
                        c2sa = mid + 1; c2sb = meet; c2fc = 3;
                        break;
                case 5:
                        path[am + 1] = bm + 1;
                        c1ea = mid; c1eb = meet - 1; c1bc = 2;
                        c2sa = mid + 1; c2sb = meet + 1; c2fc = 1;
                        break;
                case 6:
                        c1ea = mid - 1; c1eb = meet; c1bc = 3;
                        c2sa = mid + 1; c2sb = meet; c2fc = 3;
                        break;
                default: /* 7 */
                        path[am + 1] = bm + 1;
                        c1ea = mid - 1; c1eb = meet; c1bc = 3;
                        c2sa = mid + 1; c2sb = meet + 1; c2fc = 1;
                        break;
                }
        }
        const bool v1 = leader && (tr > 0) && c1sa < c1ea && c1sb < c1eb;
        const bool v2 = leader && (tr > 0) && c2sa < c2ea && c2sb < c2eb;
        // queue slots and row-buffer cells of the children: exclusive scans over the wave
        const int nsl = (v1 ? 1 : 0) + (v2 ? 1 : 0);
        const int nrw = (v1 ? c1eb - c1sb + 1 : 0) + (v2 ? c2eb - c2sb + 1 : 0);
        int sc1 = nsl, sc2 = nrw;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
                const int y1 = __shfl_up(sc1, d, 64), y2 = __shfl_up(sc2, d, 64);
                if (wlane >= d) { sc1 += y1; sc2 += y2; }
        }
        const int tot1 = __shfl(sc1, 63, 64), tot2 = __shfl(sc2, 63, 64);
        int slot = n_next + sc1 - nsl, row = row_next + sc2 - nrw;
        if (v1) {
                KaSubL c; c.a = c1sa | (c1ea << 16); c.b = c1sb | (c1eb << 16); c.c = row | (fcode << 16) | (c1bc << 18);
                ka_subl_store(qn, slot++, c); row += c1eb - c1sb + 1;
        }
        if (v2) {
                KaSubL c; c.a = c2sa | (c2ea << 16); c.b = c2sb | (c2eb << 16); c.c = row | (c2fc << 16) | (bcode << 18);
                ka_subl_store(qn, slot, c);
        }
        n_next += tot1; row_next += tot2;
        // margins (best - second best) of the meetups that had a second candidate
        double m = (leader && Bt.mx2 > -KA_F) ? (double)(Bt.mx - Bt.mx2) : 0.0;
        int mc = (leader && Bt.mx2 > -KA_F) ? 1 : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { m += __shfl_xor(m, d, 64); mc += __shfl_xor(mc, d, 64); }
        msum += m; mcount += mc;
}

// The whole subtree below `root` by the calling wave.  area: the wave's LDS region (KA_WAVE_LDS bytes).
template <int KIND, int NRES>
__device__ __forceinline__ void ka_subtree(TaskShared& S, const KaSub root, const int lane, char* area, const float* tss)
{
        constexpr int RW = (KIND == KA_PP) ? 4 * ((NRES + 3) / 4) + 4 : (KIND == KA_SP ? 28 : 0);
        constexpr int G0 = RW - 4;
        constexpr int NV = (NRES + 3) / 4;
        KaSubCtx X;
        X.a0 = __builtin_amdgcn_readfirstlane(root.starta); X.b0 = __builtin_amdgcn_readfirstlane(root.startb);
        X.R = __builtin_amdgcn_readfirstlane(root.enda) - X.a0; X.C = __builtin_amdgcn_readfirstlane(root.endb) - X.b0;
        X.La = __builtin_amdgcn_readfirstlane(S.La); X.Lb = __builtin_amdgcn_readfirstlane(S.Lb);
        X.rfin = root.fin; X.rbin = root.bin;
        X.gpo = ka_uniform_f(S.gpo); X.gpe = ka_uniform_f(S.gpe); X.tgpe = ka_uniform_f(S.tgpe);
        X.kc_open = 0.0f; X.kc_ext = 0.0f; X.kc_text = 0.0f;
        if (KIND != KA_PP) {
                col_terms<KIND>(S, 0, X.kc_open, X.kc_ext, X.kc_text);
                X.kc_open = ka_uniform_f(X.kc_open); X.kc_ext = ka_uniform_f(X.kc_ext); X.kc_text = ka_uniform_f(X.kc_text);
        }
        // ---- carve the region (same arithmetic as ka_sub_bytes) ----
        int o = 0;
        X.rowsL = (ka_lf*)area; X.rowres = (ka_lu8*)area;
        o += (KIND == KA_SS) ? ((X.R + 2 + 15) & ~15) : (X.R + 2) * RW * 4;
        X.colsL = (ka_lf*)(area + o); X.colres = (ka_lu8*)(area + o);
        o += (KIND == KA_PP) ? (X.C + 2) * RW * 4 : ((X.C + 3 + 15) & ~15);
        X.q[0] = (ka_li*)(area + o); o += KA_SUB_NQ * (int)sizeof(KaSubL);
        X.q[1] = (ka_li*)(area + o); o += KA_SUB_NQ * (int)sizeof(KaSubL);
        const int rbytes = ((X.C + 1 + KA_SUB_MAXROWS) * 12 + 15) & ~15;
        X.F = (ka_lf*)(area + o); o += rbytes;
        X.B = (ka_lf*)(area + o);

        // ---- stage the operand windows ----
        const float m1 = ka_uniform_f(S.p1_mult), m2 = ka_uniform_f(S.p2_mult);
        if (KIND == KA_SS) {
                for (int i = lane; i < X.R; i += 64) X.rowres[i] = S.s1[X.a0 + i];
        } else {
                // records a0 .. a0 + R + 1 of the row profile: chunk NV carries the gap terms times nsip of the other operand
                const int src0 = (KIND == KA_PP) ? 0 : 32;              // PP rows: the counts; SP rows: the scores
                constexpr int NVR = (KIND == KA_PP) ? NV : 6;
                static_assert(KIND != KA_SP || RW == 28, "seq-profile row records are 24 scores + the gap chunk");
                const int per = NVR + 1;
                const int nit = (X.R + 2) * per;
                for (int it = lane; it < nit; it += 64) {
                        const int k = it / per, ch = it % per;
                        const float* rec = S.p1 + ((long long)(X.a0 + k) << 6);
                        float4v val;
                        if (ch < NVR) val = *(const float4v*)(rec + src0 + 4 * ch);
                        else { val.x = rec[55] * m1; val.y = rec[56] * m1; val.z = rec[57] * m1; val.w = 0.0f; }
                        *(ka_lf4*)(X.rowsL + k * RW + (ch < NVR ? 4 * ch : G0)) = val;
                }
        }
        if (KIND == KA_PP) {
                const int per = NV + 1;
                const int nit = (X.C + 2) * per;
                for (int it = lane; it < nit; it += 64) {
                        const int k = it / per, ch = it % per;
                        const float* rec = S.p2 + ((long long)(X.b0 + k) << 6);
                        float4v val;
                        if (ch < NV) val = *(const float4v*)(rec + 32 + 4 * ch);
                        else { val.x = rec[55] * m2; val.y = rec[56] * m2; val.z = rec[57] * m2; val.w = 0.0f; }
                        *(ka_lf4*)(X.colsL + k * RW + (ch < NV ? 4 * ch : G0)) = val;
                }
        } else {
                // colres[1 + j] = residue at position b0 + j of the column sequence, j = 0 .. C-1 (ka_sub_pass clamps into that range)
                for (int i = lane; i < X.C; i += 64) X.colres[1 + i] = S.s2[X.b0 + i];
        }
        if (lane == 0) {
                KaSubL e; e.a = 0 | (X.R << 16); e.b = 0 | (X.C << 16); e.c = 0;
                ka_subl_store(X.q[0], 0, e);
        }
        ka_wave_lds_sync();

        int ncur = 1, level = 0;
        double msum = 0.0;
        int mcount = 0;
        while (ncur > 0) {
                const ka_li* qc = X.q[level & 1];
                ka_li* qn = X.q[(level + 1) & 1];
                // ---- passes: the slot size follows the level's longest pass ----
                int maxrows = 0;
                for (int k = lane; k < ncur; k += 64) { const KaSubL e = ka_subl_load(qc, k); const int r = (e.a >> 16) - (e.a & 0xffff); maxrows = max(maxrows, r - r / 2); }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) maxrows = max(maxrows, __shfl_xor(maxrows, off, 64));
                const int npass = 2 * ncur;
                if (maxrows > 8) { for (int p0 = 0; p0 < npass; p0 += 4) ka_sub_pass<KIND, NRES, 16>(X, qc, npass, p0, lane, tss); }
                else if (maxrows > 2) { for (int p0 = 0; p0 < npass; p0 += 16) ka_sub_pass<KIND, NRES, 4>(X, qc, npass, p0, lane, tss); }
                else { for (int p0 = 0; p0 < npass; p0 += 64) ka_sub_pass<KIND, NRES, 1>(X, qc, npass, p0, lane, tss); }
                ka_wave_lds_sync();
                // ---- meetups and children ----
                int n_next = 0, row_next = 0;
                if (ncur <= 1) { ka_sub_meet<KIND, NRES, 64>(S, X, qc, ncur, 0, qn, n_next, row_next, msum, mcount, lane); }
                else if (ncur <= 4) { ka_sub_meet<KIND, NRES, 16>(S, X, qc, ncur, 0, qn, n_next, row_next, msum, mcount, lane); }
                else if (ncur <= 16) { ka_sub_meet<KIND, NRES, 4>(S, X, qc, ncur, 0, qn, n_next, row_next, msum, mcount, lane); }
                else { for (int k0 = 0; k0 < ncur; k0 += 64) ka_sub_meet<KIND, NRES, 1>(S, X, qc, ncur, k0, qn, n_next, row_next, msum, mcount, lane); }
                ka_wave_lds_sync();
                ncur = n_next;
                ++level;
        }
        if (lane == 0 && mcount) { atomicAdd(&S.lctl->msum, msum); atomicAdd(&S.lctl->mcount, mcount); }
}
