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

#ifndef KA_SUB_EARLY
#define KA_SUB_EARLY 0                                           // ka_sub_pass: the next step's column record is read right behind this step's wait (1) or behind the dot products (0); round 6, measured: no difference (passes of a 430 x 420 task 464 / 466 us) -- the reads are covered either way
#endif
#define KA_SUB_MAXROWS 64                                       // a subtree in ONE wave region: decided when the sub-problem is emitted (ka_child_is_subtree)
#define KA_SUB_WIDEROWS 128                                     // ... in TWO regions (round 5): decided per recursion level at run time (ka_run_items)
#if KA_TP
#define KA_SUB_NQ 40                                            // (the throughput kernel's 12 KB regions: the floor only matters for windows of at most 32 rows)
#else
#define KA_SUB_NQ 72                                            // queue entries per level: a level has at most one sub-problem per row
#endif
#ifndef KA_WIDE_SUB
#define KA_WIDE_SUB 0                                           // measured: slower everywhere (profiles/r05_wide_subtree_levels.log) -- the engine below pays ~1300 cycles per step
#endif
__device__ __forceinline__ int ka_sub_nq(int R) { return R + 8 > KA_SUB_NQ ? R + 8 : KA_SUB_NQ; }
__device__ __forceinline__ int ka_sub_rowcells(int R, int C) { return C + 1 + (R > KA_SUB_MAXROWS ? R : KA_SUB_MAXROWS); }   // row-buffer cells of a level: one per column and sub-problem

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
        b += 2 * ka_sub_nq(R) * (int)sizeof(KaSubL);
        b += 2 * ((ka_sub_rowcells(R, C) * 12 + 15) & ~15);
        return b;
}

__device__ __forceinline__ bool ka_sub_fits(int kind, int nres, int R, int C)
{
        return R >= 1 && R <= KA_SUB_MAXROWS && C >= 1 && C < 4096 && ka_sub_bytes(kind, nres, R, C) <= KA_WAVE_LDS;
}

// inclusive prefix sum over the 64 lanes with DPP only (row_shr 1 / 2 / 4 / 8 inside rows of 16, then row_bcast:15 and
// row_bcast:31 carry the row totals up; no LDS crossbar trips)
__device__ __forceinline__ int ka_wave_scan_incl(int x)
{
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
        x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
        return x;
}

// maximum over the 64 lanes, wave-uniform result (same DPP ladder; the last lane ends with the total)
__device__ __forceinline__ int ka_wave_max_i(int x)
{
        x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false));
        x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false));
        x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false));
        x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false));
        x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false));
        x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false));
        return __builtin_amdgcn_readlane(x, 63);
}

template <int ctrl, int row_mask>
__device__ __forceinline__ double ka_dpp_f64(double x)
{
        // lanes the DPP move does not write (no source lane / masked row) read 0.0
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), ctrl, row_mask, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), ctrl, row_mask, 0xf, false);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 64 lanes (lane 63's total, broadcast)
__device__ __forceinline__ double ka_wave_sum_f64(double x)
{
        x += ka_dpp_f64<0x111, 0xf>(x);
        x += ka_dpp_f64<0x112, 0xf>(x);
        x += ka_dpp_f64<0x114, 0xf>(x);
        x += ka_dpp_f64<0x118, 0xf>(x);
        x += ka_dpp_f64<0x142, 0xa>(x);
        x += ka_dpp_f64<0x143, 0xc>(x);
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
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

// The passes pass0 .. pass0 + (64 >> sshift) - 1 of the current level (pass p = sub-problem p / 2, direction p & 1),
// 1 << sshift lanes apiece, ONE DP row per lane (the level's passes have at most 1 << sshift rows), in lock-step.
// The cell is ka_packed's; the profile-profile dot product packs two residues per v_pk_mul_f32 and adds the two
// products one after the other (as ka_strip<.., Q = 1>).  The column record of the NEXT step is fetched behind the
// dot-product chain (loads the compiler tracks; KA_UNTRACKED_READS in ka_pass.h has the history).
// NB > 0 (round 4): the anchor-consistency bonus of the lane's row (aln_profileprofile.c:108-110 and its seq-seq / seq-profile
// twins: added behind the substitution terms), the row's entries held in registers as in ka_strip / ka_packed -- until then a job
// with a consistency table ran its deep recursion levels level by level across the workgroup.
template <int KIND, int NRES, int NB = 0>
__device__ __forceinline__ void ka_sub_pass(const KaSubCtx& X, const ka_li* qc, const int npass, const int pass0, const int sshift, const int lane, const float* tss,
                                            const int2* ent = nullptr)
{
        constexpr int RW = (KIND == KA_PP) ? 4 * ((NRES + 3) / 4) + 4 : (KIND == KA_SP ? 28 : 0);
        constexpr int G0 = RW - 4;                                     // the gap chunk of a record
        constexpr int NV = (NRES + 3) / 4;
        constexpr int NPAIR = NRES / 2;
        const int p = pass0 + (lane >> sshift);
        const int ls = lane & ((1 << sshift) - 1);
        const bool live = p < npass;
        const KaSubL e = ka_subl_load(qc, live ? (p >> 1) : 0);
        const int dir = p & 1;
        const int sa = e.a & 0xffff, ea = e.a >> 16, sb = e.b & 0xffff, eb = e.b >> 16;
        const int roff = e.c & 0xffff;
        const int ncols = eb - sb;
        const int mid = ((ea - sa) / 2) + sa;
        const int r0 = (dir == KA_FWD) ? sa : mid;
        const int r1 = (dir == KA_FWD) ? mid : ea;
        const int nrows = r1 - r0;                                    // 0 .. 1 << sshift
        const bool near_t = (dir == KA_FWD) ? (X.b0 + sb == 0) : (X.b0 + eb == X.Lb);
        const bool far_t = (dir == KA_FWD) ? (X.b0 + eb == X.Lb) : (X.b0 + sb == 0);
        const KaState inj = (dir == KA_FWD) ? ka_sub_state((e.c >> 16) & 3, X.rfin) : ka_sub_state((e.c >> 18) & 3, X.rbin);
        ka_lf* const rowbuf = ((dir == KA_FWD) ? X.F : X.B) + 3 * roff;

#define SREC(v_) ((dir == KA_FWD) ? (sb + (v_)) : (eb + 1 - (v_)))     /* column record, relative to b0 */
#define SIDX(v_) ((dir == KA_FWD) ? (v_) : (ncols - (v_)))

        const bool writer = live && (ls == max(nrows - 1, 0));        // owner of the pass's last row (or of the init row)
        const int uA = min(ls, max(nrows - 1, 0));
        const int iA = (dir == KA_FWD) ? (r0 + uA) : (r1 - 1 - uA);   // DP row, relative to a0
        const int recA = min(max(iA + 1, 0), X.R + 1);
        const int prevA = min(max((dir == KA_FWD) ? recA - 1 : recA + 1, 0), X.R + 1);

        KaBonus<NB> bon;
        if (NB) bon.load(ent, min(max(X.a0 + iA, 0), X.La - 1), dir);
        float oA, eA, tA, orpA;
        float2v p1p[KIND == KA_PP ? (NPAIR > 0 ? NPAIR : 1) : 1];     // the counts of residues (2i, 2i+1)
        float p1last = 0.0f;
        int res1A = 0;
        const ka_lf* srowA = nullptr;
        if (KIND == KA_SS) {
                oA = -X.gpo; eA = -X.gpe; tA = -X.tgpe; orpA = -X.gpo;
                res1A = X.rowres[min(max(iA, 0), X.R - 1)] * KA_T_STRIDE;
        } else {
                const ka_lf* pA = X.rowsL + recA * RW;
                const float4v ga = *(const ka_lf4*)(pA + G0);
                oA = ga.x; eA = ga.y; tA = ga.z;
                orpA = X.rowsL[prevA * RW + G0];
                if (KIND == KA_PP) {
                        float4v va[NV];
#pragma unroll
                        for (int i = 0; i < NV; ++i) va[i] = ((const ka_lf4*)pA)[i];
#pragma unroll
                        for (int i = 0; i < NPAIR; ++i) { p1p[i].x = va[(2 * i) >> 2][(2 * i) & 3]; p1p[i].y = va[(2 * i + 1) >> 2][(2 * i + 1) & 3]; }
                        if (NRES & 1) p1last = va[(NRES - 1) >> 2][(NRES - 1) & 3];
                } else {
                        srowA = pA;                                   // seq-profile: the score row stays in LDS, indexed by the column's residue
                }
        }

        float cAa = -KA_F, cAga = -KA_F, cAgb = -KA_F;
        float dga = -KA_F, dgga = -KA_F, dggb = -KA_F;
        float inia = inj.a, iniga = inj.ga, inigb = inj.gb;
        float copen_prev = 0.0f;
        float4v q[2][KIND == KA_PP ? NV + 1 : 1];
        int resq[2] = {0, 0};

        // profile columns: the record of column counter vcol (clamped), untracked reads + a manual wait (see ka_strip)
        const unsigned cols_u = (unsigned)(unsigned long long)X.colsL;
        auto rec_addr = [&](int vcol) -> unsigned { const int vv = min(max(vcol, 0), ncols); return cols_u + (unsigned)(SREC(vv) * (RW * 4)); };
        auto pp_read = [&](float4v* dstq, unsigned a, auto& dep) {
                if (!KA_UNTRACKED_READS) {
                        // (tracked loads: see KA_UNTRACKED_READS in ka_pass.h)
                        const ka_lf4* const cr = (const ka_lf4*)(unsigned long)a;
                        asm volatile("" : "+v"(dep) : : "memory");
#pragma unroll
                        for (int ch = 0; ch <= NV; ++ch) dstq[ch] = cr[ch];
                        return;
                }
                if (NV == 2) {
                        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32"
                                     : "=&v"(dstq[0]), "=&v"(dstq[1]), "=&v"(dstq[2]), "+v"(dep) : "v"(a) : "memory");
                } else if (NV == 5) {
                        asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:32\n\t"
                                     "ds_read_b128 %3, %7 offset:48\n\tds_read_b128 %4, %7 offset:64\n\tds_read_b128 %5, %7 offset:80"
                                     : "=&v"(dstq[0]), "=&v"(dstq[1]), "=&v"(dstq[2]), "=&v"(dstq[3]), "=&v"(dstq[4]), "=&v"(dstq[5]), "+v"(dep) : "v"(a) : "memory");
                } else {
                        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
                                     "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                                     "ds_read_b128 %6, %8 offset:96"
                                     : "=&v"(dstq[0]), "=&v"(dstq[1]), "=&v"(dstq[2]), "=&v"(dstq[3]), "=&v"(dstq[4]), "=&v"(dstq[5]), "=&v"(dstq[6]), "+v"(dep) : "v"(a) : "memory");
                }
        };
        auto pp_wait = [&](float4v* qq) {
                if (!KA_UNTRACKED_READS) {
                        if (NV == 2) asm volatile("" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]) : : "memory");
                        else if (NV == 5) asm volatile("" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]), "+v"(qq[4]), "+v"(qq[5]) : : "memory");
                        else asm volatile("" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]), "+v"(qq[4]), "+v"(qq[5]), "+v"(qq[6]) : : "memory");
                        return;
                }
                if (NV == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]) : : "memory");
                else if (NV == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]), "+v"(qq[4]), "+v"(qq[5]) : : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]), "+v"(qq[4]), "+v"(qq[5]), "+v"(qq[6]) : : "memory");
        };
        static_assert(KIND != KA_PP || NV == 2 || NV == 5 || NV == 6, "alphabets of 5, 20 and 23 residues");
        auto res_fetch = [&](int vcol) -> int {
                // residue of column record rec sits at rec - 1 of the sequence (clamped like ka_packed does)
                const int vv = min(max(vcol, 0), ncols);
                return X.colres[min(max(SREC(max(vv, 1)) - 1, 0), X.C - 1) + 1];
        };
        if (KIND == KA_PP) { float nodep = 0.0f; pp_read(q[0], rec_addr(-ls), nodep); }
        else resq[0] = res_fetch(-ls);

        const int nsteps = ka_wave_max_i(live ? (ncols + max(nrows, 1)) : 0);

        auto step = [&](const int t, auto par_tag) {
                constexpr int P = decltype(par_tag)::value;
                const int v = t - ls;
                const bool vin = live && (v >= 0) && (v <= ncols);
                if (KIND != KA_PP) resq[1 - P] = res_fetch(v + 1);

                float copen, cext, ctext;
                if (KIND == KA_PP) {
                        pp_wait(q[P]); copen = q[P][NV].x; cext = q[P][NV].y; ctext = q[P][NV].z;
#if KA_SUB_EARLY
                        // (the next step's record right behind this step's wait, as ka_wstrip's KA_W_EARLY: a whole step covers the reads)
                        __builtin_amdgcn_sched_barrier(0);
                        pp_read(q[1 - P], rec_addr(v + 1), copen);
                        __builtin_amdgcn_sched_barrier(0);
#endif
                }
                else { copen = X.kc_open; cext = X.kc_ext; ctext = X.kc_text; }
                {
                        const float gx = near_t ? ctext : cext, gy = near_t ? ctext : copen;
                        const float g = kmax(iniga + gx, inia + gy);
                        const bool v0 = (v == 0), vmid = (v < ncols);
                        inia = v0 ? inj.a : -KA_F;
                        iniga = v0 ? inj.ga : (vmid ? g : -KA_F);
                        inigb = v0 ? inj.gb : -KA_F;
                }
                const float sha = wave_shr1(cAa), shga = wave_shr1(cAga), shgb = wave_shr1(cAgb);
                const float upa = (ls == 0) ? inia : sha, upga = (ls == 0) ? iniga : shga, upgb = (ls == 0) ? inigb : shgb;

                float a1 = kmax3(dga, dgga + copen_prev, dggb + orpA);
                if (KIND == KA_SS) {
                        a1 += tss[res1A + resq[P]];
                } else if (KIND == KA_SP) {
                        a1 += srowA[resq[P]];
                } else {
                        if (NRES & 1) a1 += p1last * q[P][(NRES - 1) >> 2][(NRES - 1) & 3];
                        if (NPAIR > 0) {
                                auto qpair = [&](int i) -> float2v {
                                        const float4v& w = q[P][(2 * i) >> 2];
                                        return ((2 * i) & 3) ? __builtin_shufflevector(w, w, 2, 3) : __builtin_shufflevector(w, w, 0, 1);
                                };
                                float2v prod = p1p[NPAIR - 1] * qpair(NPAIR - 1);
#pragma unroll
                                for (int i = NPAIR - 1; i >= 1; --i) {
                                        const float2v nprod = p1p[i - 1] * qpair(i - 1);
                                        a1 += prod.y;
                                        a1 += prod.x;
                                        prod = nprod;
                                }
                                a1 += prod.y;
                                a1 += prod.x;
                        }
#if !KA_SUB_EARLY
                        __builtin_amdgcn_sched_barrier(0);
                        pp_read(q[1 - P], rec_addr(v + 1), a1);
                        __builtin_amdgcn_sched_barrier(0);
#endif
                }
                if (NB) { const int jb = (dir == KA_FWD) ? (X.b0 + sb + v) : (X.b0 + eb - v); a1 += bon.template at<true>(jb); }
                const bool at0 = (v == 0), atN = (v == ncols);
                const bool edge = at0 | atN;
                const bool term = (at0 & near_t) | (atN & far_t);           // (bitwise: as short-circuit logic this became exec-masked branches in every edge step)
                // selects only (a branchy cell costs more than the arithmetic it skips); max(x, y) + c == max(x + c, y + c) bit
                // for bit (rounding is monotonic), so the terminal and the inner form of the gb state share one expression
                const float nAa = at0 ? -KA_F : a1;
                const float ga_in = kmax(cAga + cext, cAa + copen);
                const float nAga = edge ? -KA_F : ga_in;
                const float gbx = term ? tA : eA, gby = term ? tA : oA;
                const float nAgb = kmax(upgb + gbx, upa + gby);
                cAa = nAa; cAga = nAga; cAgb = nAgb;
                dga = upa; dgga = upga; dggb = upgb;
                copen_prev = copen;
                const float wa = (nrows == 0) ? inia : cAa, wga = (nrows == 0) ? iniga : cAga, wgb = (nrows == 0) ? inigb : cAgb;
                if (vin && writer) {
                        ka_lf* w = rowbuf + 3 * SIDX(v);
                        w[0] = wa; w[1] = wga; w[2] = wgb;
                }
        };
        int t = 0;
        for (; t + 1 < nsteps; t += 2) {
                step(t, std::integral_constant<int, 0>());
                step(t + 1, std::integral_constant<int, 1>());
        }
        if (t < nsteps) step(t, std::integral_constant<int, 0>());
        if (KIND == KA_PP) pp_wait(q[nsteps & 1]);                    // nothing of this pass is in flight when the LDS is reused
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
        // group reduction with DPP shifts towards the higher lanes (no LDS crossbar trips): the LAST lane of the group ends
        // with the group's answer -- lanes without a source lane merge the identity; what the other lanes hold is not used
#define KA_BEST_STEP(ctrl_, rmask_)                                                                                           \
        {                                                                                                                     \
                const float omx = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-KA_F), __float_as_int(Bt.mx), ctrl_, rmask_, 0xf, false));   \
                const float omx2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-KA_F), __float_as_int(Bt.mx2), ctrl_, rmask_, 0xf, false)); \
                const int okey = __builtin_amdgcn_update_dpp(0x7fffffff, Bt.key, ctrl_, rmask_, 0xf, false);                   \
                best_merge(Bt, omx, omx2, okey);                                                                              \
        }
        if (GL >= 2) KA_BEST_STEP(0x111, 0xf)
        if (GL >= 4) KA_BEST_STEP(0x112, 0xf)
        if (GL >= 8) KA_BEST_STEP(0x114, 0xf)
        if (GL >= 16) KA_BEST_STEP(0x118, 0xf)
        if (GL >= 64) { KA_BEST_STEP(0x142, 0xa) KA_BEST_STEP(0x143, 0xc) }
#undef KA_BEST_STEP
        const bool leader = (lane == GL - 1) && valid;
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
        // (both counts in one word: at most 128 slots and < 65536 row cells per level)
        const int sc = ka_wave_scan_incl(nsl | (nrw << 8));
        const int sc1 = sc & 0xff, sc2 = sc >> 8;
        const int tots = __builtin_amdgcn_readlane(sc, 63);
        const int tot1 = tots & 0xff, tot2 = tots >> 8;
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
        const bool has2 = leader && Bt.mx2 > -KA_F;
        msum += ka_wave_sum_f64(has2 ? (double)(Bt.mx - Bt.mx2) : 0.0);
        mcount += __builtin_popcountll(__ballot(has2));
}

// Context of a subtree: origin, sizes, penalties; the wave's region carved (same arithmetic as ka_sub_bytes); both operand
// windows staged into LDS; the root in q[0][0].
template <int KIND, int NRES>
__device__ __forceinline__ void ka_sub_setup(TaskShared& S, const KaSub& root, const int lane, char* area, KaSubCtx& X)
{
        constexpr int RW = (KIND == KA_PP) ? 4 * ((NRES + 3) / 4) + 4 : (KIND == KA_SP ? 28 : 0);
        constexpr int G0 = RW - 4;
        constexpr int NV = (NRES + 3) / 4;
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
        X.q[0] = (ka_li*)(area + o); o += ka_sub_nq(X.R) * (int)sizeof(KaSubL);
        X.q[1] = (ka_li*)(area + o); o += ka_sub_nq(X.R) * (int)sizeof(KaSubL);
        const int rbytes = (ka_sub_rowcells(X.R, X.C) * 12 + 15) & ~15;
        X.F = (ka_lf*)(area + o); o += rbytes;
        X.B = (ka_lf*)(area + o);

        // ---- stage the operand windows: 16-byte chunks, four loads in flight per lane before their LDS writes ----
        const float m1 = ka_uniform_f(S.p1_mult), m2 = ka_uniform_f(S.p2_mult);
        // one record = `per` chunks: nsc chunks of 4 floats from field `src0` on, then (open, ext, text) * mult
        // (round 6: two loops without a branch between their loads -- the score chunks, then the gap fields -- through global pointers: a
        // flat load counts on vmcnt AND lgkmcnt, and a branch between two loads costs a wait for everything in flight; ka_update_profile
        // has the measurement of the same change)
        auto stage = [&](const float* prof, const int rec0, const int nrec, const int src0, const int nsc, const float mult, ka_lf* dst) {
                typedef const __attribute__((address_space(1))) float* ka_gfp;
                typedef const __attribute__((address_space(1))) float4v* ka_gf4p;
                const ka_gfp gp = (ka_gfp)prof + ((long long)rec0 << 6);
                const int nit = nrec * nsc;
                for (int base = 0; base < nit; base += 256) {
                        float4v val[4];
                        int off[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                                const int it = min(base + 64 * u + lane, nit - 1);
                                const int k = it / nsc, ch = it - k * nsc;
                                val[u] = *(ka_gf4p)(gp + ((long long)k << 6) + src0 + 4 * ch);
                                off[u] = k * RW + 4 * ch;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                                if (base + 64 * u + lane < nit) *(ka_lf4*)(dst + off[u]) = val[u];
                }
                // (open, ext, text) * mult: fields 55 .. 57 of every record
                for (int base = 0; base < nrec; base += 128) {
                        float g[2][3];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                                const int k = min(base + 64 * u + lane, nrec - 1);
                                const ka_gfp r = gp + ((long long)k << 6);
                                g[u][0] = r[55]; g[u][1] = r[56]; g[u][2] = r[57];
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                                const int k = base + 64 * u + lane;
                                if (k >= nrec) continue;
                                float4v w;
                                w.x = g[u][0] * mult; w.y = g[u][1] * mult; w.z = g[u][2] * mult; w.w = 0.0f;
                                *(ka_lf4*)(dst + k * RW + G0) = w;
                        }
                }
        };
        if (KIND == KA_SS) {
                for (int i = lane; i < X.R; i += 64) X.rowres[i] = S.s1[X.a0 + i];
        } else {
                // records a0 .. a0 + R + 1 of the row profile: PP rows carry the counts, SP rows the 23 scores
                stage(S.p1, X.a0, X.R + 2, (KIND == KA_PP) ? 0 : 32, (KIND == KA_PP) ? NV : 6, m1, X.rowsL);
        }
        if (KIND == KA_PP) {
                stage(S.p2, X.b0, X.C + 2, 32, NV, m2, X.colsL);
        } else {
                // colres[1 + j] = residue at position b0 + j of the column sequence, j = 0 .. C-1 (ka_sub_pass clamps into that range)
                for (int i = lane; i < X.C; i += 64) X.colres[1 + i] = S.s2[X.b0 + i];
        }
        if (lane == 0) {
                KaSubL e; e.a = 0 | (X.R << 16); e.b = 0 | (X.C << 16); e.c = 0;
                ka_subl_store(X.q[0], 0, e);
        }
        ka_wave_lds_sync();
}

// The whole subtree below `root` by the calling wave.  area: the wave's LDS region (KA_WAVE_LDS bytes).
template <int KIND, int NRES, int NB = 0>
__device__ __forceinline__ void ka_subtree(TaskShared& S, const KaSub root, const int lane, char* area, const float* tss)
{
        KaSubCtx X;
        const bool tmg = S.sub_tm != 0;                               // KA_FLAG_TIMING on the profiled task: where a subtree's time goes
        long long tq0 = 0, tq1 = 0, tpass = 0, tmeet = 0;
        if (tmg) tq0 = __builtin_amdgcn_s_memtime();
        ka_sub_setup<KIND, NRES>(S, root, lane, area, X);
        if (tmg) tq1 = __builtin_amdgcn_s_memtime();

        int ncur = 1, level = 0;
        double msum = 0.0;
        int mcount = 0;
        while (ncur > 0) {
                const ka_li* qc = X.q[level & 1];
                ka_li* qn = X.q[(level + 1) & 1];
                // ---- passes: the slot size follows the level's longest pass ----
                int maxrows = 0;
                for (int k = lane; k < ncur; k += 64) { const KaSubL e = ka_subl_load(qc, k); const int r = (e.a >> 16) - (e.a & 0xffff); maxrows = max(maxrows, r - r / 2); }
                maxrows = ka_wave_max_i(maxrows);
                const int npass = 2 * ncur;
                // slot = the power of two that holds the level's longest pass, one row per lane
                int sshift = 0;
                while ((1 << sshift) < maxrows) ++sshift;
                long long tl0 = 0, tl1 = 0;
                if (tmg) tl0 = __builtin_amdgcn_s_memtime();
                for (int p0 = 0; p0 < npass; p0 += 64 >> sshift) ka_sub_pass<KIND, NRES, NB>(X, qc, npass, p0, sshift, lane, tss, S.ent);
                ka_wave_lds_sync();
                if (tmg) tl1 = __builtin_amdgcn_s_memtime();
                // ---- meetups and children ----
                int n_next = 0, row_next = 0;
                if (ncur <= 1) { ka_sub_meet<KIND, NRES, 64>(S, X, qc, ncur, 0, qn, n_next, row_next, msum, mcount, lane); }
                else if (ncur <= 4) { ka_sub_meet<KIND, NRES, 16>(S, X, qc, ncur, 0, qn, n_next, row_next, msum, mcount, lane); }
                else if (ncur <= 16) { ka_sub_meet<KIND, NRES, 4>(S, X, qc, ncur, 0, qn, n_next, row_next, msum, mcount, lane); }
                else { for (int k0 = 0; k0 < ncur; k0 += 64) ka_sub_meet<KIND, NRES, 1>(S, X, qc, ncur, k0, qn, n_next, row_next, msum, mcount, lane); }
                ka_wave_lds_sync();
                if (tmg) { const long long tl2 = __builtin_amdgcn_s_memtime(); tpass += tl1 - tl0; tmeet += tl2 - tl1; }
                ncur = n_next;
                ++level;
        }
        if (tmg && lane == 0) {
                const long long tot = __builtin_amdgcn_s_memtime() - tq0;
#ifndef KA_L_PROF
                atomicAdd(&S.sub_t[0], 1ull); atomicAdd(&S.sub_t[1], (unsigned long long)(tq1 - tq0));
                atomicAdd(&S.sub_t[2], (unsigned long long)tpass); atomicAdd(&S.sub_t[3], (unsigned long long)tmeet);
                atomicAdd(&S.sub_t[4], (unsigned long long)tot); atomicMax(&S.sub_t[5], (unsigned long long)tot);
                atomicAdd(&S.sub_t[6], (unsigned long long)(level * 1000000 + X.R * 1000 + min(X.C, 999)));
#endif
        }
        if (lane == 0 && mcount) { atomicAdd(&S.lctl->msum, msum); atomicAdd(&S.lctl->mcount, mcount); }
}

// ------------------------------------------------------------------------------------------------------------------
// The same subtree DEPTH FIRST, for refinement's flip trials (aln_refine.c:93-346): a trial flips every stride-th
// uncertain meetup in recursion order, so the decisions are sequential (ka_hirschberg_dfs) -- but the operands, the row
// buffers, the stack and the candidates of a small subtree need not leave LDS while one wave walks it.  Per node:
// decision (flip rule, margin log, path entries, the children's windows), the passes of BOTH children in one
// ka_sub_pass call, their candidate scans 32 lanes apiece (runner-up key included), both on the stack with their
// candidates -- the child the recursion enters first on top.  `root` arrives with its candidates (its passes ran with
// its sibling's).  area: LDS the workgroup does not use while wave 0 walks (ka_sub_bytes of the window).
// ------------------------------------------------------------------------------------------------------------------
template <int KIND, int NRES>
__device__ __forceinline__ void ka_subtree_dfs(TaskShared& S, const KaSub root, const Best rootB, const bool root_is_top, const int lane,
                                               char* area, const float* tss)
{
        constexpr int RW = (KIND == KA_PP) ? 4 * ((NRES + 3) / 4) + 4 : (KIND == KA_SP ? 28 : 0);
        constexpr int G0 = RW - 4;
        KaSubCtx X;
        ka_sub_setup<KIND, NRES>(S, root, lane, area, X);
        ka_li* fly = X.q[0];                                          // the (up to two) sub-problems whose passes run
        ka_li* stack = X.q[1];                                        // 8 ints per entry: KaSubL, then mx, mx2, key, key2
        if (lane == 0) {
                stack[0] = 0 | (X.R << 16); stack[1] = 0 | (X.C << 16); stack[2] = 0;
                stack[4] = __float_as_int(rootB.mx); stack[5] = __float_as_int(rootB.mx2); stack[6] = rootB.key; stack[7] = rootB.key2;
        }
        ka_wave_lds_sync();
        int h = 1;                                                    // stack height (wave-uniform)
        bool top = root_is_top;
        while (h > 0) {
                --h;
                // ---- the decision of the node on top of the stack (every lane computes it; lane 0 has the side effects) ----
                const int ea_ = stack[8 * h], eb_ = stack[8 * h + 1], ec_ = stack[8 * h + 2];
                Best B;
                B.mx = __int_as_float(stack[8 * h + 4]); B.mx2 = __int_as_float(stack[8 * h + 5]); B.key = stack[8 * h + 6]; B.key2 = stack[8 * h + 7];
                const int sa = ea_ & 0xffff, ea = ea_ >> 16, sb = eb_ & 0xffff, eb = eb_ >> 16;
                const int fcode = (ec_ >> 16) & 3, bcode = (ec_ >> 18) & 3;
                const int mid = ((ea - sa) / 2) + sa;
                int meet = -1, tr = -1;
                if (B.key != 0x7fffffff) { const int ord = B.key & 7; meet = sb + (B.key >> 3); tr = ord + 1 + (ord >= 3 ? 1 : 0); }
                if (top && lane == 0) { S.ctl->top_meet = (meet >= 0) ? X.b0 + meet : -1; S.ctl->top_tr = tr; S.ctl->top_score = B.mx; }
                top = false;
                const bool has2 = B.mx2 > -KA_F;
                const float margin = B.mx - B.mx2;
                // (the trial's state: read by every lane before lane 0 changes it -- one wave, LDS in program order)
                const float thr = S.rf.thr;
                const int trial = S.rf.trial, stride = S.rf.stride, counter = S.rf.counter, mcount = S.rf.mcount;
                const float msum = S.rf.msum;
                const bool uncertain = thr > 0.0f && B.key2 != 0x7fffffff && has2 && margin < thr;     // aln_seqseq.c:376-414
                if (uncertain && trial > 0 && counter % stride == trial - 1) {
                        const int ord2 = B.key2 & 7;
                        meet = sb + (B.key2 >> 3);
                        tr = ord2 + 1 + (ord2 >= 3 ? 1 : 0);
                }
                if (lane == 0) {
                        if (has2) {
                                if (S.mlog && mcount < S.mlog_cap) S.mlog[mcount] = margin;           // aln_seqseq.c:378-380
                                S.rf.msum = msum + margin; S.rf.mcount = mcount + 1;
                        }
                        if (uncertain) S.rf.counter = counter + 1;
                }
                // aln_continue (aln_controller.c:194-436): path entries and the two child windows
                int c1sa = sa, c1ea = sa, c1sb = sb, c1eb = sb, c1bc = 1;
                int c2sa = ea, c2ea = ea, c2sb = eb, c2eb = eb, c2fc = 1;
                if (tr > 0) {
                        int* path = S.raw;
                        const int am = X.a0 + mid, bm = X.b0 + meet;
                        const bool w = (lane == 0);
                        switch (tr) {
                        case 1:
                                if (w) { path[am] = bm; path[am + 1] = bm + 1; }
                                c1ea = mid - 1; c1eb = meet - 1; c1bc = 1;
                                c2sa = mid + 1; c2sb = meet + 1; c2fc = 1;
                                break;
                        case 2:
                                if (w) path[am] = bm;
                                c1ea = mid - 1; c1eb = meet - 1; c1bc = 1;
                                c2sa = mid; c2sb = meet + 1; c2fc = 2;
                                break;
                        case 3:
                                if (w) path[am] = bm;
                                c1ea = mid - 1; c1eb = meet - 1; c1bc = 1;
                                c2sa = mid + 1; c2sb = meet; c2fc = 3;
                                break;
                        case 5:
                                if (w) path[am + 1] = bm + 1;
                                c1ea = mid; c1eb = meet - 1; c1bc = 2;
                                c2sa = mid + 1; c2sb = meet + 1; c2fc = 1;
                                break;
                        case 6:
                                c1ea = mid - 1; c1eb = meet; c1bc = 3;
                                c2sa = mid + 1; c2sb = meet; c2fc = 3;
                                break;
                        default: /* 7 */
                                if (w) path[am + 1] = bm + 1;
                                c1ea = mid - 1; c1eb = meet; c1bc = 3;
                                c2sa = mid + 1; c2sb = meet + 1; c2fc = 1;
                                break;
                        }
                }
                const bool v1 = (tr > 0) && c1sa < c1ea && c1sb < c1eb;
                const bool v2 = (tr > 0) && c2sa < c2ea && c2sb < c2eb;
                const int n = (v1 ? 1 : 0) + (v2 ? 1 : 0);
                if (n == 0) { ka_wave_lds_sync(); continue; }
                if (lane == 0) {
                        int slot = 0, row = 0;
                        if (v1) { fly[0] = c1sa | (c1ea << 16); fly[1] = c1sb | (c1eb << 16); fly[2] = row | (fcode << 16) | (c1bc << 18); slot = 1; row = c1eb - c1sb + 1; }
                        if (v2) { fly[3 * slot] = c2sa | (c2ea << 16); fly[3 * slot + 1] = c2sb | (c2eb << 16); fly[3 * slot + 2] = row | (c2fc << 16) | (bcode << 18); }
                }
                ka_wave_lds_sync();
                // ---- the passes of the children ----
                {
                        const int r1 = v1 ? (c1ea - c1sa) : 0, r2 = v2 ? (c2ea - c2sa) : 0;
                        const int maxrows = max(r1 - r1 / 2, r2 - r2 / 2);
                        int sshift = 0;
                        while ((1 << sshift) < maxrows) ++sshift;
                        const int npass = 2 * n;
                        for (int p0 = 0; p0 < npass; p0 += 64 >> sshift) ka_sub_pass<KIND, NRES>(X, fly, npass, p0, sshift, lane, tss);
                }
                ka_wave_lds_sync();
                // ---- their candidates, 32 lanes per child ----
                {
                        const int gidx = lane >> 5, gl = lane & 31;
                        const bool valid = gidx < n;
                        const KaSubL e = ka_subl_load(fly, valid ? gidx : 0);
                        const int csa = e.a & 0xffff, cea = e.a >> 16, csb = e.b & 0xffff, ceb = e.b >> 16;
                        const int roff = e.c & 0xffff;
                        const int startb = X.b0 + csb, endb = X.b0 + ceb;
                        const int cmid = ((cea - csa) / 2) + csa;
                        const ka_lf* f = X.F + 3 * roff;
                        const ka_lf* b = X.B + 3 * roff;
                        const float middle = (float)(endb - startb) / 2.0f + (float)startb;
                        float g3, g7, g6n, g6f;
                        if (KIND == KA_SS) {
                                g3 = -X.gpo; g7 = -X.gpo;
                                g6n = (startb == 0) ? -X.tgpe : -X.gpe;
                                g6f = (endb == X.Lb) ? -X.tgpe : -X.gpe;
                        } else {
                                const ka_lf* Rr = X.rowsL + (cmid + 1) * RW + G0;
                                g3 = Rr[0]; g7 = Rr[-RW];
                                g6n = (startb == 0) ? Rr[2] : Rr[1];
                                g6f = (endb == X.Lb) ? Rr[2] : Rr[1];
                        }
                        Best Bt = { -KA_F, -KA_F, 0x7fffffff, 0x7fffffff };
                        for (int i = csb + gl; valid && i <= ceb; i += 32) {
                                const int x = i - csb;
                                const float fa = f[3 * x], fga = f[3 * x + 1], fgb = f[3 * x + 2];
                                const float ba = b[3 * x], bga = b[3 * x + 1], bgb = b[3 * x + 2];
                                float sub = fabsf(middle - (float)(X.b0 + i));
                                sub = sub / 1000.0f;
                                const int kb = x * 8;
                                if (i < ceb) {
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
                        // reduction towards the last lane of each 32-lane group, runner-up key included
#define KA_BEST2_STEP(ctrl_, rmask_)                                                                                          \
                        {                                                                                                     \
                                const float omx = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-KA_F), __float_as_int(Bt.mx), ctrl_, rmask_, 0xf, false));   \
                                const float omx2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-KA_F), __float_as_int(Bt.mx2), ctrl_, rmask_, 0xf, false)); \
                                const int okey = __builtin_amdgcn_update_dpp(0x7fffffff, Bt.key, ctrl_, rmask_, 0xf, false);   \
                                const int okey2 = __builtin_amdgcn_update_dpp(0x7fffffff, Bt.key2, ctrl_, rmask_, 0xf, false); \
                                best_merge(Bt, omx, omx2, okey, okey2);                                                       \
                        }
                        KA_BEST2_STEP(0x111, 0xf)
                        KA_BEST2_STEP(0x112, 0xf)
                        KA_BEST2_STEP(0x114, 0xf)
                        KA_BEST2_STEP(0x118, 0xf)
                        KA_BEST2_STEP(0x142, 0xa)                     // row_bcast:15 into rows 1 and 3: lanes 31 and 63 hold their group's answer
#undef KA_BEST2_STEP
                        // the child the recursion enters first (index 0) ends on top of the stack
                        if (gl == 31 && valid) {
                                const int pos = h + (n - 1 - gidx);
                                stack[8 * pos] = e.a; stack[8 * pos + 1] = e.b; stack[8 * pos + 2] = e.c;
                                stack[8 * pos + 4] = __float_as_int(Bt.mx); stack[8 * pos + 5] = __float_as_int(Bt.mx2);
                                stack[8 * pos + 6] = Bt.key; stack[8 * pos + 7] = Bt.key2;
                        }
                }
                h += n;
                ka_wave_lds_sync();
        }
}
