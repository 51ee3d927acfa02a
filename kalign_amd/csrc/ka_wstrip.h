// ka_wstrip.h -- profile-profile strips with a HELPER WAVE (round 4).
//
// ka_strip (ka_pass.h) does everything itself: every 32 steps it issues the global->LDS copy of the next column batch, every
// 64 steps it flushes its last row behind a release fence and reloads its boundary batch behind an acquire.  Those "event
// steps" cost 1700-3500 cycles against ~404 for a plain step (profiles/r03b_strip_phases.log): 20-30 % of every pass, and
// their code (and its spills) sits inside the strip's loop.  A lone wave issues one instruction per ~4.4 cycles of ANY kind,
// while its SIMD could issue a scalar, an LDS and a memory instruction of ANOTHER wave in the same cycles -- and at the
// levels that matter for latency (at most four work items per 8-wave workgroup: one strip per SIMD) waves 4..7 of the
// workgroup have nothing to do.
//
// So here the strip wave only computes.  Everything it touches is in LDS:
//   * its column ring (same layout as ka_strip's: chunk-major, 128 columns), filled by its helper (wave + 4), which also
//     applies set_gap_penalties_n's multiplier to the three gap fields (three multiplies per COLUMN instead of per step);
//   * its boundary row ("the row above"): the out ring of the strip above when that runs on the wave before it, else an in
//     ring in the helper's region that the helper fills -- for the first strip of a pass with the pass's generated row -1
//     (aln_seqseq.c:40-58: a serial chain along the columns, which ka_strip's first strips carried in every step), for a
//     strip whose upper neighbour runs in another workgroup with that strip's last row, fetched from the HBM row buffer
//     behind its progress flag;
//   * its out ring: the strip's last row, one ds_write per step by the lane that owns it (no shift register, no
//     rotate-and-insert for partial strips, no flush).  The strip below reads it in place, or the helper copies it to the
//     HBM row buffer (and publishes it to a strip in another workgroup: release fence + flag -- in the helper).
// Steps run in OCTETS: eight unrolled steps (immediate LDS offsets for boundary and out ring), then the strip publishes its
// step count (one ds_write) and compares two LDS words read one step earlier with what the next octet needs ("columns
// loaded / out slots free" from the helper, "boundary columns available" from the producer): ~8 instructions per 8 steps,
// no branch taken unless it must wait.  A strip therefore starts 63 + ~8 columns behind the strip above it, not 63 + 64.
//
// Same arithmetic as ka_strip, statement for statement (binary32, the reference's order, no contraction).
#pragma once

#define KA_W_CH 8                                               // steps per octet
#ifndef KA_W_TAG
#define KA_W_TAG 0                                              // debugging: every ring slot of a boundary row carries its column; a strip that reads another column reports error 9
#endif
#if KA_W_TAG
#define KA_W_DSW "ds_write_b128"
#else
#define KA_W_DSW "ds_write_b96"
#endif
#ifndef KA_W_FORMS
#define KA_W_FORMS 1                                            // head / tail forms of the step (0, experiment: the general edge form)
#endif
#ifndef KA_W_EARLY
#define KA_W_EARLY 1                                            // next step's LDS reads right behind this step's wait (0: after the dot products, as in ka_strip)
#endif
#define KA_W_BIG (1 << 30)
typedef float float3v __attribute__((ext_vector_type(3)));

// control words of helper mode: 16 ints at lds_waves - KA_LDS_HO_BACK (the words ka_strip<.., HO> uses in ITS mode; a level
// runs in one mode or the other, ka_hirschberg zeroes them between levels):
//   [w]      steps strip wave w has completed (its t_pub)
//   [4 + w]  GO: columns strip w may read / steps it may take, from its helper (see ka_whelper)
//   [8 + w]  boundary columns in strip w's helper-fed in ring
#define KA_W_TPUB(w_) (w_)
#define KA_W_GO(w_) (4 + (w_))
#define KA_W_IN(w_) (8 + (w_))
#define KA_W_INRING 0                                           // the in ring: first 4 KB of the HELPER wave's region (256 slots x 16 B)

// Q: DP rows per lane.  Q = 2: 128-row strips, ~83 instructions per step.  Q = 1: 64-row strips, ~57 instructions per step (the
// products pack two RESIDUES per v_pk_mul_f32, the sums of the one row stay scalar) and twice the strips per pass: a pass of K
// 128-row strips costs (C + 64 + (K - 1) * D) steps either way in STEPS (D ~ 80: the lane skew plus the hand-over) with 2K
// strips in place of K, but a step is 0.63 of the two-row step -- with ka_strip's 127-column hand-over delay the narrow strips
// were a wash (round 3), with this one they win wherever the cluster has a SIMD per strip (TaskShared::srows says which).
//
// ka_wstrip and ka_whelper take their arguments as a struct by value so that they can be built as REAL functions: inlined into
// the task kernel -- one __global__ with three DP kinds, the recursion, meetups, path coding and the profile merge inlined
// into it -- the strip's loop shares the kernel's register allocation with everything that is live across it and sits at the
// edge of 256 VGPRs (a version with eight more live values tipped over: 700 VGPR spills, 80 scratch accesses per STEP, and
// wrong results on top).  tools/check_hot_loops.py disassembles the built object and counts scratch accesses per strip step.
// KA_W_NOINLINE=1 (experiment, -D at build time): as real functions with an allocation of their own.  Measured (r04_ab5): the
// strip's octets then carry no scratch access at all (inlined, the head and tail forms keep up to five per step), but every
// OTHER path of the kernel pays for the calls in its loop -- ka_strip's levels +10 %, the 1024 x 2000 nucleotide tree 37.8 ->
// 42.4 ms, the protein headline unchanged -- so the default stays inlined.
#ifndef KA_W_NOINLINE
#define KA_W_NOINLINE 0
#endif
// KA_W_UNTRACKED_EDGE=1 rebuilds round 4's bug on purpose (untracked LDS reads in the head / tail octets too, where the
// allocator spills): what tools/check_lds_hazards.py is tested against.  Never in a product build.
// KA_W_UNTRACKED (1): the steady-state octets issue the next step's reads as inline asm the compiler does not track, right
// behind the step's wait (KA_UNTRACKED_READS in ka_pass.h has the hazard this can open; tools/check_lds_hazards.py must
// find the built objects clean).  0: plain loads there too.
#ifndef KA_W_UNTRACKED
#define KA_W_UNTRACKED 1
#endif
#ifndef KA_W_UNTRACKED_EDGE
#define KA_W_UNTRACKED_EDGE 0
#endif
#if KA_W_NOINLINE
#define KA_W_CALL __attribute__((noinline))
#else
#define KA_W_CALL __forceinline__
#endif
// The HELPER alone as a real function (KA_WH_NOINLINE, default 1): its call sits where waves 4..7 have nothing else live (they
// return to the level's barrier right behind it).
#ifndef KA_WH_NOINLINE
#define KA_WH_NOINLINE 1
#endif
#if KA_WH_NOINLINE
#define KA_WH_CALL __attribute__((noinline))
#else
#define KA_WH_CALL __forceinline__
#endif
typedef __attribute__((address_space(3))) char ka_lchar;
typedef __attribute__((address_space(3))) int ka_lint;

struct KaWStripArgs {
        const float* p1;                 // row profile
        const int2* ent;                 // consistency bonus entries (NB > 0)
        int* watchdog;
        long long* pslot;                // KA_PROF builds
        float m1;                        // TaskShared::p1_mult
        int Lb, prio;
        int starta, enda, startb, endb, dir, k;
        unsigned wlds_u;                 // LDS offset of the strip wave's region
        unsigned in_ring_u, in_word_u;   // the row above: ring base and the word that counts its columns
        int in_bias;
        unsigned ctl_u;                  // control words of helper mode
        int w;                           // strip wave index in the workgroup
};

__device__ __forceinline__ int ka_u(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned ka_u(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }

template <int NRES, int NB, int Q = 2>
__device__ KA_W_CALL void ka_wstrip(const KaWStripArgs a)
{
        // (everything in `a` is wave-uniform, but arrives in vector registers: back to scalars)
        const int lane = threadIdx.x & 63;
        const int starta = ka_u(a.starta), enda = ka_u(a.enda), startb = ka_u(a.startb), endb = ka_u(a.endb), dir = ka_u(a.dir), k = ka_u(a.k);
        const unsigned in_ring_u = ka_u(a.in_ring_u), in_word_u = ka_u(a.in_word_u), ctl_u = ka_u(a.ctl_u);
        const int in_bias = ka_u(a.in_bias), w = ka_u(a.w);
        const float* const p1 = ka_uniform_ptr(a.p1);
        long long* const pslot = ka_uniform_ptr(a.pslot);
        // the strip wave goes first on its SIMD: its helper (same SIMD, priority 0) takes the issue slots it leaves
        // (KA_HW_PRIO in the environment, experiments: 0 .. 3, default 3)
        {
                const int prio = ka_u(a.prio);
                if (prio == 3) __builtin_amdgcn_s_setprio(3); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 1) __builtin_amdgcn_s_setprio(1);
        }
        const int ncols = endb - startb;
        const int mid = ((enda - starta) / 2) + starta;
        const int r0 = (dir == KA_FWD) ? starta : mid;
        const int r1 = (dir == KA_FWD) ? mid : enda;
        const int nrows = r1 - r0;                                    // > 0 (the caller keeps empty passes on ka_strip)
        const int Lb = ka_u(a.Lb);
        const bool near_t = (dir == KA_FWD) ? (startb == 0) : (endb == Lb);
        const bool far_t = (dir == KA_FWD) ? (endb == Lb) : (startb == 0);
        int* const wdu = ka_uniform_ptr(a.watchdog);
        const float m1 = ka_uniform_f(a.m1);

        constexpr int SROWS = 64 * Q;
        const int u0 = k * SROWS;
        const int nr = min(SROWS, nrows - u0);
        const int nl = (Q == 2) ? ((nr + 1) >> 1) : nr;
        const int lastl = nl - 1;
        const bool last_is_b = (Q == 2) && (nr & 1) == 0;
        const bool actB = (Q == 2) && 2 * lane + 1 < nr;
        const int uA = u0 + min(Q * lane, nr - 1);
        const int uB = u0 + min(Q * lane + 1, nr - 1);              // (Q = 1: row B does not exist; its operand loads are dead code)
        const int iA = (dir == KA_FWD) ? (r0 + uA) : (r1 - 1 - uA);
        const int iB = (dir == KA_FWD) ? (r0 + uB) : (r1 - 1 - uB);
        const int recA = iA + 1, recB = iB + 1;
        const int prevA = (dir == KA_FWD) ? recA - 1 : recA + 1;
        const int prevB = (dir == KA_FWD) ? recB - 1 : recB + 1;

        const unsigned wlds_u = ka_u(a.wlds_u);
        const unsigned out_u = wlds_u + KA_HO_RING;
        const unsigned tpub_u = ctl_u + 4 * KA_W_TPUB(w), go_u = ctl_u + 4 * KA_W_GO(w);
        // The lane that owns the strip's last row writes it: exec narrowed to that lane around the ds_write, with SCALAR instructions
        // (1 << lastl built inside the asm from a 32-bit scalar: a 64-bit mask handed in as an "s" operand arrived in a VGPR pair once
        // the kernel ran short of SGPRs).  NOT v_cmpx: a vector compare writing exec right in front of the ds_write inside one asm
        // block -- where the compiler's hazard recognizer does not look -- left the write with the OLD exec every now and then: all 64
        // lanes stored, lane 63 won, and a PARTIAL strip (owner below lane 63) handed on garbage (caught by the schedule stress
        // test as a top-level row hash that differed in one run of three).

        // ---- stationary row operand (as in ka_strip) ----
        float oA, eA, tA, oB, eB, tB, orpA, orpB;
        float2v p1v[Q == 2 ? NRES : 1];                               // Q = 2: the counts of residue c in rows (A, B)
        constexpr int NPAIR = NRES / 2;
        float2v p1p[Q == 1 ? (NPAIR > 0 ? NPAIR : 1) : 1];            // Q = 1: the counts of residues (2i, 2i+1) in row A
        float p1last = 0.0f;                                          // Q = 1, odd alphabets: the count of residue NRES-1
        {
                const float* pA = p1 + ((long long)recA << 6);
                const float* pB = p1 + ((long long)recB << 6);
                oA = pA[55] * m1; eA = pA[56] * m1; tA = pA[57] * m1;
                oB = pB[55] * m1; eB = pB[56] * m1; tB = pB[57] * m1;
                orpA = p1[((long long)prevA << 6) + 55] * m1;
                orpB = p1[((long long)prevB << 6) + 55] * m1;
                constexpr int NV = (NRES + 3) / 4;
                if constexpr (Q == 2) {
                        float4v va[NV], vb[NV];
#pragma unroll
                        for (int i = 0; i < NV; ++i) { va[i] = ((const float4v*)pA)[i]; vb[i] = ((const float4v*)pB)[i]; }
#pragma unroll
                        for (int c = 0; c < NRES; ++c) {
                                p1v[c].x = va[c >> 2][c & 3];
                                p1v[c].y = actB ? vb[c >> 2][c & 3] : 0.0f;
                        }
                } else {
                        float4v va[NV];
#pragma unroll
                        for (int i = 0; i < NV; ++i) va[i] = ((const float4v*)pA)[i];
#pragma unroll
                        for (int i = 0; i < NPAIR; ++i) { p1p[i].x = va[(2 * i) >> 2][(2 * i) & 3]; p1p[i].y = va[(2 * i + 1) >> 2][(2 * i + 1) & 3]; }
                        if (NRES & 1) p1last = va[(NRES - 1) >> 2][(NRES - 1) & 3];
                }
        }
        KaBonus<NB> bonA, bonB;
        if (NB) { bonA.load(a.ent, iA, dir); if (Q == 2) bonB.load(a.ent, iB, dir); }

        float cAa = -KA_F, cAga = -KA_F, cAgb = -KA_F;
        float cBa = -KA_F, cBga = -KA_F, cBgb = -KA_F;
        float dga = -KA_F, dgga = -KA_F, dggb = -KA_F;
        float copen_prev = 0.0f;
        float4v q[2][KA_REC_CHUNKS];                                  // column record: this step / next step
        float4v bq[2];                                                // boundary state of column t: this step / next step
        int gv = 0, iv = 0;                                           // the two control words, read one step before they are tested
        unsigned in_oct = 0, out_oct = 0;                             // bases of an octet's immediate-offset LDS accesses

        auto ring_read = [&](float4v* dstq, int vcol, auto& dep) {
                const unsigned a = wlds_u | (((unsigned)vcol & 127u) << 4);
                if (NRES <= 8) {
                        asm volatile("ds_read_b128 %0, %5\n\t"
                                     "ds_read_b128 %1, %5 offset:2048\n\t"
                                     "ds_read_b128 %2, %5 offset:10240\n\t"
                                     "ds_read_b128 %3, %5 offset:12288"
                                     : "=&v"(dstq[0]), "=&v"(dstq[1]), "=&v"(dstq[5]), "=&v"(dstq[6]), "+v"(dep)
                                     : "v"(a)
                                     : "memory");
                        return;
                }
                asm volatile("ds_read_b128 %0, %8\n\t"
                             "ds_read_b128 %1, %8 offset:2048\n\t"
                             "ds_read_b128 %2, %8 offset:4096\n\t"
                             "ds_read_b128 %3, %8 offset:6144\n\t"
                             "ds_read_b128 %4, %8 offset:8192\n\t"
                             "ds_read_b128 %5, %8 offset:10240\n\t"
                             "ds_read_b128 %6, %8 offset:12288"
                             : "=&v"(dstq[0]), "=&v"(dstq[1]), "=&v"(dstq[2]), "=&v"(dstq[3]), "=&v"(dstq[4]), "=&v"(dstq[5]), "=&v"(dstq[6]),
                               "+v"(dep)
                             : "v"(a)
                             : "memory");
        };
        // everything read from LDS one step ago has landed (the control words ride along: no use can move above the wait)
        auto ring_wait = [&](float4v* qq, float4v& bb) {
                if (NRES <= 8) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[5]), "+v"(qq[6]), "+v"(bb), "+v"(gv), "+v"(iv) : : "memory");
                        return;
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]), "+v"(qq[4]), "+v"(qq[5]), "+v"(qq[6]), "+v"(bb), "+v"(gv), "+v"(iv)
                             :
                             : "memory");
        };
        auto lds_word = [&](const unsigned addr) -> int {
                int x;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(addr) : "memory");
                return __builtin_amdgcn_readfirstlane(x);
        };
        // Before steps t0 .. t0+CH-1: they read column records and boundary states of columns <= t0 + CH and write out slots
        // of steps <= t0 + CH - 1.  Columns clamp at ncols, steps end at nsteps.
        const int nsteps = ncols + nl;
        auto wait_for = [&](const int t0) {
                const int needg = min(t0 + KA_W_CH + 1, nsteps + 1);
                const int needi = min(t0 + KA_W_CH + 1, ncols + 1);
                int g = __builtin_amdgcn_readfirstlane(gv), i = __builtin_amdgcn_readfirstlane(iv) - in_bias;
                if (g >= needg && i >= needi) return;
#ifdef KA_PROF
                const long long tw0 = __builtin_amdgcn_s_memtime();
                const bool for_in = (g >= needg);
#endif
                int spins = 0;
                while (g < needg || i < needi) {
                        __builtin_amdgcn_s_sleep(1);
                        if (ka_spin_expired(wdu, ++spins, 1 << 22, 5)) break;
                        g = lds_word(go_u); i = lds_word(in_word_u) - in_bias;
                }
#ifdef KA_PROF
                // KA_PROF builds (tools/strip_phases.py): [256+0] cycles waited, [256+3] waits, [256+1] of them for the boundary row only
                if (pslot && lane == 0) { pslot[256 + 0] += __builtin_amdgcn_s_memtime() - tw0; pslot[256 + 3] += 1; pslot[256 + 1] += for_in ? 1 : 0; }
#endif
        };
        auto prefetch_words = [&]() {
                // (plain loads: they are in flight across the octet loop's back edge and its exit, where the compiler may move registers)
                gv = *(const volatile ka_lint*)(unsigned long)go_u;
                iv = *(const volatile ka_lint*)(unsigned long)in_word_u;
        };
        auto publish = [&](const int tdone) {
                asm volatile("ds_write_b32 %0, %1" : : "v"(tpub_u), "v"(tdone) : "memory");
        };

        // One wavefront step.  ST: steady state (no edge cases); P: the half of q / bq that holds this step's operands;
        // I: position in an octet (0 .. CH-1; immediate LDS offsets, the octet's checks are compile-time) or -1 (single step)
        // FORM: 0 steady state (every active lane strictly inside the column range); 1 head (lanes still entering: the lane at
        // column 0 is the only special one -- ncols >= nl, so nobody is at the last column yet); 2 tail (lanes leaving: the lane at
        // the last column is the special one); 3 both at once (passes with fewer columns than the strip has lanes).  Lanes outside
        // the column range compute on whatever their ring slot holds: state only flows DOWN the lanes, so nothing of it reaches a
        // lane inside the range (see ka_strip), and forms 1 and 2 do without the clamped column index of form 3.
        auto step = [&](const int t, auto st_tag, auto par_tag, auto i_tag, auto lb_tag) {
                constexpr bool LASTB = decltype(lb_tag)::value;
                constexpr int FORM = decltype(st_tag)::value;
                constexpr bool ST = FORM == 0;
                constexpr int P = decltype(par_tag)::value;
                constexpr int I = decltype(i_tag)::value;
                const int v = t - lane;

                ring_wait(q[P], bq[P]);
#if KA_W_TAG
                if (t <= ncols && __builtin_amdgcn_readfirstlane(__float_as_int(bq[P].w)) != t) {
                        if (lane == 0) { atomicCAS(wdu, 0, 9); }
                }
#endif
                // The strip's last row as the PREVIOUS step left it: its owner writes column t - 1 - lastl into out slot (t - 1) & 255.
                // (Here and not at the end of the step that computed it: a ds_write right in front of the next step's
                // s_waitcnt lgkmcnt(0) would be waited for; here the whole dot product lies between it and the next wait.)
                {
                        const int vLp = t - 1 - lastl;
                        // (head: the last row's owner has not reached column 0 yet; steady state and tail: always inside the range)
                        if (FORM != 1 && (FORM != 3 || (vLp >= 0 && vLp <= ncols))) {
#if KA_W_TAG
                                float4v o;
                                o.w = __int_as_float(vLp);
#else
                                float3v o;
#endif
                                o.x = (Q == 2 && LASTB) ? cBa : cAa; o.y = (Q == 2 && LASTB) ? cBga : cAga; o.z = (Q == 2 && LASTB) ? cBgb : cAgb;
                                unsigned long long sv, sm;
                                if constexpr (I >= 1) {
                                        asm volatile("s_lshl_b64 %1, 1, %4\n\ts_and_saveexec_b64 %0, %1\n\t" KA_W_DSW " %2, %3 offset:%5\n\ts_mov_b64 exec, %0"
                                                     : "=&s"(sv), "=&s"(sm) : "v"(out_oct), "v"(o), "s"(__builtin_amdgcn_readfirstlane(lastl)), "n"((I - 1) * 16) : "memory", "scc");
                                } else {
                                        const unsigned oa = out_u + ((((unsigned)t - 1u) & 255u) << 4);
                                        asm volatile("s_lshl_b64 %1, 1, %4\n\ts_and_saveexec_b64 %0, %1\n\t" KA_W_DSW " %2, %3\n\ts_mov_b64 exec, %0"
                                                     : "=&s"(sv), "=&s"(sm) : "v"(oa), "v"(o), "s"(__builtin_amdgcn_readfirstlane(lastl)) : "memory", "scc");
                                }
                        }
                }
                // steps < t are computed and written: say so, then make sure the next CH steps have their operands
                if (I == 0 || (I < 0 && (t & (KA_W_CH - 1)) == 0)) { publish(t); wait_for(t); }
                const float copen = q[P][5].w, cext = q[P][6].x;                    // (the helper has applied the multiplier; ctext only feeds row -1)

                // the row above A: lane l-1's row B, lane 0 takes the boundary state of column t
                // (the lane's last row: B, or A when a lane owns one row)
                const float lra = (Q == 2) ? cBa : cAa, lrga = (Q == 2) ? cBga : cAga, lrgb = (Q == 2) ? cBgb : cAgb;
                const float upa = wave_shr1_old(bq[P].x, lra), upga = wave_shr1_old(bq[P].y, lrga), upgb = wave_shr1_old(bq[P].z, lrgb);

                // The gap states that do not wait for the dot products come FIRST in the source: the scheduler then has them to put
                // into the wait states at the end of the dependent v_pk_add chain (six s_nop per step before; round 4).
                const bool at0 = (FORM == 1 || FORM == 3) && (v == 0), atN = (FORM == 2 || FORM == 3) && (v == ncols);
                const bool edge = at0 | atN;
                float nAga, nAgb, nBga = -KA_F;
                // (edge forms: selects on the OPERANDS, no branches: max(x, y) + c == max(x + c, y + c) bit for bit -- rounding is
                // monotonic -- so the terminal case `max(gb, a) + t` is the inner case with both penalties replaced by t; written as
                // `term ? .. : ..` over the two results the compiler made four exec-masked regions per step of it.  The penalties of
                // a lane at the first / last column are t where that column is a terminal one, else the inner ones.)
                // (near_t / far_t are wave-uniform: `uniform ? t : e` is one select on a scalar condition, made here in the edge steps
                // rather than kept in eight more registers across the steady loop -- the kernel has none to spare)
                float xeA = eA, xoA = oA, xeB = eB, xoB = oB;
                if (FORM == 1 || FORM == 3) {
                        xeA = at0 ? (near_t ? tA : eA) : xeA; xoA = at0 ? (near_t ? tA : oA) : xoA;
                        if (Q == 2) { xeB = at0 ? (near_t ? tB : eB) : xeB; xoB = at0 ? (near_t ? tB : oB) : xoB; }
                }
                if (FORM == 2 || FORM == 3) {
                        xeA = atN ? (far_t ? tA : eA) : xeA; xoA = atN ? (far_t ? tA : oA) : xoA;
                        if (Q == 2) { xeB = atN ? (far_t ? tB : eB) : xeB; xoB = atN ? (far_t ? tB : oB) : xoB; }
                }
                if (ST) {
                        nAga = kmax(cAga + cext, cAa + copen);
                        nAgb = kmax(upgb + eA, upa + oA);
                        if (Q == 2) nBga = kmax(cBga + cext, cBa + copen);
                } else {
                        nAga = edge ? -KA_F : kmax(cAga + cext, cAa + copen);
                        nAgb = kmax(upgb + xeA, upa + xoA);
                        if (Q == 2) nBga = edge ? -KA_F : kmax(cBga + cext, cBa + copen);
                }
                // Next step's operands into the other half of q / bq.  ka_strip reads them AFTER the dot products (its waits were
                // once the compiler's, and those waited for fresh loads too); here the step's one wait is the manual lgkmcnt(0) at
                // its top, so the reads can go out right behind it and have the whole dot product to land: with four strips
                // reading 8 KB per step each the LDS pipe is half busy and a read issued late was still in flight at the next wait.
                // UNTRACKED reads (inline asm: the compiler believes the registers are written when the asm ends, the data lands later,
                // ring_wait at the top of the next step is the matching wait) only where the registers cannot be touched in between:
                // the steady-state octets, which tools/check_hot_loops.py shows free of scratch traffic.  Everywhere else -- edge
                // forms, single steps -- plain loads the compiler tracks: a version that read untracked in the head and tail octets
                // too, where the allocator spills, had its in-flight registers SPILLED AND RELOADED now and then: stale column
                // records in the first steps of a strip, a wrong prefix of its last row (schedule stress test, 1 run in 8).
                auto next_reads = [&](auto& dep) {
                        const int vcol = FORM != 3 ? (v + 1) : min(max(v + 1, 0), ncols);
                        if constexpr (KA_W_UNTRACKED && (FORM == 0 || KA_W_UNTRACKED_EDGE) && I >= 0) {
                                __builtin_amdgcn_sched_barrier(0);
                                ring_read(q[1 - P], vcol, dep);
                                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(bq[1 - P]) : "v"(in_oct), "n"(I * 16) : "memory");
                                __builtin_amdgcn_sched_barrier(0);
                        } else {
                                typedef const __attribute__((address_space(3))) float4v ka_l4;
                                ka_l4* const cr = (ka_l4*)(unsigned long)(wlds_u | (((unsigned)vcol & 127u) << 4));
#pragma unroll
                                for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) if (ka_chunk_used<NRES>(ch)) q[1 - P][ch] = cr[ch * 128];
                                bq[1 - P] = *(ka_l4*)(unsigned long)(in_ring_u + ((((unsigned)t + 64u) & 255u) << 4));
                        }
                };
                if constexpr (Q == 2) {
                        float2v acc;
                        acc.x = kmax3(dga, dgga + copen_prev, dggb + orpA);
                        acc.y = kmax3(cAa, cAga + copen_prev, cAgb + orpB);
                        if (KA_W_EARLY) next_reads(acc);
                        {
                                float2v prod;
                                prod = ka_mul_bcast<(NRES - 1) & 3>(p1v[NRES - 1], q[P][(NRES - 1) >> 2]);
#pragma unroll
                                for (int c = NRES - 1; c >= 1; --c) {
                                        float2v nprod;
                                        switch ((c - 1) & 3) {
                                        case 0: nprod = ka_mul_bcast<0>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        case 1: nprod = ka_mul_bcast<1>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        case 2: nprod = ka_mul_bcast<2>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        default: nprod = ka_mul_bcast<3>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        }
                                        acc = acc + prod;
                                        prod = nprod;
                                }
                                acc = acc + prod;
                                if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); acc.x += bonA.template at<(FORM >= 2)>(jb); acc.y += bonB.template at<(FORM >= 2)>(jb); }
                                if (!KA_W_EARLY) next_reads(acc);
                        }
                        const float nAa = at0 ? -KA_F : acc.x;
                        const float nBa = at0 ? -KA_F : acc.y;
                        const float nBgb = kmax(nAgb + xeB, nAa + xoB);           // B: the row above is A's fresh state
                        cAa = nAa; cAga = nAga; cAgb = nAgb;
                        cBa = nBa; cBga = nBga; cBgb = nBgb;
                } else {
                        // ---- the one cell of this lane: residue NRES-1 first (aln_profileprofile.c:99-107 walks the non-zero counts
                        // downwards); a pair of residues per v_pk_mul_f32, products one pair ahead of the (dependent) sums ----
                        float a1 = kmax3(dga, dgga + copen_prev, dggb + orpA);
                        if (KA_W_EARLY) next_reads(a1);
                        if (NRES & 1) a1 += p1last * q[P][(NRES - 1) >> 2][(NRES - 1) & 3];
                        if (NPAIR > 0) {
                                auto qpair = [&](int i) -> float2v {
                                        const float4v& w4 = q[P][(2 * i) >> 2];
                                        return ((2 * i) & 3) ? __builtin_shufflevector(w4, w4, 2, 3) : __builtin_shufflevector(w4, w4, 0, 1);
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
                        if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); a1 += bonA.template at<(FORM >= 2)>(jb); }
                        if (!KA_W_EARLY) next_reads(a1);
                        const float nAa = at0 ? -KA_F : a1;
                        cAa = nAa; cAga = nAga; cAgb = nAgb;
                        (void)xeB; (void)xoB; (void)nBga;
                }
                dga = upa; dgga = upga; dggb = upgb;
                copen_prev = copen;

                if (I == KA_W_CH - 1 || (I < 0 && ((t + 1) & (KA_W_CH - 1)) == 0)) prefetch_words();
        };

        auto single = [&](const int t, auto st_tag, auto lb_tag) {
                if (t & 1) step(t, st_tag, std::integral_constant<int, 1>(), std::integral_constant<int, -1>(), lb_tag);
                else step(t, st_tag, std::integral_constant<int, 0>(), std::integral_constant<int, -1>(), lb_tag);
        };
        // eight steps with compile-time parities and immediate LDS offsets; t0 is a multiple of 8
        auto octet = [&](const int t0, auto st_tag, auto lb_tag) {
                in_oct = in_ring_u + ((((unsigned)t0 + 64u) & 255u) << 4);
                out_oct = out_u + (((unsigned)t0 & 255u) << 4);
                step(t0 + 0, st_tag, std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), lb_tag);
                step(t0 + 1, st_tag, std::integral_constant<int, 1>(), std::integral_constant<int, 1>(), lb_tag);
                step(t0 + 2, st_tag, std::integral_constant<int, 0>(), std::integral_constant<int, 2>(), lb_tag);
                step(t0 + 3, st_tag, std::integral_constant<int, 1>(), std::integral_constant<int, 3>(), lb_tag);
                step(t0 + 4, st_tag, std::integral_constant<int, 0>(), std::integral_constant<int, 4>(), lb_tag);
                step(t0 + 5, st_tag, std::integral_constant<int, 1>(), std::integral_constant<int, 5>(), lb_tag);
                step(t0 + 6, st_tag, std::integral_constant<int, 0>(), std::integral_constant<int, 6>(), lb_tag);
                step(t0 + 7, st_tag, std::integral_constant<int, 1>(), std::integral_constant<int, 7>(), lb_tag);
        };
        // a phase: single steps up to the next multiple of 8, octets, single steps for the rest.  The head of a strip (its lanes
        // entering at column 0: what the strip below waits for before it can start) and its tail run the edge form of the step
        // in octets too -- as single steps with run-time parity they cost 2.3 steady steps each (profiles/r04_strip_phases_v1.log).
        auto run = [&](int& t, const int tend, auto st_tag, auto lb_tag) {
                for (; t < tend && (t & (KA_W_CH - 1)); ++t) single(t, st_tag, lb_tag);
                for (; t + KA_W_CH <= tend; t += KA_W_CH) octet(t, st_tag, lb_tag);
                // (the last octet's untracked reads have landed before anything but an octet step may touch their registers)
                if (KA_W_UNTRACKED && decltype(st_tag)::value == 0) { ring_wait(q[0], bq[0]); ring_wait(q[1], bq[1]); }
                for (; t < tend; ++t) single(t, st_tag, lb_tag);
        };
        auto singles = [&](int& t, const int tend, auto st_tag, auto lb_tag) {
                for (; t < tend; ++t) single(t, st_tag, lb_tag);
        };
        auto phases = [&](auto lb_tag) {
                const int t_steady0 = min(nl, nsteps);
                const int t_steady1 = ncols;
                int t = 0;
                if (KA_W_FORMS && ncols >= nl) {
                        run(t, t_steady0, std::integral_constant<int, 1>(), lb_tag);
#ifdef KA_PROF
                        if (pslot && lane == 0) pslot[256 + 2] += __builtin_amdgcn_s_memtime() - pslot[2];      // the head
                        const long long toct0 = __builtin_amdgcn_s_memtime();
                        const int t_in = t;
#endif
                        run(t, t_steady1, std::integral_constant<int, 0>(), lb_tag);
#ifdef KA_PROF
                        // cycles in the steady phase (its waits included) / its steps
                        if (pslot && lane == 0) { pslot[6] += __builtin_amdgcn_s_memtime() - toct0; pslot[7] += t - t_in; }
                        const long long ttail0 = __builtin_amdgcn_s_memtime();
#endif
                        run(t, nsteps, std::integral_constant<int, 2>(), lb_tag);
#ifdef KA_PROF
                        if (pslot && lane == 0) pslot[256 + 4] += __builtin_amdgcn_s_memtime() - ttail0;             // the tail
#endif
                } else if (!KA_W_FORMS && ncols >= nl) {
                        // (experiment: the general edge form for head and tail)
                        singles(t, t_steady0, std::integral_constant<int, 3>(), lb_tag);
                        run(t, t_steady1, std::integral_constant<int, 0>(), lb_tag);
                        singles(t, nsteps, std::integral_constant<int, 3>(), lb_tag);
                } else {
                        // fewer columns than lanes: some lane is at column 0 while another is at the last one -- the general edge form,
                        // step by step (short passes: deep recursion levels)
                        singles(t, nsteps, std::integral_constant<int, 3>(), lb_tag);
                }
        };
        static_assert(KA_W_CH == 8, "the octet is written out for eight steps");

        // the operands of step 0: column 0 (every lane: v <= 0 clamps to it) and the boundary state of column 0
        {
                int spins = 0;
                while (true) {
                        gv = lds_word(go_u); iv = lds_word(in_word_u);
                        if (gv >= min(KA_W_CH + 1, nsteps + 1) && iv - in_bias >= min(KA_W_CH + 1, ncols + 1)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (ka_spin_expired(wdu, ++spins, 1 << 22, 5)) break;
                }
#ifdef KA_PROF
                if (pslot && lane == 0 && pslot[2] == 0) pslot[2] = __builtin_amdgcn_s_memtime();
#endif
                // (plain loads: see next_reads)
                typedef const __attribute__((address_space(3))) float4v ka_l4;
                ka_l4* const cr = (ka_l4*)(unsigned long)wlds_u;
#pragma unroll
                for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) if (ka_chunk_used<NRES>(ch)) q[0][ch] = cr[ch * 128];
                bq[0] = *(ka_l4*)(unsigned long)(in_ring_u + ((63u & 255u) << 4));
        }
        if constexpr (Q == 2) { if (last_is_b) phases(std::true_type()); else phases(std::false_type()); }
        else phases(std::false_type());
        // the last step's column (vL = ncols), then done: everything the strip wrote to its out ring is in LDS before the count
        // says so (LDS runs a wave's instructions in order)
        {
#if KA_W_TAG
                float4v o;
                o.w = __int_as_float(ncols);
#else
                float3v o;
#endif
                o.x = (Q == 2 && last_is_b) ? cBa : cAa; o.y = (Q == 2 && last_is_b) ? cBga : cAga; o.z = (Q == 2 && last_is_b) ? cBgb : cAgb;
                unsigned long long sv, sm;
                const unsigned oa = out_u + ((((unsigned)nsteps - 1u) & 255u) << 4);
                asm volatile("s_lshl_b64 %1, 1, %4\n\ts_and_saveexec_b64 %0, %1\n\t" KA_W_DSW " %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(sv), "=&s"(sm) : "v"(oa), "v"(o), "s"(__builtin_amdgcn_readfirstlane(lastl)) : "memory", "scc");
        }
        publish(nsteps);
        __builtin_amdgcn_s_setprio(0);
}

// ------------------------------------------------------------------------------------------
// The helper of strip wave w (wave w + 4 of the workgroup): feeds the strip's rings and drains its out ring.
//   in_mode   0: the strip is the first of its pass -- generate row -1 into the in ring, column batch by column batch
//             1: the strip above runs on wave w - 1: nothing to do (the strip reads that wave's out ring and step count)
//             2: the strip above runs in another workgroup: copy its last row from the HBM row buffer behind its flag
//   out_local the strip below runs on wave w + 1 (reads my strip's out ring in place); else copy the out ring to the HBM
//             row buffer -- and, when a strip below exists (in another workgroup), publish it: release fence + flag
// GO word: min(columns loaded (BIG once all are), out slots free expressed as a step bound + 1), see ka_wstrip::wait_for.
// ------------------------------------------------------------------------------------------
struct KaWHelperArgs {
        const float* p2;                 // column profile
        KaState* rows;                   // the sub-problem's row buffer (HBM): the pass's LAST row, read by the meetups
        KaState* xrows;                  // ... and its twin for rows handed from workgroup to workgroup (same offsets; see TaskShared::xfbuf)
        int* prog;                       // progress flags of the pass's strips (HBM)
        int* watchdog;
        float m2;                        // TaskShared::p2_mult
        float inj_a, inj_ga, inj_gb;     // the pass's injected boundary state
        int Lb;
        int starta, enda, startb, endb, dir, k, ns;
        unsigned slds_u, hlds_u, ctl_u;  // LDS offsets: the strip wave's region, this (helper) wave's region, the control words
        int w, in_mode, out_local;
};

template <int NRES, int Q = 2>
__device__ KA_WH_CALL void ka_whelper(const KaWHelperArgs a)
{
        const int lane = threadIdx.x & 63;
        const int starta = ka_u(a.starta), enda = ka_u(a.enda), startb = ka_u(a.startb), endb = ka_u(a.endb), dir = ka_u(a.dir), k = ka_u(a.k), ns = ka_u(a.ns);
        const int w = ka_u(a.w), in_mode = ka_u(a.in_mode);
        const bool out_local = ka_u(a.out_local) != 0;
        const float inj_a = ka_uniform_f(a.inj_a), inj_ga = ka_uniform_f(a.inj_ga), inj_gb = ka_uniform_f(a.inj_gb);
        KaState* const rows = ka_uniform_ptr(a.rows);
        KaState* const xrows = ka_uniform_ptr(a.xrows);
        int* const prog = ka_uniform_ptr(a.prog);
        ka_lchar* const slds = (ka_lchar*)(unsigned long)ka_u(a.slds_u);
        ka_lchar* const hlds = (ka_lchar*)(unsigned long)ka_u(a.hlds_u);
        ka_lint* const ctl = (ka_lint*)(unsigned long)ka_u(a.ctl_u);
        const int ncols = endb - startb;
        const int mid = ((enda - starta) / 2) + starta;
        const int r0 = (dir == KA_FWD) ? starta : mid;
        const int r1 = (dir == KA_FWD) ? mid : enda;
        const int nrows = r1 - r0;
        const int Lb = ka_u(a.Lb);
        const bool near_t = (dir == KA_FWD) ? (startb == 0) : (endb == Lb);
        const int u0 = k * (64 * Q);
        const int nr = min(64 * Q, nrows - u0);
        const int nl = (Q == 2) ? ((nr + 1) >> 1) : nr;
        const int lastl = nl - 1;
        const int nsteps = ncols + nl;
        const bool last_strip = (k + 1 == ns);
        const float m2 = ka_uniform_f(a.m2);
        const float* const p2 = ka_uniform_ptr(a.p2);
        int* const wdu = ka_uniform_ptr(a.watchdog);
        ka_gfloat* const grows = (ka_gfloat*)rows;

#define REC(v_) ((dir == KA_FWD) ? (startb + (v_)) : (endb + 1 - (v_)))
#define IDX(v_) ((dir == KA_FWD) ? (v_) : (ncols - (v_)))
        const int j = lane & 15, part = lane >> 4;                    // column within a 16-column batch / which chunks of it: part p loads chunks p and p + 4
        int Lc = 0;                                                   // columns loaded into the strip's ring
        int Li = 0;                                                   // boundary columns in the in ring (in_mode 0, 2)
        int To = 0;                                                   // out columns copied to HBM (!out_local)
        float cg = -KA_F;                                             // in_mode 0: ga of the last generated column
        int lastG = -1, lastI = -1;
        int idle = 0;

        while (true) {
                bool progress = false;
                // (the control words as scalars: the helper's bookkeeping runs on the scalar unit, next to the strip's vector work)
                const int tp = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[KA_W_TPUB(w)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const bool done = tp >= nsteps;

                // ---- the column ring: 16 columns per batch, up to two batches per round.  Column c overwrites column c - 128, last
                // read in step c - 66: batch [Lc, Lc + 15] may go out once the strip has published step Lc - 50, and is needed by its
                // check at step Lc - 8 ----
                for (int rep = 0; rep < 2 && Lc <= ncols && Lc + 15 <= tp + 65; ++rep) {
                        const int c = Lc + j;
                        float4v r0 = {0.0f, 0.0f, 0.0f, 0.0f}, r1 = {0.0f, 0.0f, 0.0f, 0.0f};
                        const bool use0 = (part == 0 && ka_chunk_used<NRES>(0)) || (part == 1 && ka_chunk_used<NRES>(1)) ||
                                          (part == 2 && ka_chunk_used<NRES>(2)) || (part == 3 && ka_chunk_used<NRES>(3));
                        const bool use1 = (part == 0 && ka_chunk_used<NRES>(4)) || (part == 1) || (part == 2);    // chunks 5 and 6: every alphabet
                        if (c <= ncols) {
                                ka_gfloat4c* g = (ka_gfloat4c*)(p2 + ((long long)REC(c) << 6) + 32);
                                if (use0) r0 = g[part];
                                if (use1) r1 = g[part + 4];
                        }
                        // set_gap_penalties_n (aln_setup.c:101-119): fields 55..57 times the other side's nsip -- chunk 5 .w (part 1), chunk 6 .x .y (part 2)
                        if (part == 1) r1.w = r1.w * m2;
                        if (part == 2) { r1.x = r1.x * m2; r1.y = r1.y * m2; }
                        if (c <= ncols) {
                                ka_lchar* dst = slds + ((c & 127) << 4);
                                if (use0) *(__attribute__((address_space(3))) float4v*)(dst + part * 2048) = r0;
                                if (use1) *(__attribute__((address_space(3))) float4v*)(dst + (part + 4) * 2048) = r1;
                        }
                        if (in_mode == 0) {
                                // Row -1 of the pass (aln_seqseq.c:40-58; ka_strip's FIRST steps): column 0 is the injected state; columns
                                // 1 .. ncols-1 carry (-FLT_MAX, g, -FLT_MAX) with the serial chain g = max(g' + gx, a' + gy) over the column
                                // before (a' = the injected a for column 1, -FLT_MAX after it: a' + gy does not depend on the chain); column
                                // ncols is all -FLT_MAX.  The chain as 16 rounds of "every lane recomputes from its left neighbour": lane l is
                                // final after round l and recomputing a final lane from a final neighbour changes nothing -- two instructions
                                // per column and no selects (a column that is NOT on the chain -- 0, ncols -- gets gx = -FLT_MAX and its fixed
                                // value as the other operand of the max).  The gap terms of column Lc + l: copen in lane 16 + l, cext / ctext in
                                // lane 32 + l.
                                const float copen_c = __shfl(r1.w, (lane & 15) + 16, 64), cext_c = __shfl(r1.x, (lane & 15) + 32, 64), ctext_c = __shfl(r1.y, (lane & 15) + 32, 64);
                                float gx = near_t ? ctext_c : cext_c;
                                const float gy = near_t ? ctext_c : copen_c;
                                float fix = ((c == 1) ? inj_a : -KA_F) + gy;
                                if (c >= ncols) { gx = -KA_F; fix = -KA_F; }
                                if (c == 0) { gx = -KA_F; fix = inj_ga; }
                                float g = -KA_F;
#pragma unroll
                                for (int round = 0; round < 16; ++round) g = kmax(wave_shr1_old(cg, g) + gx, fix);
                                cg = lane_bcast(g, 15);
                                if (lane < 16 && c <= ncols)
                                        *(__attribute__((address_space(3))) float4v*)(hlds + KA_W_INRING + (((c + 63) & 255) << 4)) = (float4v){c == 0 ? inj_a : -KA_F, g, c == 0 ? inj_gb : -KA_F, __int_as_float(c)};
                                Li = min(Lc + 16, ncols + 1);
                        }
                        Lc += 16;
                        progress = true;
                }

                // ---- boundary row from another workgroup: in column c overwrites column c - 256, read in step c - 257 ----
                if (in_mode == 2 && Li <= ncols) {
                        const int avail = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog + (k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        const int n = min(min(avail - Li, 64), tp + 256 - Li);
                        if (n > 0) {
                                // (agent-scope loads that go past L1 and L2, matching the producer's write-through stores: no
                                // cache invalidate -- an acquire fence at agent scope drops the whole L1 and the L2's clean lines)
                                if (lane < n) {
                                        const int c = Li + lane;
                                        float* r = (float*)xrows + 3 * IDX(c);
                                        const float x0 = __hip_atomic_load(r + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        const float x1 = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        const float x2 = __hip_atomic_load(r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        *(__attribute__((address_space(3))) float4v*)(hlds + KA_W_INRING + (((c + 63) & 255) << 4)) = (float4v){x0, x1, x2, __int_as_float(c)};
                                }
                                Li += n;
                                progress = true;
                        }
                }

                // ---- the out ring to the HBM row buffer (16 columns or more at a time; the rest when the strip is done) ----
                if (!out_local) {
                        const int outc = done ? (ncols + 1) : min(max(tp - lastl, 0), ncols + 1);
                        const int n = min(outc - To, 64);
                        if (n >= (last_strip ? 64 : 8) || (done && n > 0)) {
                                if (last_strip) {
                                        // nobody reads the pass's last row before the level's barrier: plain stores
                                        if (lane < n) {
                                                const int c = To + lane;
                                                const float4v x = *(const __attribute__((address_space(3))) float4v*)(slds + KA_HO_RING + (((c + lastl) & 255) << 4));
                                                ka_gfloat* wr = grows + 3 * IDX(c);
                                                wr[0] = x.x; wr[1] = x.y; wr[2] = x.z;
                                        }
                                        To += n;
                                } else {
                                        // The strip below runs in another workgroup of the cluster.  Agent-scope WRITE-THROUGH stores, a wait for
                                        // their acknowledgements, then the flag -- not a release fence: at agent scope that is a write-back
                                        // of the XCD's whole L2 (ka_strip pays it once per 64 columns; "gets slower the more the other CUs
                                        // have written"), and this hand-over wants to publish every 8 columns.
                                        if (lane < n) {
                                                const int c = To + lane;
                                                const float4v x = *(const __attribute__((address_space(3))) float4v*)(slds + KA_HO_RING + (((c + lastl) & 255) << 4));
                                                float* wr = (float*)xrows + 3 * IDX(c);
                                                __hip_atomic_store(wr + 0, x.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                __hip_atomic_store(wr + 1, x.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                __hip_atomic_store(wr + 2, x.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        }
                                        To += n;
                                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                        if (lane == 0) __hip_atomic_store(prog + k, To, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                progress = true;
                        }
                }

                // ---- what the strip may do next ----
                const int taken = out_local ? __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[KA_W_TPUB(w + 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) : To;
                const int G = min(Lc > ncols ? KA_W_BIG : Lc, taken + 257 + lastl);
                if (G != lastG || Li != lastI) {
                        // (ring contents first, then the counts: LDS keeps a wave's order; the fence keeps the compiler's)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                                if (Li != lastI) __hip_atomic_store(&ctl[KA_W_IN(w)], Li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (G != lastG) __hip_atomic_store(&ctl[KA_W_GO(w)], G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        lastG = G; lastI = Li;
                }
                if (done && (out_local || To > ncols)) {
                        // (the rows this wave stored are read behind the level's barrier, whose release is ONE thread's fence: an L2
                        // write-back that does not wait for other waves' stores in flight -- so this wave's are acknowledged first)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        break;
                }
                if (!progress) {
                        __builtin_amdgcn_s_sleep(8);                  // ~500 cycles: the strip publishes every ~3000
                        if (ka_spin_expired(wdu, ++idle, 1 << 22, 5)) break;
                } else idle = 0;
        }
#undef REC
#undef IDX
}
