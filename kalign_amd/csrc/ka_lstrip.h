// ka_lstrip.h -- the LEAN profile-profile strip of the throughput kernels (round 6; unit 10 of ka_kernels.hip, -DKA_TP=1).
//
// What bounds the launches that hold more tasks than the GPU has workgroup slots (the queued launch, forests, shared contexts)
// is slots / latency: a task is as long under load as alone, its waves are parked 46 % of their cycles, and the
// slots are two four-wave workgroups per CU because a strip wave of ka_strip / ka_wstrip wants a 14 KB column ring (128 columns,
// chunk-major) next to 4 KB of hand-over ring and the kernel around it 256 VGPRs.  This strip is built for THREE workgroups per
// CU (<= 168 VGPRs, <= 53 KB of LDS per four-wave workgroup):
//   * the column ring holds 80 columns, record-major: column v at (v mod 80) * 112 B, its seven 16-byte chunks (profile fields
//     [32..59]) side by side.  A 112-byte lane stride walks all banks once per 8 (16) lanes, so the per-lane ds_read_b128 are as
//     conflict-free as the chunk-major ring's -- and ONE global_load_lds_dwordx4 fills eight columns (lane = column * 7 + chunk:
//     56 lanes, 896 contiguous bytes of LDS, eight 112-byte runs of HBM), where ka_strip needs seven instructions of 32 scattered
//     lines each per 32 columns;
//   * steps run in OCTETS (ka_wstrip's form: eight unrolled steps, compile-time parities, immediate LDS offsets for the out ring)
//     and what ka_wstrip's helper wave does is done by the strip itself at the top of every octet, wave-uniform and mostly on the
//     scalar unit: one ring batch issued (columns t+8 .. t+15, waited for at the octet's last step), the strip's last row drained
//     from its out ring to the row buffer 64 columns at a time (the owner lane writes one ds_write_b96 per step, exec narrowed by
//     scalar instructions), the boundary batch of a non-first strip fetched one octet before it is needed;
//   * first strips generate the pass's row -1 (aln_seqseq.c:40-58) on the way (lane 0 is at column t: three instructions per step).
// Hand-over between the strips of a pass goes through the row buffer behind the pass's progress words (workgroup scope: a task of
// these kernels never leaves its workgroup).  Same arithmetic as ka_strip / ka_wstrip, statement for statement: binary32, the
// reference's order (aln_profileprofile.c:17-298), no contraction.
#pragma once

#define KA_L_LA 2                                               // ring batches in flight: batch t/8 + LA goes out at the top of the octet that starts at step t
#define KA_L_RC (72 + 8 * KA_L_LA)                              // columns in the ring: 64 lanes' worth + the batches in flight + the one being read (88)
#define KA_L_CB 112                                             // bytes per column record in the ring: chunks 0..6 of fields [32..59]
#define KA_L_RING (KA_L_RC * KA_L_CB)                           // 9856
#define KA_L_OSLOTS 128                                         // out ring: the strip's last row, slot (step & 127), 16 B apiece
#define KA_L_OUT KA_L_RING
#define KA_L_BYTES (KA_L_RING + KA_L_OSLOTS * 16)               // 11904 B of the wave's region
static_assert(!KA_TP || KA_L_BYTES <= KA_WAVE_LDS, "the lean strip's rings outgrew the wave's LDS region");
#define KA_L_WAIT_RING (0x0F70 | ((KA_L_LA - 1) & 15))          // s_waitcnt vmcnt(LA - 1): all but the LA - 1 youngest batches have landed
#define KA_L_CH 8                                               // steps per octet = columns per ring batch

// The strip is a REAL function, one per (last row is its lane's row B, first strip of its pass): inlined into the task kernel -- one
// __global__ with three DP kinds, the recursion, the meetups, path coding and the profile merge in it -- its loop shares the kernel's
// register allocation with everything that is live across it: at the kernel's budget of 168 VGPRs every octet carried 7-26 scratch
// accesses per step (1099 spilled VGPRs in the unit).  As a function it has an allocation of its own; its arguments arrive in a
// struct by value; every lambda inside is inlined into it (left to itself the compiler outlined some of the phases and merged the
// rest into one body with 1.4 KB of scratch).
//
// ONE copy of the column record (28 registers, not a ping-pong pair): the dot product walks the residues 19 .. 0, i.e. the chunks
// 4 .. 0, and a chunk is read again -- the NEXT step's column, into the same registers -- right behind its last use; LDS returns a
// wave's reads in order, so the chunk the next step needs first was asked for first and has had a whole step to land.
#define KA_L_INLINE __attribute__((always_inline))
struct KaLStripArgs {
        const float* p1;                 // row profile
        const float* p2;                 // column profile
        const int2* ent;                 // consistency bonus entries (NB > 0)
        int* watchdog;
        KaState* rows;                   // the sub-problem's row buffer: f or b, at its slice
        int* prog;                       // progress words of the pass's strips
        char* wlds;                      // the wave's LDS region
        float m1, m2;                    // TaskShared::p1_mult / p2_mult
        float inj_a, inj_ga, inj_gb;     // the pass's injected boundary state
        int Lb;
        int starta, enda, startb, endb, dir, k;
        unsigned long long* prof;        // KA_L_PROF builds: TaskShared::sub_t of the profiled task (phase cycles of its strips), or null
};

template <int NRES, int NB, bool LASTB, bool FIRST>
__device__ __attribute__((noinline)) void ka_lstrip_v(const KaLStripArgs a)
{
        // (everything in `a` is wave-uniform, but arrives in vector registers: back to scalars)
#ifdef KA_L_PROF
        const long long tcall0 = __builtin_amdgcn_s_memtime();
#endif
        const int lane = threadIdx.x & 63;
        const int starta = ka_u(a.starta), enda = ka_u(a.enda), startb = ka_u(a.startb), endb = ka_u(a.endb), dir = ka_u(a.dir), k = ka_u(a.k);
        const float inj_a = ka_uniform_f(a.inj_a), inj_ga = ka_uniform_f(a.inj_ga), inj_gb = ka_uniform_f(a.inj_gb);
        KaState* const rows = ka_uniform_ptr(a.rows);
        int* const prog = ka_uniform_ptr(a.prog);
        char* const wlds = ka_uniform_ptr(a.wlds);
        const int ncols = endb - startb;
        const int mid = ((enda - starta) / 2) + starta;
        const int r0 = (dir == KA_FWD) ? starta : mid;
        const int r1 = (dir == KA_FWD) ? mid : enda;
        const int nrows = r1 - r0;                                    // > 0: strip items have rows (ka_pass_is_strip)
        const int Lb = ka_u(a.Lb);
        const bool near_t = (dir == KA_FWD) ? (startb == 0) : (endb == Lb);
        const bool far_t = (dir == KA_FWD) ? (endb == Lb) : (startb == 0);
#define REC(v_) ((dir == KA_FWD) ? (startb + (v_)) : (endb + 1 - (v_)))
#define IDX(v_) ((dir == KA_FWD) ? (v_) : (ncols - (v_)))
        const float* const p1 = ka_uniform_ptr(a.p1);
        const float* const p2u = ka_uniform_ptr(a.p2);
        int* const wdu = ka_uniform_ptr(a.watchdog);
        const float m1 = ka_uniform_f(a.m1), m2 = ka_uniform_f(a.m2);
        ka_gfloat* const grows = (ka_gfloat*)rows;

        constexpr int SROWS = 128;
        const int u0 = k * SROWS;
        const int nr = min(SROWS, nrows - u0);
        const int nl = (nr + 1) >> 1;
        const int lastl = nl - 1;                                     // lane holding the strip's last row
        const bool last_strip = (u0 + SROWS >= nrows);
        const bool actB = 2 * lane + 1 < nr;
        const int uA = u0 + min(2 * lane, nr - 1);
        const int uB = u0 + min(2 * lane + 1, nr - 1);
        const int iA = (dir == KA_FWD) ? (r0 + uA) : (r1 - 1 - uA);
        const int iB = (dir == KA_FWD) ? (r0 + uB) : (r1 - 1 - uB);
        const int recA = iA + 1, recB = iB + 1;
        const int prevA = (dir == KA_FWD) ? recA - 1 : recA + 1;
        const int prevB = (dir == KA_FWD) ? recB - 1 : recB + 1;
        const int nsteps = ncols + nl;                                // t = 0 .. ncols + nl - 1

        const unsigned wlds_u = (unsigned)(unsigned long long)wlds;
        const unsigned out_u = wlds_u + KA_L_OUT;

        // ---- the column ring: batch b = columns 8b .. 8b+7, lane = column * 7 + chunk ----
        const int cib = lane / 7, rch = lane - 7 * cib;               // (lanes 56..63: cib = 8, never inside a batch)
        int rb_slot = 0;                                              // ring slot of the next batch's first column (uniform)
        int rb_next = 0;                                              // the next batch to issue
        auto ring_issue = [&]() KA_L_INLINE {
                const int c = rb_next * KA_L_CH + cib;
                if (cib < KA_L_CH && c <= ncols) {
                        const float* g = p2u + ((long long)REC(c) << 6) + 32 + 4 * rch;
                        __builtin_amdgcn_global_load_lds((ka_glb_ptr)g, (ka_lds_ptr)(wlds + rb_slot * KA_L_CB), 16, 0, 0);
                }
                rb_next += 1;
                rb_slot += KA_L_CH;
                if (rb_slot == KA_L_RC) rb_slot = 0;
        };
        __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);                      // whatever this wave had in flight towards its region has landed
        ring_issue();                                                 // columns 0 .. 7
        for (int b = 1; b < KA_L_LA; ++b) if (rb_next * KA_L_CH <= ncols) ring_issue(); else { rb_next += 1; rb_slot += KA_L_CH; }

        // ---- stationary row operand (as in ka_strip) ----
        float oA, eA, tA, oB, eB, tB, orpA, orpB;
        float2v p1v[NRES];                                            // the counts of residue c in rows (A, B)
        {
                const float* pA = p1 + ((long long)recA << 6);
                const float* pB = p1 + ((long long)recB << 6);
                oA = pA[55] * m1; eA = pA[56] * m1; tA = pA[57] * m1;
                oB = pB[55] * m1; eB = pB[56] * m1; tB = pB[57] * m1;
                orpA = p1[((long long)prevA << 6) + 55] * m1;
                orpB = p1[((long long)prevB << 6) + 55] * m1;
                constexpr int NV = (NRES + 3) / 4;
                float4v va[NV], vb[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) { va[i] = ((const float4v*)pA)[i]; vb[i] = ((const float4v*)pB)[i]; }
#pragma unroll
                for (int c = 0; c < NRES; ++c) {
                        p1v[c].x = va[c >> 2][c & 3];
                        p1v[c].y = actB ? vb[c >> 2][c & 3] : 0.0f;
                }
        }
        KaBonus<NB> bonA, bonB;
        if (NB) { bonA.load(a.ent, iA, dir); bonB.load(a.ent, iB, dir); }

        float cAa = -KA_F, cAga = -KA_F, cAgb = -KA_F;
        float cBa = -KA_F, cBga = -KA_F, cBgb = -KA_F;
        float dga = -KA_F, dgga = -KA_F, dggb = -KA_F;
        float copen_prev = 0.0f;
        // FIRST: row -1 of the pass, the state of column t; else: the batch of the row above (lane 0 = column t, rotated every step)
        float bta = FIRST ? inj_a : -KA_F, btga = FIRST ? inj_ga : -KA_F, btgb = FIRST ? inj_gb : -KA_F;
        float nbta = -KA_F, nbtga = -KA_F, nbtgb = -KA_F;             // !FIRST: the next batch, fetched one octet ahead
        // the column record of this step's column, as exactly the registers the step reads (a 16-byte read whose tail is dead lets the
        // allocator hand the dead registers out again -- and every write to one of them then waits for the read to land: two exposed
        // LDS latencies per step in the first version): whole chunks of four scores, a lone fifth score (nucleotides), fields 55 .. 57
        constexpr int NQ = (NRES == 23) ? 6 : NRES / 4;               // whole chunks (23 residues: chunk 5 = scores 20 .. 22 + field 55)
        float4v q[NQ > 0 ? NQ : 1];
        float qs = 0.0f;                                              // NRES == 5: the score of residue 4
        float g55 = 0.0f;                                             // NRES != 23: field 55 (the base gap-open penalty)
        float2v g56 = {0.0f, 0.0f};                                   // fields 56, 57
        int To = 0;                                                   // columns of the last row drained to the row buffer
        unsigned out_oct = 0;
        // byte offset in the ring of the column the reads of step t take (the NEXT step's): column t + 1 - lane
        unsigned ra = (unsigned)((((1 - lane) % KA_L_RC) + KA_L_RC) % KA_L_RC) * KA_L_CB;

        typedef const __attribute__((address_space(3))) float4v ka_l4;
        typedef const __attribute__((address_space(3))) float2v ka_l2;
        typedef const __attribute__((address_space(3))) float ka_l1;
        static_assert(NRES == 5 || NRES == 20 || NRES == 23, "the record's register form is written out for these alphabets");
        auto prog_wait = [&](const int need) KA_L_INLINE {
                if (lane == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(prog + (k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                                __builtin_amdgcn_s_sleep(2);
                                if (ka_spin_expired(wdu, ++spins, 1 << 22, 5)) break;
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        // 64 columns (or what is left) of the strip's last row from the out ring to the row buffer; columns < upto are in the ring
        auto drain = [&](const int upto) KA_L_INLINE {
                const int c = To + lane;
                if (c < upto) {
                        const float4v x = *(ka_l4*)(unsigned long)(out_u + ((((unsigned)(c + lastl)) & (KA_L_OSLOTS - 1)) << 4));
                        ka_gfloat* w = grows + 3 * IDX(c);
                        w[0] = x.x; w[1] = x.y; w[2] = x.z;
                }
                To = min(To + 64, upto);
                // (vmcnt counts stores too and does not promise their order against loads: the ring's partial waits want only loads in flight)
                __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);
                if (!last_strip) {
                        // the strip below (another wave of this workgroup) may read them; the LAST strip of a pass has no reader before
                        // the level's barrier
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_store(prog + k, To, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
        };
        // top of every octet (t a multiple of 8), wave-uniform: steps < t are computed, their last-row states are in the out ring
        auto bookkeeping = [&](const int t) KA_L_INLINE {
                const int avail = min(t - lastl, ncols + 1);          // columns of the last row written so far
                if (avail - To >= 64) drain(avail);
                if (!FIRST) {
                        if ((t & 63) == 0 && t > 0) { bta = nbta; btga = nbtga; btgb = nbtgb; }
                        if ((t & 63) == 56 && t + 8 <= ncols) {
                                const int nxt = t + 8;
                                prog_wait(min(nxt + 64, ncols + 1));
                                const ka_gfloat* r = grows + 3 * IDX(min(nxt + lane, ncols));
                                nbta = r[0]; nbtga = r[1]; nbtgb = r[2];
                        }
                }
                // the ring batch LAST: a release fence or a use of the boundary batch above waits for everything outstanding
                // (rb_next == t / 8 + LA here; batches wholly beyond the last column are not fetched, nobody inside the range reads them)
                if (rb_next * KA_L_CH <= ncols) ring_issue(); else { rb_next += 1; rb_slot += KA_L_CH; if (rb_slot == KA_L_RC) rb_slot = 0; }
        };

        // One wavefront step.  FORM: 0 steady state, 1 head (lanes entering: the lane at column 0 is the special one), 2 tail
        // (the lane at the last column), 3 both (fewer columns than lanes); I: position in an octet (0 .. 7, compile time) or -1
        auto step = [&](const int t, auto st_tag, auto i_tag) KA_L_INLINE {
                constexpr int FORM = decltype(st_tag)::value;
                constexpr bool ST = FORM == 0;
                constexpr int I = decltype(i_tag)::value;
                const int v = t - lane;

                // the strip's last row as the PREVIOUS step left it: its owner writes column t - 1 - lastl into out slot (t - 1) & 127
                {
                        const int vLp = t - 1 - lastl;
                        if (FORM != 1 && (FORM != 3 || (vLp >= 0 && vLp <= ncols))) {
                                float3v o;
                                o.x = LASTB ? cBa : cAa; o.y = LASTB ? cBga : cAga; o.z = LASTB ? cBgb : cAgb;
                                unsigned long long sv, sm;
                                if constexpr (I >= 1) {
                                        asm volatile("s_lshl_b64 %1, 1, %4\n\ts_and_saveexec_b64 %0, %1\n\tds_write_b96 %2, %3 offset:%5\n\ts_mov_b64 exec, %0"
                                                     : "=&s"(sv), "=&s"(sm) : "v"(out_oct), "v"(o), "s"(__builtin_amdgcn_readfirstlane(lastl)), "n"((I - 1) * 16) : "memory", "scc");
                                } else {
                                        const unsigned oa = out_u + ((((unsigned)t - 1u) & (KA_L_OSLOTS - 1)) << 4);
                                        asm volatile("s_lshl_b64 %1, 1, %4\n\ts_and_saveexec_b64 %0, %1\n\tds_write_b96 %2, %3\n\ts_mov_b64 exec, %0"
                                                     : "=&s"(sv), "=&s"(sm) : "v"(oa), "v"(o), "s"(__builtin_amdgcn_readfirstlane(lastl)) : "memory", "scc");
                                }
                        }
                }
                if (I == 0 || (I < 0 && (t & (KA_L_CH - 1)) == 0)) bookkeeping(t);
                float copen = (NRES == 23 ? q[NQ - 1].w : g55) * m2, cext = g56.x * m2, ctext = g56.y * m2;
                // the batch that holds lane 0's next column was issued at the top of this octet: waited for at the octet's last step,
                // in front of the step's first read of the next column
                if (I == KA_L_CH - 1 || (I < 0 && ((t + 1) & (KA_L_CH - 1)) == 0)) __builtin_amdgcn_s_waitcnt(KA_L_WAIT_RING);
                const unsigned cra = wlds_u + ra;                     // the next step's column
                ka_l4* const cr = (ka_l4*)(unsigned long)cra;
                // a piece of the record is read again right behind its last use (the empty asm holds the read behind `dep`, the barrier
                // holds it in place)
                auto reload = [&](const int ch, auto& dep) KA_L_INLINE {
                        asm volatile("" : "+v"(dep) : : "memory");
                        q[ch] = cr[ch];
                        __builtin_amdgcn_sched_barrier(0);
                };
                {
                        asm volatile("" : "+v"(ctext) : : "memory");
                        if (NRES != 23) g55 = *(ka_l1*)(unsigned long)(cra + 92);
                        g56 = *(ka_l2*)(unsigned long)(cra + 96);
                        __builtin_amdgcn_sched_barrier(0);
                }

                // ---- the row above A: lane l-1's row B; lane 0 takes the state of column t of the row above the strip ----
                float upa, upga, upgb;
                if (FIRST) {
                        // row -1 of the pass (aln_seqseq.c:40-58): lane 0 is at column t
                        if (!ST && t == 0) {
                                bta = inj_a; btga = inj_ga; btgb = inj_gb;
                        } else if (ST || t < ncols) {
                                // max(x, y) + c == max(x + c, y + c) bit for bit (rounding is monotonic)
                                const float gx = near_t ? ctext : cext, gy = near_t ? ctext : copen;
                                const float g = kmax(btga + gx, bta + gy);
                                bta = -KA_F; btga = g; btgb = -KA_F;
                        } else {
                                bta = -KA_F; btga = -KA_F; btgb = -KA_F;
                        }
                        upa = wave_shr1_old(bta, cBa); upga = wave_shr1_old(btga, cBga); upgb = wave_shr1_old(btgb, cBgb);
                } else {
                        const float r_a = wave_rol1(bta), r_ga = wave_rol1(btga), r_gb = wave_rol1(btgb);
                        upa = wave_shr1_old(bta, cBa); upga = wave_shr1_old(btga, cBga); upgb = wave_shr1_old(btgb, cBgb);
                        bta = r_a; btga = r_ga; btgb = r_gb;
                }

                const bool at0 = (FORM == 1 || FORM == 3) && (v == 0), atN = (FORM == 2 || FORM == 3) && (v == ncols);
                const bool edge = at0 | atN;
                float nAga, nAgb, nBga;
                // (edge forms: selects on the OPERANDS, as in ka_wstrip: the terminal case `max(gb, a) + t` is the inner case with both
                // penalties replaced by t)
                float xeA = eA, xoA = oA, xeB = eB, xoB = oB;
                if (FORM == 1 || FORM == 3) {
                        xeA = at0 ? (near_t ? tA : eA) : xeA; xoA = at0 ? (near_t ? tA : oA) : xoA;
                        xeB = at0 ? (near_t ? tB : eB) : xeB; xoB = at0 ? (near_t ? tB : oB) : xoB;
                }
                if (FORM == 2 || FORM == 3) {
                        xeA = atN ? (far_t ? tA : eA) : xeA; xoA = atN ? (far_t ? tA : oA) : xoA;
                        xeB = atN ? (far_t ? tB : eB) : xeB; xoB = atN ? (far_t ? tB : oB) : xoB;
                }
                if (ST) {
                        nAga = kmax(cAga + cext, cAa + copen);
                        nAgb = kmax(upgb + eA, upa + oA);
                        nBga = kmax(cBga + cext, cBa + copen);
                } else {
                        nAga = edge ? -KA_F : kmax(cAga + cext, cAa + copen);
                        nAgb = kmax(upgb + xeA, upa + xoA);
                        nBga = edge ? -KA_F : kmax(cBga + cext, cBa + copen);
                }
                float2v acc;
                acc.x = kmax3(dga, dgga + copen_prev, dggb + orpA);
                acc.y = kmax3(cAa, cAga + copen_prev, cAgb + orpB);
                {
                        // residue NRES-1 first (aln_profileprofile.c:99-107 walks the non-zero counts downwards); products one term ahead
                        // of the (dependent) sums
                        float2v prod;
                        if (NRES == 5) {
                                float2v w; w.x = qs; w.y = qs;
                                prod = p1v[4] * w;
                                asm volatile("" : "+v"(prod) : : "memory");
                                qs = *(ka_l1*)(unsigned long)(cra + 16);
                                __builtin_amdgcn_sched_barrier(0);
                        } else {
                                prod = ka_mul_bcast<(NRES - 1) & 3>(p1v[NRES - 1], q[(NRES - 1) >> 2]);
                        }
#pragma unroll
                        for (int c = NRES - 1; c >= 1; --c) {
                                float2v nprod;
                                switch ((c - 1) & 3) {
                                case 0: nprod = ka_mul_bcast<0>(p1v[c - 1], q[(c - 1) >> 2]); break;
                                case 1: nprod = ka_mul_bcast<1>(p1v[c - 1], q[(c - 1) >> 2]); break;
                                case 2: nprod = ka_mul_bcast<2>(p1v[c - 1], q[(c - 1) >> 2]); break;
                                default: nprod = ka_mul_bcast<3>(p1v[c - 1], q[(c - 1) >> 2]); break;
                                }
                                if (((c - 1) & 3) == 0) reload((c - 1) >> 2, nprod);     // (the chunk's last use: residue 4 * chunk)
                                acc = acc + prod;
                                prod = nprod;
                        }
                        acc = acc + prod;
                        if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); acc.x += bonA.template at<(FORM >= 2)>(jb); acc.y += bonB.template at<(FORM >= 2)>(jb); }
                }
                const float nAa = at0 ? -KA_F : acc.x;
                const float nBa = at0 ? -KA_F : acc.y;
                const float nBgb = kmax(nAgb + xeB, nAa + xoB);           // B: the row above is A's fresh state
                cAa = nAa; cAga = nAga; cAgb = nAgb;
                cBa = nBa; cBga = nBga; cBgb = nBgb;
                dga = upa; dgga = upga; dggb = upgb;
                copen_prev = copen;
                const unsigned x = ra + KA_L_CB;
                ra = min(x, x - KA_L_RING);                           // (unsigned: x - 8960 wraps unless x == 8960)
        };

        auto single = [&](const int t, auto st_tag) KA_L_INLINE { step(t, st_tag, std::integral_constant<int, -1>()); };
        auto octet = [&](const int t0, auto st_tag) KA_L_INLINE {
                out_oct = out_u + (((unsigned)t0 & (KA_L_OSLOTS - 1)) << 4);
                step(t0 + 0, st_tag, std::integral_constant<int, 0>());
                step(t0 + 1, st_tag, std::integral_constant<int, 1>());
                step(t0 + 2, st_tag, std::integral_constant<int, 2>());
                step(t0 + 3, st_tag, std::integral_constant<int, 3>());
                step(t0 + 4, st_tag, std::integral_constant<int, 4>());
                step(t0 + 5, st_tag, std::integral_constant<int, 5>());
                step(t0 + 6, st_tag, std::integral_constant<int, 6>());
                step(t0 + 7, st_tag, std::integral_constant<int, 7>());
        };
        auto run = [&](int& t, const int tend, auto st_tag) KA_L_INLINE {
                for (; t < tend && (t & (KA_L_CH - 1)); ++t) single(t, st_tag);
                for (; t + KA_L_CH <= tend; t += KA_L_CH) octet(t, st_tag);
                for (; t < tend; ++t) single(t, st_tag);
        };
        static_assert(KA_L_CH == 8, "the octet is written out for eight steps");

        // the operands of step 0: column 0 of the ring, and (other than first strips) the first batch of the row above
        if (!FIRST) {
                prog_wait(min(64, ncols + 1));
                const ka_gfloat* r = grows + 3 * IDX(min(lane, ncols));
                bta = r[0]; btga = r[1]; btgb = r[2];
        }
        __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);                      // ring batch 0 has landed
        {
                ka_l4* const c0 = (ka_l4*)(unsigned long)wlds_u;
#pragma unroll
                for (int ch = 0; ch < NQ; ++ch) q[ch] = c0[ch];
                if (NRES == 5) qs = *(ka_l1*)(unsigned long)(wlds_u + 16);
                if (NRES != 23) g55 = *(ka_l1*)(unsigned long)(wlds_u + 92);
                g56 = *(ka_l2*)(unsigned long)(wlds_u + 96);
        }
        {
                const int t_steady0 = min(nl, nsteps);
                const int t_steady1 = ncols;
                int t = 0;
                if (ncols >= nl) {
#ifdef KA_L_PROF
                        const long long tp0 = __builtin_amdgcn_s_memtime();
#endif
                        run(t, t_steady0, std::integral_constant<int, 1>());
#ifdef KA_L_PROF
                        const long long tp1 = __builtin_amdgcn_s_memtime();
#endif
                        run(t, t_steady1, std::integral_constant<int, 0>());
#ifdef KA_L_PROF
                        const long long tp2 = __builtin_amdgcn_s_memtime();
#endif
                        run(t, nsteps, std::integral_constant<int, 2>());
#ifdef KA_L_PROF
                        // per strip: [0] strips, [1] head / [2] steady / [3] tail cycles, [4] whole call, [5] longest call, [6] steady steps
                        if (a.prof && lane == 0 && k == 0 && nr == 128) {
                                unsigned long long* pf = ka_uniform_ptr(a.prof);
                                const long long tp3 = __builtin_amdgcn_s_memtime();
                                atomicAdd(&pf[0], 1ull); atomicAdd(&pf[1], (unsigned long long)(tp1 - tp0)); atomicAdd(&pf[2], (unsigned long long)(tp2 - tp1));
                                atomicAdd(&pf[3], (unsigned long long)(tp3 - tp2)); atomicAdd(&pf[4], (unsigned long long)(tp3 - tcall0)); atomicMax(&pf[5], (unsigned long long)(tp3 - tcall0));
                                atomicAdd(&pf[6], (unsigned long long)(t_steady1 - t_steady0));
                        }
#endif
                } else {
                        // fewer columns than lanes: the general edge form, step by step (short passes of deep recursion levels)
                        for (; t < nsteps; ++t) single(t, std::integral_constant<int, 3>());
                }
        }
        // the last step's column (vL = ncols), then what is left of the row
        {
                float3v o;
                o.x = LASTB ? cBa : cAa; o.y = LASTB ? cBga : cAga; o.z = LASTB ? cBgb : cAgb;
                unsigned long long sv, sm;
                const unsigned oa = out_u + ((((unsigned)nsteps - 1u) & (KA_L_OSLOTS - 1)) << 4);
                asm volatile("s_lshl_b64 %1, 1, %4\n\ts_and_saveexec_b64 %0, %1\n\tds_write_b96 %2, %3\n\ts_mov_b64 exec, %0"
                             : "=&s"(sv), "=&s"(sm) : "v"(oa), "v"(o), "s"(__builtin_amdgcn_readfirstlane(lastl)) : "memory", "scc");
        }
        while (To <= ncols) drain(ncols + 1);
#undef REC
#undef IDX
}

// the strip of work item (sub-problem window, direction, k): which of the four forms it is
template <int NRES, int NB>
__device__ __forceinline__ void ka_lstrip(const KaLStripArgs& a)
{
        const int mid = ((a.enda - a.starta) / 2) + a.starta;
        const int nrows = (a.dir == KA_FWD) ? mid - a.starta : a.enda - mid;
        const int nr = min(128, nrows - a.k * 128);
        const bool last_is_b = (nr & 1) == 0;
        if (a.k == 0) { if (last_is_b) ka_lstrip_v<NRES, NB, true, true>(a); else ka_lstrip_v<NRES, NB, false, true>(a); }
        else { if (last_is_b) ka_lstrip_v<NRES, NB, true, false>(a); else ka_lstrip_v<NRES, NB, false, false>(a); }
}
