// ka_pass.h -- one 128-row STRIP of a linear-space Gotoh pass (forward or backward) of one
// Hirschberg sub-problem, executed by ONE wave64 as an anti-diagonal wavefront.
//
// Mapping (same (u, v) formulation as oracle/kalign_oracle.c:ko_pass):
//   * a pass over nrows x ncols is cut into strips of 128 rows; strip k of a pass is one work
//     item; the strips of a pass form a pipeline across the waves of the workgroup: strip k
//     consumes the last row of strip k-1 from the sub-problem's row buffer (in place) 64
//     columns at a time, guarded by a per-strip progress word (workgroup-scope release/acquire);
//   * lane l owns the TWO rows u0+2l (A) and u0+2l+1 (B) and keeps their operand data
//     stationary in registers (profile-profile: 23 residue counts per row, packed as float2 so
//     the 23-term dot products of A and B run as v_pk_mul_f32 / v_pk_add_f32);
//   * at step t lane l is at column v = t - l; the state of the row above A arrives from lane
//     l-1 by a DPP wave shift (v_mov_b32_dpp wave_shr:1), B takes A's fresh state;
//   * the column operand is streamed: profile-profile stages the 28 useful floats of every
//     column record (fields 32..59: pre-summed substitution scores + base gap penalties,
//     aln_setup.c:40-99) into a per-wave LDS ring with direct global->LDS loads
//     (global_load_lds_dwordx4), 32 columns per batch, chunk-major so the per-lane reads are
//     conflict-free ds_read_b128, issued one step ahead; sequence operands flow lane to lane
//     through one more DPP shift;
//   * the strip's last row is collected in registers (one column per lane) and handed on 64 columns at a time: to the
//     row buffer (coalesced, then published), or -- neighbouring waves of one workgroup -- through a ring in LDS (HO, below).
//
// All arithmetic is binary32 in the reference's order, no contraction (see ka_kernels.hip).
#pragma once

#include <type_traits>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define KA_STRIP_ROWS 128                                       // Q = 2: two DP rows per lane
#define KA_STRIP1_ROWS 64                                       // Q = 1: one DP row per lane (narrow step, see ka_strip)
#define KA_RING_BATCH 32
#define KA_RING_SLOTS 4
#define KA_REC_CHUNKS 7                                         // 7 x 16 B = profile fields [32..59]
#define KA_SLOT_BYTES (KA_REC_CHUNKS * KA_RING_BATCH * 16)      // 3584
#define KA_RING_BYTES (KA_RING_SLOTS * KA_SLOT_BYTES)           // 14336 B: the column ring of a strip wave
// KA_TP (round 6, unit 10: the THROUGHPUT kernel -- three four-wave workgroups per CU): profile-profile strips run as ka_lstrip
// (ka_lstrip.h: an 80-column record-major ring + a 2 KB out ring), so a wave's region shrinks to 11 KB; the seq-profile score table
// takes 21 floats per row (the kernel only runs jobs without B / Z / X: residues < 20), packed passes stage 64 records per region.
#ifndef KA_TP
#define KA_TP 0
#endif
#if KA_TP
#define KA_WAVE_LDS 12288
#define KA_SP_STRIDE 21
#define KA_PK_RECS 96
#else
#define KA_WAVE_LDS 18432                                       // per-wave LDS region: the ring, or the staging area of a wave-local subtree (ka_subtree.h)
#define KA_SP_STRIDE 25                                         // floats per row in the seq-profile score table
#define KA_PK_RECS 128                                          // packed passes: staged column records per region
#endif
#define KA_SP_FILL (KA_SP_STRIDE < 23 ? KA_SP_STRIDE : 23)      // scores copied into a row of the seq-profile table
#define KA_WAVE_LDS_LEAN 6144                                   // the same in the seq-seq kernels (no ring; residues instead of records)
#define KA_T_STRIDE 24                                          // floats per row in the seq-seq score table

// Column-record reads of the DP steps (ka_strip, ka_sub_pass; ka_wstrip outside its steady octets): plain loads the compiler
// tracks (0, the default), or inline-asm ds_read_b128 it does not, waited for by a hand-placed s_waitcnt one step later (1).
// The untracked form was round 3's: between the asm that issues the reads and the asm that waits, the compiler believes the
// destination registers written -- and where a loop exit or a join wants the record in other registers it COPIES them
// (v_mov_b64 of all 28) while the data may still be on its way; tools/check_lds_hazards.py finds those copies in the built
// objects (2387 uses of in-flight registers in the round-3 form of the fast-mode kernel).  They only bite when LDS is slow
// enough (four waves streaming 7 KB per step each) -- the intermittent wrong rows round 4 chased in ka_wstrip.  The tracked
// form keeps the reads where they were (sched_barrier) and marks the old wait's place with an empty asm that USES the
// registers: the compiler puts the s_waitcnt it needs in front of it, and in front of any copy it makes before.
#ifndef KA_BONUS_ASM
#define KA_BONUS_ASM 1
#endif
#ifndef KA_UNTRACKED_READS
#define KA_UNTRACKED_READS 0
#endif
#define KA_WAIT_VM0 0x0F70                                      // s_waitcnt vmcnt(0) only (gfx9 encoding)

typedef __attribute__((address_space(3))) void* ka_lds_ptr;
typedef const __attribute__((address_space(1))) void* ka_glb_ptr;
typedef __attribute__((address_space(1))) float ka_gfloat;
typedef const __attribute__((address_space(1))) float4v ka_gfloat4c;          // global (not flat) loads: a flat load counts on
typedef const __attribute__((address_space(1))) unsigned char ka_gbytec;       // lgkmcnt too and is caught by every LDS wait

__device__ __forceinline__ int wave_shr1_i(int x)
{
        return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
}

// lane l <- lane l-1 of src; lane 0 keeps its own `old` (used to splice the boundary state in)
__device__ __forceinline__ float wave_shr1_old(float old, float src)
{
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x138, 0xf, 0xf, false));
}
// lane l <- lane l+1 of src; lane 63 keeps its own `old` (shift register fed at the top lane)
__device__ __forceinline__ float wave_shl1_old(float old, float src)
{
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x130, 0xf, 0xf, false));
}
// (e, e) of a float4 as a shuffle: selects straight into the op_sel broadcast of v_pk_mul_f32 (built from a scalar
// extract, the .w elements went through a v_mov_b32 first)
// a * (v[E], v[E]): one v_pk_mul_f32 with an op_sel broadcast.  The compiler finds the broadcast for elements 0..2
// but routes .w through a v_mov_b32 first, so that one is spelled out: src1 = the (z, w) half of the float4,
// op_sel:[0,1] takes its high half for the low lane (op_sel_hi defaults to [1,1]).
template <int E>
__device__ __forceinline__ float2v ka_mul_bcast(const float2v a, const float4v& v)
{
        if (E == 3) {
                const float2v zw = __builtin_shufflevector(v, v, 2, 3);
                float2v r;
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(zw));
                return r;
        }
        const float sc = v[E];
        float2v w; w.x = sc; w.y = sc;
        return a * w;
}

// rotate: lane l <- lane (l+1) mod 64
__device__ __forceinline__ float wave_rol1(float x)
{
        // a rotation writes every lane: no `old` operand (which would be tied to the destination and cost a copy
        // whenever the source stays live)
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x134, 0xf, 0xf, true));
}

// rotate: lane l <- lane (l-1) mod 64
__device__ __forceinline__ float wave_ror1(float x)
{
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x13C, 0xf, 0xf, true));
}

// Chunks (16 B = 4 fields) of a column record's fields [32..59] that an alphabet of NRES residues reads: the scores
// [32 .. 32+NRES) and the gap fields [55..57] (chunks 5 and 6).  Nucleotides (NRES = 5): chunks 0, 1, 5, 6 -- the other
// three would be dead global->LDS copies and LDS reads on every column of every step.
template <int NRES>
__device__ __forceinline__ constexpr bool ka_chunk_used(int ch) { return ch * 4 < NRES || ch >= 5; }

// srows: rows per strip of the task (TaskShared::srows: 128 or 64)
__device__ __forceinline__ int ka_strips_of(int nrows, int srows) { return nrows <= 0 ? 1 : (nrows + srows - 1) / srows; }

// NB > 0: anchor-consistency build -- every DP row carries NB (column, value) bonus entries with distinct
// columns (ka_cons_prepare); the cell at s-index j adds the value of the entry whose column is j, which
// is what the reference's dense `pa += consistency[i*stride + j]` adds (aln_seqseq.c:83-85,199-201).
//
// NB > KA_NB (round 5: the kernels for more than five anchors): STREAM -- the row's entries are not held in registers but walked.  They
// lie sorted by column in the row's slice of S.ent (ka_cons_entries: [1] = (INT_MIN, count), [2 .. count + 1] the entries ascending,
// [count + 2] = (INT_MAX, 0), pads at [0] and [count + 3]); a lane's column only ever moves one way during a pass, so it keeps the
// entry it will meet next (and the one after, already fetched: the load of a new one is a trip to L2 whose latency then hides behind a
// step) and advances past the columns it has left behind.  Eight registers per DP row for any number of anchors.
template <int NB>
struct KaBonus {
        static constexpr bool STREAM = NB > KA_NB;
        int col[(NB > 0 && !STREAM) ? NB : 1];
        float val[(NB > 0 && !STREAM) ? NB : 1];
        const int2* sp;                                  // STREAM: the entry after `n*` (the next one to fetch is sp + sstep)
        int sstep, ccol, ncol;
        float cval, nval;
        __device__ __forceinline__ void load(const int2* ent, int row, int dir = KA_FWD)
        {
                if constexpr (STREAM) {
                        const int2* e = ent + (long long)row * NB;
                        const int cnt = e[1].y;
                        sstep = (dir == KA_FWD) ? 1 : -1;
                        const int2* c = (dir == KA_FWD) ? e + 2 : e + 1 + cnt;       // first entry in walking order (a sentinel when there is none)
                        const int2 x = c[0], y = c[sstep];
                        ccol = x.x; cval = __int_as_float(x.y); ncol = y.x; nval = __int_as_float(y.y);
                        sp = c + sstep;
                        return;
                }
#pragma unroll
                for (int e = 0; e < NB; ++e) { const int2 x = ent[(long long)row * NB + e]; col[e] = x.x; val[e] = __int_as_float(x.y); }
        }
        // STREAM: the value of the entry at column j, after leaving the entries behind j behind.  (A lane outside its column range asks
        // for columns before the window -- no advance -- or beyond it -- it runs into the sentinel; what it gets is never used.)
        __device__ __forceinline__ float at_stream(int j)
        {
                while (sstep > 0 ? (ccol < j) : (ccol > j)) {
                        ccol = ncol; cval = nval;
                        // (never past the pads: an advance happens only while ccol is not yet the sentinel the walk ends on)
                        sp += sstep;
                        const int2 y = *sp;
                        ncol = y.x; nval = __int_as_float(y.y);
                }
                return ccol == j ? cval : 0.0f;
        }
        // EDGE = false: the wrap-around entry (slot NB-1, column Lb) is left out -- it can only match in the last column
        // of a pass, which steady-state steps never are
        // Five entries at a time as five compares into five SGPR pairs, then five selects: written as `(col[e] == j) ? val[e] : b`
        // the compiler makes cmp -> vcc -> cndmask pairs, and a VALU write of vcc wants two wait states before a cndmask reads it
        // (gfx950): 14 s_nop in the 117-instruction steady step of the default-mode strip.  Here every select reads a mask
        // written five instructions earlier.  (KA_BONUS_ASM=0: the plain form.)
        template <bool EDGE>
        __device__ __forceinline__ float at(int j)
        {
                if constexpr (STREAM) return at_stream(j);
                float b = 0.0f;
                constexpr int N = EDGE ? NB : NB - 1;
                int e0 = 0;
#if KA_BONUS_ASM
#pragma unroll
                for (; e0 + 5 <= N; e0 += 5) {
                        unsigned long long m0, m1, m2, m3, m4;
                        asm("v_cmp_eq_u32_e64 %1, %6, %7\n\t"
                            "v_cmp_eq_u32_e64 %2, %6, %8\n\t"
                            "v_cmp_eq_u32_e64 %3, %6, %9\n\t"
                            "v_cmp_eq_u32_e64 %4, %6, %10\n\t"
                            "v_cmp_eq_u32_e64 %5, %6, %11\n\t"
                            "v_cndmask_b32_e64 %0, %0, %12, %1\n\t"
                            "v_cndmask_b32_e64 %0, %0, %13, %2\n\t"
                            "v_cndmask_b32_e64 %0, %0, %14, %3\n\t"
                            "v_cndmask_b32_e64 %0, %0, %15, %4\n\t"
                            "v_cndmask_b32_e64 %0, %0, %16, %5"
                            : "+v"(b), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4)
                            : "v"(j), "v"(col[e0]), "v"(col[e0 + 1]), "v"(col[e0 + 2]), "v"(col[e0 + 3]), "v"(col[e0 + 4]),
                              "v"(val[e0]), "v"(val[e0 + 1]), "v"(val[e0 + 2]), "v"(val[e0 + 3]), "v"(val[e0 + 4]));
                }
#endif
#pragma unroll
                for (int e = e0; e < N; ++e) b = (col[e] == j) ? val[e] : b;
                return b;
        }
};

// Q: DP rows per lane.  Q = 2 (128-row strips) is the throughput shape: ~100 instructions per 128 cells.  Q = 1 (64-row
// strips) is the latency shape for tasks that own idle SIMDs: ~55 instructions per step -- a lone wave issues one
// instruction per ~5 cycles, so the step, and with it the C + R/Q + hand-over steps of a pass, costs about half -- at
// twice the waves.  The profile-profile dot product then packs two RESIDUES per v_pk_mul_f32 (the counts of residues
// 2i, 2i+1 against the matching half of the column record's float4) and adds the two products one after the other.
//
// HO (round 3): two strips of one pass that were dealt to NEIGHBOURING waves of one workgroup (wave w-1 -> wave w) hand the
// boundary row over through LDS instead of the HBM row buffer -- the same 64-column batches at the same steps, but a batch
// is three ds_write / ds_read and a progress word in LDS instead of global stores behind a release fence (which waits
// for the stores to be acknowledged: ~1700 cycles per event in a strip that waits for nobody, ~2900 in one that does,
// three events per 64 steps; profiles/r03b_strip_phases.log) and global loads behind a flag in HBM.  The step itself is
// the same code either way: the choice is a wave-uniform branch inside the event steps.
//   * producer: its 256-slot ring (16 B per column) sits in its own LDS region behind the column ring (KA_HO_RING); after the
//     batch it raises `columns written` in its control word;
//   * consumer: waits for that word, reads its 64 columns, then raises `columns read` in ITS control word -- the producer
//     checks it before it re-uses a slot (256 columns later; never waits in practice);
//   * the words and nothing else are reset between the levels (ka_hirschberg), while no strip runs.
// Only for levels whose items are all dealt statically (one item per wave: a producer's LDS region stays untouched
// until the level's barrier).  (A first version handed over column by column behind a tag per slot -- no batches, a strip
// started 63 + 4 instead of 63 + 64 columns behind the one above -- and cost 16 instructions per step on the two sides:
// 13 % fewer steps at 17 % more per step on the root task, profiles/r03b_variants_ho_per_column.log.)
#define KA_HO_RING KA_RING_BYTES                                // 256 slots x 16 B behind the column ring: [14336, 18432) of the wave's region
#define KA_HO_SLOTS 256
#define KA_LDS_HO_BACK 64                                       // control words, this many bytes below the wave regions: [wave] columns written, [8 + wave] columns read
static_assert(KA_TP || KA_HO_RING + KA_HO_SLOTS * 16 <= KA_WAVE_LDS, "hand-over ring outgrew the wave's LDS region");
static_assert(128 * KA_SP_STRIDE * 4 <= KA_WAVE_LDS && KA_REC_CHUNKS * KA_PK_RECS * 16 <= KA_WAVE_LDS, "seq-profile table / packed staging outgrew the wave's LDS region");

// SAVE (round 5, Hirschberg prefix reuse, ka_meetup.h): svrows != nullptr -- the pass leaves the row after `sv_rows` of its rows in
// svrows[0 .. ncols] (indexed like `rows`): the strip that holds that row has its owner lane store the fresh state every step
// (one exec-narrowed 12-byte store next to ~100 instructions; the strips of the pass that do not hold it do nothing).
template <int KIND, int NRES, int NB, int Q = 2, bool HO = false, bool SAVE = false>
__device__ __forceinline__ void ka_strip(const TaskShared& S, const int starta, const int enda, const int startb, const int endb,
                                         const float inj_a, const float inj_ga, const float inj_gb,
                                         const int dir, const int k, KaState* rows, int* prog,
                                         const int lane, char* wlds, const float* tss, const bool acq_agent, const bool rel_agent,
                                         long long* pslot = nullptr, const bool in_lds = false, const bool out_lds = false, int* ho_ctl_w = nullptr,
                                         KaState* svrows = nullptr, const int sv_rows = 0)
{
        const int ncols = endb - startb;
        const int mid = ((enda - starta) / 2) + starta;
        const int r0 = (dir == KA_FWD) ? starta : mid;
        const int r1 = (dir == KA_FWD) ? mid : enda;
        const int nrows = r1 - r0;
        const int Lb = __builtin_amdgcn_readfirstlane(S.Lb);
        const bool near_t = (dir == KA_FWD) ? (startb == 0) : (endb == Lb);
        const bool far_t = (dir == KA_FWD) ? (endb == Lb) : (startb == 0);

#define REC(v_) ((dir == KA_FWD) ? (startb + (v_)) : (endb + 1 - (v_)))
#define IDX(v_) ((dir == KA_FWD) ? (v_) : (ncols - (v_)))

#ifdef KA_TRACE_STRIP
        if (S.trace && blockIdx.x == 0 && lane == 0) { volatile int* tr0 = (volatile int*)S.trace; tr0[32 + (threadIdx.x >> 6)] = -(nrows * 1000 + ncols) - 1; tr0[40 + (threadIdx.x >> 6)] = dir * 100000 + starta * 100 + startb; tr0[48 + (threadIdx.x >> 6)] = -7; __threadfence_system(); }
#endif
        if (nrows == 0) {
                // only the "row -1" initialisation survives (aln_seqseq.c:40-58): a serial chain
                if (lane == 0) {
                        KaState ini;
                        ini.a = inj_a; ini.ga = inj_ga; ini.gb = inj_gb;
                        rows[IDX(0)] = ini;
                        for (int v = 1; v < ncols; ++v) {
                                float copen, cext, ctext;
                                col_terms<KIND>(S, REC(v), copen, cext, ctext);
                                const float g = near_t ? kmax(ini.ga, ini.a) + ctext : kmax(ini.ga + cext, ini.a + copen);
                                ini.a = -KA_F; ini.ga = g; ini.gb = -KA_F;
                                rows[IDX(v)] = ini;
                        }
                        ini.a = -KA_F; ini.ga = -KA_F; ini.gb = -KA_F;
                        rows[IDX(ncols)] = ini;
                }
#ifdef KA_TRACE_STRIP
                if (S.trace && blockIdx.x == 0 && lane == 0) { volatile int* tr0 = (volatile int*)S.trace; tr0[48 + (threadIdx.x >> 6)] = -8; __threadfence_system(); }
#endif
                return;
        }

        // (the operand pointers the event code needs, as SGPR pairs: see ka_uniform_ptr)
        const float* const p2u = (KIND == KA_PP) ? ka_uniform_ptr(S.p2) : nullptr;
        const uint8_t* const s2u = (KIND != KA_PP) ? ka_uniform_ptr(S.s2) : nullptr;
        int* const wdu = ka_uniform_ptr(S.watchdog);
        float* const sp_tbl = (float*)wlds;                           // seq-profile: this lane's two score rows
        const float m1 = ka_uniform_f(S.p1_mult), m2 = ka_uniform_f(S.p2_mult);
        ka_gfloat* const grows = (ka_gfloat*)rows;

        constexpr int SROWS = 64 * Q;
        const int u0 = k * SROWS;
        const int nr = min(SROWS, nrows - u0);
        const int nl = (Q == 2) ? ((nr + 1) >> 1) : nr;
        const int lastl = nl - 1;                                     // lane holding the strip's last row
        const bool last_is_b = (Q == 2) && (nr & 1) == 0;
        const bool first = (k == 0);
        const bool last_strip = (u0 + SROWS >= nrows);                // no strip below this one
        // SAVE: the pass row sv_rows - 1, if this strip holds it: its lane, and which of the lane's two rows it is
        const int sv_u = sv_rows - 1 - u0;
        const bool sv_on = SAVE && svrows != nullptr && sv_u >= 0 && sv_u < nr;
        const int sv_lane = (Q == 2) ? (sv_u >> 1) : sv_u;
        const bool sv_b = (Q == 2) && (sv_u & 1) != 0;
        ka_gfloat* const gsv = (ka_gfloat*)svrows;
        // Hand-over batches: a strip starts 63 columns (the lane skew) plus one batch behind the strip above it, and every
        // hand-over is an event step on both sides (a flush behind a release fence; a wait and a reload) that breaks the
        // branch-free step pairs.  64 columns per batch, inside a workgroup as well as across workgroups: 16-column batches
        // inside a workgroup (start delay 79 instead of 127 columns) measured 4 % slower on the 4096 x 400 tree and 12 %
        // slower on 4096 x 2000 nucleotides -- consumers run behind their producers anyway, so the start delay is not
        // what a pass waits for, the event steps are.  (Tried on top, no gain: a 16-column FIRST batch, fetching the next
        // batch one batch early, raising a batch's flag one batch late.)
        const int CBM = 63;                                           // consumer side: batch mask of the strip above
        const int PBM = 63;                                           // producer side
        // acq_agent: the strip above (k-1) runs in ANOTHER workgroup of the cluster; rel_agent: the strip below (k+1) does
        // (or may: work items beyond the statically dealt ones are pulled by whoever is free).  Neighbours inside this
        // workgroup hand over at workgroup scope: no L2 write-back, no L1 invalidate.
        const bool actB = (Q == 2) && 2 * lane + 1 < nr;
        const int uA = u0 + min(Q * lane, nr - 1);
        const int uB = u0 + min(Q * lane + 1, nr - 1);              // (Q = 1: row B does not exist; its operand loads are dead code)
        const int iA = (dir == KA_FWD) ? (r0 + uA) : (r1 - 1 - uA);
        const int iB = (dir == KA_FWD) ? (r0 + uB) : (r1 - 1 - uB);
        const int recA = iA + 1, recB = iB + 1;
        const int prevA = (dir == KA_FWD) ? recA - 1 : recA + 1;
        const int prevB = (dir == KA_FWD) ? recB - 1 : recB + 1;

        const unsigned wlds_u = (unsigned)(unsigned long long)wlds;
        auto ring_issue = [&](int nb) {
                // start the global->LDS copy of column batch nb (32 columns x 7 chunks)
                if (lane < KA_RING_BATCH && nb * KA_RING_BATCH <= ncols) {
                        const int vv = min(nb * KA_RING_BATCH + lane, ncols);
                        const float* g = p2u + ((long long)REC(vv) << 6) + 32;
                        // ring layout: chunk-major, 128 columns per chunk row (2048 B): column v of chunk ch at
                        // ch * 2048 + (v & 127) * 16 -- the read address is one AND + one shift-OR
                        char* dst = wlds + (nb & (KA_RING_SLOTS - 1)) * (KA_RING_BATCH * 16);
#pragma unroll
                        for (int ch = 0; ch < KA_REC_CHUNKS; ++ch)
                                if (ka_chunk_used<NRES>(ch))
                                        __builtin_amdgcn_global_load_lds((ka_glb_ptr)(g + 4 * ch), (ka_lds_ptr)(dst + ch * 2048), 16, 0, 0);
                }
        };
        // prime the column ring BEFORE the row operand is fetched: the two latencies (L2 / HBM) overlap
        if (KIND == KA_PP) {
                __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);              // earlier strip's ring traffic is done
                ring_issue(0);
                ring_issue(1);
        }
        // ---- stationary row operand ----
        float oA, eA, tA, oB, eB, tB, orpA, orpB;
        float2v p1v[Q == 2 ? NRES : 1];                               // Q = 2: the counts of residue c in rows (A, B)
        constexpr int NPAIR = NRES / 2;
        float2v p1p[Q == 1 ? (NPAIR > 0 ? NPAIR : 1) : 1];            // Q = 1: the counts of residues (2i, 2i+1) in row A
        float p1last = 0.0f;                                          // Q = 1, odd alphabets: the count of residue NRES-1
        int res1A = 0, res1B = 0;
        if (KIND == KA_SS) {
                oA = oB = -S.gpo; eA = eB = -S.gpe; tA = tB = -S.tgpe; orpA = orpB = -S.gpo;
                res1A = S.s1[iA] * KA_T_STRIDE; res1B = S.s1[iB] * KA_T_STRIDE;
        } else {
                const float* pA = S.p1 + ((long long)recA << 6);
                const float* pB = S.p1 + ((long long)recB << 6);
                oA = pA[55] * m1; eA = pA[56] * m1; tA = pA[57] * m1;
                oB = pB[55] * m1; eB = pB[56] * m1; tB = pB[57] * m1;
                orpA = S.p1[((long long)prevA << 6) + 55] * m1;
                orpB = S.p1[((long long)prevB << 6) + 55] * m1;
                if (KIND == KA_PP) {
                        // the NRES counts of a row are the head of its 256-B record: 16-B loads (a dword load per count made
                        // 2 x NRES load instructions of 64 scattered lines each at the start of every strip)
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
                } else {
                        // seq-profile: score = P1[row][32 + residue], residue varies per step ->
                        // keep this lane's score rows (two, or one with Q = 1) in its private LDS lines
                        float* tA_ = sp_tbl + (Q * lane) * KA_SP_STRIDE;
                        float4v sa[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) sa[i] = ((const float4v*)(pA + 32))[i];
#pragma unroll
                        for (int c = 0; c < KA_SP_FILL; ++c) tA_[c] = sa[c >> 2][c & 3];
                        if constexpr (Q == 2) {
                                float* tB_ = tA_ + KA_SP_STRIDE;
                                float4v sb[6];
#pragma unroll
                                for (int i = 0; i < 6; ++i) sb[i] = ((const float4v*)(pB + 32))[i];
#pragma unroll
                                for (int c = 0; c < KA_SP_FILL; ++c) tB_[c] = sb[c >> 2][c & 3];
                        }
                }
        }

        KaBonus<NB> bonA, bonB;
        if (NB) { bonA.load(S.ent, iA, dir); if (Q == 2) bonB.load(S.ent, iB, dir); }

        // sequence columns: the three column gap terms are task constants.  Read them from the LDS-resident
        // TaskShared ONCE -- inside the step the compiler re-loads them (ds_read + s_waitcnt) every step.
        float kc_open = 0.0f, kc_ext = 0.0f, kc_text = 0.0f;
        if (KIND != KA_PP) {
                col_terms<KIND>(S, 0, kc_open, kc_ext, kc_text);
                kc_open = ka_uniform_f(kc_open); kc_ext = ka_uniform_f(kc_ext); kc_text = ka_uniform_f(kc_text);
        }

        // cell states as plain scalars (a struct here ends up in scratch memory)
        float cAa = -KA_F, cAga = -KA_F, cAgb = -KA_F;
        float cBa = -KA_F, cBga = -KA_F, cBgb = -KA_F;
        float dga = -KA_F, dgga = -KA_F, dggb = -KA_F;
        float inia = inj_a, iniga = inj_ga, inigb = inj_gb;
        // boundary batch (strips > 0): lane 0 always holds the state of column t (rotated every step)
        float bta = -KA_F, btga = -KA_F, btgb = -KA_F;
        // output batch of the strip's last row.  FULL strips: a shift register fed at lane 63;
        // partial strips: lane j holds column 64*b + j
        float oba = -KA_F, obga = -KA_F, obgb = -KA_F;
        float copen_prev = 0.0f;
        int res2 = 0, resb = 0, resn = 0;
        float scA = 0.0f, scB = 0.0f;                                 // sequence columns: this step's two scores, looked up one step ahead
        float4v q[2][KA_REC_CHUNKS];                                  // column record: current / next step (ping-pong)

        // The ring reads are issued as inline asm so that the compiler does not track them: its
        // own s_waitcnt for this step's half of q (loaded one step ago) would otherwise also wait
        // for the loads just issued for the next step (lgkmcnt is a plain in-order counter), exposing
        // the full LDS latency every step.  ring_wait() is the matching manual wait; it takes the
        // registers as in/out operands so that no use can be scheduled above it.
        auto ring_read = [&](float4v* dstq, int vcol, auto& dep) {
                // (the wave's LDS region is 2048-B aligned: OR instead of ADD)
                const unsigned a = wlds_u | (((unsigned)vcol & 127u) << 4);
                if (!KA_UNTRACKED_READS) {
                        typedef const __attribute__((address_space(3))) float4v ka_l4;
                        ka_l4* const cr = (ka_l4*)(unsigned long)a;
                        asm volatile("" : "+v"(dep) : : "memory");      // (the loads stay behind `dep`, as the asm form's operand made them)
#pragma unroll
                        for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) if (ka_chunk_used<NRES>(ch)) dstq[ch] = cr[ch * 128];
                        return;
                }
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
                               "+v"(dep)                              // orders the loads after the value `dep` (see step())
                             : "v"(a)
                             : "memory");
        };
        auto ring_wait = [&](float4v* qq) {
                if (!KA_UNTRACKED_READS) {
                        // (a use of the record: the compiler's own wait lands here at the latest)
                        if (NRES <= 8) asm volatile("" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[5]), "+v"(qq[6]) : : "memory");
                        else asm volatile("" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]), "+v"(qq[4]), "+v"(qq[5]), "+v"(qq[6]) : : "memory");
                        return;
                }
                if (NRES <= 8) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[5]), "+v"(qq[6]) : : "memory");
                        return;
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]), "+v"(qq[4]), "+v"(qq[5]), "+v"(qq[6])
                             :
                             : "memory");
        };

        // ---- HO: batches through LDS (see the head of ka_strip) ----
        const unsigned ho_out_u = wlds_u + KA_HO_RING;                // my ring (I produce)
        const unsigned ho_in_u = wlds_u - KA_WAVE_LDS + KA_HO_RING;   // the ring of the wave before me (I consume)
        const unsigned ho_my_u = (unsigned)(unsigned long long)ho_ctl_w;        // [0] columns I have written; [+32 B] columns I have read
        auto lds_word = [&](const unsigned addr) -> int {
                int x;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(addr) : "memory");
                return __builtin_amdgcn_readfirstlane(x);
        };
        auto lds_wait_for = [&](const unsigned addr, const int need) {
                int spins = 0;
                while (lds_word(addr) < need) {
                        __builtin_amdgcn_s_sleep(1);
                        if (ka_spin_expired(wdu, ++spins, 1 << 22, 5)) break;
                }
        };

        if (KIND == KA_PP) {
                __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);              // the two ring batches issued at the top have landed
                { float2v nodep = {0.0f, 0.0f}; ring_read(q[0], min(max(-lane, 0), ncols), nodep); }
        } else {
                // Sequence columns run one step ahead: at step t the residue and the two score look-ups of
                // step t+1 are prepared (the LDS latency of the look-up and, every 64 steps, the L2 latency
                // of the residue batch would otherwise sit in every step of a lone wave).  resn = batch of
                // columns 64m+1+lane, fetched 64 steps before it is rotated in.
                resn = ((ka_gbytec*)s2u)[REC(min(1 + lane, ncols)) - 1];
                if (KIND == KA_SS) { scA = tss[res1A]; if (Q == 2) scB = tss[res1B]; }        // step 0: no lane is at a real column yet
                else { scA = sp_tbl[(Q * lane) * KA_SP_STRIDE]; if (Q == 2) scB = sp_tbl[(2 * lane + 1) * KA_SP_STRIDE]; }
        }

#ifdef KA_PROF
        if (pslot && lane == 0 && pslot[2] == 0) pslot[2] = __builtin_amdgcn_s_memtime();
#endif
        // One wavefront step.
        //   ST   : steady state, every active lane is strictly inside the column range (no edge cases)
        //   FULL : the strip has all 64 * Q rows (64 active lanes, last row = the last row of lane 63)
        //   P    : which half of q[] holds this step's column record (the other half receives the next)
        //   EV   : this step may carry a periodic event (ring batch hand-over, boundary / residue batch
        //          reload, flush of the output batch); EV = false steps are branch-free
        auto step = [&](const int t, auto st_tag, auto full_tag, auto par_tag, auto ev_tag, auto first_tag, auto lb_tag) {
                constexpr bool LASTB = decltype(lb_tag)::value;                // partial strips: the last row is its lane's row B (always so in full strips)
                constexpr bool FIRST = decltype(first_tag)::value;             // strip 0 of its pass: the row above is the pass's generated row -1
                constexpr bool ST = decltype(st_tag)::value;
                constexpr bool FULL = decltype(full_tag)::value;
                constexpr int P = decltype(par_tag)::value;
                constexpr bool EV = decltype(ev_tag)::value;
                const int v = t - lane;
#ifdef KA_PROF
                const long long tq0 = __builtin_amdgcn_s_memtime();
#endif

                // ---- column data for column v ----
                float copen, cext, ctext;
                if (KIND == KA_PP) {
                        ring_wait(q[P]);                       // this step's column record (issued one step ago)
#ifdef KA_PROF
                        if (EV && ST && pslot && lane == 0) pslot[256 + 0] += __builtin_amdgcn_s_memtime() - tq0;      // (head slot reused: top-of-step wait)
#endif
                        copen = q[P][5].w * m2; cext = q[P][6].x * m2; ctext = q[P][6].y * m2;
                } else {
                        copen = kc_open; cext = kc_ext; ctext = kc_text;
                }
                const float scA_cur = scA, scB_cur = scB;
                if (KIND != KA_PP) {
                        // prepare step t+1: lane 0 takes column t+1 from the batch, lanes > 0 their upper neighbour's residue
                        if (EV && (t & 63) == 0) {
                                resb = resn;
                                resn = ((ka_gbytec*)s2u)[REC(min(t + 65 + lane, ncols)) - 1];
                        } else {
                                resb = __builtin_amdgcn_update_dpp(resb, resb, 0x134, 0xf, 0xf, false);   // wave_rol:1
                        }
                        res2 = __builtin_amdgcn_update_dpp(resb, res2, 0x138, 0xf, 0xf, false);
                        if (KIND == KA_SS) { scA = tss[res1A + res2]; if (Q == 2) scB = tss[res1B + res2]; }
                        else { scA = sp_tbl[(Q * lane) * KA_SP_STRIDE + res2]; if (Q == 2) scB = sp_tbl[(2 * lane + 1) * KA_SP_STRIDE + res2]; }
                }

                // ---- state of the row above A: lane l-1's row B, lane 0 takes the boundary ----
                if (FIRST) {
                        // (!ST: a steady step is never step 0 -- without the hint the test, two selects and their operand moves
                        // sat in every step of every first strip, the strip that sets its pass's pace)
                        if (!ST && t == 0) {
                                inia = inj_a; iniga = inj_ga; inigb = inj_gb;
                        } else if (ST || t < ncols) {
                                // max(x, y) + c == max(x + c, y + c) bit for bit (rounding is monotonic): one select-free form
                                // for the terminal and the inner case instead of a branch per step
                                const float gx = near_t ? ctext : cext, gy = near_t ? ctext : copen;
                                const float g = kmax(iniga + gx, inia + gy);
                                inia = -KA_F; iniga = g; inigb = -KA_F;
                        } else {
                                inia = -KA_F; iniga = -KA_F; inigb = -KA_F;
                        }
                        bta = inia; btga = iniga; btgb = inigb;
                } else {
                        if (EV && (t & CBM) == 0) {
                                if (HO && in_lds) {
                                        if (t <= ncols) {
                                                // the strip above runs on the wave before this one: its batch sits in its LDS ring
                                                const int need = min(t + CBM + 1, ncols + 1);
                                                lds_wait_for(ho_my_u - 4, need);
                                                const unsigned a = ho_in_u + (((unsigned)min(t + lane, ncols) & (KA_HO_SLOTS - 1)) << 4);
                                                float2v x01;
                                                float x2;
                                                asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                                                             : "=&v"(x01), "=&v"(x2) : "v"(a) : "memory");
                                                bta = x01.x; btga = x01.y; btgb = x2;
                                                // ... and it may re-use the slots of everything below `need`
                                                if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(ho_my_u + 32), "v"(need) : "memory");
                                        }
                                } else if (t <= ncols) {
                                        // the previous strip must have published columns t .. t+CBM
                                        const int need = min(t + CBM + 1, ncols + 1);
                                        if (lane == 0) {
                                                // bounded spin: a stuck pipeline must surface as an error, never as a hung GPU
                                                int spins = 0;
                                                if (acq_agent) {
                                                        while (__hip_atomic_load(prog + (k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                                                                __builtin_amdgcn_s_sleep(4);
                                                                if (ka_spin_expired(wdu, ++spins, 1 << 22, 5)) break;
                                                        }
                                                } else {
                                                        while (__hip_atomic_load(prog + (k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                                                                __builtin_amdgcn_s_sleep(2);
                                                                if (ka_spin_expired(wdu, ++spins, 1 << 22, 5)) break;
                                                        }
                                                }
                                        }
                                        // the producer may be a wave of another workgroup (another CU) of the cluster
                                        if (acq_agent) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                                        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                        const ka_gfloat* r = grows + 3 * IDX(min(t + lane, ncols));
                                        bta = r[0]; btga = r[1]; btgb = r[2];
                                }
                        }
                }
                // lanes > 0 take lane l-1's row B, lane 0 the boundary state of column t (lane 0 of the batch).  The
                // batch is rotated for the NEXT step before it is used: its registers then die at the DPP that
                // splices lane 0 in (the `old` operand is tied to the destination) and no copy is needed.
                float upa, upga, upgb;
                // (the lane's last row: B, or A when a lane owns one row)
                const float lra = (Q == 2) ? cBa : cAa, lrga = (Q == 2) ? cBga : cAga, lrgb = (Q == 2) ? cBgb : cAgb;
                if (FIRST) {
                        upa = wave_shr1_old(bta, lra); upga = wave_shr1_old(btga, lrga); upgb = wave_shr1_old(btgb, lrgb);
                } else {
                        const float nbta = wave_rol1(bta), nbtga = wave_rol1(btga), nbtgb = wave_rol1(btgb);
                        upa = wave_shr1_old(bta, lra); upga = wave_shr1_old(btga, lrga); upgb = wave_shr1_old(btgb, lrgb);
                        bta = nbta; btga = nbtga; btgb = nbtgb;
                }

                if constexpr (Q == 2) {
                        // ---- the two cells of this lane ----
                        float2v acc;
                        acc.x = kmax3(dga, dgga + copen_prev, dggb + orpA);
                        acc.y = kmax3(cAa, cAga + copen_prev, cAgb + orpB);
                        if (KIND == KA_SS) {
                                acc.x += scA_cur;
                                acc.y += scB_cur;
                                if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); acc.x += bonA.template at<!ST>(jb); acc.y += bonB.template at<!ST>(jb); }
                        } else if (KIND == KA_SP) {
                                acc.x += scA_cur;
                                acc.y += scB_cur;
                                if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); acc.x += bonA.template at<!ST>(jb); acc.y += bonB.template at<!ST>(jb); }
                        } else {
                                // products one term ahead of the (dependent) sums: keeps a v_pk_mul between two
                                // v_pk_add of the chain instead of an s_nop
                                float2v prod;
                                prod = ka_mul_bcast<(NRES - 1) & 3>(p1v[NRES - 1], q[P][(NRES - 1) >> 2]);
        #pragma unroll
                                for (int c = NRES - 1; c >= 1; --c) {
                                        float2v nprod;
                                        switch ((c - 1) & 3) {                  // (compile-time after unrolling)
                                        case 0: nprod = ka_mul_bcast<0>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        case 1: nprod = ka_mul_bcast<1>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        case 2: nprod = ka_mul_bcast<2>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        default: nprod = ka_mul_bcast<3>(p1v[c - 1], q[P][(c - 1) >> 2]); break;
                                        }
                                        acc = acc + prod;
                                        prod = nprod;
                                }
                                acc = acc + prod;
                                if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); acc.x += bonA.template at<!ST>(jb); acc.y += bonB.template at<!ST>(jb); }
                                // Fetch the next step's column record into the other half of q.  The loads must
                                // stay AFTER the dot products: placed above them, the s_waitcnt for this step's
                                // half (loaded one step ago) also waits for the fresh loads and exposes the whole
                                // LDS latency every step.  sched_barrier pins the machine scheduler; the fake
                                // dependency on acc keeps the IR passes from sinking the chain below the loads.
                                {
                                __builtin_amdgcn_sched_barrier(0);
                                const int tn = t + 1;
#ifdef KA_PROF
                                const long long tq2 = __builtin_amdgcn_s_memtime();
#endif
                                if (EV && (tn & (KA_RING_BATCH - 1)) == 0) __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);      // batch tn/32 (issued >= 32 steps ago) has landed; the next one is issued at the end of the step
#ifdef KA_PROF
                                if (EV && ST && pslot && lane == 0) { const long long tq3 = __builtin_amdgcn_s_memtime(); pslot[256 + 1] += tq3 - tq2; pslot[256 + 4] += tq2 - tq0; }   // (tail slot: vmcnt wait; singles slot: start .. after the chain)
#endif
                                ring_read(q[1 - P], ST ? (v + 1) : min(max(v + 1, 0), ncols), acc);
                                __builtin_amdgcn_sched_barrier(0);
                                }
                        }
                        float nAa, nAga, nAgb, nBa, nBga, nBgb;
                        if (ST) {
                                nAa = acc.x;
                                nAga = kmax(cAga + cext, cAa + copen);
                                nAgb = kmax(upgb + eA, upa + oA);
                                nBa = acc.y;
                                nBga = kmax(cBga + cext, cBa + copen);
                                nBgb = kmax(nAgb + eB, nAa + oB);
                        } else {
                                const bool at0 = (v == 0), atN = (v == ncols);
                                const bool edge = at0 | atN;
                                const bool term = (at0 & near_t) | (atN & far_t);           // (bitwise: as short-circuit logic this became exec-masked branches in every edge step)
                                nAa = at0 ? -KA_F : acc.x;
                                nAga = edge ? -KA_F : kmax(cAga + cext, cAa + copen);
                                // (the terminal case `max(gb, a) + t` as the inner case with both penalties replaced by t:
                                // max(x, y) + c == max(x + c, y + c) bit for bit, and two selects on the operands instead of
                                // the exec-masked regions the compiler made of `term ? .. : ..` over the results; round 4)
                                nAgb = kmax(upgb + (term ? tA : eA), upa + (term ? tA : oA));
                                // B: the row above is A's fresh state
                                nBa = at0 ? -KA_F : acc.y;
                                nBga = edge ? -KA_F : kmax(cBga + cext, cBa + copen);
                                nBgb = kmax(nAgb + (term ? tB : eB), nAa + (term ? tB : oB));
                        }
                        // No predication on "this lane is inside its row/column range": state only flows DOWN the lanes
                        // (lane l -> l+1) and a lane's first real column (v = 0) rebuilds all six states from the lane above,
                        // so whatever lanes outside the range compute is never consumed by a lane inside it; the last-row
                        // collection below reads an active lane and is range-checked itself.
                        cAa = nAa; cAga = nAga; cAgb = nAgb;
                        cBa = nBa; cBga = nBga; cBgb = nBgb;
                        dga = upa; dgga = upga; dggb = upgb;
                        copen_prev = copen;

                } else {
                        // ---- the one cell of this lane (Q = 1) ----
                        float a1 = kmax3(dga, dgga + copen_prev, dggb + orpA);
                        if (KIND != KA_PP) {
                                a1 += scA_cur;
                                if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); a1 += bonA.template at<!ST>(jb); }
                        } else {
                                // residue NRES-1 first (aln_profileprofile.c:99-107 walks the non-zero counts downwards); a pair of
                                // residues per v_pk_mul_f32, products one pair ahead of the (dependent) sums
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
                                if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); a1 += bonA.template at<!ST>(jb); }
                                // next step's column record: after the chain, as in the two-row step
                                {
                                __builtin_amdgcn_sched_barrier(0);
                                const int tn = t + 1;
                                if (EV && (tn & (KA_RING_BATCH - 1)) == 0) __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);
                                ring_read(q[1 - P], ST ? (v + 1) : min(max(v + 1, 0), ncols), a1);
                                __builtin_amdgcn_sched_barrier(0);
                                }
                        }
                        float nAa, nAga, nAgb;
                        if (ST) {
                                nAa = a1;
                                nAga = kmax(cAga + cext, cAa + copen);
                                nAgb = kmax(upgb + eA, upa + oA);
                        } else {
                                const bool at0 = (v == 0), atN = (v == ncols);
                                const bool edge = at0 | atN;
                                const bool term = (at0 & near_t) | (atN & far_t);           // (bitwise: as short-circuit logic this became exec-masked branches in every edge step)
                                nAa = at0 ? -KA_F : a1;
                                nAga = edge ? -KA_F : kmax(cAga + cext, cAa + copen);
                                nAgb = kmax(upgb + (term ? tA : eA), upa + (term ? tA : oA));
                        }
                        cAa = nAa; cAga = nAga; cAgb = nAgb;
                        dga = upa; dgga = upga; dggb = upgb;
                        copen_prev = copen;
                }

                // ---- SAVE: the row the pass leaves for the sub-problem's child (prefix reuse) ----
                if constexpr (SAVE) {
                        if (sv_on) {
                                const int vs = t - sv_lane;                    // (wave-uniform: the owner's column)
                                if (ST || (vs >= 0 && vs <= ncols)) {
                                        if (lane == sv_lane) {
                                                ka_gfloat* w = gsv + 3 * IDX(vs);
                                                if (Q == 2 && sv_b) { w[0] = cBa; w[1] = cBga; w[2] = cBgb; }
                                                else { w[0] = cAa; w[1] = cAga; w[2] = cAgb; }
                                        }
                                }
                        }
                }
                // ---- collect the strip's last row and hand it on 64 columns at a time (through the row buffer, or through LDS) ----
                const int vL = t - lastl;
#ifdef KA_PROF
                const long long tfl0_outer = __builtin_amdgcn_s_memtime();
#endif
                if (ST || (vL >= 0 && vL <= ncols)) {
                        if (FULL) {
                                // shift register: lane 63 (the last row's owner) feeds its fresh state in
                                oba = wave_shl1_old(Q == 2 ? cBa : cAa, oba); obga = wave_shl1_old(Q == 2 ? cBga : cAga, obga); obgb = wave_shl1_old(Q == 2 ? cBgb : cAgb, obgb);
                        } else {
                                // partial strips: the collected batch rotates one lane up per step and the owner of the last row
                                // (lane lastl) drops its fresh state in -- lane (lastl + j) mod 64 holds column vL - j.  (Round 2
                                // broadcast the state with v_readlane and selected the receiving lane: ~20 % slower per step, and the
                                // partial strip is the LAST strip of its pass -- the one the pass waits for.)
                                // (the rotations first, as statements of their own: inside `own ? x : rotate()` they would run with
                                // the owner's lane switched off, and a DPP read of a disabled lane returns 0)
                                const bool own = (lane == lastl);
                                const float ra = wave_ror1(oba), rga = wave_ror1(obga), rgb = wave_ror1(obgb);
                                oba = own ? (LASTB ? cBa : cAa) : ra;
                                obga = own ? (LASTB ? cBga : cAga) : rga;
                                obgb = own ? (LASTB ? cBgb : cAgb) : rgb;
                        }
                        if (EV && ((vL & PBM) == 0 || vL == ncols)) {
                                // Batches end at columns 0, 64, 128, ...: in a full strip that is step t = 63 mod 64 -- the step of
                                // the column ring's event, whose wait for outstanding memory operations comes BEFORE these stores
                                // (round 2 flushed one step earlier: the ring event then waited for the stores to complete, every
                                // 64 steps, in every strip).
                                // FULL: lane i holds column vL - 63 + i; partial: lane i holds column vL - ((i - lastl) mod 64)
                                const int c0 = (vL == 0) ? 0 : (((vL - 1) & ~PBM) + 1);
                                const int col = FULL ? (vL - 63 + lane) : (vL - ((lane - lastl) & 63));
                                if (HO && out_lds) {
                                        // the strip below runs on the next wave: the batch goes to my LDS ring.  Its slots held the
                                        // columns 256 lower: the consumer must be through with them (it is, by ~190 columns).
                                        if (vL >= KA_HO_SLOTS) lds_wait_for(ho_my_u + 32 + 4, vL - (KA_HO_SLOTS - 1));
                                        if (col >= c0 && col <= vL) {
                                                const unsigned a = ho_out_u + (((unsigned)col & (KA_HO_SLOTS - 1)) << 4);
                                                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write_b32 %0, %3 offset:8" : : "v"(a), "v"(oba), "v"(obga), "v"(obgb) : "memory");
                                        }
                                        // (LDS executes a wave's instructions in order: who sees the count sees the batch)
                                        if (lane == 0) { const int pp = vL + 1; asm volatile("ds_write_b32 %0, %1" : : "v"(ho_my_u), "v"(pp) : "memory"); }
                                } else {
                                if (col >= c0 && col <= vL) {
                                        ka_gfloat* w = grows + 3 * IDX(col);
                                        w[0] = oba; w[1] = obga; w[2] = obgb;
                                }
                                // publish: the next strip (another wave of this workgroup, or of another
                                // workgroup of the cluster) may read them.  The LAST strip of a pass has no reader before the
                                // level's barrier (which orders its stores for the meetups): no fence, no wait for the stores, no flag.
                                if (last_strip) {
                                } else if (rel_agent) {
                                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                        if (lane == 0) __hip_atomic_store(prog + k, vL + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                } else {
                                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                        if (lane == 0) __hip_atomic_store(prog + k, vL + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                                }
                        }
                }
#ifdef KA_PROF
                if (EV && ST && pslot && lane == 0) pslot[256 + 7] += __builtin_amdgcn_s_memtime() - tfl0_outer;
#endif
                // the column ring's next batch goes out LAST: a release fence of the flush above waits for everything outstanding,
                // and these loads take a microsecond
#ifdef KA_PROF
                const long long tri0 = __builtin_amdgcn_s_memtime();
#endif
                if (KIND == KA_PP && EV && ((t + 1) & (KA_RING_BATCH - 1)) == 0) ring_issue(((t + 1) >> 5) + 1);
#ifdef KA_PROF
                if (EV && ST && pslot && lane == 0) pslot[256 + 6] += __builtin_amdgcn_s_memtime() - tri0;
#endif
        };

        // q[t & 1] holds step t's column record: a step's parity is its t's, so a single step between two
        // pairs needs no copy of the 28 record registers (round 2 started every phase on half 0 and copied after odd steps
        // and after every event step: four copies, each behind an exposed LDS wait, per 64 steps)
        // The head (lanes still entering at column 0) and the tail (lanes leaving at the last column) of a strip: the edge form of
        // the step (selects for the first / last column, clamped column indices).  Its periodic events fall on the same known
        // steps as in the steady phase, plus the strip's last step (the final flush) -- everything else runs WITHOUT the event
        // tests (round 2 ran all 128 edge steps of a strip as event steps, ~900 cycles each; the head of a strip is what its
        // consumer's start waits for, at every hand-over of every pass).
        auto run = [&](int& t, const int tend, auto st_tag, auto full_tag, auto first_tag, auto lb_tag) {
                constexpr bool FIRST = decltype(first_tag)::value;
                const int e4 = ncols + lastl;                                 // the last step: vL == ncols
                while (t < tend) {
                        const int e1 = t | 31;
                        const int e2 = (KIND == KA_PP && FIRST) ? 0x7fffffff : ((t + CBM) & ~CBM);
                        const int e3 = t + ((lastl - t) & PBM);
                        const int e5 = (t <= e4) ? e4 : 0x7fffffff;
                        const int ev = min(min(e1, e2), min(e3, e5));
                        const int fend = min(ev, tend);
                        if ((t & 1) && t < fend) {
                                step(t, st_tag, full_tag, std::integral_constant<int, 1>(), std::false_type(), first_tag, lb_tag);
                                ++t;
                        }
                        for (; t + 1 < fend; t += 2) {
                                step(t, st_tag, full_tag, std::integral_constant<int, 0>(), std::false_type(), first_tag, lb_tag);
                                step(t + 1, st_tag, full_tag, std::integral_constant<int, 1>(), std::false_type(), first_tag, lb_tag);
                        }
                        if (t < fend) {
                                step(t, st_tag, full_tag, std::integral_constant<int, 0>(), std::false_type(), first_tag, lb_tag);
                                ++t;
                        }
                        if (t < tend && t == ev) {
                                if (t & 1) step(t, st_tag, full_tag, std::integral_constant<int, 1>(), std::true_type(), first_tag, lb_tag);
                                else step(t, st_tag, full_tag, std::integral_constant<int, 0>(), std::true_type(), first_tag, lb_tag);
                                ++t;
                        }
                }
        };
        // steady state: the periodic events fall on known steps (ring hand-over at t = 31 mod 32 -- in full strips also the
        // output flush, at t = lastl mod 64 --, batch reloads at t = 0 mod 64); everything between two event steps runs
        // as branch-free step pairs.  (HO changes what a reload and a flush DO -- LDS instead of HBM -- not when they happen.)
        auto run_steady = [&](int& t, const int tend, auto full_tag, auto first_tag, auto lb_tag) {
                constexpr bool FIRST = decltype(first_tag)::value;
                while (t < tend) {
                        const int e1 = t | 31;
                        // (profile columns, first strip of the pass: nothing is reloaded every 64 steps -- no event)
                        const int e2 = (KIND == KA_PP && FIRST) ? 0x7fffffff : ((t + CBM) & ~CBM);
                        const int e3 = t + ((lastl - t) & PBM);
                        const int ev = min(e1, min(e2, e3));
                        const int fend = min(ev, tend);
                        if ((t & 1) && t < fend) {
                                step(t, std::true_type(), full_tag, std::integral_constant<int, 1>(), std::false_type(), first_tag, lb_tag);
                                ++t;
                        }
#ifdef KA_PROF
                        const long long tpair0 = __builtin_amdgcn_s_memtime();
                        const int t_in = t;
#endif
                        for (; t + 1 < fend; t += 2) {
                                step(t, std::true_type(), full_tag, std::integral_constant<int, 0>(), std::false_type(), first_tag, lb_tag);
                                step(t + 1, std::true_type(), full_tag, std::integral_constant<int, 1>(), std::false_type(), first_tag, lb_tag);
                        }
#ifdef KA_PROF
                        // cycles inside the branch-free step pairs / steps taken there (the rest of the strip: events, single steps, edges)
                        if (pslot && lane == 0) { pslot[6] += __builtin_amdgcn_s_memtime() - tpair0; pslot[7] += t - t_in; }
#endif
                        if (t < fend) {
                                step(t, std::true_type(), full_tag, std::integral_constant<int, 0>(), std::false_type(), first_tag, lb_tag);
                                ++t;
                        }
#ifdef KA_PROF
                        // KA_PROF builds, slots [256 ..] of the wave's profile record (tools/strip_phases.py): per event step of the
                        // steady phase -- [2] cycles, [3] count, [0] wait for the column record, [4] start .. end of the dot products,
                        // [1] wait for the ring batch, [7] collection + flush, [6] issue of the next ring batch
                        const long long tev0 = __builtin_amdgcn_s_memtime();
                        const int ev_in = t;
#endif
                        if (t < tend && t == ev) {
                                if (t & 1) step(t, std::true_type(), full_tag, std::integral_constant<int, 1>(), std::true_type(), first_tag, lb_tag);
                                else step(t, std::true_type(), full_tag, std::integral_constant<int, 0>(), std::true_type(), first_tag, lb_tag);
                                ++t;
                        }
#ifdef KA_PROF
                        if (pslot && lane == 0) { pslot[256 + 2] += __builtin_amdgcn_s_memtime() - tev0; pslot[256 + 3] += t - ev_in; }
#endif
                }
        };

        const int nsteps = ncols + nl;                                // t = 0 .. ncols + nl - 1
        const int t_steady0 = min(nl, nsteps);                        // first step with every active lane at v >= 1
        const int t_steady1 = ncols;                                  // one past the last step with every lane at v <= ncols-1
        auto phases = [&](auto full_tag, auto first_tag, auto lb_tag) {
                int t = 0;
                run(t, t_steady0, std::false_type(), full_tag, first_tag, lb_tag);
                run_steady(t, t_steady1, full_tag, first_tag, lb_tag);
                run(t, nsteps, std::false_type(), full_tag, first_tag, lb_tag);
        };
        // (which row of its lane a partial strip's last row is: two instances, so that the step does not select)
        auto go = [&](auto full_tag, auto first_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                if constexpr (FULL || Q == 1) phases(full_tag, first_tag, std::integral_constant<bool, Q == 2>());
                else { if (last_is_b) phases(full_tag, first_tag, std::true_type()); else phases(full_tag, first_tag, std::false_type()); }
        };
        if (nr == SROWS) {
                if (first) go(std::true_type(), std::true_type());
                else go(std::true_type(), std::false_type());
        } else {
                if (first) go(std::false_type(), std::true_type());
                else go(std::false_type(), std::false_type());
        }
#undef REC
#undef IDX
}

// ------------------------------------------------------------------------------------------
// Packed passes: below the top few recursion levels a task has hundreds of tiny sub-problems
// (a few rows x a few dozen columns).  One wave per pass would be almost pure latency, so
// small passes (<= 2*SLOT rows) are packed SLOT lanes apiece, 64/SLOT passes per wave, and
// stepped in lock-step.  Same cell code as ka_strip, but everything that is wave-uniform
// there (window, direction, terminal flags, row-buffer base) is per-lane here, the column
// record comes straight from L2 one step ahead instead of the LDS ring, and the slot's last
// lane writes the last row state by state.
// ------------------------------------------------------------------------------------------
template <int KIND, int NRES, int SLOT, int NB>
__device__ __forceinline__ void ka_packed(const TaskShared& S, const KaSub* qc, const int2* pack, const int nslots,
                                          const int job, const int lane, const float* tss, char* wlds,
                                          const int nreg = 1, const int reg_stride = 0)
{
        constexpr int SPW = 64 / SLOT;                                // slots per wave
        const int slot = job * SPW + lane / SLOT;
        const int ls = lane % SLOT;                                   // lane within the slot
        const bool live = slot < nslots;
        const int2 d = pack[live ? slot : 0];
        const KaSub* sp = qc + d.x;
        const int dir = d.y;
        const int starta = sp->starta, enda = sp->enda, startb = sp->startb, endb = sp->endb;
        const int ncols = endb - startb;
        const int mid = ((enda - starta) / 2) + starta;
        const int r0 = (dir == KA_FWD) ? starta : mid;
        const int r1 = (dir == KA_FWD) ? mid : enda;
        const int nrows = r1 - r0;                                    // 0 .. 2*SLOT
        const int nl = (nrows + 1) >> 1;                              // active lanes of the slot (0 for an init-only pass)
        const bool near_t = (dir == KA_FWD) ? (startb == 0) : (endb == S.Lb);
        const bool far_t = (dir == KA_FWD) ? (endb == S.Lb) : (startb == 0);
        const float inj_a = (dir == KA_FWD) ? sp->fin.a : sp->bin.a;
        const float inj_ga = (dir == KA_FWD) ? sp->fin.ga : sp->bin.ga;
        const float inj_gb = (dir == KA_FWD) ? sp->fin.gb : sp->bin.gb;
        ka_gfloat* const grows = (ka_gfloat*)(((dir == KA_FWD) ? S.fbuf : S.bbuf) + sp->roff);

#define REC(v_) ((dir == KA_FWD) ? (startb + (v_)) : (endb + 1 - (v_)))
#define IDX(v_) ((dir == KA_FWD) ? (v_) : (ncols - (v_)))

        const bool actB = live && (2 * ls + 1 < nrows);
        const bool writer = live && (ls == (nl > 0 ? nl - 1 : 0));    // owner of the pass's last row (or of the init row)
        const bool last_is_b = (nrows & 1) == 0;
        const int uA = min(2 * ls, max(nrows - 1, 0));
        const int uB = min(2 * ls + 1, max(nrows - 1, 0));
        const int iA = (dir == KA_FWD) ? (r0 + uA) : (r1 - 1 - uA);
        const int iB = (dir == KA_FWD) ? (r0 + uB) : (r1 - 1 - uB);
        const int recA = iA + 1, recB = iB + 1;
        const int prevA = (dir == KA_FWD) ? recA - 1 : recA + 1;
        const int prevB = (dir == KA_FWD) ? recB - 1 : recB + 1;
        const float m1 = S.p1_mult, m2 = S.p2_mult;

        float oA, eA, tA, oB, eB, tB, orpA, orpB;
        float2v p1v[NRES];
        int res1A = 0, res1B = 0;
        const float* pA = nullptr;
        const float* pB = nullptr;
        if (KIND == KA_SS) {
                oA = oB = -S.gpo; eA = eB = -S.gpe; tA = tB = -S.tgpe; orpA = orpB = -S.gpo;
                res1A = S.s1[min(iA, S.La - 1)] * KA_T_STRIDE; res1B = S.s1[min(iB, S.La - 1)] * KA_T_STRIDE;
        } else {
                pA = S.p1 + ((long long)min(recA, S.La + 1) << 6);
                pB = S.p1 + ((long long)min(recB, S.La + 1) << 6);
                oA = pA[55] * m1; eA = pA[56] * m1; tA = pA[57] * m1;
                oB = pB[55] * m1; eB = pB[56] * m1; tB = pB[57] * m1;
                orpA = S.p1[((long long)min(prevA, S.La + 1) << 6) + 55] * m1;
                orpB = S.p1[((long long)min(prevB, S.La + 1) << 6) + 55] * m1;
                if (KIND == KA_PP) {
                        constexpr int NV = (NRES + 3) / 4;                // 16-B loads, as in ka_strip
                        float4v va[NV], vb[NV];
#pragma unroll
                        for (int i = 0; i < NV; ++i) { va[i] = ((const float4v*)pA)[i]; vb[i] = ((const float4v*)pB)[i]; }
#pragma unroll
                        for (int c = 0; c < NRES; ++c) {
                                p1v[c].x = va[c >> 2][c & 3];
                                p1v[c].y = actB ? vb[c >> 2][c & 3] : 0.0f;
                        }
                } else {
                        // seq-profile: score = P1[row][32 + residue] with a different residue every step ->
                        // this lane's two score rows go to its private LDS lines (as in ka_strip)
                        float* tA_ = (float*)wlds + (2 * lane) * KA_SP_STRIDE;
                        float* tB_ = tA_ + KA_SP_STRIDE;
#pragma unroll
                        for (int c = 0; c < KA_SP_FILL; ++c) { tA_[c] = pA[32 + c]; tB_[c] = pB[32 + c]; }
                }
        }

        KaBonus<NB> bonA, bonB;
        if (NB) { bonA.load(S.ent, min(max(iA, 0), S.La - 1), dir); bonB.load(S.ent, min(max(iB, 0), S.La - 1), dir); }
        float kc_open = 0.0f, kc_ext = 0.0f, kc_text = 0.0f;           // see ka_strip
        if (KIND != KA_PP) {
                col_terms<KIND>(S, 0, kc_open, kc_ext, kc_text);
                kc_open = ka_uniform_f(kc_open); kc_ext = ka_uniform_f(kc_ext); kc_text = ka_uniform_f(kc_text);
        }

        float cAa = -KA_F, cAga = -KA_F, cAgb = -KA_F;
        float cBa = -KA_F, cBga = -KA_F, cBgb = -KA_F;
        float dga = -KA_F, dgga = -KA_F, dggb = -KA_F;
        float inia = inj_a, iniga = inj_ga, inigb = inj_gb;
        float copen_prev = 0.0f;
        float4v q[2][KA_REC_CHUNKS];
        int resq[4] = {0, 0, 0, 0};                                   // sequence columns: residues 3 steps ahead (L2 latency)

        // Profile-profile: when the column records of all the job's slots fit into LDS, stage them once (every
        // slot's lanes copy their slot's columns) and read LDS per step; otherwise stream them from L2 one step
        // ahead (a packed step then costs ~2000 cycles instead of ~800: the per-lane gathers cannot be run far
        // enough ahead without spilling).  The wave's own region holds 128 records; on levels that keep only
        // a half / a quarter of the workgroup's waves busy the caller lends it the idle waves' regions as well
        // (nreg regions, reg_stride bytes apart).
        int lds_base = 0;                                             // first staged record of this lane's slot
        bool staged = false;
        if (KIND == KA_PP && wlds != nullptr) {
                const int cnt = live ? (ncols + 1) : 0;               // identical in all lanes of a slot
                int total = 0;
#pragma unroll
                for (int sidx = 0; sidx < SPW; ++sidx) {
                        const int c = __shfl(cnt, sidx * SLOT, 64);
                        if (sidx < lane / SLOT) lds_base += c;
                        total += c;
                }
                staged = total <= nreg * KA_PK_RECS;                      // 128 records per region, 64 in the throughput kernel (wave-uniform)
                if (staged) {
                        for (int vv = ls; vv < cnt; vv += SLOT) {
                                ka_gfloat4c* g = (ka_gfloat4c*)(S.p2 + ((long long)REC(vv) << 6) + 32);
                                const int idx = lds_base + vv;
                                char* dst = wlds + (idx / KA_PK_RECS) * reg_stride + (idx % KA_PK_RECS) * 16;
#pragma unroll
                                for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) if (ka_chunk_used<NRES>(ch)) *(float4v*)(dst + ch * (KA_PK_RECS * 16)) = g[ch];
                        }
                        // written and read by different lanes of this wave only
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_s_waitcnt(0);
                }
        }
        // (STG is a template tag, not a run-time test: with both sources in one step the loaded values meet in a phi
        // and the compiler waits for the global loads right where they are issued -- no prefetch at all)
        auto fetch = [&](float4v* dstq, int& dstres, int vcol, auto stg_tag) {
                constexpr bool STG = decltype(stg_tag)::value;
                // column operand for column counter vcol (clamped)
                const int vv = min(max(vcol, 0), ncols);
                if (KIND == KA_PP) {
                        if (STG) {
                                const int idx = lds_base + vv;
                                const char* src = wlds + (idx / KA_PK_RECS) * reg_stride + (idx % KA_PK_RECS) * 16;
#pragma unroll
                                for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) if (ka_chunk_used<NRES>(ch)) dstq[ch] = *(const float4v*)(src + ch * (KA_PK_RECS * 16));
                        } else {
                                ka_gfloat4c* g = (ka_gfloat4c*)(S.p2 + ((long long)REC(vv) << 6) + 32);
#pragma unroll
                                for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) if (ka_chunk_used<NRES>(ch)) dstq[ch] = g[ch];
                        }
                } else {
                        dstres = ((ka_gbytec*)S.s2)[min(max(REC(max(vv, 1)) - 1, 0), S.Lb - 1)];
                }
        };
        if (staged) fetch(q[0], resq[0], -ls, std::true_type()); else fetch(q[0], resq[0], -ls, std::false_type());
        if (KIND != KA_PP) { fetch(q[0], resq[1], 1 - ls, std::false_type()); fetch(q[0], resq[2], 2 - ls, std::false_type()); }

        int nsteps = live ? (ncols + max(nl, 1)) : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, off, 64));

        // P4 = t mod 4: register-ring slot of this step's residue (sequence columns, fetched 3 steps ahead);
        // profile columns ping-pong between the two halves of q
        auto step = [&](const int t, auto par_tag, auto stg_tag) {
                constexpr int P4 = decltype(par_tag)::value;
                constexpr int P = P4 & 1;
                const int v = t - ls;
                const bool vin = live && (v >= 0) && (v <= ncols);
                if (KIND == KA_PP) fetch(q[1 - P], resq[0], v + 1, stg_tag);
                else fetch(q[0], resq[(P4 + 3) & 3], v + 3, stg_tag);

                float copen, cext, ctext;
                if (KIND == KA_PP) { copen = q[P][5].w * m2; cext = q[P][6].x * m2; ctext = q[P][6].y * m2; }
                else { copen = kc_open; cext = kc_ext; ctext = kc_text; }

                // "row -1" of the pass, generated by the slot's first lane (v == t there)
                // (selects, not branches: the three cases differ per lane; max(x, y) + c == max(x + c, y + c) bit for bit)
                {
                        const float gx = near_t ? ctext : cext, gy = near_t ? ctext : copen;
                        const float g = kmax(iniga + gx, inia + gy);
                        const bool v0 = (v == 0), vmid = (v < ncols);
                        inia = v0 ? inj_a : -KA_F;
                        iniga = v0 ? inj_ga : (vmid ? g : -KA_F);
                        inigb = v0 ? inj_gb : -KA_F;
                }
                float upa = wave_shr1(cBa), upga = wave_shr1(cBga), upgb = wave_shr1(cBgb);
                if (ls == 0) { upa = inia; upga = iniga; upgb = inigb; }

                float2v acc;
                acc.x = kmax3(dga, dgga + copen_prev, dggb + orpA);
                acc.y = kmax3(cAa, cAga + copen_prev, cAgb + orpB);
                if (KIND == KA_SS) {
                        acc.x += tss[res1A + resq[P4]];
                        acc.y += tss[res1B + resq[P4]];
                } else if (KIND == KA_SP) {
                        acc.x += ((const float*)wlds)[(2 * lane) * KA_SP_STRIDE + resq[P4]];
                        acc.y += ((const float*)wlds)[(2 * lane + 1) * KA_SP_STRIDE + resq[P4]];
                } else {
#pragma unroll
                        for (int c = NRES - 1; c >= 0; --c) {
                                const float sc = q[P][c >> 2][c & 3];
                                float2v w; w.x = sc; w.y = sc;
                                acc = acc + p1v[c] * w;
                        }
                }
                if (NB) { const int jb = (dir == KA_FWD) ? (startb + v) : (endb - v); acc.x += bonA.template at<true>(jb); acc.y += bonB.template at<true>(jb); }
                const bool at0 = (v == 0), atN = (v == ncols);
                const bool edge = at0 | atN;
                const bool term = (at0 & near_t) | (atN & far_t);           // (bitwise: as short-circuit logic this became exec-masked branches in every edge step)
                const float nAa = at0 ? -KA_F : acc.x;
                const float nAga = edge ? -KA_F : kmax(cAga + cext, cAa + copen);
                // (selects on the penalties, not on the results: see ka_strip)
                const float nAgb = kmax(upgb + (term ? tA : eA), upa + (term ? tA : oA));
                const float nBa = at0 ? -KA_F : acc.y;
                const float nBga = edge ? -KA_F : kmax(cBga + cext, cBa + copen);
                const float nBgb = kmax(nAgb + (term ? tB : eB), nAa + (term ? tB : oB));
                // (no range predication, as in ka_strip: state flows down the lanes of a slot only, lane 0 of every
                // slot takes the generated row, and v == 0 rebuilds all six states)
                cAa = nAa; cAga = nAga; cAgb = nAgb;
                cBa = nBa; cBga = nBga; cBgb = nBgb;
                dga = upa; dgga = upga; dggb = upgb;
                copen_prev = copen;
                if (vin && writer) {
                        ka_gfloat* w = grows + 3 * IDX(v);
                        if (nrows == 0) { w[0] = inia; w[1] = iniga; w[2] = inigb; }
                        else {
                                w[0] = last_is_b ? cBa : cAa;
                                w[1] = last_is_b ? cBga : cAga;
                                w[2] = last_is_b ? cBgb : cAgb;
                        }
                }
        };
        auto loop = [&](auto stg_tag) {
                int t = 0;
                for (; t + 3 < nsteps; t += 4) {
                        step(t, std::integral_constant<int, 0>(), stg_tag);
                        step(t + 1, std::integral_constant<int, 1>(), stg_tag);
                        step(t + 2, std::integral_constant<int, 2>(), stg_tag);
                        step(t + 3, std::integral_constant<int, 3>(), stg_tag);
                }
                if (t < nsteps) step(t, std::integral_constant<int, 0>(), stg_tag);
                if (t + 1 < nsteps) step(t + 1, std::integral_constant<int, 1>(), stg_tag);
                if (t + 2 < nsteps) step(t + 2, std::integral_constant<int, 2>(), stg_tag);
        };
        if (KIND == KA_PP && staged) loop(std::true_type()); else loop(std::false_type());
#undef REC
#undef IDX
}
