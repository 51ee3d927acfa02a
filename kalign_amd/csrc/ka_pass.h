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
//   * the strip's last row is collected in registers (one column per lane) and written to the
//     row buffer 64 columns at a time (coalesced), then published.
//
// All arithmetic is binary32 in the reference's order, no contraction (see ka_kernels.hip).
#pragma once

#include <type_traits>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define KA_STRIP_ROWS 128
#define KA_RING_BATCH 32
#define KA_RING_SLOTS 4
#define KA_REC_CHUNKS 7                                         // 7 x 16 B = profile fields [32..59]
#define KA_SLOT_BYTES (KA_REC_CHUNKS * KA_RING_BATCH * 16)      // 3584
#define KA_WAVE_LDS (KA_RING_SLOTS * KA_SLOT_BYTES)             // 14336 B per wave
#define KA_SP_STRIDE 25                                         // floats per row in the seq-profile score table
#define KA_T_STRIDE 24                                          // floats per row in the seq-seq score table

#define KA_WAIT_VM0 0x0F70                                      // s_waitcnt vmcnt(0) only (gfx9 encoding)

typedef __attribute__((address_space(3))) void* ka_lds_ptr;
typedef const __attribute__((address_space(1))) void* ka_glb_ptr;
typedef __attribute__((address_space(1))) float ka_gfloat;

__device__ __forceinline__ int wave_shr1_i(int x)
{
        return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ int ka_strips_of(int nrows) { return nrows <= 0 ? 1 : (nrows + KA_STRIP_ROWS - 1) / KA_STRIP_ROWS; }

template <int KIND, int NRES>
__device__ __forceinline__ void ka_strip(const TaskShared& S, const int starta, const int enda, const int startb, const int endb,
                                         const float inj_a, const float inj_ga, const float inj_gb,
                                         const int dir, const int k, KaState* rows, int* prog,
                                         const int lane, char* wlds, const float* tss)
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

        float* const sp_tbl = (float*)wlds;                           // seq-profile: this lane's two score rows
        const float m1 = ka_uniform_f(S.p1_mult), m2 = ka_uniform_f(S.p2_mult);
        ka_gfloat* const grows = (ka_gfloat*)rows;

        const int u0 = k * KA_STRIP_ROWS;
        const int nr = min(KA_STRIP_ROWS, nrows - u0);
        const int nl = (nr + 1) >> 1;
        const int lastl = nl - 1;                                     // lane holding the strip's last row
        const bool last_is_b = (nr & 1) == 0;
        const bool first = (k == 0);
        const bool actA = 2 * lane < nr;
        const bool actB = 2 * lane + 1 < nr;
        const int uA = u0 + min(2 * lane, nr - 1);
        const int uB = u0 + min(2 * lane + 1, nr - 1);
        const int iA = (dir == KA_FWD) ? (r0 + uA) : (r1 - 1 - uA);
        const int iB = (dir == KA_FWD) ? (r0 + uB) : (r1 - 1 - uB);
        const int recA = iA + 1, recB = iB + 1;
        const int prevA = (dir == KA_FWD) ? recA - 1 : recA + 1;
        const int prevB = (dir == KA_FWD) ? recB - 1 : recB + 1;

        // ---- stationary row operand ----
        float oA, eA, tA, oB, eB, tB, orpA, orpB;
        float2v p1v[NRES];
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
#pragma unroll
                        for (int c = 0; c < NRES; ++c) {
                                p1v[c].x = pA[c];
                                p1v[c].y = actB ? pB[c] : 0.0f;
                        }
                } else {
                        // seq-profile: score = P1[row][32 + residue], residue varies per step ->
                        // keep this lane's two score rows in its private LDS lines
                        float* tA_ = sp_tbl + (2 * lane) * KA_SP_STRIDE;
                        float* tB_ = tA_ + KA_SP_STRIDE;
#pragma unroll
                        for (int c = 0; c < 23; ++c) { tA_[c] = pA[32 + c]; tB_[c] = pB[32 + c]; }
                }
        }

        // cell states as plain scalars (a struct here ends up in scratch memory)
        float cAa = -KA_F, cAga = -KA_F, cAgb = -KA_F;
        float cBa = -KA_F, cBga = -KA_F, cBgb = -KA_F;
        float dga = -KA_F, dgga = -KA_F, dggb = -KA_F;
        float inia = inj_a, iniga = inj_ga, inigb = inj_gb;
        float bta = -KA_F, btga = -KA_F, btgb = -KA_F;                // boundary batch (strips > 0)
        float oba = -KA_F, obga = -KA_F, obgb = -KA_F;                // output batch: lane j holds column 64*b + j
        float copen_prev = 0.0f;
        int res2 = 0, resb = 0;
        float4v q[KA_REC_CHUNKS];

        auto ring_issue = [&](int nb) {
                // start the global->LDS copy of column batch nb (32 columns x 7 chunks)
                if (lane < KA_RING_BATCH && nb * KA_RING_BATCH <= ncols) {
                        const int vv = min(nb * KA_RING_BATCH + lane, ncols);
                        const float* g = S.p2 + ((long long)REC(vv) << 6) + 32;
                        char* dst = wlds + (nb & (KA_RING_SLOTS - 1)) * KA_SLOT_BYTES;
#pragma unroll
                        for (int ch = 0; ch < KA_REC_CHUNKS; ++ch)
                                __builtin_amdgcn_global_load_lds((ka_glb_ptr)(g + 4 * ch), (ka_lds_ptr)(dst + ch * 512), 16, 0, 0);
                }
        };
        auto ring_read = [&](int vcol) {
                const char* src = wlds + ((vcol >> 5) & (KA_RING_SLOTS - 1)) * KA_SLOT_BYTES + (vcol & 31) * 16;
#pragma unroll
                for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) q[ch] = *(const float4v*)(src + ch * 512);
        };

        if (KIND == KA_PP) {
                __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);              // earlier strip's ring traffic is done
                ring_issue(0);
                ring_issue(1);
                __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);
                ring_read(min(max(-lane, 0), ncols));
        }

        // one wavefront step; STEADY = every active lane is strictly inside the column range
        auto step = [&](const int t, auto steady_tag) {
                constexpr bool ST = decltype(steady_tag)::value;
                const int v = t - lane;
                const bool inr = ST ? actA : ((v >= 0) && (v <= ncols) && actA);

                // ---- column data for column v ----
                float copen, cext, ctext;
                if (KIND == KA_PP) {
                        copen = q[5].w * m2; cext = q[6].x * m2; ctext = q[6].y * m2;
                } else {
                        col_terms<KIND>(S, 0, copen, cext, ctext);
                        if ((t & 63) == 0) {
                                const int vv = min(max(t + lane, 1), ncols);
                                resb = S.s2[REC(vv) - 1];
                        }
                        res2 = wave_shr1_i(res2);
                        const int r0_ = __builtin_amdgcn_readlane(resb, t & 63);
                        if (lane == 0) res2 = r0_;
                }

                // ---- boundary state for lane 0 at column t ----
                float b0a, b0ga, b0gb;
                if (first) {
                        if (t == 0) {
                                inia = inj_a; iniga = inj_ga; inigb = inj_gb;
                        } else if (ST || t < ncols) {
                                const float g = near_t ? kmax(iniga, inia) + ctext : kmax(iniga + cext, inia + copen);
                                inia = -KA_F; iniga = g; inigb = -KA_F;
                        } else {
                                inia = -KA_F; iniga = -KA_F; inigb = -KA_F;
                        }
                        b0a = inia; b0ga = iniga; b0gb = inigb;
                } else {
                        if ((t & 63) == 0 && t <= ncols) {
                                // the previous strip must have published columns t .. t+63
                                const int need = min(t + 64, ncols + 1);
                                if (lane == 0) {
                                        // bounded spin: a stuck pipeline must surface as an error, never as a hung GPU
                                        int spins = 0;
                                        while (__hip_atomic_load(prog + (k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                                                __builtin_amdgcn_s_sleep(2);
                                                if (++spins > (1 << 22)) { *S.watchdog = 5; break; }
                                        }
                                }
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                const ka_gfloat* r = grows + 3 * IDX(min(t + lane, ncols));
                                bta = r[0]; btga = r[1]; btgb = r[2];
                        }
                        b0a = lane_bcast(bta, t & 63);
                        b0ga = lane_bcast(btga, t & 63);
                        b0gb = lane_bcast(btgb, t & 63);
                }

                float upa = wave_shr1(cBa), upga = wave_shr1(cBga), upgb = wave_shr1(cBgb);
                if (lane == 0) { upa = b0a; upga = b0ga; upgb = b0gb; }

                // ---- the two cells of this lane ----
                float2v acc;
                acc.x = kmax3(dga, dgga + copen_prev, dggb + orpA);
                acc.y = kmax3(cAa, cAga + copen_prev, cAgb + orpB);
                if (KIND == KA_SS) {
                        acc.x += tss[res1A + res2];
                        acc.y += tss[res1B + res2];
                } else if (KIND == KA_SP) {
                        acc.x += sp_tbl[(2 * lane) * KA_SP_STRIDE + res2];
                        acc.y += sp_tbl[(2 * lane + 1) * KA_SP_STRIDE + res2];
                } else {
#pragma unroll
                        for (int c = NRES - 1; c >= 0; --c) {
                                const float sc = q[c >> 2][c & 3];
                                float2v w; w.x = sc; w.y = sc;
                                acc = acc + p1v[c] * w;
                        }
                        // q is dead: fetch the next step's column record now so the LDS latency
                        // hides behind the rest of this step
                        const int tn = t + 1;
                        if ((tn & (KA_RING_BATCH - 1)) == 0) {
                                __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);      // batch tn/32 (issued >= 32 steps ago) has landed
                                ring_issue((tn >> 5) + 1);
                        }
                        ring_read(ST ? (v + 1) : min(max(v + 1, 0), ncols));
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
                        const bool edge = at0 || atN;
                        const bool term = (at0 && near_t) || (atN && far_t);
                        nAa = at0 ? -KA_F : acc.x;
                        nAga = edge ? -KA_F : kmax(cAga + cext, cAa + copen);
                        nAgb = term ? kmax(upgb, upa) + tA : kmax(upgb + eA, upa + oA);
                        // B: the row above is A's fresh state
                        nBa = at0 ? -KA_F : acc.y;
                        nBga = edge ? -KA_F : kmax(cBga + cext, cBa + copen);
                        nBgb = term ? kmax(nAgb, nAa) + tB : kmax(nAgb + eB, nAa + oB);
                }
                if (inr) {
                        cAa = nAa; cAga = nAga; cAgb = nAgb;
                        cBa = nBa; cBga = nBga; cBgb = nBgb;
                }
                dga = upa; dgga = upga; dggb = upgb;
                copen_prev = copen;

                // ---- collect the strip's last row: column vL of lane `lastl` goes to lane vL & 63 ----
                const int vL = t - lastl;
                if (ST || (vL >= 0 && vL <= ncols)) {
                        const float la = lane_bcast(last_is_b ? cBa : cAa, lastl);
                        const float lga = lane_bcast(last_is_b ? cBga : cAga, lastl);
                        const float lgb = lane_bcast(last_is_b ? cBgb : cAgb, lastl);
                        if (lane == (vL & 63)) { oba = la; obga = lga; obgb = lgb; }
                        if ((vL & 63) == 63 || vL == ncols) {
                                const int c0 = vL & ~63;
                                if (c0 + lane <= vL) {
                                        ka_gfloat* w = grows + 3 * IDX(c0 + lane);
                                        w[0] = oba; w[1] = obga; w[2] = obgb;
                                }
                                // publish: the next strip (another wave of this workgroup) may read them
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                if (lane == 0) __hip_atomic_store(prog + k, vL + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                }
        };

        const int nsteps = ncols + nl;                                // t = 0 .. ncols + nl - 1
        const int t_steady0 = nl;                                     // first step with every active lane at v >= 1
        const int t_steady1 = ncols - 1;                              // last step with every active lane at v <= ncols-1
        int t = 0;
#ifdef KA_TRACE_STRIP
        volatile int* tr = (volatile int*)S.trace;
        const int wv = threadIdx.x >> 6;
        if (tr && blockIdx.x == 0 && lane == 0) { tr[32 + wv] = nrows * 1000 + ncols; tr[40 + wv] = dir * 100000 + starta * 100 + startb; tr[48 + wv] = -5; __threadfence_system(); }
        for (; t < min(t_steady0, nsteps); ++t) { step(t, std::false_type()); if (tr && blockIdx.x == 0 && lane == 0) { tr[48 + wv] = t; __threadfence_system(); } }
        for (; t <= t_steady1; ++t) { step(t, std::true_type()); if (tr && blockIdx.x == 0 && lane == 0) { tr[48 + wv] = 10000 + t; __threadfence_system(); } }
        for (; t < nsteps; ++t) { step(t, std::false_type()); if (tr && blockIdx.x == 0 && lane == 0) { tr[48 + wv] = 20000 + t; __threadfence_system(); } }
        if (tr && blockIdx.x == 0 && lane == 0) { tr[56 + wv] = nsteps; __threadfence_system(); }
#else
        for (; t < min(t_steady0, nsteps); ++t) step(t, std::false_type());
        for (; t <= t_steady1; ++t) step(t, std::true_type());
        for (; t < nsteps; ++t) step(t, std::false_type());
#endif
#undef REC
#undef IDX
}
