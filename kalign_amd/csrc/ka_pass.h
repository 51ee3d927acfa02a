// ka_pass.h -- one linear-space Gotoh pass (forward or backward) of one Hirschberg
// sub-problem, executed by ONE wave64 as an anti-diagonal wavefront.
//
// Mapping (same (u, v) formulation as oracle/kalign_oracle.c:ko_pass):
//   * rows are cut into strips of 128; lane l owns the TWO rows u0+2l (A) and u0+2l+1 (B) and
//     keeps their operand data stationary in registers (profile-profile: 23 residue counts per
//     row, packed as float2 so the 23-term dot products of A and B run as v_pk_mul/v_pk_add);
//   * at step t lane l is at column v = t - l; the state of the row above A arrives from lane
//     l-1 by a DPP wave shift (v_mov_b32_dpp wave_shr:1), B takes A's fresh state;
//   * the column operand is streamed: profile-profile stages the 28 useful floats of every
//     column record (fields 32..59: pre-summed substitution scores + base gap penalties,
//     aln_setup.c:40-99) into a per-wave LDS ring with direct global->LDS loads
//     (global_load_lds_dwordx4), 32 columns per batch, chunk-major so the per-lane reads are
//     conflict-free ds_read_b128; sequence operands flow lane to lane through one more DPP shift;
//   * the strip's last row goes to the sub-problem's row buffer (HBM/L2), which is also the
//     boundary the next strip reads (in place, 64 states prefetched per 64 steps).
//
// All arithmetic is binary32 in the reference's order, no contraction (see ka_kernels.hip).
#pragma once

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ int wave_shr1_i(int x)
{
        return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
}

template <int KIND, int NRES>
__device__ void ka_pass(const TaskShared& S, const KaSub& sb, const int dir, KaState* rows, const int lane,
                        char* wlds, const float* tss)
{
        const int startb = sb.startb, endb = sb.endb;
        const int ncols = endb - startb;
        const int mid = ((sb.enda - sb.starta) / 2) + sb.starta;
        const int r0 = (dir == KA_FWD) ? sb.starta : mid;
        const int r1 = (dir == KA_FWD) ? mid : sb.enda;
        const int nrows = r1 - r0;
        const bool near_t = (dir == KA_FWD) ? (startb == 0) : (endb == S.Lb);
        const bool far_t = (dir == KA_FWD) ? (endb == S.Lb) : (startb == 0);
        const KaState inj = (dir == KA_FWD) ? sb.fin : sb.bin;

#define REC(v_) ((dir == KA_FWD) ? (startb + (v_)) : (endb + 1 - (v_)))
#define IDX(v_) ((dir == KA_FWD) ? (v_) : (ncols - (v_)))

        if (nrows == 0) {
                // only the "row -1" initialisation survives (aln_seqseq.c:40-58): a serial chain
                if (lane == 0) {
                        KaState ini = inj;
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
                return;
        }

        float* const sp_tbl = (float*)wlds;                           // seq-profile: this lane's two score rows
        const float m1 = S.p1_mult, m2 = S.p2_mult;

        for (int u0 = 0; u0 < nrows; u0 += 128) {
                const int nr = min(128, nrows - u0);
                const int nl = (nr + 1) >> 1;
                const bool first = (u0 == 0);
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

                KaState cA = { -KA_F, -KA_F, -KA_F };
                KaState cB = cA, dg = cA;
                KaState ini = inj;
                KaState batch = cA, nextb = cA;
                float copen_prev = 0.0f;
                int res2 = 0, resb = 0;

                if (!first) nextb = rows[IDX(min(lane, ncols))];

                if (KIND == KA_PP) {
                        // prime the ring: batches 0 and 1
                        __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                                if (lane < KA_RING_BATCH) {
                                        const int vv = min(bb * KA_RING_BATCH + lane, ncols);
                                        const float* g = S.p2 + ((long long)REC(vv) << 6) + 32;
                                        char* dst = wlds + bb * KA_SLOT_BYTES;
#pragma unroll
                                        for (int ch = 0; ch < KA_REC_CHUNKS; ++ch)
                                                __builtin_amdgcn_global_load_lds((ka_glb_ptr)(g + 4 * ch), (ka_lds_ptr)(dst + ch * 512), 16, 0, 0);
                                }
                        }
                }

                const int nsteps = ncols + nl;
                for (int t = 0; t < nsteps; ++t) {
                        const int v = t - lane;
                        const bool inr = (v >= 0) && (v <= ncols) && actA;
                        const int vc = min(max(v, 0), ncols);

                        if (KIND == KA_PP) {
                                if ((t & (KA_RING_BATCH - 1)) == 0) {
                                        // batch t/32 was issued >= 32 steps ago (or just primed): make it visible,
                                        // then start batch t/32 + 1 (t > 0; batch 1 was primed)
                                        __builtin_amdgcn_s_waitcnt(KA_WAIT_VM0);
                                        if (t > 0) {
                                                const int nb = (t >> 5) + 1;
                                                if (lane < KA_RING_BATCH && nb * KA_RING_BATCH <= ncols) {
                                                        const int vv = min(nb * KA_RING_BATCH + lane, ncols);
                                                        const float* g = S.p2 + ((long long)REC(vv) << 6) + 32;
                                                        char* dst = wlds + (nb & (KA_RING_SLOTS - 1)) * KA_SLOT_BYTES;
#pragma unroll
                                                        for (int ch = 0; ch < KA_REC_CHUNKS; ++ch)
                                                                __builtin_amdgcn_global_load_lds((ka_glb_ptr)(g + 4 * ch), (ka_lds_ptr)(dst + ch * 512), 16, 0, 0);
                                                }
                                        }
                                }
                        }

                        // ---- column data for column v ----
                        float copen, cext, ctext;
                        float4v q[KA_REC_CHUNKS];
                        if (KIND == KA_PP) {
                                const char* src = wlds + ((vc >> 5) & (KA_RING_SLOTS - 1)) * KA_SLOT_BYTES + (vc & 31) * 16;
#pragma unroll
                                for (int ch = 0; ch < KA_REC_CHUNKS; ++ch) q[ch] = *(const float4v*)(src + ch * 512);
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
                        KaState b0;
                        if (first) {
                                if (t == 0) {
                                        ini = inj;
                                } else if (t < ncols) {
                                        const float g = near_t ? kmax(ini.ga, ini.a) + ctext : kmax(ini.ga + cext, ini.a + copen);
                                        ini.a = -KA_F; ini.ga = g; ini.gb = -KA_F;
                                } else {
                                        ini.a = -KA_F; ini.ga = -KA_F; ini.gb = -KA_F;
                                }
                                b0 = ini;
                        } else {
                                if ((t & 63) == 0) {
                                        batch = nextb;
                                        nextb = rows[IDX(min(t + 64 + lane, ncols))];
                                }
                                b0.a = lane_bcast(batch.a, t & 63);
                                b0.ga = lane_bcast(batch.ga, t & 63);
                                b0.gb = lane_bcast(batch.gb, t & 63);
                        }

                        KaState up;
                        up.a = wave_shr1(cB.a); up.ga = wave_shr1(cB.ga); up.gb = wave_shr1(cB.gb);
                        if (lane == 0) up = b0;

                        // ---- the two cells of this lane ----
                        KaState nA, nB;
                        {
                                float2v acc;
                                acc.x = kmax3(dg.a, dg.ga + copen_prev, dg.gb + orpA);
                                acc.y = kmax3(cA.a, cA.ga + copen_prev, cA.gb + orpB);
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
                                }
                                const bool at0 = (v == 0), atN = (v == ncols);
                                // A
                                const float gbA_gen = kmax(up.gb + eA, up.a + oA);
                                const float gbA_ter = kmax(up.gb, up.a) + tA;
                                nA.a = at0 ? -KA_F : acc.x;
                                nA.ga = (at0 || atN) ? -KA_F : kmax(cA.ga + cext, cA.a + copen);
                                nA.gb = ((at0 && near_t) || (atN && far_t)) ? gbA_ter : gbA_gen;
                                // B: the row above is A's fresh state
                                const float gbB_gen = kmax(nA.gb + eB, nA.a + oB);
                                const float gbB_ter = kmax(nA.gb, nA.a) + tB;
                                nB.a = at0 ? -KA_F : acc.y;
                                nB.ga = (at0 || atN) ? -KA_F : kmax(cB.ga + cext, cB.a + copen);
                                nB.gb = ((at0 && near_t) || (atN && far_t)) ? gbB_ter : gbB_gen;
                        }
                        if (inr) {
                                cA = nA;
                                cB = nB;
                                if (lane == nl - 1) rows[IDX(v)] = actB ? nB : nA;
                        }
                        dg = up;
                        copen_prev = copen;
                }
                // the next strip (same wave) reads rows[] written by this one
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0);
        }
#undef REC
#undef IDX
}
