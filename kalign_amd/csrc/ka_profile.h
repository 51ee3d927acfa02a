// ka_profile.h -- sum-of-pairs score of a trial, P4 profile merge, leaf profiles, the anchor-consistency bonus entries of a task, the residue -> column tables.
// One of the text sections of the task kernels, included by ka_kernels.hip in this order: ka_shared.h, ka_pass.h, ka_best.h,
// ka_subtree.h, ka_wstrip.h, ka_meetup.h, ka_hirschberg.h, ka_path.h, ka_profile.h, ka_task.h.  Not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------
// compute_sp_score (sp_score.c:22-201): residue counts per column of both groups from the members' residue -> column
// tables (build_profile expands every member through its gaps[]; D.colof is the same information), then ONE sequential
// fp32 walk along the coded path -- substitution terms in (i, j) order, then the gap term, exactly as the reference
// accumulates them (the total decides which trial wins; it is not reassociated).
// S.sp_freq: [col][24] for operand a (23 counts + residues in the column), then the same for b.
// ------------------------------------------------------------------------------------------
__device__ void ka_sp_build(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T)
{
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const long long total = 24ll * (S.len_a + S.len_b);
        for (long long x = tid; x < total; x += KA_NT) S.sp_freq[x] = 0;
        __syncthreads();
        const int na = T.nsip_a, nb = T.nsip_b;
        const int* ma = D.sip + D.sip_off[T.a];
        const int* mb = D.sip + D.sip_off[T.b];
        for (int m = wave; m < na + nb; m += KA_NW) {
                const bool in_a = m < na;
                const int si = in_a ? ma[m] : mb[m - na];
                int* fr = S.sp_freq + (in_a ? 0 : 24 * S.len_a);
                const int* col = D.colof + D.seq_off[si];
                const uint8_t* res = D.codes + D.seq_off[si];
                const int len = D.node_len[si];
                for (int p = lane; p < len; p += 64) {
                        const int c = col[p], r = res[p];
                        if (r < 23) { atomicAdd(&fr[24 * c + r], 1); atomicAdd(&fr[24 * c + 23], 1); }
                }
        }
        __syncthreads();
}

// The walk adds ONE term at a time to ONE float (sp_score.c:134-189): the order of the additions is part of the result
// and the chain cannot be split.  What can be parallel is everything around the additions: one thread per path column
// works out its column's terms -- the (i, j) products in the reference's order, then the gap term(s), with the sign
// folded in (x - y == x + (-y)) -- into an LDS buffer, and one thread adds the buffer up in order (loads run ahead of
// the dependent adds: the chain costs an add per term instead of a trip to L2 per term).
// lds: KA_SP_TB floats + 2 * blockDim.x ints.
#define KA_SP_TB 24576
__device__ void ka_sp_score(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, char* lds)
{
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        float* const buf = (float*)lds;
        int* const wtot = (int*)(buf + KA_SP_TB);                     // per-wave totals of the scan
        const int* path = S.coded;
        const int* fa0 = S.sp_freq;
        const int* fb0 = S.sp_freq + 24 * S.len_a;
        const int nsa = T.nsip_a, nsb = T.nsip_b;
        const float gpo = T.gpo, gpe = T.gpe, tgpe = T.tgpe;
        const float* subm = D.subm;
        const int plen = path[0];
        float total = 0.0f;                                            // (thread 0's)
        for (int c0 = 1; c0 <= plen; c0 += KA_NT) {
                const int c = c0 + tid;
                const bool in = c <= plen;
                const int code = in ? path[c] : 0;
                const int step = code & 3;
                const float pen = (code & 32) ? tgpe : gpe;
                const int prev = (in && c > 1) ? (path[c - 1] & 3) : 0;
                const int* fa = fa0 + 24 * ((in && step != 1) ? S.srcA[c] - 1 : 0);
                const int* fb = fb0 + 24 * ((in && step != 2) ? S.srcB[c] - 1 : 0);
                // the column's term count
                int nza = 0, nzb = 0, cnt = 0;
                if (in) {
                        if (step == 0) {
                                for (int i = 0; i < 23; ++i) nza += fa[i] != 0;
                                for (int j = 0; j < 23; ++j) nzb += fb[j] != 0;
                                cnt = nza * nzb + 1;
                        } else if (step == 1) cnt = (prev == 1) ? 1 : 2;
                        else cnt = (prev == 2) ? 1 : 2;
                }
                // exclusive scan over the block
                int sc = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(sc, d, 64); if (lane >= d) sc += y; }
                __syncthreads();                                       // (the previous block's sum is done with buf / wtot)
                if (lane == 63) wtot[wave] = sc;
                __syncthreads();
                int off = sc - cnt, all = 0;
                for (int w = 0; w < KA_NW; ++w) { const int x = wtot[w]; if (w < wave) off += x; all += x; }
                for (int base = 0; base < all; base += KA_SP_TB) {
                        // this thread's terms with a buffer index in [base, base + KA_SP_TB)
                        if (in && off < base + KA_SP_TB && off + cnt > base) {
                                int k = off - base;
                                auto put = [&](float v) { if (k >= 0 && k < KA_SP_TB) buf[k] = v; ++k; };
                                if (step == 0) {
                                        for (int i = 0; i < 23; ++i) {
                                                const int ai = fa[i];
                                                if (ai == 0) continue;
                                                for (int j = 0; j < 23; ++j) {
                                                        const int bj = fb[j];
                                                        if (bj == 0) continue;
                                                        put((float)(ai * bj) * subm[i * 23 + j]);
                                                }
                                        }
                                        const int n_res_a = fa[23], n_res_b = fb[23];
                                        const int n_gap_a = nsa - n_res_a, n_gap_b = nsb - n_res_b;
                                        put(-((float)(n_res_a * n_gap_b + n_gap_a * n_res_b) * pen));
                                } else if (step == 1) {
                                        const int n_pairs = nsa * fb[23];
                                        if (prev != 1) put(-((float)n_pairs * gpo));
                                        put(-((float)n_pairs * pen));
                                } else {
                                        const int n_pairs = fa[23] * nsb;
                                        if (prev != 2) put(-((float)n_pairs * gpo));
                                        put(-((float)n_pairs * pen));
                                }
                        }
                        __syncthreads();
                        if (tid == 0) {
                                const int n = min(KA_SP_TB, all - base);
                                int k = 0;
                                for (; k + 8 <= n; k += 8) {
                                        const float4v x = *(const float4v*)(buf + k), y = *(const float4v*)(buf + k + 4);
                                        total += x.x; total += x.y; total += x.z; total += x.w;
                                        total += y.x; total += y.y; total += y.z; total += y.w;
                                }
                                for (; k < n; ++k) total += buf[k];
                        }
                        if (base + KA_SP_TB < all) __syncthreads();        // the buffer is refilled
                }
        }
        if (tid == 0) S.sp_value = total;
        __syncthreads();
}

// ------------------------------------------------------------------------------------------
// P4: update_n (aln_setup.c:230-436), one thread per (output column, field).
// ------------------------------------------------------------------------------------------
// tss_syn (round 6; first pass without sequence weights): the leaf operands' records are NOT in HBM -- a sequence's record is a
// function of its residue (make_profile_n, aln_setup.c:40-99: a count of one, the residue's row of pre-summed scores, the three gap
// fields), so the merge makes the four floats it wants of it from the residue and the seq-seq score table in LDS, and the tasks that
// consume a sequence neither write nor read 256 bytes per position for it (ka_make_leaf_profile: 60 of a leaf task's 475 us).
__device__ __forceinline__ float4v ka_leaf_rec4(const uint8_t* __restrict__ seq, const int len, const int rec, const int k4,
                                                const float gpo, const float gpe, const float tgpe, const float* tss)
{
        // k4 = first of four consecutive fields.  Fields: [c] = 1 (the residue's count), [32 .. 54] = tss[c][0 .. 22] (its row of scores
        // -- 16-byte aligned in the table: one read for four of them), [55 .. 57] = -gpo, -gpe, -tgpe, the rest 0; the two boundary
        // records (rec 0, len + 1) carry the gap fields only.
        const bool inner = (rec >= 1 && rec <= len);
        const int c = inner ? seq[rec - 1] : 0;
        float4v out = {0.0f, 0.0f, 0.0f, 0.0f};
        if (k4 >= 32 && k4 < 56) {
                if (inner) out = *(const float4v*)(tss + c * KA_T_STRIDE + (k4 - 32));       // (k4 == 52: [55] is overwritten below; the table's 24th float is 0)
                if (k4 == 52) out.w = -gpo;
        } else if (k4 == 56) {
                out.x = -gpe; out.y = -tgpe;
        } else if (inner && (c >> 2) == (k4 >> 2)) {
                out.x = (c & 3) == 0 ? 1.0f : 0.0f; out.y = (c & 3) == 1 ? 1.0f : 0.0f; out.z = (c & 3) == 2 ? 1.0f : 0.0f; out.w = (c & 3) == 3 ? 1.0f : 0.0f;
        }
        return out;
}

// update_n's two successive adjustments of a gap column at the end / start of its run (code & 16, then code & 4; aln_setup.c:230-436): elem()'s
// statements on a value that is already in a register -- a synthesized record (tss_syn) is in no memory elem() could read it from.
// (A function with scalar arguments, not a lambda: captured, `(code & 1) ? sipa : sipb` became an indexed load from a closure in scratch.)
__device__ __forceinline__ float ka_adj2(float val, const int k, const int code, const float sip, const float gpo0, const float tgpe0)
{
        for (int pass = 0; pass < 2; ++pass) {
                const int bit = pass == 0 ? 16 : 4;
                if (!(code & bit)) continue;
                float gp;
                if (code & 32) {
                        if (k == 25) val += sip;
                        gp = tgpe0 * sip;
                        if (k == 23) val += sip;
                        gp += gpo0 * sip;
                } else {
                        if (k == 23) val += sip;
                        gp = gpo0 * sip;
                }
                if (k >= 32 && k < 55) val -= gp;
        }
        return val;
}

#if defined(KA_UNIT) && (KA_UNIT == 3 || KA_UNIT == 8)
#define KA_MERGE_LEAN 1                                        // the 128-register units run seq-seq tasks only: the batches' loop for two sequences alone (all four variants spilled there)
#else
#define KA_MERGE_LEAN 0
#endif
#ifndef KA_MERGE_BATCH
#define KA_MERGE_BATCH 4                                       // output items a thread of the merge has in flight (0: one)
#endif
__device__ void ka_update_profile(const TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, const int alnlen, const float* tss_syn = nullptr)
{
        const float* pa = S.profa;
        const float* pb = S.profb;
        float* np = S.newp;
        const float sipa = (float)T.nsip_a, sipb = (float)T.nsip_b;
        float sA = 1.0f, sB = 1.0f;
        bool rebalance = false;
        if (D.usw > 0.0f && T.nsip_a > 0 && T.nsip_b > 0) {
                const float pseudo = D.usw;
                const float total = (float)(T.nsip_a + T.nsip_b);
                const float denom = total + 2.0f * pseudo;
                sA = total * (sipa + pseudo) / (denom * sipa);
                sB = total * (sipb + pseudo) / (denom * sipb);
                rebalance = true;
        }
        // fields 27..29 of an operand as the reference would see them at this point: zero for a
        // leaf (make_profile_n), [55..57] * nsip_other for a profile (set_gap_penalties_n ran on it
        // for this merge, aln_run.c:239-253).  They are dead values (always rewritten before
        // use) but part of the merged record, so they are reproduced for bit-identical profiles.
        const bool leaf_a = (T.nsip_a == 1), leaf_b = (T.nsip_b == 1);
        auto fa = [&](const float* rec, int k) __attribute__((always_inline)) -> float {
                if (k >= 27 && k <= 29) return leaf_a ? 0.0f : rec[k + 28] * sipb;
                return rec[k];
        };
        auto fb = [&](const float* rec, int k) __attribute__((always_inline)) -> float {
                if (k >= 27 && k <= 29) return leaf_b ? 0.0f : rec[k + 28] * sipa;
                return rec[k];
        };
        // one thread per (output column, 4 consecutive fields): the column's op code and source
        // records are looked up once, the four field values are independent
        const float* __restrict__ pa_r = pa;
        const float* __restrict__ pb_r = pb;
        float* __restrict__ np_r = np;
        const int* __restrict__ coded = S.coded;
        const int* __restrict__ srcA = S.srcA;
        const int* __restrict__ srcB = S.srcB;
        auto elem = [&](const int c, const int k, const int code, const float* __restrict__ ra, const float* __restrict__ rb) __attribute__((always_inline)) -> float {
                float val;
                if (c == 0 || c == alnlen + 1) {
                        const float va = fa(ra, k), vb = fb(rb, k);
                        val = (rebalance && k < 23) ? (va * sA + vb * sB) : (va + vb);
                } else if (!code) {
                        if (rebalance && k < 23) {
                                val = ra[k] * sA + rb[k] * sB;
                        } else {
                                val = fa(ra, k) + fb(rb, k);
                                if (rebalance && k >= 32 && k < 55) {
                                        const float dA = sA - 1.0f, dB = sB - 1.0f;
                                        const int jj = k - 32;
                                        float delta = 0.0f;
                                        for (int aa = 0; aa < 23; ++aa) {
                                                delta += (ra[aa] * dA + rb[aa] * dB) * D.subm[23 * aa + jj];
                                        }
                                        val += delta;
                                }
                        }
                } else {
                        const bool gap_in_a = (code & 1) != 0;
                        const float sip = gap_in_a ? sipa : sipb;
                        val = gap_in_a ? fb(rb, k) : fa(ra, k);
                        // as the reference: up to two successive adjustments (close, then open)
                        if (!(code & 20)) {
                                if (code & 32) {
                                        if (k == 25) val += sip;
                                        if (k >= 32 && k < 55) val -= D.tgpe0 * sip;
                                } else {
                                        if (k == 24) val += sip;
                                        if (k >= 32 && k < 55) val -= D.gpe0 * sip;
                                }
                        } else {
                                for (int pass = 0; pass < 2; ++pass) {
                                        const int bit = pass == 0 ? 16 : 4;
                                        if (!(code & bit)) continue;
                                        float gp;
                                        if (code & 32) {
                                                if (k == 25) val += sip;
                                                gp = D.tgpe0 * sip;
                                                if (k == 23) val += sip;
                                                gp += D.gpo0 * sip;
                                        } else {
                                                if (k == 23) val += sip;
                                                gp = D.gpo0 * sip;
                                        }
                                        if (k >= 32 && k < 55) val -= gp;
                                }
                        }
                }
                return val;
        };
        const float adj_gpo = D.gpo0, adj_tgpe = D.tgpe0;
        const long long total4 = (long long)(alnlen + 2) * 16;
        const float gpe_a = D.gpe0 * sipa, gpe_b = D.gpe0 * sipb, tgpe_a = D.tgpe0 * sipa, tgpe_b = D.tgpe0 * sipb;
#if KA_MERGE_BATCH
#if KA_MERGE_LEAN
        if (!rebalance && tss_syn && leaf_a && leaf_b && (D.merge_batch & 8)) {
#else
        if (!rebalance && (S.G == 1 || (D.merge_batch & 4)) && (D.merge_batch & (tss_syn && (leaf_a || leaf_b) ? 2 : 1))) {
#endif
                // Round 6: the same statements, KA_MERGE_BATCH output items per thread at a time.  The walk is a chain of dependent trips
                // to memory per item (op code and source records -> the two records, written by other XCDs: they come from HBM -> the
                // store) and a task's 4 or 8 waves have ~27 items per thread: one after the other that is 45-70 us of every task, on
                // every dependency chain of the tree, with the memory pipe all but idle.  Here the codes of a batch are fetched
                // together, then its records, then it is stored: loads without branches around them (clamped indices, selects
                // behind) and through global pointers -- a flat load counts on vmcnt AND lgkmcnt and the compiler then waits
                // for everything at every use.  (The stride is a multiple of 16: a thread's four fields are the same in every item.)
                // Measured (tools/phase_means.py with KA_MERGE=0 / 1 / 3 / 7 / 15 on one context): profile-profile tasks of the queued launch
                // 52 -> 29 us (bit 0), seq-profile tasks 72 -> 41 us (bit 1), the workgroups of a cluster 20 -> 16 us (bit 2), the seq-seq tasks
                // of the leaf launch 51 -> 32 us (bit 3: the 128-register units build the loop for two sequences only -- with all four
                // variants it spilled, 61 us).  A first version gave the gain back to scratch: a lambda that captures by reference keeps its
                // closure in memory as soon as two captured scalars are SELECTED between (`gap_in_a ? sipa : sipb` became an indexed load
                // of an address, then a flat load through it) -- 424 B per lane and call, 0.4 GB of HBM writes per headline tree; hence
                // ka_adj2 as a function with scalar arguments and the c_ copies at the top of the loop's lambda.
                typedef const __attribute__((address_space(1))) int* ka_gint;
                typedef const __attribute__((address_space(1))) float* ka_gf;
                typedef const __attribute__((address_space(1))) float4v* ka_gf4;
                typedef const __attribute__((address_space(1))) uint8_t* ka_gu8;
                constexpr int U = KA_MERGE_BATCH;
                const long long stride = (long long)S.G * KA_NT;
                const long long first = (long long)S.member * KA_NT + threadIdx.x;
                const int k4 = (int)(first & 15) << 2;
                const bool syn_a = tss_syn && leaf_a, syn_b = tss_syn && leaf_b;
                const ka_gint g_coded = (ka_gint)coded, g_srcA = (ka_gint)srcA, g_srcB = (ka_gint)srcB;
                const ka_gf g_pa = (ka_gf)pa_r, g_pb = (ka_gf)pb_r;
                const ka_gu8 g_seqa = (ka_gu8)(D.codes + (syn_a ? D.seq_off[T.a] : 0)), g_seqb = (ka_gu8)(D.codes + (syn_b ? D.seq_off[T.b] : 0));
                const int la = S.len_a, lb = S.len_b;
                // a sequence's record from its residue (ka_leaf_rec4, the residue already here)
                auto leaf4 = [&](const int cres, const bool inner) __attribute__((always_inline)) -> float4v {
                        float4v out = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (k4 >= 32 && k4 < 56) {
                                if (inner) out = *(const ka_lf4*)((const ka_lf*)tss_syn + cres * KA_T_STRIDE + (k4 - 32));   // (the table is in LDS: ds_read, not a flat load)
                                if (k4 == 52) out.w = -T.gpo;
                        } else if (k4 == 56) {
                                out.x = -T.gpe; out.y = -T.tgpe;
                        } else if (inner && (cres >> 2) == (k4 >> 2)) {
                                out.x = (cres & 3) == 0 ? 1.0f : 0.0f; out.y = (cres & 3) == 1 ? 1.0f : 0.0f; out.z = (cres & 3) == 2 ? 1.0f : 0.0f; out.w = (cres & 3) == 3 ? 1.0f : 0.0f;
                        }
                        return out;
                };
                auto run = [&](auto sa_tag, auto sb_tag) __attribute__((always_inline)) {   // (inlined: as a call its captures went through 424 B of scratch per lane -- 0.4 GB of HBM writes per headline tree)
                // (scalars by value first: selected between through the closure's references they became indexed loads of addresses kept in scratch)
                const float c_sipa = sipa;
                const float c_sipb = sipb;
                const float c_gpe_a = gpe_a;
                const float c_gpe_b = gpe_b;
                const float c_tgpe_a = tgpe_a;
                const float c_tgpe_b = tgpe_b;
                const float c_adj_gpo = adj_gpo;
                const float c_adj_tgpe = adj_tgpe;
                const int c_la = la;
                const int c_lb = lb;
                const int c_k4 = k4;
                const int c_alnlen = alnlen;
                const bool c_leaf_a = leaf_a;
                const bool c_leaf_b = leaf_b;
                const long long c_stride = stride;
                const long long c_first = first;
                const long long c_total4 = total4;
                constexpr bool SA = decltype(sa_tag)::value, SB = decltype(sb_tag)::value;   // (the operand is a sequence whose records are made here)
                for (long long x0 = c_first; x0 < c_total4; x0 += U * c_stride) {
                        int code[U], reca[U], recb[U], col[U];
                        int lc[U], lia[U], lib[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                                const long long x4 = x0 + u * c_stride;
                                col[u] = x4 < c_total4 ? (int)(x4 >> 4) : -1;
                                const int cc = min(max(col[u], 0), c_alnlen + 1);
                                lc[u] = g_coded[cc]; lia[u] = g_srcA[cc]; lib[u] = g_srcB[cc];
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                                const int c = col[u];
                                const bool inner = c > 0 && c <= c_alnlen;
                                code[u] = inner ? lc[u] : 0;
                                reca[u] = c == c_alnlen + 1 ? c_la + 1 : (inner ? max(lia[u], 0) : 0);
                                recb[u] = c == c_alnlen + 1 ? c_lb + 1 : (inner ? max(lib[u], 0) : 0);
                        }
                        float4v A[U], B[U];
                        int resa[U], resb[U];
                        // (one block of loads, no branch inside: a branch between two loads costs a wait for everything in flight --
                        // which operand is a sequence is a template argument of this loop)
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                                if constexpr (SA) resa[u] = g_seqa[min(max(reca[u] - 1, 0), c_la - 1)];
                                else A[u] = *(ka_gf4)(g_pa + ((long long)reca[u] << 6) + c_k4);
                                if constexpr (SB) resb[u] = g_seqb[min(max(recb[u] - 1, 0), c_lb - 1)];
                                else B[u] = *(ka_gf4)(g_pb + ((long long)recb[u] << 6) + c_k4);
                        }
                        float xa[U][2], xb[U][2];
                        if (c_k4 == 24 || c_k4 == 28) {                     // fields 27 / 28, 29 (see fa / fb)
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                        const int f = c_k4 == 24 ? 55 : 56;
                                        // (a sequence's fields 27 .. 29 are zero whatever its record holds: read anyway where the record exists, selected below)
                                        if constexpr (SA) { xa[u][0] = 0.0f; xa[u][1] = 0.0f; }
                                        else { xa[u][0] = g_pa[((long long)reca[u] << 6) + f]; xa[u][1] = g_pa[((long long)reca[u] << 6) + f + 1]; }
                                        if constexpr (SB) { xb[u][0] = 0.0f; xb[u][1] = 0.0f; }
                                        else { xb[u][0] = g_pb[((long long)recb[u] << 6) + f]; xb[u][1] = g_pb[((long long)recb[u] << 6) + f + 1]; }
                                }
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                                if constexpr (SA) A[u] = leaf4(resa[u], reca[u] >= 1 && reca[u] <= c_la);
                                if constexpr (SB) B[u] = leaf4(resb[u], recb[u] >= 1 && recb[u] <= c_lb);
                        }
                        if (c_k4 == 24) {
#pragma unroll
                                for (int u = 0; u < U; ++u) { A[u].w = c_leaf_a ? 0.0f : xa[u][0] * c_sipb; B[u].w = c_leaf_b ? 0.0f : xb[u][0] * c_sipa; }
                        } else if (c_k4 == 28) {
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                        A[u].x = c_leaf_a ? 0.0f : xa[u][0] * c_sipb; A[u].y = c_leaf_a ? 0.0f : xa[u][1] * c_sipb;
                                        B[u].x = c_leaf_b ? 0.0f : xb[u][0] * c_sipa; B[u].y = c_leaf_b ? 0.0f : xb[u][1] * c_sipa;
                                }
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                                const int c = col[u];
                                if (c < 0) continue;
                                float4v out;
                                if (code[u] & 20) {                      // (a gap column at the end / start of its run)
                                        const bool gap_in_a = (code[u] & 1) != 0;
                                        const float sip = gap_in_a ? c_sipa : c_sipb;
                                        out = gap_in_a ? B[u] : A[u];
                                        out.x = ka_adj2(out.x, c_k4 + 0, code[u], sip, c_adj_gpo, c_adj_tgpe); out.y = ka_adj2(out.y, c_k4 + 1, code[u], sip, c_adj_gpo, c_adj_tgpe);
                                        out.z = ka_adj2(out.z, c_k4 + 2, code[u], sip, c_adj_gpo, c_adj_tgpe); out.w = ka_adj2(out.w, c_k4 + 3, code[u], sip, c_adj_gpo, c_adj_tgpe);
                                } else if (c == 0 || c == c_alnlen + 1 || !code[u]) {
                                        out = A[u] + B[u];
                                } else {
                                        const bool gap_in_a = (code[u] & 1) != 0, term = (code[u] & 32) != 0;
                                        const float sip = gap_in_a ? c_sipa : c_sipb;
                                        const float g = term ? (gap_in_a ? c_tgpe_a : c_tgpe_b) : (gap_in_a ? c_gpe_a : c_gpe_b);
                                        out = gap_in_a ? B[u] : A[u];
                                        if (c_k4 == 24) { if (term) out.y += sip; else out.x += sip; }        // [25] / [24]
                                        else if (c_k4 >= 32 && c_k4 < 52) { out.x -= g; out.y -= g; out.z -= g; out.w -= g; }
                                        else if (c_k4 == 52) { out.x -= g; out.y -= g; out.z -= g; }           // [55] is not a score
                                }
                                *(float4v*)(np_r + ((x0 + u * c_stride) << 2)) = out;
                        }
                }
                };
#if KA_MERGE_LEAN
                run(std::true_type(), std::true_type());
#else
                if (syn_a && syn_b) run(std::true_type(), std::true_type());
                else if (syn_a) run(std::true_type(), std::false_type());
                else if (syn_b) run(std::false_type(), std::true_type());
                else run(std::false_type(), std::false_type());
#endif
                return;
        }
#endif
        for (long long x4 = (long long)S.member * KA_NT + threadIdx.x; x4 < total4; x4 += (long long)S.G * KA_NT) {
                const int c = (int)(x4 >> 4);
                const int k4 = (int)(x4 & 15) << 2;
                int code = 0;
                int reca = 0, recb = 0;
                if (c == alnlen + 1) { reca = S.len_a + 1; recb = S.len_b + 1; }
                else if (c != 0) {
                        code = coded[c];
                        const int ia = srcA[c], ib = srcB[c];
                        reca = ia < 0 ? 0 : ia; recb = ib < 0 ? 0 : ib;
                }
                const float* ra = pa_r + ((long long)reca << 6);
                const float* rb = pb_r + ((long long)recb << 6);
                float4v out;
                if (!rebalance) {
                        // The common case (no sequence weights; the coded path carries only the flags the reference
                        // really sets), four fields at a time -- same operations as elem() below, without the per-field
                        // branching: a match / boundary column is the sum of the two records, a gap column the present
                        // side with its gap counter bumped and the scores lowered by (t)gpe * members of the absent side.
                        float4v A, B;
                        if (tss_syn && leaf_a) A = ka_leaf_rec4(D.codes + D.seq_off[T.a], S.len_a, reca, k4, T.gpo, T.gpe, T.tgpe, tss_syn);
                        else A = *(const float4v*)(ra + k4);
                        if (tss_syn && leaf_b) B = ka_leaf_rec4(D.codes + D.seq_off[T.b], S.len_b, recb, k4, T.gpo, T.gpe, T.tgpe, tss_syn);
                        else B = *(const float4v*)(rb + k4);
                        if (k4 == 24) {                                  // field 27 (see fa / fb)
                                A.w = leaf_a ? 0.0f : ra[55] * sipb; B.w = leaf_b ? 0.0f : rb[55] * sipa;
                        } else if (k4 == 28) {                           // fields 28, 29
                                A.x = leaf_a ? 0.0f : ra[56] * sipb; A.y = leaf_a ? 0.0f : ra[57] * sipb;
                                B.x = leaf_b ? 0.0f : rb[56] * sipa; B.y = leaf_b ? 0.0f : rb[57] * sipa;
                        }
                        if (code & 20) {                                 // (a gap column at the end / start of its run)
                                const bool gap_in_a = (code & 1) != 0;
                                const float sip = gap_in_a ? sipa : sipb;
                                out = gap_in_a ? B : A;
                                out.x = ka_adj2(out.x, k4 + 0, code, sip, adj_gpo, adj_tgpe); out.y = ka_adj2(out.y, k4 + 1, code, sip, adj_gpo, adj_tgpe);
                                out.z = ka_adj2(out.z, k4 + 2, code, sip, adj_gpo, adj_tgpe); out.w = ka_adj2(out.w, k4 + 3, code, sip, adj_gpo, adj_tgpe);
                        } else if (c == 0 || c == alnlen + 1 || !code) {
                                out = A + B;
                        } else {
                                const bool gap_in_a = (code & 1) != 0, term = (code & 32) != 0;
                                const float sip = gap_in_a ? sipa : sipb;
                                const float g = term ? (gap_in_a ? tgpe_a : tgpe_b) : (gap_in_a ? gpe_a : gpe_b);
                                out = gap_in_a ? B : A;
                                if (k4 == 24) { if (term) out.y += sip; else out.x += sip; }        // [25] / [24]
                                else if (k4 >= 32 && k4 < 52) { out.x -= g; out.y -= g; out.z -= g; out.w -= g; }
                                else if (k4 == 52) { out.x -= g; out.y -= g; out.z -= g; }           // [55] is not a score
                        }
                } else {
                        out.x = elem(c, k4 + 0, code, ra, rb);
                        out.y = elem(c, k4 + 1, code, ra, rb);
                        out.z = elem(c, k4 + 2, code, ra, rb);
                        out.w = elem(c, k4 + 3, code, ra, rb);
                }
                *(float4v*)(np_r + (x4 << 2)) = out;
        }
}

// Leaf profile (make_profile_n, aln_setup.c:40-99), one float4 per thread.  The pre-summed
// substitution scores subm[c][j] - soff come from the seq-seq table in LDS (same expression, same
// bits).  Non-leaf operands need nothing here: set_gap_penalties_n is folded into the loads.
__device__ void ka_make_leaf_profile(float* __restrict__ prof, int len, const uint8_t* __restrict__ seq,
                                     float gpo, float gpe, float tgpe, const float* tss)
{
        const long long total4 = (long long)(len + 2) * 16;
        for (long long x4 = threadIdx.x; x4 < total4; x4 += KA_NT) {
                const int r = (int)(x4 >> 4);
                const int k4 = (int)(x4 & 15) << 2;
                const bool inner = (r >= 1 && r <= len);
                const int c = inner ? seq[r - 1] : 0;
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                        const int k = k4 + u;
                        float val = 0.0f;
                        if (k == 55) val = -gpo;
                        else if (k == 56) val = -gpe;
                        else if (k == 57) val = -tgpe;
                        else if (inner) {
                                if (k == c) val = 1.0f;
                                else if (k >= 32 && k < 55) val = tss[c * KA_T_STRIDE + (k - 32)];
                        }
                        v[u] = val;
                }
                float4v out; out.x = v[0]; out.y = v[1]; out.z = v[2]; out.w = v[3];
                *(float4v*)(prof + (x4 << 2)) = out;
        }
}

// ------------------------------------------------------------------------------------------
// Anchor consistency, per task (anchor_consistency.c:352-561 + do_align's bonus block,
// aln_run.c:262-295).  The reference materialises a dense La x Lb bonus matrix on the host for
// every task; per anchor every row has at most ONE non-zero entry, so the device keeps <= K
// (column, value) entries per DP row instead and the passes carry them in registers.
//
// ka_cons_votes = get_node_anchor_positions for both operands and all anchors: for a leaf the
// position map itself; for a profile a vote over its member sequences, where "best" is the anchor
// position of the FIRST member (in the reference's sip order) that has one in the column.  Being
// first in a fixed order is a min-reduction over the member index, and `agree` / `total` are
// counts, so the vote runs in parallel over (member, residue) with LDS atomics:
// key = member_index << 32 | position, counts = total | agree << 16.  Which column a residue sits
// in comes from D.colof (kept up to date by ka_update_colof).  The workgroups of a cluster share
// the work by operand and by anchor; tables that do not fit into LDS live in the task's HBM scratch.
// ------------------------------------------------------------------------------------------
// The two sweeps over the (member, residue) pairs of members [m0, m1) of `node` for ONE anchor `anchor` (table index tb of the
// tables at key / cnt / agr, dp_len cells each): sweep 0 first-member key + total, sweep 1 agreement with the winner.
// Latency-bound gathers: each wave pre-loads the metadata of 64 of its members lane-parallel and keeps 8 x 64 residues of loads
// in flight before the atomics.  NBL anchors at a time (the tables of anchors KS0 + b * KSTEP, b < nb).
template <int NBL>
__device__ __forceinline__ void ka_vote_sweep(const KaTreeDev& D, const int* members, const int m0, const int m1, const int sweep,
                                              unsigned long long* key, unsigned int* cnt, unsigned int* agr, const bool wide, const bool in_lds,
                                              const int dp_len, const int nb, const int ks0, const int kstep)
{
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int mine = (m1 - m0 - wave + KA_NW - 1) / KA_NW;          // members of this wave
        for (int base = 0; base < mine; base += 64) {
                const int ml = min(base + lane, mine - 1);
                const int mi_l = m0 + wave + KA_NW * ml;
                const int si_l = members[mi_l];
                const int len_l = D.node_len[si_l];
                const long long mo_l = D.cons_map_off[si_l];
                const int so_l = D.seq_off[si_l];
                const int cntm = min(64, mine - base);
                for (int jm = 0; jm < cntm; ++jm) {
                        const int mi = m0 + wave + KA_NW * (base + jm);
                        const int len = __shfl(len_l, jm, 64);
                        const long long mo = __shfl(mo_l, jm, 64);
                        const int* map = D.cons_maps + mo;
                        const int* col = D.colof + __shfl(so_l, jm, 64);
                        // (round 4: eight residues per lane in flight instead of four -- a 400-residue member is one trip)
                        constexpr int KU = 8;
                        for (int p0 = lane; p0 < len; p0 += 64 * KU) {
                                int cc[KU], aa[KU][NBL];
#pragma unroll
                                for (int u = 0; u < KU; ++u) {
                                        const int pp = p0 + 64 * u;
                                        const bool ok = pp < len;
                                        cc[u] = ok ? col[pp] : 0;
#pragma unroll
                                        for (int b = 0; b < NBL; ++b)
                                                aa[u][b] = (ok && b < nb) ? map[(long long)(ks0 + b * kstep) * len + pp] : -1;
                                }
#pragma unroll
                                for (int u = 0; u < KU; ++u) {
#pragma unroll
                                        for (int b = 0; b < NBL; ++b) {
                                                const int a = aa[u][b];
                                                if (a < 0) continue;
                                                const int x = b * dp_len + cc[u];
                                                if (sweep == 0) {
                                                        atomicMin(&key[x], ((unsigned long long)(unsigned int)mi << 32) | (unsigned int)a);
                                                        atomicAdd(&cnt[x], 1u);
                                                } else {
                                                        // (HBM tables: the atomics were performed at L2; read them back past the L1)
                                                        const unsigned long long kk = in_lds ? key[x]
                                                                : __hip_atomic_load(&key[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                        if ((unsigned int)a == (unsigned int)(kk & 0xffffffffull)) { if (wide) atomicAdd(&agr[x], 1u); else atomicAdd(&cnt[x], 0x10000u); }
                                                }
                                        }
                                }
                        }
                }
        }
}

// Round 4: R >= 2 workgroups per (operand, anchor) -- clusters of 4K workgroups and more at the top of a big tree, where one
// workgroup per unit takes ~0.75 us per member of the bigger operand (7 ms of the 10 ms root task of a 16384-sequence tree).
// Every workgroup votes over its range of the members into tables in LDS, the R partial tables of a unit meet in the task's
// HBM table (16 B per column: key, total, agree) behind cluster barriers: first member = min over the partial keys, totals add
// up; the merged keys come back into LDS for the agreement sweep, whose counts add up the same way.  Same numbers as the
// one-workgroup path: min and + do not care how the members were grouped.
__device__ void ka_cons_votes_split(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, char* lds, const int R)
{
        const int tid = threadIdx.x;
        const int K = D.cons_K;
        const long long n = (long long)S.len_a + S.len_b + 8;
        const int unit = S.member / R, chunk = S.member % R;
        const bool active = unit < 2 * K;
        const bool is_rows = unit < K;
        const int anchor = is_rows ? unit : unit - K;
        const int node = (is_rows != (S.swapped != 0)) ? T.a : T.b;
        const int nmem = (node == T.a) ? T.nsip_a : T.nsip_b;
        const int dp_len = is_rows ? S.La : S.Lb;
        int* apos = is_rows ? S.apos_r : S.apos_c;
        float* conf = is_rows ? S.conf_r : S.conf_c;
        // the unit's table in the task's HBM scratch: rows-side units first ([K][La]), then the columns side ([K][Lb])
        char* hb = S.vote + (is_rows ? (long long)anchor * S.La : (long long)K * S.La + (long long)anchor * S.Lb) * 16;
        unsigned long long* hkey = (unsigned long long*)hb;
        unsigned int* hcnt = (unsigned int*)(hkey + dp_len);
        unsigned int* hagr = hcnt + dp_len;
        const bool leaf = nmem == 1;
        const bool voting = active && !leaf;
        unsigned long long* key = (unsigned long long*)lds;
        unsigned int* cnt = (unsigned int*)(key + dp_len);
        const int* members = D.sip + D.sip_off[node];
        const int m0 = (int)((long long)chunk * nmem / R), m1 = (int)((long long)(chunk + 1) * nmem / R);
        if (active && leaf && chunk == 0) {
                const int* map = D.cons_maps + D.cons_map_off[node] + (long long)anchor * dp_len;
                for (int i = tid; i < dp_len; i += KA_NT) { const int a = map[i]; apos[anchor * n + i] = a; conf[anchor * n + i] = (a >= 0) ? 1.0f : 0.0f; }
        }
        if (voting) {
                for (int x = chunk * KA_NT + tid; x < dp_len; x += R * KA_NT) { hkey[x] = ~0ull; hcnt[x] = 0u; hagr[x] = 0u; }
                for (int x = tid; x < dp_len; x += KA_NT) { key[x] = ~0ull; cnt[x] = 0u; }
                __syncthreads();
                ka_vote_sweep<1>(D, members, m0, m1, 0, key, cnt, nullptr, false, true, dp_len, 1, anchor, 1);
                __syncthreads();
        }
        ka_cluster_sync(S);                                                  // the HBM tables are initialised
        if (voting) {
                for (int x = tid; x < dp_len; x += KA_NT)
                        if (cnt[x]) { atomicMin(&hkey[x], key[x]); atomicAdd(&hcnt[x], cnt[x]); }
        }
        ka_cluster_sync(S);                                                  // first members and totals of every unit are complete
        if (voting) {
                for (int x = tid; x < dp_len; x += KA_NT) { key[x] = __hip_atomic_load(&hkey[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cnt[x] = 0u; }
                __syncthreads();
                ka_vote_sweep<1>(D, members, m0, m1, 1, key, cnt, nullptr, false, true, dp_len, 1, anchor, 1);
                __syncthreads();
                for (int x = tid; x < dp_len; x += KA_NT)
                        if (cnt[x] >> 16) atomicAdd(&hagr[x], cnt[x] >> 16);
        }
        ka_cluster_sync(S);                                                  // agreement counts complete
        if (voting) {
                for (int x = chunk * KA_NT + tid; x < dp_len; x += R * KA_NT) {
                        const unsigned long long kk = __hip_atomic_load(&hkey[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int tot = (int)__hip_atomic_load(&hcnt[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int ag = (int)__hip_atomic_load(&hagr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const long long o = (long long)anchor * n + x;
                        if (tot > 0 && ag > 0) { apos[o] = (int)(unsigned int)(kk & 0xffffffffull); conf[o] = (float)ag / (float)tot; }
                        else { apos[o] = -1; conf[o] = 0.0f; }
                }
        }
        __syncthreads();
}

template <bool LEAN, int NBK = KA_NB>
__device__ void ka_cons_votes(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, char* lds, const long long lds_bytes)
{
        const int tid = threadIdx.x;
        const int K = D.cons_K;
        const long long n = (long long)S.len_a + S.len_b + 8;           // stride of the per-anchor arrays
        if (!LEAN && K > 0 && S.G >= 4 * K && max(T.nsip_a, T.nsip_b) < 65536 && 12ll * max(S.La, S.Lb) <= lds_bytes) {
                ka_cons_votes_split(S, D, T, lds, S.G / (2 * K));
                return;
        }
        const int half = (S.G >= 2) ? (S.G >> 1) : 1;
        for (int side = 0; side < 2; ++side) {
                if (S.G >= 2 && side != S.member / half) continue;
                const int sub = (S.G >= 2) ? (S.member % half) : 0;
                const bool is_rows = (side == 0);
                const int node = (is_rows != (S.swapped != 0)) ? T.a : T.b;  // rows: a unless swapped
                const int nmem = (node == T.a) ? T.nsip_a : T.nsip_b;
                const int dp_len = is_rows ? S.La : S.Lb;
                int* apos = is_rows ? S.apos_r : S.apos_c;
                float* conf = is_rows ? S.conf_r : S.conf_c;
                // this workgroup's anchors: sub, sub + half, ...  (closed form: an indexed array would live in scratch
                // memory and put a scratch load in front of every gather)
                const int nk = (K - sub + half - 1) / half;
#define KS(b_) (sub + (b_) * half)
                if (nmem == 1) {
                        // leaf: direct lookup (a leaf's dp_len is its length)
                        for (int b = 0; b < nk; ++b) {
                                const int* map = D.cons_maps + D.cons_map_off[node] + (long long)KS(b) * dp_len;
                                for (int i = tid; i < dp_len; i += KA_NT) {
                                        const int a = map[i];
                                        apos[KS(b) * n + i] = a; conf[KS(b) * n + i] = (a >= 0) ? 1.0f : 0.0f;
                                }
                        }
                        continue;
                }
                if (LEAN) continue;                                      // lean levels hold leaf-leaf tasks only
                // a cell's `total` and `agree` counts share one 32-bit word (16 bits each) -- below 65536 members; from there on
                // `agree` has a word of its own (anchor_consistency.c:352-470 counts in ints)
                const bool wide = nmem >= 65536;
                const long long cell = wide ? 16 : 12;
                const int* members = D.sip + D.sip_off[node];
                int fit = (int)(lds_bytes / (cell * dp_len));             // anchors whose tables fit into LDS together
                const bool in_lds = fit >= 1;
                if (!in_lds) fit = nk;
                // (a sweep keeps the positions of its anchors in registers, KU x NBL of them: five anchors at a time, as in the kernels of
                // the default mode -- with ten, the round-4 form of the second set, the sweep alone spilled ~1000 VGPRs)
                constexpr int NBL = (NBK - 1 > 5) ? 5 : NBK - 1;
                if (fit > NBL) fit = NBL;
                for (int b0 = 0; b0 < nk; b0 += fit) {
                        const int nb = min(fit, nk - b0);
                        unsigned long long* key = in_lds ? (unsigned long long*)lds : (unsigned long long*)S.vote;
                        unsigned int* cnt = (unsigned int*)(key + (long long)nb * dp_len);
                        unsigned int* agr = cnt + (long long)nb * dp_len;        // (wide only)
                        for (int x = tid; x < nb * dp_len; x += KA_NT) { key[x] = ~0ull; cnt[x] = 0u; if (wide) agr[x] = 0u; }
                        __syncthreads();
                        for (int sweep = 0; sweep < 2; ++sweep) {
                                ka_vote_sweep<NBL>(D, members, 0, nmem, sweep, key, cnt, agr, wide, in_lds, dp_len, nb, KS(b0), half);
                                __syncthreads();
                        }
                        for (int x = tid; x < nb * dp_len; x += KA_NT) {
                                const int b = x / dp_len, c = x - b * dp_len;
                                unsigned long long kk;
                                unsigned int cc, ca = 0u;
                                if (in_lds) { kk = key[x]; cc = cnt[x]; if (wide) ca = agr[x]; }
                                else {
                                        kk = __hip_atomic_load(&key[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        cc = __hip_atomic_load(&cnt[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        if (wide) ca = __hip_atomic_load(&agr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                                const int tot = wide ? (int)cc : (int)(cc & 0xffffu), ag = wide ? (int)ca : (int)(cc >> 16);
                                const long long o = KS(b0 + b) * n + c;
                                if (tot > 0 && ag > 0) { apos[o] = (int)(unsigned int)(kk & 0xffffffffull); conf[o] = (float)ag / (float)tot; }
                                else { apos[o] = -1; conf[o] = 0.0f; }
                        }
                        __syncthreads();
                }
        }
}

#undef KS

// ------------------------------------------------------------------------------------------
// Round 5: CARRIED votes.  get_node_anchor_positions (anchor_consistency.c:352-470) counts, per anchor and profile column, over the
// node's members in sip order: best = the anchor position of the FIRST member that has one there, total = members that have one,
// agree = members whose position is best.  The member list of a merged node is its operands' lists, each REVERSED, a's first
// (aln_run.c:428-436: sip[c] = rev(sip[a]) ++ rev(sip[b])) -- the first voter of c in a column is the LAST voter of a there (of b where a
// has none), the last voter of c the FIRST voter of b (of a).  So a node carries, per anchor and column, both ends of its list:
//     f / fc = position of its first voter / voters at that position        (what the task that aligns the node reads: best, agree)
//     l / lc = position of its last voter / voters at that position
//     n      = voters
// and a column of c -- a column of a, of b, or one of each -- follows from its operands' without touching a member:
//     one side only (X):  f = l_X, fc = lc_X, l = f_X, lc = fc_X, n = n_X
//     both:               f = l_a, fc = lc_a + #{voters of b at l_a};   l = f_b, lc = fc_b + #{voters of a at f_b};   n = n_a + n_b
//     #{voters of X at p} = fc_X if p == f_X,  lc_X if p == l_X,  0 if the voters at f_X and l_X are all of X's voters,
//                           otherwise a COUNT over X's members: the cell is marked, and one sweep over the members' residues -- a
//                           column and K marks per residue, an atomic where a mark and the position match -- settles all marks.
// The same integers as the reference's loop over all members, member by member.  A node's table -- five planes [K][plen] of ints --
// lies behind its profile records in the arena (KaTreeDev::node_vote).  A task whose operands are leaves or carry tables reads its
// anchor positions from them (ka_cons_from_tables) instead of sweeping twice over every member and anchor with LDS atomics
// (ka_cons_votes: 0.25-0.45 ms of a task with ~2000 members; used for nodes that arrive without a table -- profiles handed over
// between ranks, injected profiles).
// MEASURED, OFF BY DEFAULT (KA_CARRY=1 switches it on; bit-identical, tests/test_gpu_stress.py runs it): the votes do shrink (root of a
// 4096-sequence tree: prep 351 -> 109 us) but in real alignments some cell of nearly every big node is marked, and the sweep that
// settles the marks -- global atomics instead of LDS ones -- costs what was saved (merge 105 -> 353 us): default-mode tree 20.6 ->
// 21.2 ms, the realignment tree of a `--precise` member 79.5 -> 78.6 ms, the same tree on one workgroup per task (shared context)
// 178 -> 220 ms (profiles/r05_carried_votes.log).  With few marks settled one by one (below; KA_CARRY=3 keeps the sweep): 20.7 / 75.4 /
// 220 ms -- the top nodes of a 4096-sequence tree have more marks than that path takes.
// ------------------------------------------------------------------------------------------
#define KA_VOTE_MARK 0x80000000u
struct KaVote { int f, fc, l, lc, n; };
__device__ __forceinline__ KaVote ka_vote_read(const KaTreeDev& D, const int node, const int* tab, const int plen, const int k, const int i)
{
        KaVote v;
        if (!tab) {                                                   // a leaf: its position map is its table
                v.f = v.l = D.cons_maps[D.cons_map_off[node] + (long long)k * plen + i];
                v.fc = v.lc = v.n = (v.f >= 0) ? 1 : 0;
                return v;
        }
        const long long K = D.cons_K, o = (long long)k * plen + i, pl = K * plen;
        v.f = tab[o];
        v.fc = (int)((unsigned int)tab[pl + o] & ~KA_VOTE_MARK);
        v.l = tab[2 * pl + o];
        v.lc = (int)((unsigned int)tab[3 * pl + o] & ~KA_VOTE_MARK);
        v.n = tab[4 * pl + o];
        return v;
}

__device__ __forceinline__ const int* ka_vote_table(const KaTreeDev& D, const int node, const int nmem)
{
        if (nmem == 1) return nullptr;
        return (const int*)(D.prof_arena + __hip_atomic_load(&D.node_vote[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// both operands are leaves or carry a table (the same answer in every workgroup of a cluster: the tables are final before the task starts)
__device__ __forceinline__ bool ka_votes_carried(const KaTreeDev& D, const KaTaskDesc& T)
{
        if (!D.carry) return false;
        const bool a = T.nsip_a == 1 || __hip_atomic_load(&D.node_vote[T.a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
        const bool b = T.nsip_b == 1 || __hip_atomic_load(&D.node_vote[T.b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
        return a && b;
}

// ka_cons_votes' result (apos / conf of both DP sides, all anchors) from the operands' tables; one workgroup
__device__ void ka_cons_from_tables(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T)
{
        const int tid = threadIdx.x;
        const int K = D.cons_K;
        const long long n = (long long)S.len_a + S.len_b + 8;
        for (int side = 0; side < 2; ++side) {
                const bool is_rows = (side == 0);
                const int node = (is_rows != (S.swapped != 0)) ? T.a : T.b;
                const int nmem = (node == T.a) ? T.nsip_a : T.nsip_b;
                const int dp_len = is_rows ? S.La : S.Lb;
                int* apos = is_rows ? S.apos_r : S.apos_c;
                float* conf = is_rows ? S.conf_r : S.conf_c;
                const int* tab = ka_vote_table(D, node, nmem);
                for (int x = tid; x < K * dp_len; x += KA_NT) {
                        const int k = x / dp_len, i = x - k * dp_len;
                        const KaVote v = ka_vote_read(D, node, tab, dp_len, k, i);
                        const long long o = (long long)k * n + i;
                        if (v.n > 0 && v.fc > 0) { apos[o] = v.f; conf[o] = (float)v.fc / (float)v.n; }
                        else { apos[o] = -1; conf[o] = 0.0f; }
                }
        }
        __syncthreads();
}

// voters of X at position p, from X's cell alone; false: only a count over X's members tells
__device__ __forceinline__ bool ka_vote_count_at(const KaVote& x, const int p, int& cnt)
{
        if (p == x.f) { cnt = x.fc; return true; }
        if (p == x.l) { cnt = x.lc; return true; }
        // nobody else left to ask: one group that is everybody (f == l), or two disjoint groups that are
        if ((x.f == x.l) ? (x.fc == x.n) : (x.fc + x.lc == x.n)) { cnt = 0; return true; }
        return false;
}

// The table of the merged node (all workgroups of the cluster; after ka_update_colof: the members' columns are the merged node's).
// vt: [5][K][alnlen] ints behind the node's profile records.
__device__ void ka_votes_merge(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, const int alnlen, int* vt)
{
        const int tid = threadIdx.x;
        const long long K = D.cons_K, pl = K * alnlen;
        const int* ta = ka_vote_table(D, T.a, T.nsip_a);
        const int* tb = ka_vote_table(D, T.b, T.nsip_b);
        // marked cells are also listed (in the scratch the counted votes use for their tables, 16 B per anchor and column: idle here): few
        // of them -- 0.1 % of the cells of a DSSim family -- are settled one by one (below), many by a sweep over the members
        int* mlist = (int*)S.vote;
        const int mcap = (pl < (1ll << 30)) ? (int)min(K * ((long long)S.len_a + S.len_b), (long long)(1 << 20)) : 0;
        int any = 0;                                                  // bit 0: cells that want b's members counted, bit 1: a's
        auto mark = [&](const long long x, const int type) {
                const int slot = atomicAdd(&S.ctl->vote_conf, 1);
                if (slot < mcap) mlist[slot] = (int)x | (type << 30);
                any |= 1 << type;
        };
        for (long long x = (long long)S.member * KA_NT + tid; x < pl; x += (long long)S.G * KA_NT) {
                const int k = (int)(x / alnlen), j = (int)(x - (long long)k * alnlen);
                const int ia = S.srcA[j + 1] - 1, ib = S.srcB[j + 1] - 1;
                KaVote A = { -1, 0, -1, 0, 0 }, B = { -1, 0, -1, 0, 0 };
                if (ia >= 0) A = ka_vote_read(D, T.a, ta, S.len_a, k, ia);
                if (ib >= 0) B = ka_vote_read(D, T.b, tb, S.len_b, k, ib);
                int f, l, n;
                unsigned int fc, lc;
                if (A.n == 0) { f = B.l; fc = (unsigned int)B.lc; l = B.f; lc = (unsigned int)B.fc; n = B.n; }
                else if (B.n == 0) { f = A.l; fc = (unsigned int)A.lc; l = A.f; lc = (unsigned int)A.fc; n = A.n; }
                else {
                        int cnt;
                        f = A.l; l = B.f; n = A.n + B.n;
                        if (ka_vote_count_at(B, f, cnt)) fc = (unsigned int)(A.lc + cnt); else { fc = (unsigned int)A.lc | KA_VOTE_MARK; mark(x, 0); }
                        if (ka_vote_count_at(A, l, cnt)) lc = (unsigned int)(B.fc + cnt); else { lc = (unsigned int)B.fc | KA_VOTE_MARK; mark(x, 1); }
                }
                vt[x] = f; vt[pl + x] = (int)fc; vt[2 * pl + x] = l; vt[3 * pl + x] = (int)lc; vt[4 * pl + x] = n;
        }
        if (any) __hip_atomic_fetch_or(&S.ctl->vote_types, any, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ka_cluster_sync(S);                                           // the table, the marks, and every member's merged columns
        const int nm = __hip_atomic_load(&S.ctl->vote_conf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nm == 0) return;
        const int todo = __hip_atomic_load(&S.ctl->vote_types, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
                // few marks: every (marked cell, member of the operand it asks about) is one binary search in the member's residue -> column
                // table (columns grow with the residue) -- at most eight per thread, otherwise the sweep
                const long long NM = max(T.nsip_a, T.nsip_b);
                if (nm <= mcap && (long long)nm * NM <= 8ll * S.G * KA_NT && !(D.carry & 2)) {
                        for (long long idx = (long long)S.member * KA_NT + tid; idx < (long long)nm * NM; idx += (long long)S.G * KA_NT) {
                                const int e = mlist[idx / NM], m = (int)(idx % NM);
                                const int type = (e >> 30) & 1;
                                const long long x = e & 0x3fffffff;
                                const int nX = type ? T.nsip_a : T.nsip_b;
                                if (m >= nX) continue;
                                const int si = (D.sip + D.sip_off[type ? T.a : T.b])[m];
                                const int* col = D.colof + D.seq_off[si];
                                const int len = D.node_len[si];
                                const int k = (int)(x / alnlen), j = (int)(x - (long long)k * alnlen);
                                int lo = 0, hi = len - 1, p = -1;
                                while (lo <= hi) {
                                        const int mid = (lo + hi) >> 1, c = col[mid];
                                        if (c == j) { p = mid; break; }
                                        if (c < j) lo = mid + 1; else hi = mid - 1;
                                }
                                if (p < 0) continue;
                                const int a = D.cons_maps[D.cons_map_off[si] + (long long)k * len + p];
                                if (a >= 0 && a == vt[(type ? 2 * pl : 0) + x]) atomicAdd((unsigned int*)&vt[(type ? 3 * pl : pl) + x], 1u);
                        }
                        return;
                }
        }
        // marked cells: b's members are counted at c's first position (plane 1 at plane 0), a's at c's last (plane 3 at plane 2).
        // Gathers: four residues per lane and five anchors' marks in flight.  (Eight residues -- forty loads -- tipped the register allocation
        // of the whole task kernel, everything is inlined into it: 1514 spilled VGPRs against 568, the strips of every default-mode job 14 %
        // slower, carried votes or not.  hipcc ignores noinline.)
        const int lane = tid & 63, wave = tid >> 6;
        const int na = (todo & 2) ? T.nsip_a : 0, nb = (todo & 1) ? T.nsip_b : 0;
        const int* ma = D.sip + D.sip_off[T.a];
        const int* mb = D.sip + D.sip_off[T.b];
        constexpr int KU = 4, KB = 5;
        for (int m = S.member * KA_NW + wave; m < na + nb; m += KA_NW * S.G) {
                const bool isb = m >= na;
                const int si = isb ? mb[m - na] : ma[m];
                const int* col = D.colof + D.seq_off[si];
                const int len = D.node_len[si];
                const int* map = D.cons_maps + D.cons_map_off[si];
                const int* pos = vt + (isb ? 0 : 2 * pl);
                int* cntp = vt + (isb ? pl : 3 * pl);
                for (int p0 = lane; p0 < len; p0 += 64 * KU) {
                        int jj[KU];
#pragma unroll
                        for (int u = 0; u < KU; ++u) { const int pp = p0 + 64 * u; jj[u] = (pp < len) ? col[pp] : -1; }
                        for (int k0 = 0; k0 < (int)K; k0 += KB) {
                                int mk[KU][KB];
#pragma unroll
                                for (int u = 0; u < KU; ++u)
#pragma unroll
                                        for (int q = 0; q < KB; ++q)
                                                mk[u][q] = (jj[u] >= 0 && k0 + q < (int)K) ? cntp[(long long)(k0 + q) * alnlen + jj[u]] : 0;
#pragma unroll
                                for (int u = 0; u < KU; ++u)
#pragma unroll
                                        for (int q = 0; q < KB; ++q) {
                                                if (mk[u][q] >= 0) continue;
                                                const long long o = (long long)(k0 + q) * alnlen + jj[u];
                                                const int a = map[(long long)(k0 + q) * len + p0 + 64 * u];
                                                if (a >= 0 && a == pos[o]) atomicAdd((unsigned int*)&cntp[o], 1u);
                                        }
                        }
                }
        }
}

// anchor_consistency_get_bonus_profile in sparse form (first workgroup of the cluster, after the votes).
// After it S.ent[row][0..KA_NB) holds the row's non-zero bonus cells with distinct columns: entries of
// different anchors that hit the same cell are summed in anchor order (the dense matrix accumulates
// k = 0..K-1 into a zeroed cell), and slot KA_NB-1 carries the cell the reference reaches when a forward
// pass indexes column Lb of row i -- flat index i*Lb + Lb is cell (i+1, 0) (aln_seqseq.c:83-85 uses the
// 1-based column).
template <int NBK = KA_NB>
__device__ void ka_cons_entries(TaskShared& S, const KaTreeDev& D)
{
        const int tid = threadIdx.x;
        const int K = D.cons_K;
        const int rows = S.La, cols = S.Lb;
        const long long n = (long long)S.len_a + S.len_b + 8;
        const int ml = D.cons_maxlen + 8;
        const float paw = D.cons_paw;
        // inverse maps anchor position -> column; of several columns the last one wins (:521-526)
        for (int x = tid; x < K * ml; x += KA_NT) S.invj[x] = -1;
        __syncthreads();
        for (int x = tid; x < K * cols; x += KA_NT) {
                const int k = x / cols, j = x - k * cols;
                const int a = S.apos_c[k * n + j];
                if (a >= 0) atomicMax(&S.invj[k * ml + a], j);
        }
        __syncthreads();
        if constexpr (KaBonus<NBK>::STREAM) {
                // the streamed set (any K up to KA_CONS_MAX_ANCHORS): a row's entries are collected in the row's own slice of the task's
                // scratch -- [1] = (INT_MIN, count), the entries from [2] on -- not in per-thread arrays of NBK entries (a kilobyte of
                // private memory per lane at 128 anchors); sorted below, once the wrap-around entry has joined them
                for (int i = tid; i < rows; i += KA_NT) {
                        int2* e = S.ent + (long long)i * NBK;
                        int cnt = 0;
                        for (int k = 0; k < K; ++k) {
                                const int a = S.apos_r[k * n + i];
                                if (a < 0) continue;
                                const int bj = __hip_atomic_load(&S.invj[k * ml + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (bj < 0) continue;
                                const float val = paw * S.conf_r[k * n + i] * S.conf_c[k * n + bj];                // :534-535
                                int hit = -1;
                                for (int m = 0; m < cnt; ++m) if (e[2 + m].x == bj) hit = m;
                                if (hit >= 0) e[2 + hit].y = __float_as_int(__int_as_float(e[2 + hit].y) + val);
                                else { e[2 + cnt] = make_int2(bj, __float_as_int(0.0f + val)); ++cnt; }
                        }
                        e[1] = make_int2((int)0x80000000, cnt);
                }
        } else
        for (int i = tid; i < rows; i += KA_NT) {
                int mc[NBK];
                float mv[NBK];
                int cnt = 0;
                for (int k = 0; k < K; ++k) {
                        const int a = S.apos_r[k * n + i];
                        if (a < 0) continue;
                        const int bj = __hip_atomic_load(&S.invj[k * ml + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (bj < 0) continue;
                        const float val = paw * S.conf_r[k * n + i] * S.conf_c[k * n + bj];                // :534-535
                        int hit = -1;
                        for (int m = 0; m < cnt; ++m) if (mc[m] == bj) hit = m;
                        if (hit >= 0) mv[hit] += val;
                        else { mc[cnt] = bj; mv[cnt] = 0.0f + val; ++cnt; }
                }
                int2* e = S.ent + (long long)i * NBK;
                for (int m = 0; m < NBK - 1; ++m) e[m] = (m < cnt) ? make_int2(mc[m], __float_as_int(mv[m])) : make_int2(-1, 0);
        }
        __syncthreads();
        if constexpr (KaBonus<NBK>::STREAM) {
                // the wrap-around entry of row i -- what row i + 1 holds at column 0 (see above) -- first into the pad slot [0] (the rows
                // still read each other's unsorted lists), then, behind a barrier, into the list, which is sorted by column and closed
                for (int i = tid; i < rows; i += KA_NT) {
                        int2 w = make_int2(-1, 0);
                        if (i + 1 < rows) {
                                const int2* nx = S.ent + (long long)(i + 1) * NBK;
                                const int nc = nx[1].y;
                                for (int m = 0; m < nc; ++m) if (nx[2 + m].x == 0) w = make_int2(cols, nx[2 + m].y);
                        }
                        S.ent[(long long)i * NBK] = w;
                }
                __syncthreads();
                for (int i = tid; i < rows; i += KA_NT) {
                        int2* e = S.ent + (long long)i * NBK;
                        int cnt = e[1].y;
                        if (e[0].x >= 0) { e[2 + cnt] = e[0]; ++cnt; }
                        for (int m = 1; m < cnt; ++m) {                           // insertion sort by column (distinct columns, <= K + 1 entries)
                                const int2 x = e[2 + m];
                                int q = m - 1;
                                while (q >= 0 && e[2 + q].x > x.x) { e[2 + q + 1] = e[2 + q]; --q; }
                                e[2 + q + 1] = x;
                        }
                        e[1] = make_int2((int)0x80000000, cnt);
                        e[2 + cnt] = make_int2(0x7fffffff, 0);
                }
                __syncthreads();
                return;
        }
        for (int i = tid; i < rows; i += KA_NT) {
                int2 w = make_int2(-1, 0);
                if (i + 1 < rows) {
                        const int2* nx = S.ent + (long long)(i + 1) * NBK;
                        for (int m = 0; m < NBK - 1; ++m) if (nx[m].x == 0) w = make_int2(cols, nx[m].y);
                }
                S.ent[(long long)i * NBK + NBK - 1] = w;
        }
        __syncthreads();
}

// make_seq / update_gaps (weave_alignment.c:41-112) in the device's form: after the merge of a and
// b, residue p of a member of a moves from column col to amap[col].  S.raw / S.raw2 (dead after the
// path coding) receive amap / bmap.  All workgroups of the cluster share the members.
__device__ void ka_update_colof(TaskShared& S, const KaTreeDev& D, const KaTaskDesc& T, const int alnlen)
{
        const int tid = threadIdx.x;
        if (S.member == 0) {
                for (int j = 1 + tid; j <= alnlen; j += KA_NT) {
                        const int ia = S.srcA[j], ib = S.srcB[j];
                        if (ia >= 1) S.raw[ia - 1] = j - 1;
                        if (ib >= 1) S.raw2[ib - 1] = j - 1;
                }
        }
        ka_cluster_sync(S);
        const int lane = tid & 63, wave = tid >> 6;
        const int na = T.nsip_a, nb = T.nsip_b;
        const int* ma = D.sip + D.sip_off[T.a];
        const int* mb = D.sip + D.sip_off[T.b];
        for (int m = S.member * KA_NW + wave; m < na + nb; m += KA_NW * S.G) {
                const int si = (m < na) ? ma[m] : mb[m - na];
                const int* mp = (m < na) ? S.raw : S.raw2;
                int* col = D.colof + D.seq_off[si];
                const int len = D.node_len[si];
                for (int p = lane; p < len; p += 64) col[p] = mp[col[p]];
        }
}
