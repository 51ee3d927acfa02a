// ka_path.h -- P3: mirror + coding of the raw path (first pass and refinement).
// One of the text sections of the task kernels, included by ka_kernels.hip in this order: ka_shared.h, ka_pass.h, ka_best.h,
// ka_subtree.h, ka_wstrip.h, ka_meetup.h, ka_hirschberg.h, ka_path.h, ka_profile.h, ka_task.h.  Not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------
// P3: mirror + coding of the raw path by the whole workgroup (see
// oracle/kalign_oracle.c:ko_code_path for the as-executed semantics of add_gap_info_to_path_n).
// Row i of the (a-indexed) raw path emits g_i gap-in-a ops followed by one op (match or
// gap-in-b); two block-wide prefix sums (ops emitted, b positions consumed) place every row's
// ops independently.  Also records, per output column, which record of profile a / b feeds
// it (srcA/srcB, -1 = none) for the parallel update_n.  `lds` = 2*blockDim.x+4 ints of scratch.
// ------------------------------------------------------------------------------------------
__device__ void ka_code_path(TaskShared& S, int* lds)
{
        const int tid = threadIdx.x;
        const int len_a = S.len_a, len_b = S.len_b;
        const int* raw = S.raw;
        if (S.swapped) {
                int* r2 = S.raw2;
                for (int i = tid; i < len_a + 2; i += KA_NT) r2[i] = -1;
                __syncthreads();
                for (int i = 1 + tid; i <= len_b; i += KA_NT) { const int c = S.raw[i]; if (c != -1) r2[c] = i; }
                __syncthreads();
                raw = r2;
        }
        int* o = S.coded;
        int* tot_ops = lds;
        int* tot_b = lds + KA_NT;
        int* zmin = lds + 2 * KA_NT;
        int* zmax = zmin + 1;
        // rows [lo, hi) of this thread (1-based rows 1..len_a)
        const int per = (len_a + KA_NT - 1) / KA_NT;
        const int lo = 1 + tid * per, hi = min(len_a + 1, lo + per);
        auto row_gaps = [&](int i, int cur, int prev) -> int {
                // gap-in-a ops emitted before row i's own op (aln_setup.c:145-188)
                if (cur == -1) return 0;
                if (i == 1) return cur - 1;
                return (cur - 1 != prev && prev != -1) ? (cur - prev - 1) : 0;
        };
        int nops = 0, nb = 0;
        for (int i = lo; i < hi; ++i) {
                const int cur = raw[i], prev = (i > 1) ? raw[i - 1] : -1;
                const int g = row_gaps(i, cur, prev);
                nops += g + 1;
                nb += g + (cur != -1 ? 1 : 0);
        }
        // exclusive prefix sums of (ops, b positions) over the threads: wave scan + one pass over the wave totals
        const int lane_ = tid & 63, wave_ = tid >> 6;
        int sc_ops = nops, sc_b = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
                const int y1 = __shfl_up(sc_ops, d, 64), y2 = __shfl_up(sc_b, d, 64);
                if (lane_ >= d) { sc_ops += y1; sc_b += y2; }
        }
        if (lane_ == 63) { tot_ops[wave_] = sc_ops; tot_b[wave_] = sc_b; }
        if (tid == 0) { *zmin = 0x7fffffff; *zmax = 0; }
        __syncthreads();
        int off = sc_ops - nops, offb = sc_b - nb, all_ops = 0, total_b = 0;
        for (int k = 0; k < KA_NW; ++k) {
                const int a = tot_ops[k], b2 = tot_b[k];
                if (k < wave_) { off += a; offb += b2; }
                all_ops += a; total_b += b2;
        }
        // trailing gap-in-a run (aln_setup.c:180-186)
        const int last = raw[len_a];
        const int tail = (last != -1 && last < len_b) ? (len_b - last) : 0;
        const int alnlen = all_ops + tail;
        int my_zmin = 0x7fffffff, my_zmax = 0;
        int j = 1 + off, rb = 1 + offb;
        for (int i = lo; i < hi; ++i) {
                const int cur = raw[i], prev = (i > 1) ? raw[i - 1] : -1;
                const int g = row_gaps(i, cur, prev);
                for (int k = 0; k < g; ++k) { o[j] = 1; S.srcA[j] = -1; S.srcB[j] = rb++; ++j; }
                if (cur == -1) { o[j] = 2; S.srcA[j] = i; S.srcB[j] = -1; }
                else { o[j] = 0; S.srcA[j] = i; S.srcB[j] = rb++; my_zmin = min(my_zmin, j); my_zmax = max(my_zmax, j); }
                ++j;
        }
        if (my_zmax > 0) { atomicMin(zmin, my_zmin); atomicMax(zmax, my_zmax); }
        for (int k = tid; k < tail; k += KA_NT) { o[1 + all_ops + k] = 1; S.srcA[1 + all_ops + k] = -1; S.srcB[1 + all_ops + k] = 1 + total_b + k; }
        if (tid == 0) { o[0] = alnlen; o[alnlen + 1] = 3; S.ctl->alnlen = alnlen; }
        __syncthreads();
        // terminal-run flag (aln_setup.c:209-219): everything before the first and after the last
        // match column; the 4/8/16 flag loop never executes in the reference
        const int z1 = *zmin, z2 = *zmax;
        for (int c = 1 + tid; c <= alnlen; c += KA_NT) if (c < z1 || c > z2) o[c] |= 32;
        __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Path coding of the refinement pass: convert_raw_path (aln_refine.c:591-672) by the whole workgroup.  Differences to
// add_gap_info_to_path_n (ka_code_path): the gap-in-a run in front of a match is counted from the last MATCHED column
// (a prefix maximum over the rows), and the open / extend / close flags are real: 4 = first op of a gap run that
// follows a match, 8 = continuation of a run of the same kind, 16 = last op before a match (an op carrying 8 gets +8,
// which is 16 as well), 32 = runs before the first / after the last match.  Every flag depends on the op kinds of the
// two neighbours only.  `lds` = 3*blockDim.x+4 ints of scratch.
// ------------------------------------------------------------------------------------------
__device__ void ka_code_path_refine(TaskShared& S, int* lds)
{
        const int tid = threadIdx.x;
        const int len_a = S.len_a, len_b = S.len_b;
        const int* raw = S.raw;
        if (S.swapped) {
                int* r2 = S.raw2;
                for (int i = tid; i < len_a + 2; i += KA_NT) r2[i] = -1;
                __syncthreads();
                for (int i = 1 + tid; i <= len_b; i += KA_NT) { const int c = S.raw[i]; if (c != -1) r2[c] = i; }
                __syncthreads();
                raw = r2;
        }
        int* o = S.coded;
        int* tot_ops = lds;
        int* tot_b = lds + KA_NT;
        int* tot_m = lds + 2 * KA_NT;
        int* zmin = lds + 3 * KA_NT;
        int* zmax = zmin + 1;
        const int per = (len_a + KA_NT - 1) / KA_NT;
        const int lo = 1 + tid * per, hi = min(len_a + 1, lo + per);
        // last matched column before this thread's rows: exclusive prefix maximum over the threads
        int mymax = 0;
        for (int i = lo; i < hi; ++i) mymax = max(mymax, raw[i]);
        const int lane_ = tid & 63, wave_ = tid >> 6;
        int sc_m = mymax;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(sc_m, d, 64); if (lane_ >= d) sc_m = max(sc_m, y); }
        if (lane_ == 63) tot_m[wave_] = sc_m;
        if (tid == 0) { *zmin = 0x7fffffff; *zmax = 0; }
        __syncthreads();
        int blast = __shfl_up(sc_m, 1, 64);
        if (lane_ == 0) blast = 0;
        for (int k = 0; k < wave_; ++k) blast = max(blast, tot_m[k]);
        blast = max(blast, 0);
        // ops and b positions of this thread's rows
        int nops = 0, nb = 0;
        {
                int bl = blast;
                for (int i = lo; i < hi; ++i) {
                        const int cur = raw[i];
                        if (cur == -1) { nops += 1; }
                        else { const int gpre = max(cur - bl - 1, 0); nops += gpre + 1; nb += gpre + 1; bl = cur; }
                }
        }
        int sc_ops = nops, sc_b = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
                const int y1 = __shfl_up(sc_ops, d, 64), y2 = __shfl_up(sc_b, d, 64);
                if (lane_ >= d) { sc_ops += y1; sc_b += y2; }
        }
        if (lane_ == 63) { tot_ops[wave_] = sc_ops; tot_b[wave_] = sc_b; }
        __syncthreads();
        int off = sc_ops - nops, offb = sc_b - nb, all_ops = 0, total_b = 0, all_max = 0;
        for (int k = 0; k < KA_NW; ++k) {
                const int a = tot_ops[k], b2 = tot_b[k];
                if (k < wave_) { off += a; offb += b2; }
                all_ops += a; total_b += b2; all_max = max(all_max, tot_m[k]);
        }
        all_max = max(all_max, 0);
        const int tail = len_b - all_max;                                 // trailing gap-in-a run (:630-634)
        const int alnlen = all_ops + tail;
        int my_zmin = 0x7fffffff, my_zmax = 0;
        {
                int j = 1 + off, rb = 1 + offb, bl = blast;
                for (int i = lo; i < hi; ++i) {
                        const int cur = raw[i];
                        if (cur == -1) { o[j] = 2; S.srcA[j] = i; S.srcB[j] = -1; ++j; }
                        else {
                                const int gpre = max(cur - bl - 1, 0);
                                for (int k = 0; k < gpre; ++k) { o[j] = 1; S.srcA[j] = -1; S.srcB[j] = rb++; ++j; }
                                o[j] = 0; S.srcA[j] = i; S.srcB[j] = rb++;
                                my_zmin = min(my_zmin, j); my_zmax = max(my_zmax, j);
                                ++j; bl = cur;
                        }
                }
        }
        if (my_zmax > 0) { atomicMin(zmin, my_zmin); atomicMax(zmax, my_zmax); }
        for (int k = tid; k < tail; k += KA_NT) { o[1 + all_ops + k] = 1; S.srcA[1 + all_ops + k] = -1; S.srcB[1 + all_ops + k] = 1 + total_b + k; }
        if (tid == 0) { o[0] = alnlen; o[alnlen + 1] = 3; S.ctl->alnlen = alnlen; }
        __syncthreads();
        const int z1 = *zmin, z2 = *zmax;
        // flags: the op kinds are final; every position reads its neighbours' kinds (low two bits) and writes itself
        for (int c = 1 + tid; c <= alnlen; c += KA_NT) {
                const int t = o[c] & 3;
                int v = t;
                if (t != 0) {
                        if (c >= 2) {
                                const int tp = o[c - 1] & 3;
                                if (tp == 0) v |= 4; else if (tp == t) v |= 8;
                        }
                        if (c <= alnlen - 1 && (o[c + 1] & 3) == 0) { if (v & 8) v += 8; else v |= 16; }
                }
                if (c < z1 || c > z2) v |= 32;
                S.raw2[c] = v;                                            // (raw2 is free once the path is mirrored / coded)
        }
        __syncthreads();
        for (int c = 1 + tid; c <= alnlen; c += KA_NT) o[c] = S.raw2[c];
        __syncthreads();
}
