// ka_kernels.hip -- CDNA4 (gfx950) kernels for Kalign's progressive-alignment hot path.
//
// One workgroup aligns one pairwise task (a, b) -> c end to end:
//   P1  operand preparation   make_profile_n / set_gap_penalties_n   (aln_setup.c:40-119)
//   P2  Hirschberg recursion  aln_runner / aln_continue              (aln_controller.c:21-436)
//         level-synchronous inside the workgroup: every wave pulls (sub-problem, direction)
//         passes of the current recursion level, then the waves run the meetups and emit
//         the next level's sub-problems
//       each pass is an anti-diagonal wavefront: lane l owns DP row u0+l of a 64-row strip
//       and walks the columns one step behind lane l-1; the three cell states move to the
//       next lane with a DPP wave shift (v_mov_b32_dpp wave_shr:1)
//   P3  path post-processing  mirror_path_n / add_gap_info_to_path_n (aln_setup.c:121-228,438-462)
//   P4  profile merge         update_n                               (aln_setup.c:230-436)
//
// Arithmetic is IEEE binary32 in the reference's source order with NO contraction
// (compile with -ffp-contract=off): the traceback is an argmax over float sums and must be
// bit-identical to the CPU reference (hard part 1).
//
// The formulation of a pass over (u, v) = (row counter, column counter) is the same as
// oracle/kalign_oracle.c:ko_pass; see there for the mapping to the reference's six functions.
#include <hip/hip_runtime.h>
#include "ka_device.h"
// #define KA_TRACE_STRIP 1   // per-step breadcrumbs into D.trace (debugging hangs)

#include "ka_shared.h"        // TaskShared, helpers
#include "ka_pass.h"          // ka_strip, ka_packed: the DP passes
#include "ka_best.h"
#include "ka_subtree.h"       // wave-local subtrees
#include "ka_wstrip.h"        // strips with helper waves
#include "ka_lstrip.h"        // the lean strip of the throughput kernel (unit 10)
#include "ka_meetup.h"
#include "ka_hirschberg.h"
#include "ka_path.h"
#include "ka_profile.h"
#include "ka_task.h"

// The kernels are compiled as four translation units from this one file (-DKA_UNIT=0..3, csrc/Makefile): every
// instantiation of ka_task_body takes about a minute of compile time, the units build in parallel.
//   unit 0: ka_task_kernel            unit 1: ka_task_kernel_cons
//   unit 2: the two half kernels      unit 3: the two lean kernels + ka_pair_kernel
#ifndef KA_UNIT
#error "compile with -DKA_UNIT=0..9 (see csrc/Makefile)"
#endif

// more than 64 KiB of dynamic LDS needs an explicit opt-in per kernel
template <typename K>
static hipError_t ka_optin(K kernel, int bytes, bool* done)
{
        if (*done) return hipSuccess;
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) *done = true;
        return e;
}

#if KA_UNIT == 0
__global__ __launch_bounds__(KA_BLOCK) void ka_task_kernel(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<false, 0>(D, blocks, chain);
}
extern "C" void ka_unit0_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel, KA_LDS_TOTAL, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_task_kernel, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, chain);
}
extern "C" long long ka_scratch_bytes_host(long long la, long long lb, long long cons_maxlen) { return ka_scratch_bytes(la, lb, cons_maxlen); }
// ... of the kernels that walk a DP row's bonus entries (KA_NB_BIG entries per row, K-sized anchor tables: `--consistency K`, 5 < K <= 128)
extern "C" long long ka_scratch_bytes_host_big(long long la, long long lb, long long cons_maxlen, long long k_anchors) { return ka_scratch_bytes(la, lb, cons_maxlen, 1, false, false, KA_NB_BIG, k_anchors); }
extern "C" long long ka_ctl_bytes_host(void) { return (long long)sizeof(KaCtl); }
extern "C" int ka_max_g_host(void) { return KA_MAX_G; }
#endif

#if KA_UNIT == 4
// refinement pass (units 4 and 5: one kernel each -- they are the longest compiles of the library, side by side they halve the
// build's critical path): one workgroup per edge, per-level launches
extern "C" void ka_unit5_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream);
__global__ __launch_bounds__(KA_BLOCK) void ka_refine_kernel(const KaTreeDev D, const int2* __restrict__ blocks, const int unused)
{
        const int2 blk = blocks[blockIdx.x];
        if (blk.x >= 0) ka_task_body_refine<0>(D, blk.x, blk.y & 0xff, blk.y >> 8);
}
extern "C" void ka_unit4_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, hipStream_t stream)
{
        static bool done0 = false;
        if (cons) { ka_unit5_launch(D, blocks_dev, nblocks, stream); return; }
        if (ka_optin(ka_refine_kernel, KA_LDS_TOTAL, &done0) != hipSuccess) return;
        hipLaunchKernelGGL(ka_refine_kernel, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, 0);
}
#endif

#if KA_UNIT == 5
// the refinement pass with the anchor-consistency bonus
__global__ __launch_bounds__(KA_BLOCK) void ka_refine_kernel_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int unused)
{
        const int2 blk = blocks[blockIdx.x];
        if (blk.x >= 0) ka_task_body_refine<KA_NB>(D, blk.x, blk.y & 0xff, blk.y >> 8);
}
extern "C" void ka_unit5_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream)
{
        static bool done1 = false;
        if (ka_optin(ka_refine_kernel_cons, KA_LDS_TOTAL, &done1) != hipSuccess) return;
        hipLaunchKernelGGL(ka_refine_kernel_cons, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, 0);
}
#endif

#if KA_UNIT == 1
// the same with the anchor-consistency bonus (default mode of the reference's CLI)
__global__ __launch_bounds__(KA_BLOCK) void ka_task_kernel_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<false, KA_NB>(D, blocks, chain);
}
extern "C" void ka_unit1_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel_cons, KA_LDS_TOTAL, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_task_kernel_cons, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, chain);
}
#endif

#if KA_UNIT == 2
// Throughput variant for levels with more tasks than CUs (big trees, forests): 4 waves, 4 rings -> TWO workgroups per
// CU.  The four strip waves of one task keep a CU's SIMDs busy only part of the time (pipeline fill and drain, deep
// recursion levels, meetups); a second resident task fills the holes.  Latency per task is no better -- levels
// with at most one task per CU use the 8-wave kernel.
// nqueue > 0: a queued launch over the first nqueue entries of `blocks` (ka_task_queue_entry); else one workgroup per entry
__global__ __launch_bounds__(KA_HALF_BLOCK, 2) void ka_task_kernel_half(const KaTreeDev D, const int2* __restrict__ blocks, const int nqueue)
{
        ka_task_queue_entry<false, 0>(D, blocks, nqueue);
}
__global__ __launch_bounds__(KA_HALF_BLOCK, 2) void ka_task_kernel_half_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int nqueue)
{
        ka_task_queue_entry<false, KA_NB>(D, blocks, nqueue);
}
// nqueue > 0: `nblocks` workgroups share the `nqueue` tasks listed in blocks_dev
extern "C" void ka_unit2_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, int nqueue, hipStream_t stream)
{
        static bool done0 = false, done1 = false;
        // a queued launch may run on narrower workgroups (KaTreeDev::qw waves: same kernel, a ring per wave, more workgroups per CU)
        const int qw = (nqueue > 0 && D->qw >= 1 && D->qw <= KA_HALF_BLOCK / 64) ? D->qw : KA_HALF_BLOCK / 64;
        const int qlds = KA_LDS_WAVES + qw * KA_WAVE_LDS;
        if (cons) {
                if (ka_optin(ka_task_kernel_half_cons, KA_LDS_HALF, &done1) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_half_cons, dim3(nblocks), dim3(64 * qw), qlds, stream, *D, blocks_dev, nqueue);
        } else {
                if (ka_optin(ka_task_kernel_half, KA_LDS_HALF, &done0) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_half, dim3(nblocks), dim3(64 * qw), qlds, stream, *D, blocks_dev, nqueue);
        }
}
#endif

#if KA_UNIT == 10
// THE THROUGHPUT KERNEL (round 6; compiled with -DKA_TP=1): the queued launch, the levels of forests and of shared contexts -- wherever
// there are more tasks than workgroup slots and the rate is slots / latency.  The task body of the 4-wave kernel with
// 11 KB of LDS per wave instead of 18 (ka_lstrip.h: an 80-column record-major ring; ka_pass.h KA_TP) and a register budget of 168:
// THREE workgroups per CU where unit 2 has two.  Fast mode, alphabets without B / Z / X (the host picks the unit: ka_tp_ok).
__global__ __launch_bounds__(KA_HALF_BLOCK, 3) void ka_task_kernel_tp(const KaTreeDev D, const int2* __restrict__ blocks, const int nqueue)
{
        ka_task_queue_entry<false, 0>(D, blocks, nqueue);
}
// nqueue > 0: `nblocks` workgroups share the `nqueue` tasks listed in blocks_dev; 0: one workgroup per entry
static long long ka_tp_launches = 0;
extern "C" long long ka_debug_tp_launches(void) { return ka_tp_launches; }
extern "C" void ka_unit10_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int nqueue, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel_tp, KA_LDS_HALF, &done) != hipSuccess) return;
        ka_tp_launches += 1;
        hipLaunchKernelGGL(ka_task_kernel_tp, dim3(nblocks), dim3(KA_HALF_BLOCK), KA_LDS_HALF, stream, *D, blocks_dev, nqueue);
}
#endif

#if KA_UNIT == 3
// (the second launch-bound is waves per SIMD: 4 -> <=128 VGPRs -> two 8-wave workgroups per CU)
__global__ __launch_bounds__(KA_LEAN_BLOCK, 4) void ka_task_kernel_lean(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<true, 0>(D, blocks, 0);
}
// the bonus entries cost ~40 VGPRs: 4 waves, 3 waves per SIMD (<=168 VGPRs) -> three workgroups per CU
__global__ __launch_bounds__(KA_PAIR_BLOCK, 3) void ka_task_kernel_lean_cons(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<true, KA_NB>(D, blocks, 0);
}
// the same body with 4 waves: four workgroups per CU, the shape ka_pair_kernel runs the same alignments in (experiment: KA_LEAN4=1)
__global__ __launch_bounds__(KA_PAIR_BLOCK, 4) void ka_task_kernel_lean4(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<true, 0>(D, blocks, 0);
}
extern "C" void ka_unit3_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int cons, hipStream_t stream)
{
        static bool done0 = false, done1 = false, done2 = false;
        if (!cons && D->lean4) {
                if (ka_optin(ka_task_kernel_lean4, KA_LDS_PAIR, &done2) != hipSuccess) return;
                const int lw = (D->lw >= 1 && D->lw <= KA_PAIR_BLOCK / 64) ? D->lw : KA_PAIR_BLOCK / 64;
                hipLaunchKernelGGL(ka_task_kernel_lean4, dim3(nblocks), dim3(64 * lw), KA_LDS_WAVES + KA_LEAN_SCRATCH(64 * lw) + lw * KA_WAVE_LDS_LEAN, stream, *D, blocks_dev, 0);
                return;
        }
        if (cons) {
                if (ka_optin(ka_task_kernel_lean_cons, KA_LDS_PAIR, &done1) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_lean_cons, dim3(nblocks), dim3(KA_PAIR_BLOCK), KA_LDS_PAIR, stream, *D, blocks_dev, 0);
        } else {
                if (ka_optin(ka_task_kernel_lean, KA_LDS_LEAN, &done0) != hipSuccess) return;
                hipLaunchKernelGGL(ka_task_kernel_lean, dim3(nblocks), dim3(KA_LEAN_BLOCK), KA_LDS_LEAN, stream, *D, blocks_dev, 0);
        }
}

// ------------------------------------------------------------------------------------------
// Batch of independent seq-seq alignments (pairwise_align_map, anchor_consistency.c:19-120)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KA_PAIR_BLOCK, 4) void ka_pair_kernel(const KaPairDev P)
{
        extern __shared__ __attribute__((aligned(16))) char ka_smem[];
        TaskShared& S = *(TaskShared*)ka_smem;
        float* tss = (float*)(ka_smem + KA_LDS_TSS);
        char* lds_waves = ka_smem + KA_LDS_WAVES;
        const int k = blockIdx.x;
        const int tid = threadIdx.x;
        if (tid == 0) {
                const int i = P.ia[k], j = P.ib[k];
                const int len_i = P.seq_len[i], len_j = P.seq_len[j];
                const int swapped = !(len_i <= len_j);
                S.ctl = &S.ctl_lds; S.G = 1; S.member = 0; S.bar_phase = 0; S.srows = KA_STRIP_ROWS; S.q1_lvl = 0; S.lvl_srows[0] = KA_STRIP_ROWS; S.lvl_srows[1] = KA_STRIP_ROWS;
                S.reuse_ok = P.reuse ? 1 : 0;
                S.sub_ok = 1; S.rec_on = 0; S.nres_t = 23; S.sub_stride = KA_WAVE_LDS_LEAN; S.sub_base = lds_waves + KA_LEAN_SCRATCH(KA_NT); S.sub_tm = 0; S.mw_ok = 1;
                S.lctl = S.ctl; S.Gw = 1; S.member_w = 0; S.split = 0;
                S.ctl_lds.fail = 0; S.ctl_lds.bar = 0;
                S.watchdog = P.error; S.trace = nullptr; S.dbgskip = 0; S.prof = nullptr;
                S.kind = KA_SS; S.swapped = swapped;
                S.len_a = len_i; S.len_b = len_j;
                S.La = swapped ? len_j : len_i;
                S.Lb = swapped ? len_i : len_j;
                S.s1 = P.codes + P.seq_off[swapped ? j : i];
                S.s2 = P.codes + P.seq_off[swapped ? i : j];
                S.p1 = nullptr; S.p2 = nullptr; S.profa = nullptr; S.profb = nullptr;
                S.subm = P.subm;
                S.gpo = P.gpo; S.gpe = P.gpe; S.tgpe = P.tgpe; S.soff = 0.0f;
                S.sp_open = 0.0f; S.sp_ext = 0.0f; S.sp_text = 0.0f;
                S.p1_mult = 1.0f; S.p2_mult = 1.0f;
                ka_carve(S, P.scratch + (long long)k * P.scratch_stride, len_i, len_j, 0);
        }
        ka_build_tss(tss, P.subm, 0.0f);
        __syncthreads();
        ka_hirschberg<KA_SS, 23, 0, false, false, false, false, true>(S, nullptr, lds_waves, tss, nullptr);
        __syncthreads();
        ka_code_path(S, (int*)lds_waves);
        if (tid == 0 && P.scores) P.scores[k] = S.ctl->top_score;
        __syncthreads();
        int* dst = P.paths_out + P.poff[k];
        for (int i = tid; i < S.ctl->alnlen + 2; i += KA_NT) dst[i] = S.coded[i];
}

extern "C" void ka_launch_pairs(const KaPairDev* P, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_pair_kernel, KA_LDS_PAIR, &done) != hipSuccess) return;
        const int pw = (P->pw >= 1 && P->pw <= KA_PAIR_BLOCK / 64) ? P->pw : KA_PAIR_BLOCK / 64;
        hipLaunchKernelGGL(ka_pair_kernel, dim3(P->npairs), dim3(64 * pw), KA_LDS_WAVES + KA_LEAN_SCRATCH(64 * pw) + pw * KA_WAVE_LDS_LEAN, stream, *P);
}
#endif

// ------------------------------------------------------------------------------------------
// Units 6..9 (round 4): the four kernels that carry the anchor-consistency bonus once more, with room for ten anchors per DP row
// (KA_NB_BIG = 11 entries: `--consistency K`, 5 < K <= 10).  Twice the bonus registers: slower steps than the K <= 5 set
// (the allocator spills in the strips), same results; the host picks the set by K (ka_tree_build_consistency).
// ------------------------------------------------------------------------------------------
#if KA_UNIT == 6
__global__ __launch_bounds__(KA_BLOCK) void ka_task_kernel_cons_big(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<false, KA_NB_BIG>(D, blocks, chain);
}
extern "C" void ka_unit6_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int chain, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel_cons_big, KA_LDS_TOTAL, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_task_kernel_cons_big, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, chain);
}
#endif

#if KA_UNIT == 7
__global__ __launch_bounds__(KA_HALF_BLOCK, 2) void ka_task_kernel_half_cons_big(const KaTreeDev D, const int2* __restrict__ blocks, const int nqueue)
{
        ka_task_queue_entry<false, KA_NB_BIG>(D, blocks, nqueue);
}
extern "C" void ka_unit7_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, int nqueue, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel_half_cons_big, KA_LDS_HALF, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_task_kernel_half_cons_big, dim3(nblocks), dim3(KA_HALF_BLOCK), KA_LDS_HALF, stream, *D, blocks_dev, nqueue);
}
#endif

#if KA_UNIT == 8
// (two workgroups of four waves per CU: the entries of ten anchors do not fit the three-per-CU budget of the K <= 5 kernel)
__global__ __launch_bounds__(KA_PAIR_BLOCK, 2) void ka_task_kernel_lean_cons_big(const KaTreeDev D, const int2* __restrict__ blocks, const int chain)
{
        ka_task_entry<true, KA_NB_BIG>(D, blocks, 0);
}
extern "C" void ka_unit8_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_task_kernel_lean_cons_big, KA_LDS_PAIR, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_task_kernel_lean_cons_big, dim3(nblocks), dim3(KA_PAIR_BLOCK), KA_LDS_PAIR, stream, *D, blocks_dev, 0);
}
#endif

#if KA_UNIT == 9
__global__ __launch_bounds__(KA_BLOCK) void ka_refine_kernel_cons_big(const KaTreeDev D, const int2* __restrict__ blocks, const int unused)
{
        const int2 blk = blocks[blockIdx.x];
        if (blk.x >= 0) ka_task_body_refine<KA_NB_BIG>(D, blk.x, blk.y & 0xff, blk.y >> 8);
}
extern "C" void ka_unit9_launch(const KaTreeDev* D, const int2* blocks_dev, int nblocks, hipStream_t stream)
{
        static bool done = false;
        if (ka_optin(ka_refine_kernel_cons_big, KA_LDS_TOTAL, &done) != hipSuccess) return;
        hipLaunchKernelGGL(ka_refine_kernel_cons_big, dim3(nblocks), dim3(KA_BLOCK), KA_LDS_TOTAL, stream, *D, blocks_dev, 0);
}
#endif
