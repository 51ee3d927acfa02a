// ka_best.h -- refinement's incremental-trial view (KaInc) and the (best, runner-up) pair of a meetup scan.
// One of the text sections of the task kernels, included by ka_kernels.hip in this order: ka_shared.h, ka_pass.h, ka_best.h,
// ka_subtree.h, ka_wstrip.h, ka_meetup.h, ka_hirschberg.h, ka_path.h, ka_profile.h, ka_task.h.  Not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------
// Meetup of one sub-problem by one wave (aln_seqseq.c:241-420 and the two profile variants),
// then aln_continue: path writes and the two child sub-problems (aln_controller.c:194-436).
// ------------------------------------------------------------------------------------------
// Incremental flip trials of refinement (ka_trial_incremental): what the level-synchronous baseline trial leaves behind, carved
// from the task's scratch behind TaskShared::inc.  n = len_a + len_b + 8 bounds the number of meetups of a trial.
struct KaInc {
        KaSub* win;        // [record] the sub-problem (pad = its recursion-order key)
        int2* mx;          // [record] (width of its subtree's key range, raw path entry of its first row before its subtree ran)
        int* msort;        // [sorted position] record
        int* skey;         // [sorted position] key
        float* mseq0;      // [sorted position] margin = the baseline's margins in recursion order
        float* mseq;       // the running trial's margins in recursion order (2n)
        int* upos;         // [u] sorted position of the u-th uncertain meetup of the baseline
        int* ucnt;         // [sorted position] uncertain meetups in front of it (n + 1)
        int* raw0;         // the baseline's raw path
};
__device__ __host__ inline long long ka_inc_bytes(long long n) { return (40 + (long long)sizeof(KaSub)) * n + 64; }
__device__ __forceinline__ KaInc ka_inc_from(char* base, const long long n)
{
        KaInc I;
        I.win = (KaSub*)base; base += (long long)sizeof(KaSub) * n;
        I.mx = (int2*)base; base += 8 * n;
        I.msort = (int*)base; base += 4 * n;
        I.skey = (int*)base; base += 4 * n;
        I.mseq0 = (float*)base; base += 4 * n;
        I.mseq = (float*)base; base += 8 * n;
        I.upos = (int*)base; base += 4 * n;
        I.ucnt = (int*)base; base += 4 * n + 16;
        I.raw0 = (int*)base;
        return I;
}
static_assert(sizeof(KaSub) % 8 == 0, "KaInc::win stride keeps the arrays behind it 8-byte aligned");

__device__ __forceinline__ KaInc ka_inc_view(const TaskShared& S) { return ka_inc_from(S.inc, (long long)S.len_a + S.len_b + 8); }

struct Best { float mx; float mx2; int key; int key2; };          // key2 (who the runner-up is) only matters to refinement trials

__device__ __forceinline__ void best_consider(Best& b, float s, int key)
{
        if (s > b.mx) { b.mx2 = b.mx; b.key2 = b.key; b.mx = s; b.key = key; }
        else if (s > b.mx2) { b.mx2 = s; b.key2 = key; }
}

// (value, key) pairs in the order the reference's sequential scan ranks them: higher value first, among equal values the
// earlier candidate (a later candidate only displaces on a strictly greater value, aln_seqseq.c:284-291)
__device__ __forceinline__ bool best_before(float v1, int k1, float v2, int k2) { return v1 > v2 || (v1 == v2 && k1 < k2); }

__device__ __forceinline__ void best_merge(Best& x, float omx, float omx2, int okey, int okey2 = 0x7fffffff)
{
        if (best_before(omx, okey, x.mx, x.key)) {
                // the other side's best wins: the runner-up is the better of our best and its runner-up
                const bool mine = best_before(x.mx, x.key, omx2, okey2);
                x.mx2 = mine ? x.mx : omx2; x.key2 = mine ? x.key : okey2;
                x.mx = omx; x.key = okey;
        } else {
                const bool theirs = best_before(omx, okey, x.mx2, x.key2);
                x.mx2 = theirs ? omx : x.mx2; x.key2 = theirs ? okey : x.key2;
        }
}
