/*
 * kalign_oracle.h -- TEST INFRASTRUCTURE ONLY (the parity checker).
 *
 * A plain-C, single-threaded restatement of Kalign's progressive-alignment hot
 * path (Hirschberg/Gotoh DP + per-task glue).  It exists so that tests can check
 * the HIP product bit-for-bit on the GPU box, where /root/reference does not
 * exist.  It is itself pinned against the real reference (oracle/_ref, golden
 * vectors under tests/golden/).  Nothing under kalign_amd/ may link, load or
 * call this.
 */
#ifndef KALIGN_ORACLE_H
#define KALIGN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as struct refh_task_rec in ref_harness.c (and ka_task_rec in
   include/kalign_amd.h) so one ctypes definition serves all three. */
typedef struct ko_task_rec {
        int a, b, c;
        int len_a, len_b;
        int nsip_a, nsip_b;
        int plen;
        int kind;          /* 0 seq-seq, 1 seq-profile, 2 profile-profile */
        int swapped;
        int meet, transition;
        int path_off;
        float gap_scale, subm_off;
        float score;
        float confidence;
        uint64_t prof_hash;
        uint64_t fhash, bhash;
} ko_task_rec;

/* scal[6] = gpo, gpe, tgpe, dist_scale, vsm_amax, use_seq_weights */
int ko_msa_tree(int numseq, const uint8_t* codes, const int* off, const int* lens,
                const float* seq_distances,
                int n_tasks, const int* tasks_abc,
                const float* subm, const float* scal,
                ko_task_rec* recs, int* paths_out, long long paths_cap,
                int* gaps_out, int dump_task, float* prof_dump);

/* The same with anchor consistency (n_anchors > 0: anchor_consistency_build + bonus in every DP).
   Optional outputs: anchor_ids_out[K]; maps_out = all position maps concatenated in (i*K+k) order;
   bonus_hash_out[n_tasks] = FNV-1a of each task's dense bonus matrix. */
int ko_msa_tree_cons(int numseq, const uint8_t* codes, const int* off, const int* lens,
                     const float* seq_distances,
                     int n_tasks, const int* tasks_abc,
                     const float* subm, const float* scal,
                     int n_anchors, float cons_weight,
                     ko_task_rec* recs, int* paths_out, long long paths_cap,
                     int* gaps_out, int dump_task, float* prof_dump,
                     int* anchor_ids_out, int* maps_out, uint64_t* bonus_hash_out);

/* refinement: refine_alignment (aln_refine.c:36-346) -- the second pass over every edge with
   convert_raw_path coding (:591-672), refine_edge's five trials (baseline + four round-robin flips of uncertain meetups,
   aln_seqseq.c:385-414) scored with compute_sp_score (sp_score.c:75-201), best trial kept.  mode 1 = KALIGN_REFINE_ALL,
   2 = KALIGN_REFINE_CONFIDENT (conf_in = the task confidences of the alignment being refined). */
int ko_msa_tree_refine(int numseq, const uint8_t* codes, const int* off, const int* lens,
                       const float* seq_distances,
                       int n_tasks, const int* tasks_abc,
                       const float* subm, const float* scal,
                       int n_anchors, float cons_weight, int mode, const float* conf_in,
                       ko_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out);
int ko_convert_raw_path(const int* raw_path, int len_a, int len_b, int* coded);

int ko_pairwise_batch(const uint8_t* codes, const int* off, const int* lens,
                      const int* ia, const int* ib, int npairs,
                      const float* subm, float gpo, float gpe, float tgpe,
                      int* paths_out, const long long* poff, float* scores_out);

/* One DP between explicit operands, for kernel-level tests.
   kind 0: seq1/seq2; kind 1: prof1 (rows) / seq2; kind 2: prof1 / prof2.
   raw_path: len_a+2 ints (path[i] = matched column, 1-based, or -1).
   f_out/b_out: top-level forward/backward rows, 3*(len_b+1) floats each. */
int ko_dp_single(int kind, const uint8_t* seq1, const uint8_t* seq2,
                 const float* prof1, const float* prof2, int len_a, int len_b,
                 const float* subm, float gpo, float gpe, float tgpe, float soff, int sip,
                 const float* bonus, int bonus_stride,
                 int* raw_path, float* f_out, float* b_out,
                 int* meet, int* transition, float* score, float* confidence);

int ko_make_profile(const uint8_t* seq, int len, const float* subm,
                    float gpo, float gpe, float tgpe, float soff, float* prof);
int ko_set_gap_penalties(float* prof, int len, int nsip);
int ko_code_path(const int* raw_path, int len_a, int len_b, int* coded);
int ko_mirror_path(const int* raw_in, int len_a, int len_b, int* raw_out);
int ko_update_profile(const float* pa, const float* pb, float* out, const int* coded,
                      int sipa, int sipb, const float* subm,
                      float gpo, float gpe, float tgpe, float use_seq_weights);

uint64_t ko_fnv1a(const void* p, uint64_t n);
/* Hirschberg prefix reuse (kalign_oracle.c: ko_hirschberg_r): on = 1, the passes a child can take over from its parent's are
   not run; every result must stay bit-identical (tests/test_oracle_golden.py).  The counters: DP cells of the passes that ran /
   that were taken over since the switch was last set. */
void ko_set_prefix_reuse(int on);
/* anchor votes carried up the tree (the device's KA_CARRY=1 rule) instead of counted over every member at every task; the second call:
   cells merged since the switch, and how many of them needed a count over an operand's members (what the device marks) */
void ko_set_carried_votes(int on);
void ko_carried_votes_cells(long long* cells, long long* counted);
/* ko_set_carried_votes(2): also tallies, per merged cell in which both operands vote, how many different positions its voters hold
   (out8[d], d = 1 .. 6, [7] = 7 and more) */
void ko_carried_votes_distinct(long long* out8);
void ko_prefix_reuse_cells(long long* run, long long* reused);

/* distance estimation: bpm_block (lib/src/bpm.c:356-582) and calc_distance's pair rule
   (sequence_distance.c:150-162: the longer sequence is the text).  Codes < 13. */
int ko_bpm_block(const uint8_t* t, const uint8_t* p, int n, int m);
int ko_bpm_batch(const uint8_t* codes, const int* off, const int* lens, const int* ia, const int* ib, int npairs, int* dist_out);

/* realignment pass (aln_wrap.c:449-495): identity distances of a finished alignment (aln_apair_dist.c:9-86) and the
   UPGMA task list built on them (bisectingKmeans.c:1150-1200, :974-1053); dm is consumed by the second call */
int ko_aln_pairwise_dist(const uint8_t* rows, int n, long long stride, int alnlen, uint8_t gap, float* dm);
int ko_tree_from_pairwise(float* dm, int n, int* tasks_abc, float* seq_distances);

#ifdef __cplusplus
}
#endif
#endif
