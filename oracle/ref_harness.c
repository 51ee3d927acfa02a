/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin driver that is compiled *together with the real Kalign sources where
 * they lie under /root/reference* (see oracle/Makefile, target `ref`) into
 * oracle/_ref/libkalign_ref.so.  Nothing here restates an algorithm: every DP,
 * profile, path and tree operation is a call into the reference's own global
 * functions.  The harness only (1) sequences those calls the way
 * kalign_run_seeded does up to the dispatcher seam (lib/src/aln_wrap.c:133-226),
 * (2) exposes the prepared state (encoded sequences, task list, scoring
 * parameters) as flat arrays, and (3) re-plays do_align (lib/src/aln_run.c:213-441,
 * a static function) task by task through the reference's public pieces so
 * that per-task paths / profiles / scores can be recorded as golden vectors.
 *
 * Used by tests/golden/make_golden.py (fixture generation, this container only)
 * and -- as the prebuilt .so -- by bench.py's cpu_baseline leg ("reference").
 * The product (kalign_amd/) never links or loads this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <float.h>
#include <time.h>

#ifdef HAVE_OPENMP
#include <omp.h>
#endif

#include "tldevel.h"
#include "kalign/kalign.h"
#include "msa_struct.h"
#include "msa_op.h"
#include "msa_alloc.h"
#include "msa_check.h"
#include "msa_sort.h"
#include "alphabet.h"
#include "task.h"
#include "bisectingKmeans.h"
#include "tlrng.h"
#include "aln_apair_dist.h"
#include "msa_op.h"
#include "aln_param.h"
#include "aln_struct.h"
#include "aln_mem.h"
#include "aln_setup.h"
#include "aln_controller.h"
#include "aln_seqseq.h"
#include "aln_seqprofile.h"
#include "aln_profileprofile.h"
#include "aln_run.h"
#include "aln_refine.h"
#include "weave_alignment.h"
#include "anchor_consistency.h"
#include "bpm.h"
#include "sequence_distance.h"

struct refh {
        struct msa* msa;
        struct aln_tasks* tasks;
        struct aln_param* ap;
        uint8_t* tree_codes;            /* the sequences in the alphabet build_tree_kmeans saw, concatenated */
        double tree_secs;               /* wall time of build_tree_kmeans */
};

static uint64_t fnv1a(const void* p, size_t n, uint64_t h)
{
        const unsigned char* c = (const unsigned char*)p;
        for(size_t i = 0; i < n; i++){ h ^= c[i]; h *= 1099511628211ULL; }
        return h;
}
#define FNV_SEED 1469598103934665603ULL

void refh_free(void* hv)
{
        struct refh* h = (struct refh*)hv;
        if(!h) return;
        if(h->ap) aln_param_free(h->ap);
        if(h->tasks) free_tasks(h->tasks);
        if(h->msa) kalign_free_msa(h->msa);
        free(h->tree_codes);
        free(h);
}

/* Everything kalign_run_seeded does before the dispatcher (aln_wrap.c:144-207),
   with deterministic sequence names so the len/name sort has no garbage ties. */
static void* prepare(char** seqs, int* lens, int numseq, int type,
                     float gpo, float gpe, float tgpe, int n_threads,
                     float dist_scale, float vsm_amax, float use_seq_weights, uint64_t tree_seed, float tree_noise);

void* refh_prepare(char** seqs, int* lens, int numseq, int type,
                   float gpo, float gpe, float tgpe, int n_threads,
                   float dist_scale, float vsm_amax, float use_seq_weights)
{
        return prepare(seqs, lens, numseq, type, gpo, gpe, tgpe, n_threads, dist_scale, vsm_amax, use_seq_weights, 0, 0.0f);
}

/* the same with the noisy guide tree of ensemble members (kalign_run_seeded with tree_seed / tree_noise,
   aln_wrap.c:171-176) */
void* refh_prepare_noisy(char** seqs, int* lens, int numseq, int type,
                         float gpo, float gpe, float tgpe, int n_threads,
                         float dist_scale, float vsm_amax, float use_seq_weights, uint64_t tree_seed, float tree_noise)
{
        return prepare(seqs, lens, numseq, type, gpo, gpe, tgpe, n_threads, dist_scale, vsm_amax, use_seq_weights, tree_seed, tree_noise);
}

/* the multipliers build_tree_kmeans_noisy draws (bisectingKmeans.c:103-115), from the reference's own generator */
int refh_noise_multipliers(uint64_t seed, float sigma, int n, float* out)
{
        struct rng_state* rng = init_rng(seed);
        if(!rng) return 1;
        for(int i = 0; i < n; i++){
                double noise = tl_random_gaussian(rng, 1.0, (double)sigma);
                if(noise < 0.1) noise = 0.1;
                out[i] = (float)noise;
        }
        free_rng(rng);
        return 0;
}

static void* prepare(char** seqs, int* lens, int numseq, int type,
                     float gpo, float gpe, float tgpe, int n_threads,
                     float dist_scale, float vsm_amax, float use_seq_weights, uint64_t tree_seed, float tree_noise)
{
        struct refh* h = calloc(1, sizeof(struct refh));
        struct msa* msa = NULL;
        if(kalign_arr_to_msa(seqs, lens, numseq, &msa) != OK) goto ERROR;
        h->msa = msa;
        msa->quiet = 1;
        for(int i = 0; i < numseq; i++){
                snprintf(msa->sequences[i]->name, MSA_NAME_LEN, "s%07d", i);
        }
        if(kalign_essential_input_check(msa, 0) != OK) goto ERROR;
        if(msa->aligned != ALN_STATUS_UNALIGNED){
                if(dealign_msa(msa) != OK) goto ERROR;
        }
        if(msa_sort_len_name(msa) != OK) goto ERROR;
        if(msa->biotype == ALN_BIOTYPE_DNA){
                msa->L = ALPHA_defDNA;
                if(convert_msa_to_internal(msa, ALPHA_defDNA) != OK) goto ERROR;
        }else if(msa->biotype == ALN_BIOTYPE_PROTEIN){
                msa->L = ALPHA_redPROTEIN;
                if(convert_msa_to_internal(msa, ALPHA_redPROTEIN) != OK) goto ERROR;
        }else{
                goto ERROR;
        }
        if(alloc_tasks(&h->tasks, msa->numseq) != OK) goto ERROR;
#ifdef HAVE_OPENMP
        omp_set_num_threads(n_threads < 1 ? 1 : n_threads);
#endif
        {
                size_t total = 0, o = 0;
                for(int i = 0; i < msa->numseq; i++) total += msa->sequences[i]->len;
                h->tree_codes = malloc(total ? total : 1);
                for(int i = 0; i < msa->numseq; i++){
                        memcpy(h->tree_codes + o, msa->sequences[i]->s, msa->sequences[i]->len);
                        o += msa->sequences[i]->len;
                }
        }
        {
                struct timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                if(tree_seed != 0 && tree_noise > 0.0f){
                        if(build_tree_kmeans_noisy(msa, &h->tasks, tree_seed, tree_noise) != OK) goto ERROR;
                }else{
                        if(build_tree_kmeans(msa, &h->tasks) != OK) goto ERROR;
                }
                clock_gettime(CLOCK_MONOTONIC, &t1);
                h->tree_secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        }
        if(msa->biotype == ALN_BIOTYPE_PROTEIN){
                if(convert_msa_to_internal(msa, ALPHA_ambigiousPROTEIN) != OK) goto ERROR;
        }
        if(aln_param_init(&h->ap, msa->biotype, n_threads, type, gpo, gpe, tgpe) != OK) goto ERROR;
        if(use_seq_weights >= 0.0f) h->ap->use_seq_weights = use_seq_weights;
        if(dist_scale > 0.0f) h->ap->dist_scale = dist_scale;
        if(vsm_amax >= 0.0f) h->ap->vsm_amax = vsm_amax;
        if(sort_tasks(h->tasks, TASK_ORDER_TREE) != OK) goto ERROR;
        return h;
ERROR:
        refh_free(h);
        return NULL;
}

double refh_tree_seconds(void* hv){ return ((struct refh*)hv)->tree_secs; }
int refh_numseq(void* hv){ return ((struct refh*)hv)->msa->numseq; }
int refh_biotype(void* hv){ return ((struct refh*)hv)->msa->biotype; }
int refh_ntasks(void* hv){ return ((struct refh*)hv)->tasks->n_tasks; }

/* lens[i], ranks[i] (= index in the caller's input order) for sorted position i */
void refh_get_seq_info(void* hv, int* lens, int* ranks)
{
        struct msa* msa = ((struct refh*)hv)->msa;
        for(int i = 0; i < msa->numseq; i++){
                lens[i] = msa->sequences[i]->len;
                ranks[i] = msa->sequences[i]->rank;
        }
}

/* the codes of all sequences (sorted order, concatenated) as they were when the guide tree was built:
   reduced protein alphabet / nucleotides (aln_wrap.c:155-160) */
int refh_get_tree_codes(void* hv, uint8_t* out)
{
        struct refh* h = (struct refh*)hv;
        size_t total = 0;
        if(!h->tree_codes) return 1;
        for(int i = 0; i < h->msa->numseq; i++) total += h->msa->sequences[i]->len;
        memcpy(out, h->tree_codes, total);
        return 0;
}

void refh_get_seq_codes(void* hv, int i, uint8_t* out)
{
        struct msa* msa = ((struct refh*)hv)->msa;
        memcpy(out, msa->sequences[i]->s, msa->sequences[i]->len);
}

int refh_get_seq_distances(void* hv, float* out)
{
        struct msa* msa = ((struct refh*)hv)->msa;
        if(!msa->seq_distances) return 0;
        memcpy(out, msa->seq_distances, sizeof(float) * msa->numseq);
        return 1;
}

/* subm: 23*23 row-major; scal: gpo,gpe,tgpe,dist_scale,vsm_amax,use_seq_weights */
void refh_get_params(void* hv, float* subm, float* scal)
{
        struct aln_param* ap = ((struct refh*)hv)->ap;
        for(int i = 0; i < 23; i++) for(int j = 0; j < 23; j++) subm[i*23+j] = ap->subm[i][j];
        scal[0] = ap->gpo; scal[1] = ap->gpe; scal[2] = ap->tgpe;
        scal[3] = ap->dist_scale; scal[4] = ap->vsm_amax; scal[5] = ap->use_seq_weights;
}

/* Scoring tables only (no sequences): used to generate the product's constant tables. */
int refh_param_table(int biotype, int type, float* subm, float* scal)
{
        struct aln_param* ap = NULL;
        if(aln_param_init(&ap, biotype, 1, type, -1.0f, -1.0f, -1.0f) != OK) return 1;
        for(int i = 0; i < 23; i++) for(int j = 0; j < 23; j++) subm[i*23+j] = ap->subm[i][j];
        scal[0] = ap->gpo; scal[1] = ap->gpe; scal[2] = ap->tgpe;
        scal[3] = ap->dist_scale; scal[4] = ap->vsm_amax; scal[5] = ap->use_seq_weights;
        aln_param_free(ap);
        return 0;
}

/* abc[3*t+{0,1,2}] in TASK_ORDER_TREE order */
void refh_get_tasks(void* hv, int* abc)
{
        struct aln_tasks* t = ((struct refh*)hv)->tasks;
        for(int i = 0; i < t->n_tasks; i++){
                abc[3*i] = t->list[i]->a; abc[3*i+1] = t->list[i]->b; abc[3*i+2] = t->list[i]->c;
        }
}

/* Replace the guide tree by a caller-supplied task list (a, b, c = numseq + index, children
   first) and optionally the per-sequence distances; everything else stays as prepared. */
int refh_set_tasks(void* hv, const int* abc, int n_tasks, const float* seq_distances)
{
        struct refh* h = (struct refh*)hv;
        struct aln_tasks* t = h->tasks;
        if(n_tasks != h->msa->numseq - 1) return 1;
        for(int i = 0; i < n_tasks; i++){
                t->list[i]->a = abc[3*i]; t->list[i]->b = abc[3*i+1]; t->list[i]->c = abc[3*i+2];
                t->list[i]->p = i; t->list[i]->n = 0; t->list[i]->score = 0.0f; t->list[i]->confidence = 0.0f;
                if(abc[3*i+2] != h->msa->numseq + i) return 1;
        }
        t->n_tasks = n_tasks;
        if(seq_distances){
                if(!h->msa->seq_distances) h->msa->seq_distances = malloc(sizeof(float) * h->msa->numseq);
                memcpy(h->msa->seq_distances, seq_distances, sizeof(float) * h->msa->numseq);
        }
        return 0;
}

/* Prepare WITHOUT the reference's own tree building: sequences arrive already encoded and
   ordered; the caller supplies tasks + distances via refh_set_tasks.  Used by bench.py's
   cpu_baseline leg so that CPU and GPU run the identical task list. */
void* refh_prepare_encoded(const uint8_t* codes, const int* off, const int* lens, int numseq,
                           int biotype, int type, float gpo, float gpe, float tgpe, int n_threads,
                           float dist_scale, float vsm_amax, float use_seq_weights)
{
        struct refh* h = calloc(1, sizeof(struct refh));
        struct msa* msa = NULL;
        char** tmp = malloc(sizeof(char*) * numseq);
        for(int i = 0; i < numseq; i++){
                tmp[i] = malloc(lens[i] + 1);
                for(int j = 0; j < lens[i]; j++) tmp[i][j] = biotype == ALN_BIOTYPE_DNA ? "ACGTN"[codes[off[i]+j] % 5] : "ARNDCQEGHILKMFPSTWYVBZX"[codes[off[i]+j] % 23];
                tmp[i][lens[i]] = 0;
        }
        if(kalign_arr_to_msa(tmp, (int*)lens, numseq, &msa) != OK) goto ERROR;
        h->msa = msa;
        msa->quiet = 1;
        msa->biotype = biotype;
        for(int i = 0; i < numseq; i++){
                snprintf(msa->sequences[i]->name, MSA_NAME_LEN, "s%07d", i);
                msa->sequences[i]->rank = i;
                memcpy(msa->sequences[i]->s, codes + off[i], lens[i]);
        }
        msa->L = biotype == ALN_BIOTYPE_DNA ? ALPHA_defDNA : ALPHA_ambigiousPROTEIN;
        if(alloc_tasks(&h->tasks, msa->numseq) != OK) goto ERROR;
        if(aln_param_init(&h->ap, msa->biotype, n_threads, type, gpo, gpe, tgpe) != OK) goto ERROR;
        if(use_seq_weights >= 0.0f) h->ap->use_seq_weights = use_seq_weights;
        if(dist_scale > 0.0f) h->ap->dist_scale = dist_scale;
        if(vsm_amax >= 0.0f) h->ap->vsm_amax = vsm_amax;
        for(int i = 0; i < numseq; i++) free(tmp[i]);
        free(tmp);
        return h;
ERROR:
        for(int i = 0; i < numseq; i++) free(tmp[i]);
        free(tmp);
        refh_free(h);
        return NULL;
}


/* ---- anchor consistency (aln_wrap.c:207-214) ---- */
/* Builds msa->consistency_table with the reference's own anchor_consistency_build
   (anchor selection + N x K pairwise_align_map).  From then on refh_run_tree and
   refh_run_tree_traced run in "default mode" (bonus matrices in every DP). */
int refh_build_consistency(void* hv, int n_anchors, float weight)
{
        struct refh* h = (struct refh*)hv;
        if(h->msa->consistency_table){
                anchor_consistency_free((struct consistency_table*)h->msa->consistency_table);
                h->msa->consistency_table = NULL;
        }
        h->ap->consistency_anchors = n_anchors;
        h->ap->consistency_weight = weight;
        if(anchor_consistency_build(h->msa, h->ap, n_anchors, weight,
                                    (struct consistency_table**)&h->msa->consistency_table) != OK) return 1;
        return 0;
}

/* K (0 when no table), anchor ids, and all position maps concatenated in (i*K + k) order,
   each of length len_i.  Pass NULL to query K only. */
int refh_get_consistency(void* hv, int* anchor_ids, int* maps_out)
{
        struct refh* h = (struct refh*)hv;
        struct consistency_table* ct = (struct consistency_table*)h->msa->consistency_table;
        if(!ct) return 0;
        if(anchor_ids) for(int k = 0; k < ct->n_anchors; k++) anchor_ids[k] = ct->anchor_ids[k];
        if(maps_out){
                int o = 0;
                for(int i = 0; i < ct->numseq * ct->n_anchors; i++){
                        for(int p = 0; p < ct->map_lengths[i]; p++) maps_out[o++] = ct->pos_maps[i][p];
                }
        }
        return ct->n_anchors;
}

/* optional sink for the FNV hash of every task's bonus matrix (traced replay) */
static uint64_t* g_bonus_hash_out = NULL;
void refh_set_bonus_hash_out(uint64_t* out){ g_bonus_hash_out = out; }

/* do_align's bonus block (aln_run.c:262-295): row/col node assignment mirrors the DP swap rules */
static int attach_bonus(struct msa* msa, struct aln_mem* m, int a, int b, int len_a, int len_b, float** bonus)
{
        struct consistency_table* ct = (struct consistency_table*)msa->consistency_table;
        int rn, cn, rows, cols;
        *bonus = NULL;
        m->consistency = NULL; m->consistency_stride = 0;
        if(!ct) return 0;
        if(msa->nsip[a] == 1 && msa->nsip[b] == 1){
                if(len_a < len_b){ rn = a; rows = len_a; cn = b; cols = len_b; }
                else{ rn = b; rows = len_b; cn = a; cols = len_a; }
        }else if(msa->nsip[a] == 1){
                rn = b; rows = len_b; cn = a; cols = len_a;
        }else if(msa->nsip[b] == 1){
                rn = a; rows = len_a; cn = b; cols = len_b;
        }else{
                if(len_a < len_b){ rn = a; rows = len_a; cn = b; cols = len_b; }
                else{ rn = b; rows = len_b; cn = a; cols = len_a; }
        }
        if(anchor_consistency_get_bonus_profile(ct, msa, rn, rows, cn, cols, bonus) != OK) return 1;
        m->consistency = *bonus;
        m->consistency_stride = cols;
        return 0;
}

static void collect_gaps(struct msa* msa, int* gaps_out)
{
        int o = 0;
        for(int i = 0; i < msa->numseq; i++){
                for(int j = 0; j <= msa->sequences[i]->len; j++){
                        gaps_out[o++] = msa->sequences[i]->gaps[j];
                }
        }
}

/* The real dispatcher (aln_run.c:43).  gaps_out: concatenated gaps[len+1] per
   sequence in sorted order.  Returns seconds spent inside create_msa_tree in *secs. */
int refh_run_tree(void* hv, int* gaps_out, double* secs)
{
        struct refh* h = (struct refh*)hv;
        double t0 = 0.0, t1 = 0.0;
#ifdef HAVE_OPENMP
        omp_set_num_threads(h->ap->nthreads < 1 ? 1 : h->ap->nthreads);
        t0 = omp_get_wtime();
#endif
        if(create_msa_tree(h->msa, h->ap, h->tasks) != OK) return 1;
#ifdef HAVE_OPENMP
        t1 = omp_get_wtime();
#endif
        if(secs) *secs = t1 - t0;
        h->msa->aligned = ALN_STATUS_ALIGNED;
        if(gaps_out) collect_gaps(h->msa, gaps_out);
        return 0;
}

/* refine_alignment (aln_refine.c:36-88) after refh_run_tree: the second, multi-trial pass over every edge
   (mode 1 = KALIGN_REFINE_ALL, 2 = KALIGN_REFINE_CONFIDENT; + 256: ap->adaptive_budget = 1, aln_refine.c:255-282);
   mode 3 = KALIGN_REFINE_INLINE: the tree aligned again from scratch with create_msa_tree_inline_refine
   (aln_run.c:448-475, three trials per edge, as aln_wrap.c:222-224 calls it).  conf_before / conf_after (n_tasks
   floats, may be NULL): task.confidence before and after; plen_out (num_profiles ints, may be NULL): msa->plen
   afterwards. */
int refh_refine(void* hv, int mode, int* gaps_out, float* conf_before, float* conf_after, int* plen_out)
{
        struct refh* h = (struct refh*)hv;
        const int adaptive = (mode >> 8) & 1;
        mode &= 255;
        if(sort_tasks(h->tasks, TASK_ORDER_TREE) != OK) return 1;
        if(conf_before) for(int i = 0; i < h->tasks->n_tasks; i++) conf_before[i] = h->tasks->list[i]->confidence;
#ifdef HAVE_OPENMP
        omp_set_num_threads(h->ap->nthreads < 1 ? 1 : h->ap->nthreads);
#endif
        if(mode == KALIGN_REFINE_INLINE){
                if(clean_aln(h->msa) != OK) return 1;
                for(int i = 0; i < h->msa->num_profiles; i++){
                        if(h->tasks->profile[i]){ MFREE(h->tasks->profile[i]); h->tasks->profile[i] = NULL; }
                }
                if(create_msa_tree_inline_refine(h->msa, h->ap, h->tasks, 3) != OK) return 1;
        }else{
                h->ap->adaptive_budget = adaptive;
                const int rc = refine_alignment(h->msa, h->ap, h->tasks, mode);
                h->ap->adaptive_budget = 0;
                if(rc != OK) return 1;
        }
        if(conf_after) for(int i = 0; i < h->tasks->n_tasks; i++) conf_after[i] = h->tasks->list[i]->confidence;
        if(plen_out) for(int i = 0; i < h->msa->num_profiles; i++) plen_out[i] = h->msa->plen[i];
        if(gaps_out) collect_gaps(h->msa, gaps_out);
        return 0;
}

/* One iteration of kalign_run_realign's loop up to the new guide tree (aln_wrap.c:449-495), after refh_run_tree:
   finalise_alignment, compute_aln_pairwise_dist, strip the gaps again, re-encode, set_sip_nsip,
   build_tree_from_pairwise.  Afterwards refh_get_tasks / refh_get_seq_distances describe the new tree and
   refh_run_tree aligns on it.  rows_sorted (may be NULL): the finalised rows in sorted order, alnlen+1 bytes each,
   as compute_aln_pairwise_dist saw them; dm_out (may be NULL): the numseq x numseq identity distances. */
int refh_realign_tree(void* hv, char** rows_sorted, int* alnlen, float* dm_out, double* secs_dist, double* secs_tree)
{
        struct refh* h = (struct refh*)hv;
        struct msa* msa = h->msa;
        float** dm = NULL;
        struct timespec t0, t1, t2;
        if(finalise_alignment(msa) != OK) return 1;
        if(alnlen) *alnlen = msa->alnlen;
        if(rows_sorted){
                for(int i = 0; i < msa->numseq; i++) memcpy(rows_sorted[i], msa->sequences[i]->seq, msa->alnlen + 1);
        }
        clock_gettime(CLOCK_MONOTONIC, &t0);
        if(compute_aln_pairwise_dist(msa, &dm) != OK) return 1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if(dm_out){
                for(int i = 0; i < msa->numseq; i++) memcpy(dm_out + (size_t)i * msa->numseq, dm[i], sizeof(float) * msa->numseq);
        }
        if(dealign_msa(msa) != OK) return 1;
        for(int si = 0; si < msa->numseq; si++){
                struct msa_seq* seq = msa->sequences[si];
                int w = 0;
                for(int r = 0; seq->seq[r] != '\0'; r++){
                        if(seq->seq[r] != '-') seq->seq[w++] = seq->seq[r];
                }
                seq->seq[w] = '\0';
                seq->len = w;
        }
        if(msa->biotype == ALN_BIOTYPE_DNA){
                if(convert_msa_to_internal(msa, ALPHA_defDNA) != OK) return 1;
        }else{
                if(convert_msa_to_internal(msa, ALPHA_ambigiousPROTEIN) != OK) return 1;
        }
        if(set_sip_nsip(msa) != OK) return 1;
        free_tasks(h->tasks);
        h->tasks = NULL;
        if(alloc_tasks(&h->tasks, msa->numseq) != OK) return 1;
        clock_gettime(CLOCK_MONOTONIC, &t2);
        if(build_tree_from_pairwise(msa, &h->tasks, dm) != OK) return 1;
        {
                struct timespec t3;
                clock_gettime(CLOCK_MONOTONIC, &t3);
                if(secs_tree) *secs_tree = (double)(t3.tv_sec - t2.tv_sec) + 1e-9 * (double)(t3.tv_nsec - t2.tv_nsec);
        }
        if(secs_dist) *secs_dist = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        free_aln_dm(dm, msa->numseq);
        if(sort_tasks(h->tasks, TASK_ORDER_TREE) != OK) return 1;
        return 0;
}

/* finalise_alignment + msa_sort_rank (aln_wrap.c:240-242); rows[i] receives the
   aligned string of INPUT sequence i (buffers of >= alnlen+1 bytes). */
int refh_finalise(void* hv, char** rows, int* alnlen)
{
        struct refh* h = (struct refh*)hv;
        if(finalise_alignment(h->msa) != OK) return 1;
        if(msa_sort_rank(h->msa) != OK) return 1;
        *alnlen = h->msa->alnlen;
        if(rows){
                for(int i = 0; i < h->msa->numseq; i++){
                        memcpy(rows[i], h->msa->sequences[i]->seq, h->msa->alnlen + 1);
                }
        }
        return 0;
}

int refh_alnlen_from_gaps(void* hv)
{
        struct msa* msa = ((struct refh*)hv)->msa;
        int n = msa->sequences[0]->len;
        for(int j = 0; j <= msa->sequences[0]->len; j++) n += msa->sequences[0]->gaps[j];
        return n;
}

/* Per-task record written by the traced replay. */
struct refh_task_rec {
        int a, b, c;
        int len_a, len_b;        /* operand lengths (a, b order) */
        int nsip_a, nsip_b;
        int plen;                /* alignment length = path[0] */
        int kind;                /* 0 seq-seq, 1 seq-profile, 2 profile-profile */
        int swapped;             /* DP ran with (b as rows, a as cols) */
        int meet, transition;    /* top-level Hirschberg meetup */
        int path_off;            /* offset of coded path (plen+2 ints) in paths_out */
        float gap_scale, subm_off;
        float score;             /* top-level meetup score */
        float confidence;        /* margin_sum / margin_count */
        uint64_t prof_hash;      /* FNV-1a of merged profile bytes, 0 for the root */
        uint64_t fhash, bhash;   /* FNV-1a of top-level f / b rows [0..dp_cols] */
};

static void set_operands(struct msa* msa, struct aln_tasks* t, struct aln_mem* m,
                         int a, int b, int* kind, int* swapped)
{
        /* operand selection / swap rules, aln_run.c:297-388 */
        int len_a = m->len_a, len_b = m->len_b;
        *swapped = 0;
        m->seq1 = NULL; m->seq2 = NULL; m->prof1 = NULL; m->prof2 = NULL;
        if(msa->nsip[a] == 1 && msa->nsip[b] == 1){
                *kind = 0;
                if(len_a < len_b){
                        m->seq1 = msa->sequences[a]->s; m->seq2 = msa->sequences[b]->s;
                }else{
                        *swapped = 1;
                        m->seq1 = msa->sequences[b]->s; m->seq2 = msa->sequences[a]->s;
                }
        }else if(msa->nsip[a] == 1){
                *kind = 1; *swapped = 1;
                m->seq2 = msa->sequences[a]->s; m->prof1 = t->profile[b]; m->sip = msa->nsip[b];
        }else if(msa->nsip[b] == 1){
                *kind = 1;
                m->seq2 = msa->sequences[b]->s; m->prof1 = t->profile[a]; m->sip = msa->nsip[a];
        }else{
                *kind = 2;
                if(len_a < len_b){
                        m->prof1 = t->profile[a]; m->prof2 = t->profile[b];
                }else{
                        *swapped = 1;
                        m->prof1 = t->profile[b]; m->prof2 = t->profile[a];
                }
        }
        if(*swapped){
                m->enda = len_b; m->endb = len_a; m->len_a = len_b; m->len_b = len_a;
        }
}

/* Replays do_align (aln_run.c:213-441) for every task, through the reference's
   own functions, recording golden data.  With a consistency table attached
   (refh_build_consistency) every DP reads its bonus matrix as do_align does.  paths_out must hold sum(len_a+len_b+3) ints.
   prof_dump (optional): receives the merged profile of task `dump_task`
   ((plen+2)*64 floats). */
int refh_run_tree_traced(void* hv, struct refh_task_rec* recs, int* paths_out,
                         int* gaps_out, int dump_task, float* prof_dump)
{
        struct refh* h = (struct refh*)hv;
        struct msa* msa = h->msa;
        struct aln_tasks* t = h->tasks;
        int poff = 0;
        for(int tid = 0; tid < t->n_tasks; tid++){
                struct aln_mem* m = NULL;
                struct refh_task_rec* r = &recs[tid];
                int a = t->list[tid]->a, b = t->list[tid]->b, c = t->list[tid]->c;
                struct aln_param scaled;
                struct aln_param* ap = h->ap;
                float* tmp = NULL;
                float* bonus = NULL;
                int kind, swapped, len_a, len_b, bonus_stride = 0;

                if(alloc_aln_mem(&m, 256) != OK) return 1;
                m->run_parallel = 0;
                m->ap = ap; m->mode = ALN_MODE_FULL;
                r->a = a; r->b = b; r->c = c;
                r->gap_scale = compute_gap_scale(msa, ap, a, b);
                r->subm_off = compute_subm_offset(msa, ap, a, b);
                if(r->gap_scale < 1.0f || r->subm_off > 0.0f){
                        scaled = *ap;
                        scaled.gpo *= r->gap_scale; scaled.gpe *= r->gap_scale; scaled.tgpe *= r->gap_scale;
                        scaled.subm_offset = r->subm_off;
                        m->ap = &scaled;
                }
                if(msa->nsip[a] == 1){
                        m->len_a = msa->sequences[a]->len;
                        if(make_profile_n(m->ap, msa->sequences[a]->s, m->len_a, 1.0f, &t->profile[a]) != OK) return 1;
                }else{
                        m->len_a = msa->plen[a];
                        set_gap_penalties_n(t->profile[a], m->len_a, msa->nsip[b]);
                }
                if(msa->nsip[b] == 1){
                        m->len_b = msa->sequences[b]->len;
                        if(make_profile_n(m->ap, msa->sequences[b]->s, m->len_b, 1.0f, &t->profile[b]) != OK) return 1;
                }else{
                        m->len_b = msa->plen[b];
                        set_gap_penalties_n(t->profile[b], m->len_b, msa->nsip[a]);
                }
                len_a = m->len_a; len_b = m->len_b;
                r->len_a = len_a; r->len_b = len_b;
                r->nsip_a = msa->nsip[a]; r->nsip_b = msa->nsip[b];

                /* ---- probe: top-level forward / backward / meetup on a scratch aln_mem ---- */
                {
                        int old_cor[5]; int meet = -1, tr = -1; float score = 0.0f;
                        if(init_alnmem(m) != OK) return 1;
                        if(attach_bonus(msa, m, a, b, len_a, len_b, &bonus) != 0) return 1;
                        bonus_stride = m->consistency_stride;
                        if(g_bonus_hash_out){
                                g_bonus_hash_out[tid] = bonus ? fnv1a(bonus, sizeof(float) * (size_t)len_a * (size_t)len_b, FNV_SEED) : 0;
                        }
                        set_operands(msa, t, m, a, b, &kind, &swapped);
                        if(m->enda > m->starta && m->endb > m->startb){
                                int mid = ((m->enda - m->starta) / 2) + m->starta;
                                old_cor[0] = m->starta; old_cor[1] = m->enda;
                                old_cor[2] = m->startb; old_cor[3] = m->endb; old_cor[4] = mid;
                                m->enda = mid; m->starta_2 = mid; m->enda_2 = old_cor[1];
                                if(kind == 0){
                                        aln_seqseq_foward(m); aln_seqseq_backward(m);
                                        aln_seqseq_meetup(m, old_cor, &meet, &tr, &score);
                                }else if(kind == 2){
                                        aln_profileprofile_foward(m); aln_profileprofile_backward(m);
                                        aln_profileprofile_meetup(m, old_cor, &meet, &tr, &score);
                                }else{
                                        aln_seqprofile_foward(m); aln_seqprofile_backward(m);
                                        aln_seqprofile_meetup(m, old_cor, &meet, &tr, &score);
                                }
                                r->fhash = fnv1a(m->f, sizeof(struct states) * (old_cor[3] + 1), FNV_SEED);
                                r->bhash = fnv1a(m->b, sizeof(struct states) * (old_cor[3] + 1), FNV_SEED);
                        }else{
                                r->fhash = 0; r->bhash = 0;
                        }
                        r->meet = meet; r->transition = tr; r->score = score;
                        m->len_a = len_a; m->len_b = len_b;
                }

                /* ---- the real thing ---- */
                if(init_alnmem(m) != OK) return 1;
                m->margin_sum = 0.0F; m->margin_count = 0;
                m->consistency = bonus; m->consistency_stride = bonus_stride;
                m->mode = ALN_MODE_FULL;
                set_operands(msa, t, m, a, b, &kind, &swapped);
                aln_runner(m);
                if(bonus){ free(bonus); bonus = NULL; }
                m->consistency = NULL;
                if(swapped){
                        if(mirror_path_n(m, len_a, len_b) != OK) return 1;
                        m->len_a = len_a; m->len_b = len_b;
                }
                r->kind = kind; r->swapped = swapped;
                r->confidence = (m->margin_count > 0) ? m->margin_sum / (float)m->margin_count : 0.0f;
                if(add_gap_info_to_path_n(m) != OK) return 1;
                m->ap = ap;

                r->plen = m->path[0];
                r->path_off = poff;
                memcpy(paths_out + poff, m->path, sizeof(int) * (m->path[0] + 2));
                poff += m->path[0] + 2;

                tmp = malloc(sizeof(float) * 64 * (m->path[0] + 2));
                r->prof_hash = 0;
                if(tid != t->n_tasks - 1){
                        update_n(t->profile[a], t->profile[b], tmp, m->ap, m->path, msa->nsip[a], msa->nsip[b]);
                        r->prof_hash = fnv1a(tmp, sizeof(float) * 64 * (m->path[0] + 2), FNV_SEED);
                        if(tid == dump_task && prof_dump){
                                memcpy(prof_dump, tmp, sizeof(float) * 64 * (m->path[0] + 2));
                        }
                }
                free(t->profile[a]); free(t->profile[b]);
                t->profile[a] = NULL; t->profile[b] = NULL;
                t->profile[c] = tmp;
                if(make_seq(msa, a, b, m->path) != OK) return 1;
                msa->plen[c] = m->path[0];
                msa->nsip[c] = msa->nsip[a] + msa->nsip[b];
                msa->sip[c] = realloc(msa->sip[c], sizeof(int) * msa->nsip[c]);
                {
                        int g = 0;
                        for(int j = msa->nsip[a]; j--;) msa->sip[c][g++] = msa->sip[a][j];
                        for(int j = msa->nsip[b]; j--;) msa->sip[c][g++] = msa->sip[b][j];
                }
                free_aln_mem(m);
        }
        msa->aligned = ALN_STATUS_ALIGNED;
        if(gaps_out) collect_gaps(msa, gaps_out);
        return 0;
}

/* N independent seq-seq alignments the way anchor_consistency.c:19-120
   (pairwise_align_map) runs them: unscaled parameters, rows = shorter with
   `len_i <= len_j` deciding the swap.  Timed with omp parallel for when
   n_threads > 1 (the reference runs this loop serially, anchor_consistency.c:246-267).
   codes: concatenated sequences; off[i] start offsets; pair (ia[k], ib[k]).
   paths_out: per pair (len_i+len_j+3) ints at poff[k] (coded path), scores_out optional. */
int refh_pairwise_batch(const uint8_t* codes, const int* off, const int* lens,
                        const int* ia, const int* ib, int npairs,
                        const float* subm, float gpo, float gpe, float tgpe,
                        int n_threads, int* paths_out, const long long* poff, double* secs)
{
        struct aln_param ap;
        float* rows[23];
        float tbl[23*23];
        int fail = 0;
        double t0 = 0.0, t1 = 0.0;
        memset(&ap, 0, sizeof(ap));
        memcpy(tbl, subm, sizeof(tbl));
        for(int i = 0; i < 23; i++) rows[i] = &tbl[i*23];
        ap.subm = rows; ap.gpo = gpo; ap.gpe = gpe; ap.tgpe = tgpe; ap.subm_offset = 0.0f;
        ap.nthreads = n_threads;
#ifdef HAVE_OPENMP
        omp_set_num_threads(n_threads < 1 ? 1 : n_threads);
        t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 4) reduction(+:fail)
#endif
        for(int k = 0; k < npairs; k++){
                struct aln_mem* m = NULL;
                int i = ia[k], j = ib[k];
                int len_i = lens[i], len_j = lens[j];
                int swapped = 0;
                if(alloc_aln_mem(&m, 256) != OK){ fail++; continue; }
                m->ap = &ap; m->mode = ALN_MODE_FULL; m->run_parallel = 0;
                if(len_i <= len_j){
                        m->len_a = len_i; m->len_b = len_j;
                        m->seq1 = codes + off[i]; m->seq2 = codes + off[j];
                }else{
                        m->len_a = len_j; m->len_b = len_i;
                        m->seq1 = codes + off[j]; m->seq2 = codes + off[i];
                        swapped = 1;
                }
                m->prof1 = NULL; m->prof2 = NULL;
                if(init_alnmem(m) != OK){ fail++; continue; }
                aln_runner_serial(m);
                if(paths_out){
                        if(swapped){
                                mirror_path_n(m, len_i, len_j);
                                m->len_a = len_i; m->len_b = len_j;
                        }
                        add_gap_info_to_path_n(m);
                        memcpy(paths_out + poff[k], m->path, sizeof(int) * (m->path[0] + 2));
                }
                free_aln_mem(m);
        }
#ifdef HAVE_OPENMP
        t1 = omp_get_wtime();
#endif
        if(secs) *secs = t1 - t0;
        return fail;
}

/* The public one-call API, for end-to-end fixtures (aln_wrap.c:110-131). */
int refh_kalign(char** seqs, int* lens, int numseq, int n_threads, int type,
                float gpo, float gpe, float tgpe, char** rows_out, int* alnlen)
{
        char** aligned = NULL;
        int n = 0;
        if(kalign(seqs, lens, numseq, n_threads, type, gpo, gpe, tgpe, &aligned, &n) != OK) return 1;
        *alnlen = n;
        for(int i = 0; i < numseq; i++){
                if(rows_out) memcpy(rows_out[i], aligned[i], n + 1);
                free(aligned[i]);
        }
        free(aligned);
        return 0;
}

/* calc_distance (sequence_distance.c:150-162 -> bpm_block, bpm.c:356) for a list of pairs; codes < 13 */
int refh_bpm_batch(const uint8_t* codes, const int* off, const int* lens, const int* ia, const int* ib, int npairs, int* dist_out)
{
        for(int k = 0; k < npairs; k++){
                dist_out[k] = (int)calc_distance((uint8_t*)(codes + off[ia[k]]), (uint8_t*)(codes + off[ib[k]]), lens[ia[k]], lens[ib[k]]);
        }
        return 0;
}
