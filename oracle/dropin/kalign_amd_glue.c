/*
 * kalign_amd_glue.c -- the reference-side binding of libkalign_amd.so: the replacement bodies a Kalign
 * maintainer adds for the six call sites of INTEGRATION.md.  This is the text INTEGRATION.md quotes.
 *
 * TEST INFRASTRUCTURE: it is compiled only by `make -C oracle dropin` (build container, where the
 * reference sources lie under /root/reference) into oracle/_ref/dropin/libkalign.so.3 -- the reference's own
 * lib/src compiled where it lies, with the six functions below taken from here instead (the reference's
 * definitions are renamed at compile time, -Dcreate_msa_tree=kalign_ref_create_msa_tree etc.), linked against
 * libkalign_amd.so and exporting lib/include/kalign/kalign.h unchanged.  The product (kalign_amd/) links
 * nothing from the reference and nothing from here.
 *
 *   create_msa_tree            lib/src/aln_run.c:43-78        -> ka_tree_upload / ka_tree_build_consistency / ka_tree_run
 *   create_msa_tree_inline_refine  lib/src/aln_run.c:448-475  -> the same with ka_tree_refine(3) in place of ka_tree_run
 *   refine_alignment           lib/src/aln_refine.c:36-88     -> ka_tree_refine(1 | 2) on the job create_msa_tree left in HBM
 *   anchor_consistency_build   lib/src/anchor_consistency.c:200-275 -> ka_tree_build_consistency (+ a host copy of the table)
 *   build_tree_kmeans          lib/src/bisectingKmeans.c:177-271    -> ka_guide_tree
 *   build_tree_kmeans_noisy    lib/src/bisectingKmeans.c:76-175     -> ka_guide_tree with the noise multipliers (ensemble members)
 *   compute_aln_pairwise_dist  lib/src/aln_apair_dist.c:9-86        -> ka_aln_guide_tree: identity distances AND the UPGMA tree, on the
 *   build_tree_from_pairwise   lib/src/bisectingKmeans.c:1150-1200     rows finalise_alignment left in HBM (`--precise`, `--realign`)
 *   finalise_alignment         lib/src/msa_op.c:546-576       -> ka_tree_aligned_rows
 *
 * KALIGN_AMD_GLUE_REPORT=1 in the environment prints, at exit, how often every seam ran on the device and how often it
 * handed the call to the reference's own function (stderr, one line).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#ifdef HAVE_OPENMP
#include <omp.h>
#endif

#include "tldevel.h"
#include "msa_struct.h"
#include "task.h"
#include "aln_param.h"
#include "anchor_consistency.h"
#include "tlrng.h"
#include "msa_op.h"
#include "msa_check.h"
#include "aln_wrap.h"

#include "kalign_amd.h"

/* What the device holds belongs to the calling THREAD: the members of an ensemble run side by side (kalign_ensemble below), each
   thread on a device context of its own. */
#define GLUE_TLS __thread

/* the reference's own definitions, renamed by the drop-in build (oracle/Makefile) */
extern int kalign_ref_finalise_alignment(struct msa* msa);
extern int kalign_ref_refine_alignment(struct msa* msa, struct aln_param* ap, struct aln_tasks* t, int refine_mode);
extern int kalign_ref_create_msa_tree(struct msa* msa, struct aln_param* ap, struct aln_tasks* t);
extern int kalign_ref_create_msa_tree_inline_refine(struct msa* msa, struct aln_param* ap, struct aln_tasks* t, int n_trials);
extern int kalign_ref_compute_aln_pairwise_dist(struct msa* msa, float*** dm_ptr);
extern int kalign_ref_build_tree_from_pairwise(struct msa* msa, struct aln_tasks** tasks, float** dm);
extern int kalign_ref_anchor_consistency_build(struct msa* msa, struct aln_param* ap, int n_anchors, float weight, struct consistency_table** ct_out);
extern int kalign_ref_kalign_ensemble(struct msa* msa, int n_threads, int type, int n_runs, float gpo, float gpe, float tgpe, uint64_t seed, int min_support,
                                      const char* save_poar_path, int refine, float dist_scale, float vsm_amax, int realign, float use_seq_weights,
                                      int consistency_anchors, float consistency_weight);

/* how often each seam ran on the device / fell back to the reference (tests/test_gpu_dropin.py reads them) */
enum { GLUE_TREE = 0, GLUE_INLINE, GLUE_REFINE, GLUE_REFINE_REF, GLUE_FINALISE, GLUE_FINALISE_REF,
       GLUE_CONS, GLUE_CONS_REF, GLUE_KMEANS, GLUE_KMEANS_NOISY, GLUE_ALNDIST, GLUE_ALNDIST_REF, GLUE_ALNTREE, GLUE_ALNTREE_REF, GLUE_INLINE_REF,
       GLUE_TREE_MULTI, GLUE_CONS_MULTI, GLUE_TREE_REF, GLUE_ENSEMBLE_MULTI, GLUE_MEMBER_AHEAD, GLUE_MEMBER_MISSED, GLUE_N };
static const char* glue_names[GLUE_N] = { "tree", "inline", "refine", "refine_ref", "finalise", "finalise_ref",
                                          "cons", "cons_ref", "kmeans", "kmeans_noisy", "alndist", "alndist_ref", "alntree", "alntree_ref", "inline_ref",
                                          "tree_multi", "cons_multi", "tree_ref", "ensemble_multi", "member_ahead", "member_missed" };
static int glue_counts[GLUE_N];
#define GLUE_COUNT(which) __atomic_fetch_add(&glue_counts[which], 1, __ATOMIC_RELAXED)
int kalign_amd_glue_count(int which)
{
        return (which >= 0 && which < GLUE_N) ? glue_counts[which] : -1;
}
static void glue_report(void)
{
        int i;
        fprintf(stderr, "kalign_amd_glue:");
        for(i = 0; i < GLUE_N; i++){
                fprintf(stderr, " %s=%d", glue_names[i], glue_counts[i]);
        }
        fprintf(stderr, "\n");
}

static GLUE_TLS ka_ctx* glue_ctx = NULL;               /* one context per process / GPU */
static GLUE_TLS ka_ctx* glue_job_ctx = NULL;           /* the context that holds glue_job_msa's alignment: glue_ctx, or rank 0's after a sharded run */
static GLUE_TLS ka_ctx* glue_rows_ctx = NULL;          /* ... and the one whose HBM holds the finalised rows of glue_rows_msa */
static GLUE_TLS const struct msa* glue_job_msa = NULL;  /* the msa whose alignment the device currently holds */
static GLUE_TLS int glue_job_numseq = 0;
static GLUE_TLS const struct msa* glue_rows_msa;        /* (defined with the realignment seams below) */
static GLUE_TLS uint64_t glue_job_stamp = 0;            /* FNV-1a over the lengths and gap arrays the device job left in that msa */

/* The device job is recognised by more than the msa's address: an msa freed without finalise_alignment leaves a stale
   pointer that a later allocation can reuse, and the host may edit gaps[] between the seams.  The stamp covers what the
   device state stands for -- every sequence's length and gaps[] as glue_collect wrote them. */
static uint64_t glue_stamp(const struct msa* msa)
{
        uint64_t h = 1469598103934665603ULL;
        int i, j;
        for(i = 0; i < msa->numseq; i++){
                const struct msa_seq* s = msa->sequences[i];
                h = (h ^ (uint64_t)(uint32_t)s->len) * 1099511628211ULL;
                for(j = 0; j <= s->len; j++){
                        h = (h ^ (uint64_t)(uint32_t)s->gaps[j]) * 1099511628211ULL;
                }
        }
        return h;
}

static int glue_same_job(const struct msa* msa)
{
        return msa == glue_job_msa && msa->numseq == glue_job_numseq && glue_stamp(msa) == glue_job_stamp;
}

/*
 * More than one GPU (north_star: "tasks ... sharded across the 8 GPUs of one node"; lib/src/aln_run.c:95-109: subtrees are
 * independent): create_msa_tree and anchor_consistency_build go through ka_multi_* -- one context and one rank of the sharded
 * path (ka_dist_*: subtree cut, RCCL hand-overs above the cut, gathered records and paths) per device, the ranks as threads
 * of this process.  KALIGN_AMD_DEVICES=n limits / forces the number of devices (1: this path off); KALIGN_AMD_GLUE_WORLD=n
 * runs n ranks on device 0 over the library's in-process transport (tests on one-GPU boxes).  The result does not depend on
 * the number of ranks.  After a sharded run rank 0's context -- it holds the job, the whole consistency table and the gathered
 * records -- takes the finished alignment over (ka_multi_adopt: the gap arrays woven on the host go back as residue -> column
 * tables), so that the seams behind the dispatcher (refinement, finalise, the realignment distances) stay on a device.
 *
 * A node's worth of devices costs something to open (a communicator, a context and an upload per rank, a thread per rank and
 * call): without KALIGN_AMD_DEVICES the path is taken only for jobs of at least KALIGN_AMD_MULTI_MIN sequences (default
 * 2048); KALIGN_AMD_DEVICES=n (n > 1) or KALIGN_AMD_GLUE_WORLD=n takes it for every job.
 */
static ka_multi* glue_multi = NULL;
static int glue_multi_tried = 0;
static int glue_multi_min = 0;                  /* sequences a job needs before the node is opened for it (0: any) */
#define GLUE_MAX_MEMBERS 16
static ka_ctx* glue_member_ctx[GLUE_MAX_MEMBERS];  /* one single-GPU context per member slot (device), made on first use */
static void glue_release(void)
{
        int i;
        /* (at exit: the communicators, contexts and threads' buffers go before the HIP runtime does) */
        for(i = 0; i < GLUE_MAX_MEMBERS; i++){
                if(glue_member_ctx[i]){ ka_ctx_destroy(glue_member_ctx[i]); glue_member_ctx[i] = NULL; }
        }
        if(glue_multi){ ka_multi_destroy(glue_multi); glue_multi = NULL; }
        if(glue_ctx){ ka_ctx_destroy(glue_ctx); glue_ctx = NULL; }
        glue_job_ctx = NULL; glue_rows_ctx = NULL;
}
static GLUE_TLS int glue_in_member = 0;          /* this thread runs ONE ensemble member on one device (kalign_ensemble below) */
static ka_multi* glue_multi_context(int numseq)
{
        if(glue_in_member){
                return NULL;
        }
        if(!glue_multi_tried){
                const char* w = getenv("KALIGN_AMD_GLUE_WORLD");
                const char* d = getenv("KALIGN_AMD_DEVICES");
                const char* mn = getenv("KALIGN_AMD_MULTI_MIN");
                int world = 0;
                int loopback = 0;
                glue_multi_tried = 1;
                if(w && atoi(w) > 1){
                        world = atoi(w);
                        loopback = 1;
                }else{
                        world = d ? atoi(d) : ka_device_count();
                        if(world > ka_device_count()){
                                world = ka_device_count();
                        }
                        if(!d){
                                glue_multi_min = mn ? atoi(mn) : 2048;
                        }
                }
                if(world > 1 && numseq < glue_multi_min){
                        glue_multi_tried = 0;            /* (a small job: decided again for the next one) */
                        return NULL;
                }
                if(world > 1 && ka_multi_create(world, NULL, loopback, &glue_multi)){
                        WARNING_MSG("kalign_amd: %d devices could not be opened together (%s): one GPU", world, ka_multi_last_error());
                        glue_multi = NULL;
                }
        }
        return (glue_multi && numseq >= glue_multi_min) ? glue_multi : NULL;
}

static void glue_register(void)
{
        atexit(glue_release);
        if(getenv("KALIGN_AMD_GLUE_REPORT")){
                atexit(glue_report);
        }
}

static int glue_context(void)
{
        static pthread_once_t once = PTHREAD_ONCE_INIT;
        pthread_once(&once, glue_register);
        if(!glue_ctx && ka_ctx_create(0, &glue_ctx)){
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        return OK;
ERROR:
        return FAIL;
}

/* sequences of the msa as one code array (msa->sequences[i]->s in whatever alphabet they are in right now) */
static int glue_flatten(struct msa* msa, uint8_t** codes_out, int** off_out, int** lens_out, long long* total_out)
{
        uint8_t* codes = NULL;
        int* off = NULL;
        int* lens = NULL;
        long long total = 0;
        int n = msa->numseq;
        int i;
        MMALLOC(off, sizeof(int) * n);
        MMALLOC(lens, sizeof(int) * n);
        for(i = 0; i < n; i++){
                off[i] = (int)total;
                lens[i] = msa->sequences[i]->len;
                total += lens[i];
        }
        MMALLOC(codes, total + 1);
        for(i = 0; i < n; i++){
                memcpy(codes + off[i], msa->sequences[i]->s, lens[i]);
        }
        *codes_out = codes; *off_out = off; *lens_out = lens; *total_out = total;
        return OK;
ERROR:
        if(off) MFREE(off);
        if(lens) MFREE(lens);
        if(codes) MFREE(codes);
        return FAIL;
}

/* a valid task list over n leaves (pairs level by level): the consistency stage needs the sequences on the device,
   not the guide tree */
static void glue_pairing_tasks(int n, int* abc)
{
        int* cur = (int*)malloc(sizeof(int) * n);
        int m = n, next = n, t = 0, i;
        for(i = 0; i < n; i++) cur[i] = i;
        while(m > 1){
                int k = 0;
                for(i = 0; i + 1 < m; i += 2){
                        abc[3 * t] = cur[i]; abc[3 * t + 1] = cur[i + 1]; abc[3 * t + 2] = next;
                        cur[k++] = next++;
                        t++;
                }
                if(m & 1) cur[k++] = cur[m - 1];
                m = k;
        }
        free(cur);
}

static void glue_params(struct aln_param* ap, float* subm, float* scal)
{
        int i, j;
        for(i = 0; i < 23; i++){
                for(j = 0; j < 23; j++){
                        subm[i * 23 + j] = ap->subm[i][j];
                }
        }
        scal[0] = ap->gpo; scal[1] = ap->gpe; scal[2] = ap->tgpe;
        scal[3] = ap->dist_scale; scal[4] = ap->vsm_amax; scal[5] = ap->use_seq_weights;
}

/* the table the device(s) hold right now -- per device set: the single-GPU context and the ranks of the node are different
   contexts, and a table built on one says nothing about the other */
static GLUE_TLS const struct consistency_table* glue_ct_single = NULL;
static GLUE_TLS const struct consistency_table* glue_ct_multi = NULL;

/*
 * anchor_consistency_build (anchor_consistency.c:200-275).  The reference aligns every sequence to K anchors here,
 * one pair after the other on the host.  Replacement: the same N x K alignments as one batch on the device
 * (ka_tree_build_consistency: anchors, seq-seq alignments with the unscaled parameters, position maps).  The maps
 * stay in HBM for the dispatcher -- which then builds every task's bonus on the device instead of the reference's
 * dense La x Lb matrix (aln_run.c:262-295) -- and a complete host copy goes into the reference's own table layout
 * (anchor_consistency.h:17-24), so that host code that reads it (refinement, aln_refine.c) keeps working.
 * Same decline rules as the reference (:209-218).
 */
int anchor_consistency_build(struct msa* msa, struct aln_param* ap, int n_anchors, float weight, struct consistency_table** ct_out)
{
        struct consistency_table* ct = NULL;
        uint8_t* codes = NULL;
        int* off = NULL;
        int* lens = NULL;
        int* abc = NULL;
        int* maps = NULL;
        int* ids_multi = NULL;
        float subm[23 * 23];
        float scal[6];
        long long total = 0;
        long long o = 0;
        int n = msa->numseq;
        int K = n_anchors;
        int i, k;
        *ct_out = NULL;
        if(K <= 0 || n < 3 || msa->seq_distances == NULL){
                return OK;
        }
        if(K > n){
                K = n;
        }
        RUN(glue_context());
        RUN(glue_flatten(msa, &codes, &off, &lens, &total));
        MMALLOC(abc, sizeof(int) * 3 * (n - 1));
        glue_pairing_tasks(n, abc);
        glue_params(ap, subm, scal);
        glue_job_msa = NULL;
        glue_rows_msa = NULL;                            /* (a new job: the rows of the last one leave HBM) */
        glue_ct_single = NULL;
        glue_ct_multi = NULL;
        if(K <= KA_CONS_MAX_ANCHORS && glue_multi_context(n)){
                /* the N x K batch sharded over the devices, every rank's share of the maps broadcast in place (ka_dist_consistency) */
                MMALLOC(ids_multi, sizeof(int) * K);
                MMALLOC(maps, sizeof(int) * (total * K + 1));
                if(ka_multi_consistency(glue_multi, n, codes, off, lens, msa->seq_distances, n - 1, abc, subm, scal, 0, K, weight, ids_multi, maps) != K){
                        ERROR_MSG("kalign_amd: %s", ka_multi_last_error());
                }
                GLUE_COUNT(GLUE_CONS_MULTI);
        }else
        if(ka_tree_upload(glue_ctx, n, codes, off, lens, msa->seq_distances, n - 1, abc, subm, scal, 0) ||
           ka_tree_build_consistency(glue_ctx, K, weight)){
                /* a request the library does not take (more than 10 anchors ...): the reference's own function, like every
                   other seam; the dispatcher then rebuilds the table it needs or declines alike (no table is resident) */
                MFREE(codes); MFREE(off); MFREE(lens); MFREE(abc);
                GLUE_COUNT(GLUE_CONS_REF);
                return kalign_ref_anchor_consistency_build(msa, ap, n_anchors, weight, ct_out);
        }
        if(!ids_multi){
                GLUE_COUNT(GLUE_CONS);
        }
        MMALLOC(ct, sizeof(struct consistency_table));
        ct->pos_maps = NULL;
        ct->map_lengths = NULL;
        ct->anchor_ids = NULL;
        ct->n_anchors = K;
        ct->numseq = n;
        ct->weight = weight;
        MMALLOC(ct->anchor_ids, sizeof(int) * K);
        MMALLOC(ct->pos_maps, sizeof(int*) * n * K);
        MMALLOC(ct->map_lengths, sizeof(int) * n * K);
        for(i = 0; i < n * K; i++){
                ct->pos_maps[i] = NULL;
                ct->map_lengths[i] = 0;
        }
        if(ids_multi){
                memcpy(ct->anchor_ids, ids_multi, sizeof(int) * K);
        }else{
                MMALLOC(maps, sizeof(int) * (total * K + 1));
                if(ka_tree_get_consistency(glue_ctx, ct->anchor_ids, maps) != K){
                        ERROR_MSG("kalign_amd: the device declined to build the consistency table");
                }
        }
        for(i = 0; i < n; i++){
                for(k = 0; k < K; k++){
                        /* (the map of an anchor against itself is the identity, as in the reference, :251-258) */
                        ct->map_lengths[i * K + k] = lens[i];
                        MMALLOC(ct->pos_maps[i * K + k], sizeof(int) * lens[i]);
                        memcpy(ct->pos_maps[i * K + k], maps + o, sizeof(int) * lens[i]);
                        o += lens[i];
                }
        }
        if(!msa->quiet){
                LOG_MSG("Anchor consistency: K=%d, weight=%.1f", K, weight);
        }
        if(ids_multi){
                glue_ct_multi = ct;
        }else{
                glue_ct_single = ct;
        }
        *ct_out = ct;
        MFREE(codes); MFREE(off); MFREE(lens); MFREE(abc); MFREE(maps);
        if(ids_multi) MFREE(ids_multi);
        return OK;
ERROR:
        if(codes) MFREE(codes);
        if(off) MFREE(off);
        if(lens) MFREE(lens);
        if(abc) MFREE(abc);
        if(maps) MFREE(maps);
        if(ids_multi) MFREE(ids_multi);
        if(ct){
                anchor_consistency_free(ct);
        }
        return FAIL;
}

/*
 * create_msa_tree (aln_run.c:43-78, do_align :213-441 per task).  Reads numseq, sequences[i]->s/len,
 * seq_distances, the task list; leaves sequences[i]->gaps[], nsip[], sip[][], plen[], task confidence -- exactly
 * the state the reference's dispatcher leaves.  Merged profiles stay in HBM.
 */
static int glue_collect(struct msa* msa, struct aln_tasks* t, const int* lens, long long total, ka_ctx* from);

/* inline_refine: 0 = create_msa_tree, n > 0 = create_msa_tree_inline_refine with n trials per edge */
static int glue_tree(struct msa* msa, struct aln_param* ap, struct aln_tasks* t, int inline_refine)
{
        struct consistency_table* ct = (struct consistency_table*)msa->consistency_table;
        int n = msa->numseq;
        int nt = t->n_tasks;
        int* off = NULL;
        int* lens = NULL;
        int* abc = NULL;
        uint8_t* codes = NULL;
        float subm[23 * 23];
        float scal[6];
        long long total = 0;
        int flags = KA_FLAG_DEVICE_GAPS;
        int i;

        RUN(sort_tasks(t, TASK_ORDER_TREE));             /* as the reference does, aln_run.c:48 */
        RUN(glue_context());
        glue_job_msa = NULL;
        glue_rows_msa = NULL;                            /* (a new job: the rows of the last one leave HBM) */

        RUN(glue_flatten(msa, &codes, &off, &lens, &total));
        MMALLOC(abc, sizeof(int) * 3 * nt);
        for(i = 0; i < nt; i++){
                abc[3 * i] = t->list[i]->a;
                abc[3 * i + 1] = t->list[i]->b;
                abc[3 * i + 2] = t->list[i]->c;
        }
        glue_params(ap, subm, scal);

        if(!inline_refine && (!ct || ct->n_anchors <= KA_CONS_MAX_ANCHORS) && glue_multi_context(n)){
                /* the whole node: every rank uploads the job, runs its subtrees and its share of the tasks above the cut; records
                   and coded paths come back gathered, the gap arrays are woven on the host.  The table the RANKS hold from
                   anchor_consistency_build is kept when it is this msa's. */
                int mflags = (ct && ct == glue_ct_multi) ? KA_FLAG_KEEP_CONSISTENCY : 0;
                glue_ct_multi = NULL;
                if(ka_multi_tree_run(glue_multi, n, codes, off, lens, msa->seq_distances, nt, abc, subm, scal, mflags, ct ? ct->n_anchors : 0, ct ? ct->weight : 0.0f)){
                        ERROR_MSG("kalign_amd: %s", ka_multi_last_error());
                }
                glue_ct_multi = ct;
                /* (glue_collect hands the alignment to rank 0's context: the seams behind the dispatcher carry on there) */
                RUN(glue_collect(msa, t, lens, total, NULL));
                GLUE_COUNT(GLUE_TREE_MULTI);
                MFREE(off); MFREE(lens); MFREE(codes); MFREE(abc);
                return OK;
        }
        /* the table anchor_consistency_build left in HBM for these sequences is kept across the upload (also by the
           realignment passes of kalign_run_realign, aln_wrap.c:449-504: same sequences, new tree); if another job has
           used the device since -- or the table sits on the ranks of the node, not on this context -- it is built again
           (same anchors, same maps) */
        if(ct && ct == glue_ct_single){
                flags |= KA_FLAG_KEEP_CONSISTENCY;
        }
        if(ka_tree_upload(glue_ctx, n, codes, off, lens, msa->seq_distances, nt, abc, subm, scal, flags)){
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        if(ct && ct != glue_ct_single){
                glue_ct_single = NULL;
                if(ka_tree_build_consistency(glue_ctx, ct->n_anchors, ct->weight)){
                        ERROR_MSG("kalign_amd: %s", ka_last_error());
                }
        }
        glue_ct_single = ct;                             /* NULL: this upload dropped whatever table there was */
        /* do_align_inline_refine (aln_run.c:515-790) is do_align with three flip trials per edge: mode 3 of ka_tree_refine */
        if((inline_refine ? ka_tree_refine(glue_ctx, 3 | KA_REFINE_TRIALS(inline_refine), NULL) : ka_tree_run(glue_ctx)) || ka_tree_sync(glue_ctx)){
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        RUN(glue_collect(msa, t, lens, total, glue_ctx));
        GLUE_COUNT(inline_refine ? GLUE_INLINE : GLUE_TREE);
        MFREE(off); MFREE(lens); MFREE(codes); MFREE(abc);
        return OK;
ERROR:
        if(off) MFREE(off);
        if(lens) MFREE(lens);
        if(codes) MFREE(codes);
        if(abc) MFREE(abc);
        return FAIL;
}

/* leave exactly the state do_align leaves (aln_run.c:391-436): gaps[], plen[], nsip[], sip[][], task confidence.
   from: the context that ran the job; NULL: the node (ka_multi_*), whose rank 0 then takes the alignment over. */
static int glue_collect(struct msa* msa, struct aln_tasks* t, const int* lens, long long total, ka_ctx* from)
{
        int multi = from == NULL;
        int n = msa->numseq;
        int nt = t->n_tasks;
        int* gaps = NULL;
        int* paths = NULL;
        ka_task_rec* recs = NULL;
        long long cap;
        int i, j, g;
        cap = multi ? ka_multi_paths_size(glue_multi) : ka_tree_paths_size(from);
        MMALLOC(recs, sizeof(ka_task_rec) * nt);
        MMALLOC(gaps, sizeof(int) * (total + n));
        MMALLOC(paths, sizeof(int) * (cap + 1));
        if(multi){
                if(ka_multi_download(glue_multi, n, lens, nt, recs, paths, cap, gaps)){
                        ERROR_MSG("kalign_amd: %s", ka_multi_last_error());
                }
        }else
        if(ka_tree_download(from, recs, paths, cap, gaps)){
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        glue_job_msa = NULL;
        if(multi){
                if(ka_multi_adopt(glue_multi, recs, gaps)){
                        WARNING_MSG("kalign_amd: %s: the stages behind the dispatcher run on the host", ka_multi_last_error());
                }else{
                        glue_job_ctx = ka_multi_ctx(glue_multi, 0);
                        glue_job_msa = msa;
                }
        }else{
                glue_job_ctx = from;
                glue_job_msa = msa;
        }
        for(i = 0, g = 0; i < n; i++){
                memcpy(msa->sequences[i]->gaps, gaps + g, sizeof(int) * (lens[i] + 1));
                g += lens[i] + 1;
        }
        for(i = 0; i < nt; i++){
                int a = recs[i].a;
                int b = recs[i].b;
                int c = recs[i].c;
                int k = 0;
                t->list[i]->confidence = recs[i].confidence;
                msa->plen[c] = recs[i].plen;
                msa->nsip[c] = msa->nsip[a] + msa->nsip[b];
                MREALLOC(msa->sip[c], sizeof(int) * msa->nsip[c]);
                for(j = msa->nsip[a]; j--;){
                        msa->sip[c][k++] = msa->sip[a][j];
                }
                for(j = msa->nsip[b]; j--;){
                        msa->sip[c][k++] = msa->sip[b][j];
                }
        }
        glue_job_numseq = n;
        glue_job_stamp = glue_stamp(msa);
        MFREE(recs); MFREE(gaps); MFREE(paths);
        return OK;
ERROR:
        glue_job_msa = NULL;
        if(recs) MFREE(recs);
        if(gaps) MFREE(gaps);
        if(paths) MFREE(paths);
        return FAIL;
}

/* a consistency table with more anchors than a DP row of the device kernels carries (KA_CONS_MAX_ANCHORS; such a table was
   built by the reference's own stage, cons_ref): the reference's dispatcher, like every other seam the library does not take */
static int glue_table_too_wide(const struct msa* msa)
{
        const struct consistency_table* ct = (const struct consistency_table*)msa->consistency_table;
        return ct && ct->n_anchors > KA_CONS_MAX_ANCHORS;
}

int create_msa_tree(struct msa* msa, struct aln_param* ap, struct aln_tasks* t)
{
        if(glue_table_too_wide(msa)){
                glue_job_msa = NULL;
                GLUE_COUNT(GLUE_TREE_REF);
                return kalign_ref_create_msa_tree(msa, ap, t);
        }
        return glue_tree(msa, ap, t, 0);
}

/*
 * create_msa_tree_inline_refine (aln_run.c:448-475; KALIGN_REFINE_INLINE, aln_wrap.c:222-224 passes three trials).
 */
int create_msa_tree_inline_refine(struct msa* msa, struct aln_param* ap, struct aln_tasks* t, int n_trials)
{
        if(n_trials < 1 || n_trials > 255 || glue_table_too_wide(msa)){
                glue_job_msa = NULL;
                GLUE_COUNT(GLUE_INLINE_REF);
                return kalign_ref_create_msa_tree_inline_refine(msa, ap, t, n_trials);
        }
        return glue_tree(msa, ap, t, n_trials);
}

/*
 * refine_alignment (aln_refine.c:36-88): the second pass over every edge with refine_edge's five flip trials
 * (:93-346).  The job create_msa_tree uploaded is still in HBM (sequences, tree, parameters, consistency table):
 * ka_tree_refine runs the pass there.  KALIGN_REFINE_CONFIDENT compares task confidences with their median; the
 * device recomputes them as the reference's exact float sums (conf_in = NULL) instead of trusting the first pass's
 * level-order sums.  ap->adaptive_budget (aln_refine.c:255-282: a trial count per edge from the baseline's margins) is a
 * flag of the same call.  Another msa (an alignment the device does not hold) goes through the reference's own
 * function, which works on the state create_msa_tree left on the host.
 */
int refine_alignment(struct msa* msa, struct aln_param* ap, struct aln_tasks* t, int refine_mode)
{
        int* lens = NULL;
        long long total = 0;
        int n = msa->numseq;
        int i;
        if(refine_mode == 0){                            /* KALIGN_REFINE_NONE */
                return OK;
        }
        if(!glue_same_job(msa) || !glue_job_ctx || (refine_mode != 1 && refine_mode != 2)){
                glue_job_msa = NULL;                     /* the host state moves on without the device */
                GLUE_COUNT(GLUE_REFINE_REF);
                return kalign_ref_refine_alignment(msa, ap, t, refine_mode);
        }
        RUN(sort_tasks(t, TASK_ORDER_TREE));
        MMALLOC(lens, sizeof(int) * n);
        for(i = 0; i < n; i++){
                lens[i] = msa->sequences[i]->len;
                total += lens[i];
        }
        if(ka_tree_refine(glue_job_ctx, refine_mode | (ap->adaptive_budget ? KA_REFINE_ADAPTIVE : 0), NULL) || ka_tree_sync(glue_job_ctx)){
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        /* (after a sharded first pass the refinement pass ran on rank 0's context: a job of that context from here on) */
        RUN(glue_collect(msa, t, lens, total, glue_job_ctx));   /* the refined gaps are what the device holds now */
        GLUE_COUNT(GLUE_REFINE);
        MFREE(lens);
        return OK;
ERROR:
        glue_job_msa = NULL;                             /* (whatever the device holds is no longer this msa's state) */
        if(lens) MFREE(lens);
        return FAIL;
}

/*
 * build_tree_kmeans (bisectingKmeans.c:177-271): the two distance batches on the device, the 2-means bisection and
 * UPGMA between them on the host in the reference's fp32 order -- the same task list and seq_distances, bit for bit.
 * msa->sequences[i]->s holds the tree alphabet at this point (aln_wrap.c:155-160).
 */
static int glue_kmeans(struct msa* msa, struct aln_tasks** tasks, const float* dm_scale)
{
        struct aln_tasks* t = *tasks;
        int n = msa->numseq;
        int* off = NULL;
        int* lens = NULL;
        int* abc = NULL;
        uint8_t* codes = NULL;
        long long total = 0;
        int n_threads = 1;
        int i;
        ASSERT(n >= 2, "build_tree_kmeans needs at least two sequences");
        RUN(glue_context());
        if(!t){
                RUN(alloc_tasks(&t, n));
        }
#ifdef HAVE_OPENMP
        n_threads = omp_get_max_threads();
#endif
        RUN(glue_flatten(msa, &codes, &off, &lens, &total));
        MMALLOC(abc, sizeof(int) * 3 * (n - 1));
        if(msa->seq_distances == NULL){
                MMALLOC(msa->seq_distances, sizeof(float) * n);
        }
        if(ka_guide_tree(glue_ctx, n, codes, off, lens, n_threads, dm_scale, abc, msa->seq_distances)){
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        for(i = 0; i < n - 1; i++){
                t->list[i]->a = abc[3 * i];
                t->list[i]->b = abc[3 * i + 1];
                t->list[i]->c = abc[3 * i + 2];
        }
        t->n_tasks = n - 1;                              /* already in TASK_ORDER_TREE order */
        *tasks = t;
        MFREE(off); MFREE(lens); MFREE(abc); MFREE(codes);
        return OK;
ERROR:
        if(off) MFREE(off);
        if(lens) MFREE(lens);
        if(abc) MFREE(abc);
        if(codes) MFREE(codes);
        return FAIL;
}

int build_tree_kmeans(struct msa* msa, struct aln_tasks** tasks)
{
        GLUE_COUNT(GLUE_KMEANS);
        return glue_kmeans(msa, tasks, NULL);
}

/*
 * build_tree_kmeans_noisy (bisectingKmeans.c:76-175; the guide trees of ensemble members): the same tree builder on
 * distances multiplied by Gaussian noise.  The multipliers are the caller's: drawn here from the reference's own
 * generator in the reference's order (one per (sequence, anchor), :105-115) and handed to the device with the batch.
 */
int build_tree_kmeans_noisy(struct msa* msa, struct aln_tasks** tasks, uint64_t seed, float noise_sigma)
{
        float* scale = NULL;
        int n = msa->numseq;
        int na = n < 32 ? n : 32;                        /* pick_anchor, bisectingKmeans.c */
        int i, j, rc;
        if(seed == 0 || noise_sigma <= 0.0f){
                GLUE_COUNT(GLUE_KMEANS_NOISY);
                return glue_kmeans(msa, tasks, NULL);
        }
        MMALLOC(scale, sizeof(float) * (size_t)n * na);
        {
                struct rng_state* rng = init_rng(seed);
                for(i = 0; i < n; i++){
                        for(j = 0; j < na; j++){
                                double noise = tl_random_gaussian(rng, 1.0, (double)noise_sigma);
                                if(noise < 0.1) noise = 0.1;
                                scale[(size_t)i * na + j] = (float)noise;
                        }
                }
                free_rng(rng);
        }
        GLUE_COUNT(GLUE_KMEANS_NOISY);
        rc = glue_kmeans(msa, tasks, scale);
        MFREE(scale);
        return rc;
ERROR:
        return FAIL;
}

/*
 * compute_aln_pairwise_dist (aln_apair_dist.c:9-86) and build_tree_from_pairwise (bisectingKmeans.c:1150-1200): the
 * realignment loop of kalign_run_realign (aln_wrap.c:449-504; `--precise`, `--realign n`).  The rows finalise_alignment
 * just made are still in HBM: ONE device call computes the N x N identity distances and the UPGMA tree on them
 * (ka_aln_guide_tree).  The reference's interface splits that in two, so the first seam returns the distances in the
 * reference's layout and keeps the tree; the second hands the tree over when it is asked about that very matrix.
 */
static GLUE_TLS float** glue_dm = NULL;                   /* the matrix the last compute_aln_pairwise_dist returned */
static GLUE_TLS int* glue_dm_abc = NULL;
static GLUE_TLS float* glue_dm_sd = NULL;
static GLUE_TLS int glue_dm_n = 0;
static GLUE_TLS const struct msa* glue_rows_msa = NULL;   /* the msa whose finalised rows the device holds */
static GLUE_TLS int glue_rows_numseq = 0;
static GLUE_TLS int glue_rows_alnlen = 0;
static GLUE_TLS uint64_t glue_rows_stamp_v = 0;

/* The rows in HBM are recognised by more than the msa's address (as glue_same_job does for the job): a freed msa's address
   can come back with an alignment read from a file.  numseq, alnlen and FNV-1a over up to 64 evenly spaced rows. */
static uint64_t glue_rows_stamp(const struct msa* msa)
{
        uint64_t h = 1469598103934665603ULL;
        int n = msa->numseq;
        int step = n > 64 ? n / 64 : 1;
        int i, j;
        for(i = 0; i < n; i += step){
                const char* r = msa->sequences[i]->seq;
                for(j = 0; j < msa->alnlen && r[j]; j++){
                        h = (h ^ (uint64_t)(uint8_t)r[j]) * 1099511628211ULL;
                }
                h = (h ^ (uint64_t)(uint32_t)j) * 1099511628211ULL;
        }
        return h;
}

static int glue_same_rows(const struct msa* msa)
{
        return glue_rows_ctx && msa == glue_rows_msa && msa->numseq == glue_rows_numseq && msa->alnlen == glue_rows_alnlen
               && glue_rows_stamp(msa) == glue_rows_stamp_v;
}

static void glue_dm_free(void)
{
        if(glue_dm_abc){ MFREE(glue_dm_abc); glue_dm_abc = NULL; }
        if(glue_dm_sd){ MFREE(glue_dm_sd); glue_dm_sd = NULL; }
}

int compute_aln_pairwise_dist(struct msa* msa, float*** dm_ptr)
{
        float** dm = NULL;
        float* flat = NULL;
        uint8_t* rows = NULL;
        int n = msa->numseq;
        int i;
        if(msa->aligned != ALN_STATUS_FINAL || n < 2 || glue_context() != OK || !glue_ctx){
                GLUE_COUNT(GLUE_ALNDIST_REF);
                return kalign_ref_compute_aln_pairwise_dist(msa, dm_ptr);
        }
        {
                static int registered = 0;
                if(!registered){ registered = 1; atexit(glue_dm_free); }
        }
        glue_dm_free();
        glue_dm = NULL;
        MMALLOC(flat, sizeof(float) * (size_t)n * n);
        MMALLOC(glue_dm_abc, sizeof(int) * 3 * (n - 1));
        MMALLOC(glue_dm_sd, sizeof(float) * n);
        if(glue_same_rows(msa)){
                /* the rows are where ka_tree_aligned_rows left them (on rank 0's device after a sharded run) */
                if(ka_aln_guide_tree(glue_rows_ctx, n, NULL, 0, 0, '-', glue_dm_abc, glue_dm_sd, flat)){
                        ERROR_MSG("kalign_amd: %s", ka_last_error());
                }
        }else{
                long long stride = (long long)msa->alnlen + 1;
                MMALLOC(rows, (size_t)n * stride);
                for(i = 0; i < n; i++){
                        memcpy(rows + (size_t)i * stride, msa->sequences[i]->seq, msa->alnlen);
                        rows[(size_t)i * stride + msa->alnlen] = 0;
                }
                if(ka_aln_guide_tree(glue_ctx, n, rows, stride, msa->alnlen, '-', glue_dm_abc, glue_dm_sd, flat)){
                        ERROR_MSG("kalign_amd: %s", ka_last_error());
                }
                MFREE(rows); rows = NULL;
        }
        MMALLOC(dm, sizeof(float*) * n);
        for(i = 0; i < n; i++){
                dm[i] = NULL;
        }
        for(i = 0; i < n; i++){
                MMALLOC(dm[i], sizeof(float) * n);
                memcpy(dm[i], flat + (size_t)i * n, sizeof(float) * n);
        }
        MFREE(flat);
        glue_dm = dm; glue_dm_n = n;
        GLUE_COUNT(GLUE_ALNDIST);
        *dm_ptr = dm;
        return OK;
ERROR:
        if(flat) MFREE(flat);
        if(rows) MFREE(rows);
        if(dm){
                for(i = 0; i < n; i++){
                        if(dm[i]) MFREE(dm[i]);
                }
                MFREE(dm);
        }
        return FAIL;
}

int build_tree_from_pairwise(struct msa* msa, struct aln_tasks** tasks, float** dm)
{
        struct aln_tasks* t = *tasks;
        int n = msa->numseq;
        int i;
        if(dm == NULL || dm != glue_dm || n != glue_dm_n || !glue_dm_abc){
                GLUE_COUNT(GLUE_ALNTREE_REF);
                return kalign_ref_build_tree_from_pairwise(msa, tasks, dm);
        }
        if(!t){
                RUN(alloc_tasks(&t, n));
        }
        if(msa->seq_distances == NULL){
                MMALLOC(msa->seq_distances, sizeof(float) * n);
        }
        memcpy(msa->seq_distances, glue_dm_sd, sizeof(float) * n);
        for(i = 0; i < n - 1; i++){
                t->list[i]->a = glue_dm_abc[3 * i];
                t->list[i]->b = glue_dm_abc[3 * i + 1];
                t->list[i]->c = glue_dm_abc[3 * i + 2];
        }
        t->n_tasks = n - 1;
        *tasks = t;
        glue_dm = NULL;                                  /* (the caller frees the matrix next; its address may come back) */
        GLUE_COUNT(GLUE_ALNTREE);
        return OK;
ERROR:
        return FAIL;
}

/*
 * finalise_alignment (msa_op.c:546-576).  The device holds the column of every residue of the alignment the
 * dispatcher above just made: the rows come back ready-made.  Any other msa (an alignment read from a file, one
 * whose gaps[] were edited on the host: refinement, consensus) goes through the reference's own function.
 */
int finalise_alignment(struct msa* msa)
{
        int n = msa->numseq;
        int* alnlen = NULL;
        uint8_t* letters = NULL;
        uint8_t* rows = NULL;
        long long total = 0;
        long long o = 0;
        int width = 0;
        int i;
        if(!glue_same_job(msa) || !glue_job_ctx){
                GLUE_COUNT(GLUE_FINALISE_REF);
                glue_rows_msa = NULL;
                return kalign_ref_finalise_alignment(msa);
        }
        GLUE_COUNT(GLUE_FINALISE);
        glue_job_msa = NULL;                             /* the rows below replace seq->seq: one shot */
        glue_rows_msa = msa;                             /* ... and stay in HBM for the realignment loop's distances */
        glue_rows_ctx = glue_job_ctx;
        ASSERT(msa->aligned == ALN_STATUS_ALIGNED, "Sequences are not aligned");
        for(i = 0; i < n; i++){
                total += msa->sequences[i]->len;
        }
        MMALLOC(letters, total + 1);
        MMALLOC(alnlen, sizeof(int) * n);
        for(i = 0; i < n; i++){
                memcpy(letters + o, msa->sequences[i]->seq, msa->sequences[i]->len);
                o += msa->sequences[i]->len;
        }
        if(ka_tree_aligned_rows(glue_rows_ctx, letters, '-', NULL, 0, alnlen)){      /* size query */
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        for(i = 0; i < n; i++){
                if(alnlen[i] > width){
                        width = alnlen[i];
                }
        }
        MMALLOC(rows, (size_t)n * (width + 1));
        if(ka_tree_aligned_rows(glue_rows_ctx, letters, '-', rows, width + 1, alnlen)){
                ERROR_MSG("kalign_amd: %s", ka_last_error());
        }
        for(i = 0; i < n; i++){
                char* s = NULL;
                MMALLOC(s, alnlen[i] + 1);
                memcpy(s, rows + (size_t)i * (width + 1), alnlen[i] + 1);       /* terminator included */
                MFREE(msa->sequences[i]->seq);
                msa->sequences[i]->seq = s;
        }
        msa->alnlen = alnlen[0];
        msa->aligned = ALN_STATUS_FINAL;
        glue_rows_numseq = n; glue_rows_alnlen = msa->alnlen; glue_rows_stamp_v = glue_rows_stamp(msa);
        MFREE(letters); MFREE(alnlen); MFREE(rows);
        return OK;
ERROR:
        if(letters) MFREE(letters);
        if(alnlen) MFREE(alnlen);
        if(rows) MFREE(rows);
        return FAIL;
}


/*
 * kalign_ensemble's member loop (lib/src/ensemble.c:286-339; `--ensemble n`, `--precise`): n_runs complete alignments of the same
 * sequences with their own gap penalties and noisy guide trees, one after the other in the reference -- BASELINE config 5 wants
 * one member per GPU.  The seam has two halves:
 *   * kalign_ensemble (this definition; the reference's is kalign_ref_kalign_ensemble): with G >= 2 member devices it runs the
 *     members AHEAD OF TIME, member k on device k mod G, every device's members on a thread of their own -- each through the
 *     reference's own kalign_run_seeded / kalign_run_realign, i.e. through the seams above on that thread's context (guide tree,
 *     consistency, task tree, rows: no collective, weak scaling) -- and then calls the reference's function;
 *   * the two calls of its member loop come here instead (kalign_amd_member_run_*, redirected when ensemble.c is compiled,
 *     oracle/Makefile): a member that was run ahead hands its finished alignment over (the msa structs trade contents), anything
 *     else -- a single device, the winner's re-run with KALIGN_REFINE_CONFIDENT (:403-451) -- is the reference's call as it was.
 * POAR extraction, scoring, consensus and the confidence values stay the reference's, on what the members produced.
 * The penalties of member k (resolve_run_params and its table, ensemble.c:32-76) are restated here: k = 0 the base values
 * (aln_param_init's, :262-268), k > 0 scaled per entry k mod 12, tree seed + k, tree noise per entry.
 * Devices: ka_device_count(), KALIGN_AMD_DEVICES=n; KALIGN_AMD_GLUE_WORLD=n: n member contexts on device 0 (one-GPU test boxes).
 * ONE GPU (round 5): the members still run ahead, on up to GLUE_SHARED_MEMBERS contexts that share the device (ka_ctx_set_shared:
 * own stream each, no workgroup ever waits for another one, so any interleaving of the members' launches is safe).  A member's
 * second tree (`--precise`: UPGMA on the rows, a chain of ~70 dependent profile-profile tasks for 2048 sequences) keeps a few
 * workgroups busy; eight members side by side fill the gaps: 2048 x ~300 protein, 8 members with one realignment pass: 132 ms per
 * member alone, 56 ms side by side (profiles/r05_ensemble_phases.log).  KALIGN_AMD_ENSEMBLE_SLOTS=n overrides (1: one after the other).
 * The HIP runtime multiplexes streams onto 4 hardware queues by default -- members 5..8 would wait behind 1..4 -- so the glue asks
 * for 8 (GPU_MAX_HW_QUEUES, only when the user has not set it) before the runtime starts.
 */
#define GLUE_SHARED_MEMBERS 8
__attribute__((constructor)) static void glue_hw_queues(void)
{
        /* a process-wide side effect of linking this library, ensembles or not: KALIGN_AMD_KEEP_HW_QUEUES=1 leaves the runtime's default */
        const char* keep = getenv("KALIGN_AMD_KEEP_HW_QUEUES");
        if(!keep || atoi(keep) == 0){
                setenv("GPU_MAX_HW_QUEUES", "8", 0);
        }
}

static const float glue_member_scale[12][4] = {
        {1.0f, 1.0f, 1.0f, 0.0f}, {0.5f, 1.5f, 0.8f, 0.20f}, {1.5f, 0.5f, 1.2f, 0.20f}, {0.7f, 0.7f, 0.5f, 0.25f},
        {1.4f, 1.4f, 1.5f, 0.25f}, {0.8f, 1.2f, 1.0f, 0.30f}, {1.3f, 0.8f, 0.7f, 0.30f}, {0.6f, 1.0f, 1.3f, 0.15f},
        {1.0f, 0.6f, 0.6f, 0.15f}, {1.8f, 1.0f, 1.0f, 0.35f}, {1.0f, 1.8f, 1.8f, 0.35f}, {0.4f, 0.4f, 0.3f, 0.20f},
};

struct glue_member {
        struct msa* aln;                         /* the finished member (NULL: not run / handed over) */
        float gpo, gpe, tgpe, noise;
        uint64_t seed;
        int rc;
};
struct glue_member_args {                        /* what every member call of one ensemble shares */
        int n_threads, type, refine, realign, anchors;
        float dist_scale, vsm_amax, usw, weight;
};
static struct glue_member* glue_members = NULL;  /* the members run ahead of the reference's loop (main thread only) */
static int glue_n_members = 0;
static struct glue_member_args glue_margs;

struct glue_slot_job { int slot, n_slots, device, shared; };

static void* glue_member_thread(void* p)
{
        struct glue_slot_job* j = (struct glue_slot_job*)p;
        int k;
        glue_in_member = 1;
        if(!glue_member_ctx[j->slot]){
                if(ka_ctx_create(j->device, &glue_member_ctx[j->slot])){
                        glue_member_ctx[j->slot] = NULL;
                }else if(j->shared){
                        ka_ctx_set_shared(glue_member_ctx[j->slot], 1);     /* several contexts on one GPU: no co-residency assumptions */
                }
        }
        glue_ctx = glue_member_ctx[j->slot];
        for(k = j->slot; k < glue_n_members; k += j->n_slots){
                struct glue_member* m = &glue_members[k];
                if(!glue_ctx || !m->aln){
                        m->rc = FAIL;
                        continue;
                }
                if(glue_margs.realign > 0){
                        m->rc = kalign_run_realign(m->aln, glue_margs.n_threads, glue_margs.type, m->gpo, m->gpe, m->tgpe, glue_margs.refine, 0,
                                                   glue_margs.dist_scale, glue_margs.vsm_amax, glue_margs.realign, glue_margs.usw,
                                                   glue_margs.anchors, glue_margs.weight);
                }else{
                        m->rc = kalign_run_seeded(m->aln, glue_margs.n_threads, glue_margs.type, m->gpo, m->gpe, m->tgpe, glue_margs.refine, 0,
                                                  m->seed, m->noise, glue_margs.dist_scale, glue_margs.vsm_amax, glue_margs.usw,
                                                  glue_margs.anchors, glue_margs.weight);
                }
        }
        /* (this thread's view of the device ends here; the context stays in the pool) */
        glue_dm_free();
        glue_ctx = NULL; glue_job_ctx = NULL; glue_rows_ctx = NULL; glue_job_msa = NULL; glue_rows_msa = NULL;
        glue_ct_single = NULL;
        return NULL;
}

static void glue_members_free(void)
{
        int k;
        if(glue_members){
                for(k = 0; k < glue_n_members; k++){
                        if(glue_members[k].aln){
                                kalign_free_msa(glue_members[k].aln);
                        }
                }
                MFREE(glue_members);
        }
        glue_members = NULL;
        glue_n_members = 0;
}

/* the member the reference's loop asks for next, if it was run ahead: the finished alignment goes into `msa` */
static int glue_member_take(struct msa* msa, float gpo, float gpe, float tgpe, int refine, uint64_t seed, float noise, int realign)
{
        int k;
        if(!glue_members || refine != glue_margs.refine || realign != glue_margs.realign){
                return 0;
        }
        for(k = 0; k < glue_n_members; k++){
                struct glue_member* m = &glue_members[k];
                if(m->aln && m->rc == OK && m->gpo == gpo && m->gpe == gpe && m->tgpe == tgpe && (realign > 0 || (m->seed == seed && m->noise == noise))
                   && m->aln->numseq == msa->numseq){
                        struct msa tmp = *msa;           /* `msa` is the reference's fresh deep copy of the same input: trade contents */
                        *msa = *m->aln;
                        *m->aln = tmp;
                        kalign_free_msa(m->aln);
                        m->aln = NULL;
                        GLUE_COUNT(GLUE_MEMBER_AHEAD);
                        return 1;
                }
        }
        return 0;
}

int kalign_amd_member_run_seeded(struct msa* msa, int n_threads, int type, float gpo, float gpe, float tgpe, int refine, int adaptive_budget,
                                 uint64_t tree_seed, float tree_noise, float dist_scale, float vsm_amax, float use_seq_weights,
                                 int consistency_anchors, float consistency_weight)
{
        if(!adaptive_budget && glue_member_take(msa, gpo, gpe, tgpe, refine, tree_seed, tree_noise, 0)){
                return OK;
        }
        return kalign_run_seeded(msa, n_threads, type, gpo, gpe, tgpe, refine, adaptive_budget, tree_seed, tree_noise, dist_scale, vsm_amax,
                                 use_seq_weights, consistency_anchors, consistency_weight);
}

int kalign_amd_member_run_realign(struct msa* msa, int n_threads, int type, float gpo, float gpe, float tgpe, int refine, int adaptive_budget,
                                  float dist_scale, float vsm_amax, int realign_iterations, float use_seq_weights,
                                  int consistency_anchors, float consistency_weight)
{
        if(!adaptive_budget && glue_member_take(msa, gpo, gpe, tgpe, refine, 0, 0.0f, realign_iterations)){
                return OK;
        }
        return kalign_run_realign(msa, n_threads, type, gpo, gpe, tgpe, refine, adaptive_budget, dist_scale, vsm_amax, realign_iterations,
                                  use_seq_weights, consistency_anchors, consistency_weight);
}

int kalign_ensemble(struct msa* msa, int n_threads, int type, int n_runs, float gpo, float gpe, float tgpe, uint64_t seed, int min_support,
                    const char* save_poar_path, int refine, float dist_scale, float vsm_amax, int realign, float use_seq_weights,
                    int consistency_anchors, float consistency_weight)
{
        struct aln_param* ap = NULL;
        pthread_t th[GLUE_MAX_MEMBERS];
        struct glue_slot_job jobs[GLUE_MAX_MEMBERS];
        const char* w = getenv("KALIGN_AMD_GLUE_WORLD");
        const char* d = getenv("KALIGN_AMD_DEVICES");
        int slots = 0, shared = 0, started = 0;
        int k, rc;
        if(w && atoi(w) > 1){
                slots = atoi(w);
                shared = 1;
        }else{
                slots = d ? atoi(d) : ka_device_count();
                if(slots > ka_device_count()){
                        slots = ka_device_count();
                }
                if(slots == 1){                  /* one GPU: the members share it */
                        const char* es = getenv("KALIGN_AMD_ENSEMBLE_SLOTS");
                        slots = es ? atoi(es) : GLUE_SHARED_MEMBERS;
                        shared = 1;
                }
        }
        if(slots > GLUE_MAX_MEMBERS){
                slots = GLUE_MAX_MEMBERS;
        }
        if(slots > n_runs){
                slots = n_runs;
        }
        glue_members_free();
        if(msa != NULL && n_runs >= 2 && slots >= 2 && glue_context() == OK){
                /* the prologue of the reference's function (ensemble.c:240-270), so that the members below start from what its loop
                   would copy: checked input, alphabet known, base penalties resolved.  It runs again inside the reference; both are
                   idempotent. */
                float usw = use_seq_weights < 0.0f ? 0.0f : use_seq_weights;
                float base[3];
                RUN(kalign_essential_input_check(msa, 0));
                if(msa->biotype == ALN_BIOTYPE_UNDEF){
                        RUN(detect_alphabet(msa));
                }
                RUN(aln_param_init(&ap, msa->biotype, n_threads, type, gpo, gpe, tgpe));
                base[0] = ap->gpo; base[1] = ap->gpe; base[2] = ap->tgpe;
                aln_param_free(ap);
                ap = NULL;
                MMALLOC(glue_members, sizeof(struct glue_member) * n_runs);
                glue_n_members = n_runs;
                for(k = 0; k < n_runs; k++){
                        const float* sc = glue_member_scale[k % 12];
                        struct glue_member* m = &glue_members[k];
                        m->aln = NULL; m->rc = FAIL;
                        m->gpo = k ? base[0] * sc[0] : base[0];
                        m->gpe = k ? base[1] * sc[1] : base[1];
                        m->tgpe = k ? base[2] * sc[2] : base[2];
                        m->seed = k ? seed + (uint64_t)k : 0;
                        m->noise = k ? sc[3] : 0.0f;
                }
                for(k = 0; k < n_runs; k++){
                        RUN(msa_cpy(&glue_members[k].aln, msa));
                        glue_members[k].aln->quiet = 1;
                }
                glue_margs.n_threads = n_threads; glue_margs.type = type; glue_margs.refine = refine; glue_margs.realign = realign;
                glue_margs.anchors = consistency_anchors; glue_margs.dist_scale = dist_scale; glue_margs.vsm_amax = vsm_amax;
                glue_margs.usw = usw; glue_margs.weight = consistency_weight;
                for(k = 0; k < slots; k++){
                        jobs[k].slot = k; jobs[k].n_slots = slots; jobs[k].device = shared ? 0 : k; jobs[k].shared = shared;
                        if(pthread_create(&th[k], NULL, glue_member_thread, &jobs[k]) != 0){
                                break;
                        }
                        started++;
                }
                for(k = 0; k < started; k++){
                        pthread_join(th[k], NULL);
                }
                /* (members of a slot whose thread did not start, or that failed, stay with rc != OK: the reference's loop runs them) */
                GLUE_COUNT(GLUE_ENSEMBLE_MULTI);
        }
        rc = kalign_ref_kalign_ensemble(msa, n_threads, type, n_runs, gpo, gpe, tgpe, seed, min_support, save_poar_path, refine, dist_scale, vsm_amax,
                                        realign, use_seq_weights, consistency_anchors, consistency_weight);
        /* Members are handed over by EQUALITY of their parameters with the table restated above (glue_member_take): should the
           reference's resolve_run_params ever drift from it, every take misses, every member is computed twice, and nothing fails.
           Say so: a member that finished here and is still here was never asked for. */
        if(rc == OK && glue_members){
                int left = 0;
                for(k = 0; k < glue_n_members; k++){
                        if(glue_members[k].aln && glue_members[k].rc == OK){
                                left++;
                        }
                }
                if(left){
                        fprintf(stderr, "kalign_amd glue: %d of %d ensemble members run ahead were not taken over by kalign_ensemble's loop "
                                        "(member parameters differ from the glue's table: they were computed twice)\n", left, glue_n_members);
                        GLUE_COUNT(GLUE_MEMBER_MISSED);
                }
        }
        glue_members_free();
        return rc;
ERROR:
        if(ap){
                aln_param_free(ap);
        }
        glue_members_free();
        return FAIL;
}
