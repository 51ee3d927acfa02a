/*
 * kalign_amd.h -- C ABI of the MI355X (gfx950) implementation of Kalign's
 * progressive-alignment hot path: the batched seq-seq / seq-profile /
 * profile-profile affine-gap Gotoh DP under a linear-space Hirschberg
 * recursion, and the guide-tree task dispatcher that feeds it.
 *
 * The seam this replaces in the reference (TimoLassmann/kalign v3.5.1) is
 * link-time and in-process:
 *
 *   ka_msa_tree()        <->  create_msa_tree(struct msa*, struct aln_param*, struct aln_tasks*)
 *                              lib/src/aln_run.c:43-78, with do_align :213-441 per task
 *   ka_pairwise_batch()  <->  the N x K loop over pairwise_align_map() -> aln_runner()
 *                              lib/src/anchor_consistency.c:19-120, :246-267
 *   ka_tree_build_consistency() <-> anchor_consistency_build(), lib/src/anchor_consistency.c:194-275, plus the
 *                              per-task bonus of do_align (aln_run.c:262-295) inside ka_tree_run
 *   ka_tree_upload/run/download: the same dispatcher split so that a caller can
 *                              keep inputs resident in HBM (bench.py times ka_tree_run only)
 *
 * All pointers are plain host pointers unless a name says "dev".  Return
 * value 0 = OK, non-zero = FAIL (the reference's convention, tldevel.h:29-45);
 * ka_last_error() gives a message.  INTEGRATION.md shows the exact glue a
 * Kalign maintainer adds in aln_run.c / anchor_consistency.c.
 *
 * Sequence encoding is the reference's internal alphabet (alphabet.c:179-245):
 * protein 0..22, nucleotide 0..4.  Profiles use the reference's 64-float record
 * layout (aln_setup.c:40-99).  Paths are the reference's coded paths
 * (add_gap_info_to_path_n, aln_setup.c:121-228): p[0] = alignment length,
 * p[1..len] ops (0 match, 1 gap in a, 2 gap in b, |32 terminal run), p[len+1] = 3.
 */
#ifndef KALIGN_AMD_H
#define KALIGN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KA_OK 0
#define KA_FAIL 1
#define KA_ERR_PATHS_CAP 2      /* caller's paths_out too small; required size in ka_tree_paths_size() */
#define KA_ERR_ROWS_STRIDE 3    /* ka_run_encoded: rows_out too narrow; alnlen_out holds the length, the alignment is still on the device */

/* flags for ka_msa_tree / ka_tree_upload */
#define KA_FLAG_DEBUG_ROWS 1    /* also keep each task's top-level f/b rows (tests: row hashes) */
#define KA_FLAG_TIMING 2        /* record per-task phase cycle counts (ka_tree_get_timing) */
#define KA_FLAG_DEVICE_GAPS 4   /* keep every residue's alignment column on the device (make_seq / update_gaps,
                                   weave_alignment.c:41-112): ka_tree_download derives gaps_out from it in O(sum of
                                   lengths) instead of folding every task's path on the host; ka_msa_tree sets it
                                   when gaps_out is requested */
#define KA_FLAG_KEEP_CONSISTENCY 8 /* the realignment pass of kalign_run_realign (aln_wrap.c:424-431,497-502): the same
                                   sequences with a new task list keep the consistency table ka_tree_build_consistency
                                   built for the previous job of this context (no table then: none now); fails if the
                                   sequences differ */
#define KA_FLAG_EXACT_CONFIDENCE 16 /* task confidence as the reference's float, bit for bit: the reference adds a task's meetup
                                   margins in recursion order (aln_run.c:391-395, aln_controller.c:194-436), the level-synchronous
                                   first pass in level order (equal within 1e-5).  With this flag every meetup records its margin with
                                   its place in the recursion order; sorted and added in fp32 after the recursion.  Costs a sort per
                                   task and the wave-local subtrees (the records are not kept there); tasks with more than 8192
                                   recorded meetups keep the level-order sum. */
#define KA_FLAG_LEAF_PROFILES 32 /* also write the profile records of the SEQUENCES (make_profile_n, aln_setup.c:40-99) that tasks consume.  Since
                                   round 6 the first pass does not (without sequence weights): a sequence's record is a function of its
                                   residue, and the merge makes what it needs of it on the fly -- ka_tree_get_profile of a node < numseq then
                                   returns unwritten memory.  Set this to read leaf profiles back. */

typedef struct ka_ctx ka_ctx;

/* Per-task result; field-for-field what do_align leaves behind plus the top-level
   Hirschberg meetup (aln_controller.c:21-120).  Same layout as the oracle's ko_task_rec. */
typedef struct ka_task_rec {
        int a, b, c;            /* task (a, b) -> c, struct task (task.h:14-22) */
        int len_a, len_b;       /* operand lengths in (a, b) order */
        int nsip_a, nsip_b;     /* member counts, msa->nsip[] */
        int plen;               /* alignment length, msa->plen[c] */
        int kind;               /* 0 seq-seq, 1 seq-profile, 2 profile-profile */
        int swapped;            /* DP ran with b as rows (aln_run.c:297-388 swap rules) */
        int meet, transition;   /* top-level meetup column / transition code 1,2,3,5,6,7 */
        int path_off;           /* offset of this task's coded path in paths_out */
        float gap_scale;        /* compute_gap_scale (aln_run.c:126-164) */
        float subm_off;         /* compute_subm_offset (aln_run.c:166-203) */
        float score;            /* top-level meetup score (== m->score of ALN_MODE_SCORE_ONLY) */
        float confidence;       /* mean Hirschberg margin, task.confidence (aln_run.c:391-395) */
        uint64_t prof_hash;     /* FNV-1a of the merged profile bytes, 0 for the root / when not requested */
        uint64_t fhash, bhash;  /* FNV-1a of top-level f / b rows (KA_FLAG_DEBUG_ROWS), else 0 */
} ka_task_rec;

int  ka_ctx_create(int device, ka_ctx** out);
void ka_ctx_destroy(ka_ctx* ctx);
/* Launch on the caller's HIP stream (a hipStream_t, e.g. torch's current stream). NULL = default stream. */
int  ka_ctx_set_stream(ka_ctx* ctx, void* hip_stream);
/* Tell the context that it does NOT have the GPU to itself (several alignments in flight on different streams /
   processes).  Multi-workgroup tasks and the chained launch make workgroups wait for each other and need all of
   them resident; a shared context runs every task on one workgroup, one launch per guide-tree level -- slower for
   a single tree, safe under any co-scheduling.  Takes effect at the next ka_tree_upload.  A shared context that was
   given no stream (ka_ctx_set_stream) creates a non-blocking stream of its own: contexts of one process that stay on
   the null stream run one after the other however many host threads drive them. */
int  ka_ctx_set_shared(ka_ctx* ctx, int shared);
/* How often a run of this context fell back to that plan on its own because a wait between workgroups never
   completed (another process was using the GPU): the run is repeated and correct, but slower -- visible here. */
int  ka_ctx_fallback_runs(ka_ctx* ctx);
/* Of the last finished run: tasks of the queued launch that workgroups of the chained launch took over because they were resident
   before the queue was down to its last round (overlapping launches; normally 0).  -1: nothing finished. */
long long ka_ctx_helped_tasks(ka_ctx* ctx);
const char* ka_last_error(void);
/* ABI revision of this header (bumped when entry points are added) */
int  ka_abi_version(void);

/*
 * The dispatcher.  numseq sequences (concatenated codes, off[i], lens[i]),
 * n_tasks = numseq-1 tasks in TASK_ORDER_TREE order (children before parents,
 * root last; task.c:114-136) as abc[3*t + {0,1,2}].
 * subm: 23*23 floats; scal[6] = gpo, gpe, tgpe, dist_scale, vsm_amax, use_seq_weights
 * (struct aln_param, aln_param.h:19-34, after the sentinel resolution of aln_param_init).
 * seq_distances: numseq floats or NULL (msa->seq_distances).
 * n_tasks < numseq-1 describes a FOREST: several independent alignments (a batch of families, ensemble members)
 * scheduled together -- node ids stay unique, every task nobody consumes is the root of its tree, and the
 * levels of all trees share launches (the GPU is mostly idle at the top of a single tree).
 * Outputs: recs[n_tasks]; coded paths packed into paths_out (capacity paths_cap ints);
 * gaps_out: concatenated gaps[len+1] per sequence (msa->sequences[i]->gaps after make_seq,
 * weave_alignment.c:41-112) or NULL.
 */
int ka_msa_tree(ka_ctx* ctx, int numseq, const uint8_t* codes, const int* off, const int* lens,
                const float* seq_distances, int n_tasks, const int* tasks_abc,
                const float* subm, const float* scal, int flags,
                ka_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out);

/* The same, staged: upload (H2D + host-side task preparation), run (device only; may be
   called repeatedly -- it resets the device state first), download (D2H + gap weaving). */
int ka_tree_upload(ka_ctx* ctx, int numseq, const uint8_t* codes, const int* off, const int* lens,
                   const float* seq_distances, int n_tasks, const int* tasks_abc,
                   const float* subm, const float* scal, int flags);
int ka_tree_run(ka_ctx* ctx);
/* refine_alignment (aln_refine.c:34-85): a second pass over every edge of the uploaded tree with refine_edge's flip
   trials (aln_refine.c:199-325: trial 0 is the baseline, trials 1-4 take the second-best meetup round-robin where the
   margin is below the baseline's mean margin; the trial with the best sum-of-pairs score is kept, the first one on
   ties).  mode 1 = KALIGN_REFINE_ALL, 2 = KALIGN_REFINE_CONFIDENT (only edges whose first-pass confidence is at or
   below the median are refined, the others are re-aligned plainly), 3 = KALIGN_REFINE_INLINE
   (create_msa_tree_inline_refine, aln_run.c:448-790, as aln_wrap.c:222-224 calls it: one pass, three trials on every
   edge, first-pass path coding, task confidence = the kept trial's sum-of-pairs score), 4 = no refinement: the first
   pass again, run depth first like the reference so that every task's confidence is the reference's exact float sum
   (ka_tree_run adds the same margins level by level: equal within 1e-5).  conf_in[n_tasks]: first-pass task
   confidences (the reference reads task->confidence); only read for mode 2; NULL = computed here with a mode-4 pass.  Asynchronous like
   ka_tree_run; ka_tree_sync / ka_tree_download then return the refined records, paths and gaps (records carry the
   kept trial's confidence). */
#define KA_REFINE_ADAPTIVE 256      /* mode | KA_REFINE_ADAPTIVE: aln_param's adaptive_budget (`kalign --adaptive-budget`,
                                       aln_refine.c:187-193, 255-282; modes 1 and 2): 1 .. 8 trials per edge, from the share of
                                       very uncertain meetups of its baseline trial */
#define KA_REFINE_TRIALS(n) (((n) & 255) << 16)   /* 3 | KA_REFINE_TRIALS(n): create_msa_tree_inline_refine with n trials per edge
                                                     (lib/src/aln_run.c:448-475 takes any count; kalign_run passes 3, the default here) */
int ka_tree_refine(ka_ctx* ctx, int mode, const float* conf_in);
int ka_tree_sync(ka_ctx* ctx);
long long ka_tree_paths_size(ka_ctx* ctx);      /* ints needed for paths_out (valid after run+sync) */
int ka_tree_download(ka_ctx* ctx, ka_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out);

/* The aligned rows themselves (finalise_alignment + make_linear_sequence, lib/src/msa_op.c:546-598), built on the
 * device from the residue->column tables of a job uploaded with KA_FLAG_DEVICE_GAPS, after a complete run.
 *   letters[sum of lens]  the residues as the caller wants them printed (struct msa_seq.seq), laid out like `codes`
 *   rows_out[numseq * row_stride]  row i holds alnlen(i) bytes -- letters, and gap_char where the reference puts
 *                         '-' -- followed by a 0 byte; row_stride >= the longest alignment + 1
 *   alnlen_out[numseq]    (may be NULL) alignment length of the tree sequence i belongs to (struct msa.alnlen)
 * rows_out == NULL: only alnlen_out is filled (size query). */
int ka_tree_aligned_rows(ka_ctx* ctx, const uint8_t* letters, uint8_t gap_char, uint8_t* rows_out,
                         long long row_stride, int* alnlen_out);
/*
 * Partial runs -- what single-tree multi-GPU sharding is made of (kalign_amd/dist.py:sharded_tree
 * drives them with one process per GPU): a rank runs the tasks of its subtree, the profile of a subtree root moves
 * to the rank that runs its parent (ka_tree_get_profile -> send/recv -> ka_tree_set_profile), rank 0 gathers all
 * records and paths and weaves the gap arrays.  Results do not depend on the number of ranks.
 *   ka_tree_run_tasks: run the listed tasks (children must be available here: leaves, earlier tasks, injected
 *     profiles); does not reset the device state (ka_tree_upload / ka_tree_run do).
 *   ka_tree_reset: forget every computed / injected node (a new sharded run over the same upload).
 *   ka_tree_node_len: alignment length of a node on this context (-1 on error).
 *   ka_tree_set_profile: inject the merged profile of an internal node, (plen+2)*64 floats.
 *   ka_tree_get_node_cols / ka_tree_set_node_cols: with a consistency table, the residue -> column table of the
 *     node's member sequences (ka_tree_node_cols_size ints) travels with the profile.
 *   ka_tree_download_tasks: records + coded paths of the listed tasks, paths packed in list order.
 *   ka_weave_gaps: host-only make_seq/update_gaps over all tasks in tree order (weave_alignment.c:41-112);
 *     recs[t] needs a, b, c, path_off.
 */
int ka_tree_run_tasks(ka_ctx* ctx, const int* task_ids, int n);
/* The listed tasks as ONE planned run -- queued and chained launches where they apply, like a whole tree (what a rank of
 * a sharded tree does with its subtrees; ka_tree_run_tasks launches level by level).  The set must be closed under
 * descendants among the tasks not run yet.  task_ids == NULL plans the whole tree again (ka_tree_run does that itself).
 * ka_tree_run_planned runs the planned tasks on top of what the context holds; no reset. */
int ka_tree_plan_tasks(ka_ctx* ctx, const int* task_ids, int n);
int ka_tree_run_planned(ka_ctx* ctx);
/* The same hand-over without the host: ka_tree_profile_dev says where the (plen+2)*64 floats of a node's profile lie in
 * this context's HBM, ka_tree_reserve_profile_dev reserves room for an incoming profile of `plen` columns on the
 * receiving context and makes the node available to ka_tree_run_tasks; the caller moves the bytes device to device
 * (RCCL send / recv over xGMI -- kalign_amd/dist.py -- or hipMemcpyPeer) before it runs the parent. */
int ka_tree_profile_dev(ka_ctx* ctx, int node, void** dev_ptr, int* plen_out);
int ka_tree_reserve_profile_dev(ka_ctx* ctx, int node, int plen, void** dev_ptr);
int ka_tree_reset(ka_ctx* ctx);
int ka_tree_node_len(ka_ctx* ctx, int node);
int ka_tree_set_profile(ka_ctx* ctx, int node, const float* prof, int plen);
long long ka_tree_node_cols_size(ka_ctx* ctx, int node);
int ka_tree_get_node_cols(ka_ctx* ctx, int node, int* out);
int ka_tree_set_node_cols(ka_ctx* ctx, int node, const int* cols);
int ka_tree_download_tasks(ka_ctx* ctx, const int* task_ids, int n, ka_task_rec* recs, int* paths_out,
                           long long paths_cap, long long* used_out);
int ka_weave_gaps(int numseq, const int* lens, int n_tasks, const ka_task_rec* recs, const int* paths, int* gaps_out);
/* After a run made elsewhere (a sharded tree: records gathered, gap arrays woven on the host) this context -- same uploaded
 * job -- becomes the holder of the finished alignment: recs[n_tasks] (plen, c) and gaps (ka_tree_download's layout) in;
 * ka_tree_aligned_rows, ka_aln_guide_tree and ka_tree_refine then carry on here as after ka_tree_run + ka_tree_sync
 * (finalise_alignment msa_op.c:546-598, refine_alignment aln_refine.c:36-88, compute_aln_pairwise_dist aln_apair_dist.c:9-86). */
int ka_tree_adopt_alignment(ka_ctx* ctx, const ka_task_rec* recs, const int* gaps);

/* Merged profile of node `node` ((plen+2)*64 floats) after a run. */
int ka_tree_get_profile(ka_ctx* ctx, int node, float* out, long long cap_floats);
/* Per-task phase timings of the last run (KA_FLAG_TIMING): out[8*t + k], shader-clock cycles:
   0 operand prep, 1 Hirschberg, 2 path coding, 3 profile merge, 4 passes, 5 meetups, 6 recursion
   levels | workgroups used << 8 | workgroups offered << 16, 7 DP rows*cols; followed by 16 x (sub-problems, pass cycles, meetup cycles) per
   recursion level of the root task, followed by 512 values that only profiling builds (-DKA_PROF) fill:
   out must hold 8*n_tasks + 48 + 512 values. */
int ka_tree_get_timing(ka_ctx* ctx, long long* out);
/* Tests only -- fault injection for the recovery paths of ka_tree_sync (takes effect at the next ka_tree_upload):
   KA_DEBUG_SMALL_ARENAS starts with device arenas that are certainly too small (overflow -> grow -> re-run),
   KA_DEBUG_STARVE_ROOT_JOIN makes the root's join wait for a workgroup that never comes (watchdog -> re-plan). */
#define KA_DEBUG_SMALL_ARENAS 1
#define KA_DEBUG_STARVE_ROOT_JOIN 2
#define KA_DEBUG_STARVE_REFINE_MEMBER 4   /* ka_tree_refine: a member of the first multi-workgroup edge never starts (watchdog -> re-plan) */
#define KA_DEBUG_CHAIN_FIRST 8            /* overlapping launches: the chained launch is enqueued BEFORE the queued one -- its workgroups take the CUs
                                             first, as a dispatcher that ignores the streams' priorities would have it (they help the queue) */
#define KA_DEBUG_POISON_ARENAS 16         /* every run starts with the profile, scratch and path arenas filled with 0xff bytes (NaN / -1): nothing may
                                             depend on what an arena held before (round 6: sequences' profile records are no longer written) */
int ka_debug_set_hooks(ka_ctx* ctx, int hooks);
/* Tools and tests: the KA_* environment switches (experiments and measurements; none is needed in production) are read
   once, at ka_ctx_create.  This reads them again and rebuilds the launch plan of the uploaded job. */
int ka_debug_reload_env(ka_ctx* ctx);
/* Debug: 64 breadcrumb words written by workgroup 0 (context created with KA_TRACE=1 in the
   environment); readable while a kernel is still running. */
int ka_debug_trace(ka_ctx* ctx, int* out64);
/* Launches of the throughput kernel (round 6: unit 10, ka_task_kernel_tp; opt-in, KA_TP=1 in the environment) since the library was
 * loaded -- tests assert that the path they mean to test is the one that ran. */
long long ka_debug_tp_launches(void);
/* Work done by the last run: sum over tasks of len_a*len_b ("useful cells") */
double ka_tree_cells(ka_ctx* ctx);
/* Milliseconds spent in the DP kernels of the last ka_tree_run, measured with HIP events
   on the launch stream; n_launches receives the number of kernel launches. */
int ka_tree_kernel_ms(ka_ctx* ctx, float* ms, int* n_launches);
/* Measurements: the duration of every launch of the last run (needs KA_LAUNCH_EV=1 in the environment when the
   context is created); returns the number of values written (<= cap), -1 on error. */
int ka_tree_launch_ms(ka_ctx* ctx, float* ms, int cap);

/*
 * Anchor consistency = the reference's default mode (anchor_consistency_build, anchor_consistency.c:194-275,
 * called from kalign_run_seeded, aln_wrap.c:207-214).  Call between ka_tree_upload and ka_tree_run:
 * selects n_anchors anchors from seq_distances, aligns every sequence to every anchor on the device, keeps
 * the position maps in HBM; from then on every DP of ka_tree_run adds the consistency bonus that do_align
 * builds with anchor_consistency_get_bonus_profile (aln_run.c:262-295).  Like the reference it declines
 * silently (returns OK, no table) when n_anchors <= 0, numseq < 3 or there are no seq_distances.
 * n_anchors <= KA_CONS_MAX_ANCHORS (the reference's default is 5, src/parameters.c:72-73; weight 2.0).  Any number of sequences per
 * alignment: the member votes of a node count in 16 bits below 65536 members and in 32 bits from there on.
 * In a forest job every alignment gets its own table (its own anchors among its own sequences).
 * ka_tree_upload drops the table again.
 */
#define KA_CONS_MAX_ANCHORS 128 /* anchors ka_tree_build_consistency takes: up to 5 on the kernels every default-mode job uses (a DP row carries that many bonus
                                   entries + the wrap-around one in registers), 6..128 on a second set that walks a row's entries, sorted by column, along
                                   with the row's columns (`--consistency K`; round 4 stopped at 10, round 5 at 32; the reference, anchor_consistency.c:200-275,
                                   has no cap but the number of sequences) */
int ka_tree_build_consistency(ka_ctx* ctx, int n_anchors, float weight);
/* The N x K batch sharded over `nparts` GPUs: every rank uploads the same job and calls this with its
 * own `part`; it selects the same anchors, aligns only its contiguous share of the sequences (balanced by length) and
 * fills their position maps.  The table is complete once every rank's share has been copied into every other rank's
 * table: ka_tree_consistency_maps_dev gives the table in HBM (total_ints int32), ka_tree_consistency_part_range the
 * int range [lo, hi) part `part` fills -- kalign_amd/dist.py broadcasts each range in place with RCCL. */
int ka_tree_build_consistency_part(ka_ctx* ctx, int n_anchors, float weight, int part, int nparts);
int ka_tree_consistency_part_range(ka_ctx* ctx, int part, int nparts, long long* lo, long long* hi);
int ka_tree_consistency_maps_dev(ka_ctx* ctx, void** maps_dev, long long* total_ints);
/* Returns K (0: no table).  anchor_ids[K] (a forest job: K per alignment that has a table, in order of the
   alignments' first sequences); maps_out: all position maps concatenated in (i*K + k) order,
   each lens[i] ints (pos_maps, anchor_consistency.h:17-24).  Either pointer may be NULL. */
int ka_tree_get_consistency(ka_ctx* ctx, int* anchor_ids, int* maps_out);

/*
 * npairs independent seq-seq alignments (pair k = sequences ia[k], ib[k]) with unscaled
 * parameters, rows = the shorter sequence with `len_i <= len_j` deciding the swap
 * (anchor_consistency.c:44-61).  paths_out receives the coded path of pair k at poff[k]
 * (room for lens[ia]+lens[ib]+3 ints each); scores_out (optional) the top-level score.
 */
int ka_pairwise_batch(ka_ctx* ctx, const uint8_t* codes, const int* off, const int* lens, int numseq,
                      const int* ia, const int* ib, int npairs,
                      const float* subm, float gpo, float gpe, float tgpe,
                      int* paths_out, const long long* poff, float* scores_out);

/*
 * Distance estimation for the guide tree: for every pair (ia[k], ib[k]) the value of
 * calc_distance() -> bpm_block() (lib/src/sequence_distance.c:150-162, lib/src/bpm.c:356-582): block-wise Myers
 * bit-vector edit distance of the shorter sequence against the longer one, integer-exact, at most 1024 pattern
 * positions.  Sequence codes must be < 13 (the reduced alphabet kalign_run converts to before tree building,
 * aln_wrap.c:155-160, or nucleotides).  d_estimation's length term is left to the caller:
 * dm = dist + min(10000, (len_a + len_b) / 2) / 10000 (sequence_distance.c:66-69,118-120).
 */
int ka_bpm_batch(ka_ctx* ctx, const uint8_t* codes, const int* off, const int* lens, int numseq,
                 const int* ia, const int* ib, int npairs, int* dist_out);

/*
 * Realignment (kalign_run_realign, lib/src/aln_wrap.c:449-495; the member runs of `--precise`): a new guide tree from
 * a finished alignment.  compute_aln_pairwise_dist (lib/src/aln_apair_dist.c:9-86: 1 - identity over the columns
 * where both rows have a residue, all N(N-1)/2 pairs) and build_tree_from_pairwise (bisectingKmeans.c:1150-1200:
 * mean distance per sequence, then UPGMA on the N x N matrix, :974-1053) both run on the device; labels and tasks as
 * create_tasks makes them.
 *   rows             numseq rows of alnlen bytes, row_stride apart, gap_char where the reference has '-'; or NULL:
 *                    the rows the last ka_tree_aligned_rows built, still in HBM (numseq must match; row_stride,
 *                    alnlen, gap_char are ignored)
 *   tasks_abc[3*(numseq-1)], seq_distances[numseq] (may be NULL)  as for ka_guide_tree
 *   dm_out[numseq*numseq]  (may be NULL) the identity distances, before UPGMA consumed them
 */
int ka_aln_guide_tree(ka_ctx* ctx, int numseq, const uint8_t* rows, long long row_stride, int alnlen, uint8_t gap_char,
                      int* tasks_abc, float* seq_distances, float* dm_out);

/*
 * kalign_run_seeded / kalign_run_realign (lib/src/aln_wrap.c:144-251, :361-527) from "sequences encoded" to "rows
 * finalised" in one call -- ka_guide_tree, ka_tree_upload, ka_tree_build_consistency, ka_tree_run, and per realignment
 * iteration rows (kept in HBM) -> ka_aln_guide_tree -> upload with KA_FLAG_KEEP_CONSISTENCY -> run; finally the rows.
 * Sequences in the order msa_sort_len_name left them.
 *   tree_codes / codes / letters   the sequences three times, all laid out by off[] / lens[]: in the alphabet of the
 *                    guide tree (reduced protein alphabet / nucleotides), in the alignment alphabet, and as printed
 *   subm, scal       as for ka_tree_upload; n_anchors 0: the reference's --fast; dm_scale NULL or the noisy tree's
 *                    multipliers (ka_guide_tree); realign_iterations 0: kalign_run_seeded
 *   rows_out[numseq * row_stride], alnlen_out[numseq]   as for ka_tree_aligned_rows; rows_out NULL leaves the alignment
 *                    on the device (alnlen_out says how long); a row_stride that turns out too small returns
 *                    KA_ERR_ROWS_STRIDE with alnlen_out filled -- ka_tree_aligned_rows then fetches the rows
 */
int ka_run_encoded(ka_ctx* ctx, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                   const int* off, const int* lens, const float* subm, const float* scal,
                   int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads,
                   uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out);
/* The same with refinement (kalign_run_seeded / kalign_run_realign's `refine` argument): refine_mode 0 none, 1 / 2
   (optionally | KA_REFINE_ADAPTIVE) refine_alignment after the last alignment (aln_wrap.c:229-232, :506-509), 3
   KALIGN_REFINE_INLINE: every alignment of the run is create_msa_tree_inline_refine (:222-226, :498-502). */
int ka_run_encoded_refine(ka_ctx* ctx, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                   const int* off, const int* lens, const float* subm, const float* scal,
                   int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads, int refine_mode,
                   uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out);

/*
 * Guide tree: build_tree_kmeans (lib/src/bisectingKmeans.c:177-271) -- anchors by length
 * (pick_anchor.c:34-70), N x 32 distances to the anchors, bisecting 2-means on that matrix (split2, :766-971) down
 * to clusters of fewer than 50 sequences, UPGMA on all-pairs distances inside each cluster (:974-1053), post-order
 * node labels.  The two distance batches run on the device (ka_bpm_batch); the clustering between them runs on the
 * host in the reference's fp32 evaluation order, so the task list is the reference's task list.
 *   codes            the sequences in the alphabet the reference builds its tree in: reduced protein alphabet
 *                    (ALPHA_redPROTEIN, aln_wrap.c:155-160) or nucleotides; sorted as msa_sort_len_name left them
 *   n_threads        host threads for the independent halves of the bisection (the result does not depend on it)
 *   dm_scale[numseq * min(32, numseq)]  NULL, or build_tree_kmeans_noisy (:76-175, the trees of ensemble members):
 *                    the multiplier of every anchor distance, row-major [sequence][anchor] -- what the reference
 *                    draws as max(0.1, tl_random_gaussian(rng, 1.0, sigma)) cast to float, in that order (:103-115).
 *                    The random numbers stay the caller's: they come from the reference's own generator.
 *   tasks_abc[3*(numseq-1)]  (a, b, c) in TASK_ORDER_TREE order -- what ka_tree_upload / ka_msa_tree take
 *   seq_distances[numseq]    msa->seq_distances (:244-255); may be NULL
 */
int ka_guide_tree(ka_ctx* ctx, int numseq, const uint8_t* codes, const int* off, const int* lens,
                  int n_threads, const float* dm_scale, int* tasks_abc, float* seq_distances);
/* The same with the caller's distance source: dist() must fill dist_out[k] with calc_distance(seq ia[k], seq ib[k])
 * (sequence_distance.c:150-162) for k < npairs and return 0; it is called twice (anchor batch, cluster batch).
 * Host only: needs no context and no GPU. */
typedef int (*ka_dist_fn)(void* user, int npairs, const int* ia, const int* ib, int* dist_out);
int ka_guide_tree_from(int numseq, const int* lens, ka_dist_fn dist, void* user, int n_threads,
                       const float* dm_scale, int* tasks_abc, float* seq_distances);

/* Kernel time (HIP events on the launch stream) of the last ka_pairwise_batch / ka_bpm_batch, milliseconds. */
float ka_pairwise_kernel_ms(ka_ctx* ctx);


/*
 * ONE alignment over the GPUs of a node, driven from C (the reference's unit of parallelism is the
 * independent subtree, lib/src/aln_run.c:95-109): one process per GPU, RCCL (ncclBroadcast / ncclSend / ncclRecv /
 * ncclAllReduce over xGMI), loaded at run time -- the single-GPU library has no link-time dependency on it.
 *   ka_dist_unique_id   rank 0: the 128-byte ncclUniqueId; the launcher hands it to every rank (file, MPI, torch ...).
 *   ka_dist_create      the communicator of this rank's context (ncclCommInitRank).  world == 1 and id == NULL: none.
 *   ka_dist_plan        once per uploaded job (every rank uploads the same job): the tree is cut into one subtree per
 *                       rank, balanced by estimated DP cells; the hand-overs above the cut; this rank's subtrees planned
 *                       as one run (ka_tree_plan_tasks).  ka_dist_plan_subtrees is the pure planning function.
 *   ka_dist_consistency anchor_consistency_build sharded: this rank's share of the N x K batch, every share broadcast
 *                       in place, HBM to HBM; fails on every rank alike when one part cannot be built.
 *   ka_dist_tree_run    one step: subtrees, the merges above the cut (profiles -- and in default mode the residue ->
 *                       column tables, packed on the device -- move device to device), then every rank holds every
 *                       record and coded path (one all-reduce each over disjoint ranges).
 *   ka_dist_download    records in task order (path_off into paths) and the coded paths, identical on every rank and for
 *                       every world size.
 */
typedef struct ka_dist ka_dist;
int ka_dist_unique_id(void* id128);
int ka_dist_create(ka_ctx* ctx, int rank, int world, const void* id128, ka_dist** out);
void ka_dist_destroy(ka_dist* d);
int ka_dist_plan_subtrees(int numseq, const int* lens, int n_tasks, const int* tasks_abc, int world, int* run_rank, int* top, int* n_top);
int ka_dist_plan(ka_dist* d);
int ka_dist_get_plan(ka_dist* d, int* run_rank, int* top, int* n_top, int* n_moves);
int ka_dist_consistency(ka_dist* d, int n_anchors, float weight);
int ka_dist_tree_run(ka_dist* d);
long long ka_dist_paths_size(ka_dist* d);
int ka_dist_download(ka_dist* d, ka_task_rec* recs, int* paths, long long paths_cap, long long* used);
double ka_dist_last_ms(ka_dist* d);
/* Steps ka_dist_tree_run repeated because a device arena overflowed on SOME rank: the ranks agree on the outcome of their parts
   (all-reduce of a status word) before the gather's collectives, the ranks that overflowed grow their arenas and every rank
   runs the step again -- no rank fails, or waits in a collective, alone. */
int ka_dist_retries(ka_dist* d);
/* Tests: the ranks as threads of ONE process, each with its own context on the same GPU (RCCL refuses two ranks on one
   device): an in-process stand-in for the communicator with the same call sequence, host-synchronous. */
void* ka_dist_loopback_new(int world);
void ka_dist_loopback_free(void* loopback);
int ka_dist_create_loopback(ka_ctx* ctx, int rank, int world, void* loopback, ka_dist** out);


/* ---- the GPUs of one node under ONE caller (ka_multi.cpp) ----------------------------------------------------------------
 * ka_dist_* wants one caller per rank; a single-process C program (kalign_run behind the drop-in glue) has
 * one.  ka_multi_* runs the ranks as threads of the caller, one context + one ka_dist per device, behind calls shaped like
 * ka_tree_build_consistency / ka_tree_upload + run + download; results are the single-GPU results bit for bit
 * (lib/src/aln_run.c:95-109).  devices == NULL: devices 0 .. world-1; loopback != 0: every rank on ONE device over the
 * in-process transport (tests on one-GPU boxes).  Errors: ka_multi_last_error().
 *   ka_multi_consistency  anchor_consistency_build (anchor_consistency.c:200-275) sharded; returns the number of anchors (0: the
 *                         job declines, like the reference), < 0 on error; anchor_ids_out / maps_out as ka_tree_get_consistency
 *   ka_multi_tree_run     create_msa_tree (aln_run.c:43-78): ka_tree_upload's arguments; n_anchors > 0 = default mode (the table
 *                         is built unless flags carry KA_FLAG_KEEP_CONSISTENCY and ka_multi_consistency left it on the ranks)
 *   ka_multi_download     rank 0's records and coded paths, gaps woven on the host (gaps_out may be NULL)
 */
/* build_tree_kmeans' 2-means bisection (bisectingKmeans.c:273-402, split2 :766-971) runs on the device inside ka_guide_tree from
   2048 sequences up (ka_kmeans.hip: all of a set's up to 40 seeds side by side, every set of a recursion level in one launch,
   the order-dependent fp32 sums as chains in the reference's sample order -- the same tree, bit for bit); KA_KMEANS=0 / 1 in
   the environment forces the host / the device.  Wall time of the bisection inside the last ka_guide_tree of this process and
   where it ran (measurements). */
double ka_guide_last_bisect_ms(int* on_device);

typedef struct ka_multi ka_multi;
int ka_device_count(void);                       /* visible GPUs (hipGetDeviceCount; 0 when there is none or no driver) */
int ka_multi_create(int world, const int* devices, int loopback, ka_multi** out);
void ka_multi_destroy(ka_multi* m);
int ka_multi_world(ka_multi* m);
long long ka_multi_runs(ka_multi* m);
const char* ka_multi_last_error(void);
int ka_multi_consistency(ka_multi* m, int numseq, const uint8_t* codes, const int* off, const int* lens, const float* seq_distances,
                         int n_tasks, const int* tasks_abc, const float* subm, const float* scal, int flags,
                         int n_anchors, float weight, int* anchor_ids_out, int* maps_out);
int ka_multi_tree_run(ka_multi* m, int numseq, const uint8_t* codes, const int* off, const int* lens, const float* seq_distances,
                      int n_tasks, const int* tasks_abc, const float* subm, const float* scal, int flags,
                      int n_anchors, float weight);
long long ka_multi_paths_size(ka_multi* m);
int ka_multi_download(ka_multi* m, int numseq, const int* lens, int n_tasks, ka_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out);
/* rank r's single-GPU context (borrowed: ka_multi_destroy frees it); ka_multi_adopt: rank 0's context takes the alignment of the
 * last sharded run over (recs / gaps as ka_multi_download returned them), see ka_tree_adopt_alignment */
ka_ctx* ka_multi_ctx(ka_multi* m, int rank);
int ka_multi_adopt(ka_multi* m, const ka_task_rec* recs, const int* gaps);

#ifdef __cplusplus
}
#endif
#endif
