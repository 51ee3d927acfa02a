// ka_api.cpp -- host side of the C ABI (include/kalign_amd.h): the guide-tree level scheduler
// that replaces create_msa_tree / recursive_aln (aln_run.c:43-124) and the gap weaving that
// follows each task (weave_alignment.c:41-112).
//
// The reference walks the guide tree post-order with OpenMP tasks, one do_align per node.
// Here the tree is cut into dependency levels on the host (level(c) = 1 + max(level(a),
// level(b))); every level is ONE kernel launch with one workgroup per task, and profiles,
// paths and all per-task state stay in HBM between levels.  Only the coded paths and the
// per-task records come back to the host, once, at the end.
#include "ka_ctx.h"

static thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return KA_FAIL; }
// ka_guide.cpp reports through the same message (library-internal, not part of the ABI)
__attribute__((visibility("hidden"))) int ka_fail_message(const char* m) { return fail(m); }

extern "C" const char* ka_last_error(void) { return g_err.c_str(); }
struct ka_ctx;
int ka_ctx_device_stream(ka_ctx* c, int* device, hipStream_t* stream);    // (library-internal: ka_guide.cpp)
extern "C" int ka_abi_version(void) { return 9; }

extern "C" int ka_ctx_create(int device, ka_ctx** out)
{
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail("no HIP device visible (the HIP path has no CPU fallback)");
        if (device < 0 || device >= n) return fail("bad device index");
        HIPCHK(hipSetDevice(device));
        ka_ctx* c = new ka_ctx();
        c->device = device;
        {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
        }
        hipError_t e = hipEventCreate(&c->ev0);
        if (e == hipSuccess) e = hipEventCreate(&c->ev1);
        read_env(c->env);
        if (e == hipSuccess && c->env.trace) {
                e = hipHostMalloc((void**)&c->h_trace, 64 * sizeof(int), hipHostMallocMapped);
                if (e == hipSuccess) memset(c->h_trace, 0xff, 64 * sizeof(int));
        }
        if (e != hipSuccess) {
                ka_ctx_destroy(c);
                return fail(std::string("ka_ctx_create: ") + hipGetErrorString(e));
        }
        *out = c;
        return KA_OK;
}

// Tests only: fault injection that used to hide behind environment variables.
__attribute__((visibility("hidden"))) int ka_ctx_device_stream(ka_ctx* c, int* device, hipStream_t* stream)
{
        if (!c) return 1;
        *device = c->device; *stream = c->stream;
        return 0;
}

extern "C" int ka_debug_set_hooks(ka_ctx* c, int hooks)
{
        if (!c) return fail("null ctx");
        c->test_hooks = hooks;
        return KA_OK;
}

// Tools and tests that flip a KA_* switch on a live context: the environment is otherwise read once, at ka_ctx_create.
// The launch plan of an uploaded job is rebuilt.
extern "C" int ka_debug_reload_env(ka_ctx* c)
{
        if (!c) return fail("null ctx");
        read_env(c->env);
        if (c->have_job) {
                if (c->ran && !c->synced && ka_tree_sync(c)) return KA_FAIL;
                for (auto& d : c->descs) d.refine = 0;
                c->refine_mode = 0;
                if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
        }
        return KA_OK;
}

// How often a run of this context had to fall back to the no-cluster / no-chain plan because workgroups that wait for
// each other were not all resident (somebody else was using the GPU).
extern "C" int ka_ctx_fallback_runs(ka_ctx* c) { return c ? c->fallback_runs : -1; }
extern "C" long long ka_ctx_helped_tasks(ka_ctx* c) { return (c && c->synced) ? (long long)c->h_counters[5] : -1; }

extern "C" void ka_ctx_destroy(ka_ctx* c)
{
        if (!c) return;
        (void)hipSetDevice(c->device);
        c->d_codes.release(); c->d_seq_off.release(); c->d_node_len.release(); c->d_level_ids.release();
        c->d_path_arena.release(); c->d_error.release(); c->d_node_prof.release(); c->d_dbg_off.release();
        c->d_prof_arena.release(); c->d_node_vote.release(); c->d_subm.release(); c->d_dbg_arena.release(); c->d_counters.release();
        c->d_scratch.release(); c->d_tasks.release(); c->d_recs.release(); c->d_timing.release();
        c->d_ctl.release(); c->d_blocks.release(); c->d_blocks_tmp.release(); c->d_join.release();
        c->p_codes.release(); c->p_off.release(); c->p_len.release(); c->p_ia.release(); c->p_ib.release(); c->p_paths.release();
        c->p_err.release(); c->p_subm.release(); c->p_scores.release(); c->p_poff.release(); c->p_scr.release();
        c->b_peq.release(); c->b_dist.release();
        c->d_letters.release(); c->d_rows.release(); c->d_alnlen.release(); c->d_pair_of.release();
        c->d_adm.release(); c->d_amean.release(); c->d_uactive.release(); c->d_ucand.release(); c->d_umerges.release();
        c->d_cons_maps.release(); c->d_colof.release(); c->d_colof_init.release(); c->d_sip.release();
        c->d_cons_map_off.release(); c->d_sip_off.release();
        for (int k = 0; k < 2; k++) { if (c->pin[k]) (void)hipHostFree(c->pin[k]); if (c->pin_ev[k]) (void)hipEventDestroy(c->pin_ev[k]); }
        if (c->h_trace) (void)hipHostFree(c->h_trace);
        for (hipEvent_t e : c->launch_ev) (void)hipEventDestroy(e);
        if (c->s_chain) (void)hipStreamDestroy(c->s_chain);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        for (hipEvent_t e : { c->e_fork, c->e_chain }) if (e) (void)hipEventDestroy(e);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        delete c;
}

// Clusters of workgroups and the chained launch let workgroups wait for each other, which is only safe while
// all of them are resident -- true when this context has the GPU to itself.  A context that shares the GPU
// with other streams or processes (several alignments in flight at once) must say so: every task then runs on
// one workgroup and every guide-tree level is its own launch.  Takes effect at the next ka_tree_upload.
extern "C" int ka_ctx_set_shared(ka_ctx* c, int shared)
{
        if (!c) return fail("null ctx");
        c->shared_gpu = shared != 0; c->shared_by_fallback = false;
        // Contexts of one process that stay on the null stream run one after the other however many host threads drive them.  A
        // shared context whose caller gave it no stream gets its own (non-blocking: it does not order against the null stream either).
        if (c->shared_gpu && !c->stream) {
                HIPCHK(hipSetDevice(c->device));
                HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
                c->own_stream = true;
        }
        return KA_OK;
}

extern "C" int ka_ctx_set_stream(ka_ctx* c, void* s)
{
        if (!c) return fail("null ctx");
        if (c->own_stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); c->own_stream = false; }
        c->stream = (hipStream_t)s;
        return KA_OK;
}

// Device -> caller's memory.  A plain hipMemcpy into pageable memory runs at 2-3 GB/s here; large copies go through
// two pinned bounce buffers instead: the DMA of chunk k+1 overlaps the host-side copy of chunk k.  Ordered after
// everything queued on the context's stream; complete on return.
int copy_to_host(ka_ctx* c, void* dst, const void* src, size_t bytes)
{
        const size_t CH = (size_t)1 << 20;
        bool staged = bytes >= 2 * CH && !c->env.no_staging;
        for (int k = 0; staged && k < 2; k++) {
                if (!c->pin[k] && hipHostMalloc((void**)&c->pin[k], CH, hipHostMallocDefault) != hipSuccess) { c->pin[k] = nullptr; staged = false; }
                if (staged && !c->pin_ev[k] && hipEventCreateWithFlags(&c->pin_ev[k], hipEventDisableTiming) != hipSuccess) { c->pin_ev[k] = nullptr; staged = false; }
        }
        if (!staged) {
                HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                return KA_OK;
        }
        const size_t n = (bytes + CH - 1) / CH;
        for (size_t k = 0; k <= n; k++) {
                if (k < n) {
                        const size_t len = std::min(CH, bytes - k * CH);
                        HIPCHK(hipMemcpyAsync(c->pin[k & 1], (const char*)src + k * CH, len, hipMemcpyDeviceToHost, c->stream));
                        HIPCHK(hipEventRecord(c->pin_ev[k & 1], c->stream));
                }
                if (k > 0) {                                          // chunk k-1 has landed: hand it to the caller while chunk k moves
                        const size_t j = k - 1, len = std::min(CH, bytes - j * CH);
                        HIPCHK(hipEventSynchronize(c->pin_ev[j & 1]));
                        memcpy((char*)dst + j * CH, c->pin[j & 1], len);
                }
        }
        return KA_OK;
}

// reset the device state so that a run is repeatable
int tree_reset(ka_ctx* c)
{
        const int numseq = c->numseq, nprof = 2 * numseq - 1;
        std::vector<int> node_len(nprof, 0);
        std::vector<long long> node_prof(nprof, -1);
        for (int i = 0; i < numseq; i++) { node_len[i] = c->lens[i]; node_prof[i] = c->leaf_prof_off[i]; }
        unsigned long long counters[8] = { (unsigned long long)c->leaf_prof_total, 0, 0, 0, 0, 0, 0, 0 };
        int zero = 0;
        HIPCHK(hipMemcpyAsync(c->d_node_len.p, node_len.data(), sizeof(int) * nprof, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_node_prof.p, node_prof.data(), sizeof(long long) * nprof, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(c->d_node_vote.p, 0xff, sizeof(long long) * nprof, c->stream));          // (-1: no carried vote table yet)
        HIPCHK(hipMemcpyAsync(c->d_counters.p, counters, sizeof(counters), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_error.p, &zero, sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(c->d_ctl.p, 0, (size_t)ka_ctl_bytes_host() * c->n_tasks, c->stream));
        HIPCHK(hipMemsetAsync(c->d_recs.p, 0, sizeof(ka_task_rec) * c->n_tasks, c->stream));
        HIPCHK(hipMemsetAsync(c->d_join.p, 0, sizeof(KaJoin) * c->n_tasks, c->stream));
        if (c->test_hooks & KA_DEBUG_POISON_ARENAS) {   // (tests: a run must not read what it has not written)
                HIPCHK(hipMemsetAsync(c->d_prof_arena.p, 0xff, c->d_prof_arena.n * sizeof(*c->d_prof_arena.p), c->stream));
                HIPCHK(hipMemsetAsync(c->d_scratch.p, 0xff, c->d_scratch.n * sizeof(*c->d_scratch.p), c->stream));
                HIPCHK(hipMemsetAsync(c->d_path_arena.p, 0xff, c->d_path_arena.n * sizeof(*c->d_path_arena.p), c->stream));
        }
        if (c->have_colof)                           // every leaf starts with residue p in column p
                HIPCHK(hipMemcpyAsync(c->d_colof.p, c->d_colof_init.p, sizeof(int) * c->colof_n, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));          // the staging vectors above are stack/heap temporaries
        c->state_valid = true;
        c->task_done.assign(c->n_tasks, 0);
        c->injected.clear();
        return KA_OK;
}

KaTreeDev tree_dev(ka_ctx* c)
{
        KaTreeDev D;
        D.codes = c->d_codes.p; D.seq_off = c->d_seq_off.p;
        D.node_len = c->d_node_len.p; D.node_prof = c->d_node_prof.p; D.node_vote = c->d_node_vote.p;
        D.prof_arena = c->d_prof_arena.p; D.counters = c->d_counters.p;
        D.prof_cap = c->prof_cap; D.scratch_cap = c->scratch_cap; D.path_cap = c->path_cap; D.dbg_cap = c->dbg_cap;
        D.scratch = c->d_scratch.p; D.path_arena = c->d_path_arena.p;
        D.dbg_arena = c->d_dbg_arena.p; D.dbg_off = c->d_dbg_off.p;
        D.tasks = c->d_tasks.p; D.recs = c->d_recs.p; D.subm = c->d_subm.p;
        D.ctl = (KaCtl*)c->d_ctl.p;
        D.join = c->d_join.p;
        D.gpo0 = c->scal[0]; D.gpe0 = c->scal[1]; D.tgpe0 = c->scal[2]; D.usw = c->scal[5];
        D.numseq = c->numseq; D.flags = c->flags; D.error = c->d_error.p;
        D.nres = c->nres;
        D.trace = c->h_trace;
        D.refine_mode = 0;
        D.refine_adaptive = 0;
        D.refine_trials = 3;
        D.prof_task = -1;
        D.wdfs = (c->env.no_wdfs ? 0 : 1) | (c->env.no_ls0 ? 0 : 2) | (c->env.no_inc ? 0 : 4) | (c->env.no_ldfs ? 0 : 8);   // measurements / tests
        D.prof_task = c->env.prof_task;                                 // measurements only (tools/levels_real.py)
        D.timing = (c->flags & KA_FLAG_TIMING) ? c->d_timing.p : nullptr;
        D.max_g = std::max(1, std::min(c->max_cluster, ka_max_g_host()));
        D.q1_mode = c->env.q1 >= 0 ? c->env.q1 : (c->nres > 5 ? 4 : 0);     // (nucleotides: five residues -- a one-row step is 0.85 of a two-row one: not worth twice the strips)
        D.ho_mode = c->env.ho >= 0 ? c->env.ho : 1;
        D.per_target = c->env.per;
        D.q_order = nullptr; D.q_n = 0; D.q_slots = 0; D.qw = c->env.qw; D.lw = c->env.lw; D.reuse = c->env.reuse; D.carry = c->env.carry;
        D.tp = c->env.tp; D.reserve = 0;
        D.overlap = c->overlap_plan ? 1 : 0;
        D.hw_mode = c->env.hw ? (1 | (c->env.hw_prio << 4)) : 0;
        D.lean4 = c->env.lean4;
        D.sub_mode = c->env.subtree;
        D.mw_mode = c->env.mw;
        D.merge_batch = c->env.merge;
        D.cons_K = c->cons_K; D.cons_maxlen = c->max_len;
        D.cons_paw = c->cons_K > 0 ? c->cons_weight / (float)c->cons_K : 0.0f;
        D.cons_maps = c->d_cons_maps.p; D.cons_map_off = c->d_cons_map_off.p;
        D.colof = c->d_colof.p; D.sip = c->d_sip.p; D.sip_off = c->d_sip_off.p;
        return D;
}

// Workgroup table of one launch: near the top of the tree there are fewer tasks than CUs, so a task
// gets a cluster of up to max_cluster workgroups (the kernel decides from the actual operand
// lengths how many of them it uses).  Workgroups of one cluster are spaced 8 blocks apart:
// block b runs on XCD b % 8 (observed, not contractual -- used for L2 locality only).
void build_blocks(const ka_ctx* c, const std::vector<int>& L, std::vector<int2>& tbl, int* lean_out)
{
        const int nt = (int)L.size();
        int lean = 1;                                    // launch kind: 0 = 8 waves, 1 = lean, 2 = half
        for (int t : L) if (c->descs[t].nsip_a != 1 || c->descs[t].nsip_b != 1) lean = 0;
        if (c->env.no_lean) lean = 0;
        if (!lean && nt > c->n_cus && !c->env.no_half) lean = 2;   // more tasks than CUs: two 4-wave workgroups per CU
        int G = 1;
        while (lean == 0 && G * 2 <= c->max_cluster && nt * G * 2 <= c->n_cus) G *= 2;
        const int groups = (nt + 7) / 8;
        tbl.assign((size_t)groups * 8 * G, make_int2(-1, 0));
        for (int j = 0; j < nt; j++)
                for (int m = 0; m < G; m++)
                        tbl[(size_t)(j % 8) + 8 * ((size_t)m + (size_t)G * (j / 8))] = make_int2(L[j], m | (G << 8));
        *lean_out = lean;
}

// KA_LAUNCH_EV: an event behind launch number c->n_launches of the run (measurements; ka_tree_launch_ms)
static int mark_launch(ka_ctx* c, hipStream_t s = nullptr)
{
        if (!c->env.launch_ev) return KA_OK;
        while ((int)c->launch_ev.size() < c->n_launches) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->launch_ev.push_back(e); }
        HIPCHK(hipEventRecord(c->launch_ev[c->n_launches - 1], s ? s : c->stream));
        return KA_OK;
}

// the chained launch: a guide-tree level and everything above it, tasks chained through their join points
static int launch_chain(ka_ctx* c, const KaTreeDev& D, bool ov, bool join_now = true)
{
        hipStream_t cs = ov ? c->s_chain : c->stream;
        KaTreeDev Dc = D;
        if (ov && c->env.overlap_help) {
                // Its workgroups help the queue when they arrive first (ka_task_entry) -- down to an EMPTY list (round 6).  Round 5 stopped
                // helping at the queue's last round (workgroup slots of the queued launch: "its own workgroups run that"), which assumes
                // those workgroups are resident.  They need not be: the chain may hold every CU (spare workgroups fill it up to n_cus; one
                // of its workgroups takes a CU's LDS), and the queued launch can start late -- seen deterministically when its kernel's code
                // object is loaded at its first launch: the chain took all CUs, did all but the last round, and everybody waited for 768
                // tasks nobody could run until the watchdog fired (2.4 s, then the shared plan).  Measured in round 5: where the helping
                // stops makes no difference in the normal order (KA_OVERLAP_HELP=2+n: stop at the last n tasks -- experiments).
                Dc.q_order = c->d_blocks.p + c->queue_off; Dc.q_n = c->queue_n; Dc.q_slots = (c->env.overlap_help >= 2) ? (c->env.overlap_help - 2) : 0;
        }
        ka_launch_task_level(&Dc, c->d_blocks.p + c->chain_blocks_off, (int)c->chain_blocks.size(), 0, 1, cs);
        c->n_launches++; if (mark_launch(c, cs)) return KA_FAIL;
        if (cs != c->stream) {
                HIPCHK(hipEventRecord(c->e_chain, c->s_chain));
                if (join_now) HIPCHK(hipStreamWaitEvent(c->stream, c->e_chain, 0));
        }
        return KA_OK;
}

// reset: start from the leaves (a whole-tree run); false: a planned subset on top of what the context already holds
int tree_launch(ka_ctx* c, bool reset)
{
        if (reset ? tree_reset(c) : (!c->state_valid && tree_reset(c))) return KA_FAIL;
        KaTreeDev D = tree_dev(c);
        D.refine_mode = c->refine_mode & 255;
        D.refine_adaptive = (c->refine_mode >> 8) & 1;
        D.refine_trials = (c->refine_mode >> 16) & 255 ? (c->refine_mode >> 16) & 255 : 3;
        c->partial = !reset;
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        c->n_launches = 0;
        // Overlapping launches (KA_OVERLAP, round 5): the chained launch goes out on a stream of its own (lowest priority) together with
        // the queued launch instead of behind it.  Its workgroups want a CU each: they become resident as the queue's workgroups leave,
        // and its tasks wait for the done flags of whatever the queued launch makes for them (KaTreeDev::overlap) -- the queue's last,
        // partly filled round of tasks and the first level of the chain share the GPU.  The queue never waits for the chain, so
        // whatever the dispatcher does first, both finish.
        const bool ov = c->overlap_plan && reset && !c->refine_mode && c->queue_first >= 0 && c->chain_level > c->queue_first;
        if (ov && !c->s_chain) {
                int lo = 0, hi = 0;
                HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));            // (lo: the numerically greatest = least urgent)
                HIPCHK(hipStreamCreateWithPriority(&c->s_chain, hipStreamNonBlocking, lo));
                HIPCHK(hipEventCreateWithFlags(&c->e_fork, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&c->e_chain, hipEventDisableTiming));
        }
        for (size_t L = 0; L < c->plan_levels.size(); L++) {
                const int n = (int)c->plan_levels[L].size();
                if (!n) continue;
                // (the scratch arena starts again with every launch -- the chained launch that overlaps the queue shares the queue's)
                if ((L || !reset) && !(ov && (int)L == c->chain_level)) HIPCHK(hipMemsetAsync(c->d_counters.p + 1, 0, sizeof(unsigned long long), c->stream));
                if (!reset && (int)L == c->queue_first) HIPCHK(hipMemsetAsync(c->d_counters.p + 4, 0, sizeof(unsigned long long), c->stream));   // (the queue's head)
                if (c->refine_mode) {
                        // refinement pass: one launch per tree level (see refine_blocks)
                        if (ka_cons_big(&D)) ka_unit9_launch(&D, c->d_refine_blocks.p + c->refine_off[L], c->refine_off[L + 1] - c->refine_off[L], c->stream);
                        else ka_unit4_launch(&D, c->d_refine_blocks.p + c->refine_off[L], c->refine_off[L + 1] - c->refine_off[L], D.cons_K > 0, c->stream);
                        c->n_launches++; if (mark_launch(c)) return KA_FAIL;
                        continue;
                }
                if ((int)L == c->queue_first) {
                        // levels queue_first .. chain_level-1: one launch, two workgroups per CU pulling from the ordered list
                        // (workgroups per CU: two of four waves; of narrower ones as many as the registers (eight waves) and the LDS (160 KB) hold)
                        // (more than fit is harmless: a workgroup that starts late finds the rest of the list, or nothing)
                        const int per_cu = ka_tp_ok(&D) ? 3 : (c->env.qw == 4 ? 2 : (c->env.qw == 2 ? 4 : 8));
                        const int nwg = std::min(c->queue_n, per_cu * c->n_cus);
                        if (ov) {
                                // (the chain's stream forks here: behind the leaf levels and the counter resets, beside the queue)
                                HIPCHK(hipEventRecord(c->e_fork, c->stream));
                                HIPCHK(hipStreamWaitEvent(c->s_chain, c->e_fork, 0));
                        }
                        const bool chain_first = ov && (c->test_hooks & KA_DEBUG_CHAIN_FIRST);
                        if (chain_first) {                               // (tests: see KA_DEBUG_CHAIN_FIRST)
                                if (launch_chain(c, D, ov, false)) return KA_FAIL;
                                // both launches wait for the same event and the queue's stream has the higher priority: enqueued back to back the
                                // queue would still be dispatched first.  Hold it back until the chain's workgroups have the CUs.
                                std::this_thread::sleep_for(std::chrono::milliseconds(20));
                        }
                        KaTreeDev Dq = D;
                        Dq.reserve = ov ? c->reserve_cus : 0;          // (CUs left to the head of the chain: only when the chain really goes out beside the queue)
                        if (ka_cons_big(&D)) ka_unit7_launch(&Dq, c->d_blocks.p + c->queue_off, nwg, c->queue_n, c->stream);
                        else if (ka_tp_ok(&D)) ka_unit10_launch(&Dq, c->d_blocks.p + c->queue_off, nwg, c->queue_n, c->stream);
                        else ka_unit2_launch(&Dq, c->d_blocks.p + c->queue_off, nwg, D.cons_K > 0, c->queue_n, c->stream);
                        c->n_launches++; if (mark_launch(c)) return KA_FAIL;
                        L = (size_t)c->chain_level - 1;
                        if (chain_first) { HIPCHK(hipStreamWaitEvent(c->stream, c->e_chain, 0)); break; }
                        continue;
                }
                if ((int)L == c->chain_level) {
                        // this level and everything above it: one launch, tasks chained through their join points
                        if (launch_chain(c, D, ov)) return KA_FAIL;
                        break;
                }
                ka_launch_task_level(&D, c->d_blocks.p + c->blocks_off[L], c->blocks_off[L + 1] - c->blocks_off[L], c->level_lean[L], 0, c->stream);
                c->n_launches++; if (mark_launch(c)) return KA_FAIL;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        for (auto& L : c->plan_levels) for (int t : L) c->task_done[t] = 1;
        return KA_OK;
}

extern "C" int ka_tree_run(ka_ctx* c)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        c->ran = false; c->synced = false;
        if (!c->plan_active.empty()) {                   // the last plan covered a subset (ka_tree_plan_tasks): plan the whole tree again
                c->plan_active.clear();
                if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
        }
        if (c->refine_mode) {                            // the plan on the device carries the refine marks of the last ka_tree_refine
                c->refine_mode = 0;
                for (auto& d : c->descs) d.refine = 0;
                if (upload_plan(c)) return KA_FAIL;
        }
        if (tree_launch(c)) return KA_FAIL;
        c->ran = true;
        return KA_OK;
}

// refine_alignment (aln_refine.c:199-325): a second pass over every edge of the tree with the flip trials of
// refine_edge; mode 1 = KALIGN_REFINE_ALL, 2 = KALIGN_REFINE_CONFIDENT (edges whose first-pass confidence is at or
// below the median), 3 = KALIGN_REFINE_INLINE (create_msa_tree_inline_refine, aln_run.c:448-790: three trials per edge
// in one pass), 4 = the first pass again with the depth-first engine (task confidences are then the reference's exact
// float sums).  conf_in: the first-pass confidence of every task (the reference reads task->confidence); only
// read for mode 2, NULL = computed here by a mode-4 pass.  The job keeps its tree, parameters and consistency table.
// Workgroup table of a refinement pass.  Within one trial the meetups of an edge are serial by construction (the flip
// counter walks them in recursion order), so an edge gets ONE workgroup per trial in flight: on a level that leaves CUs
// idle the flip trials of a refined edge run side by side on 2 or 4 workgroups (ka_task_body_refine), otherwise one
// workgroup runs them one after the other.
#define KA_REFINE_MAX_G 4
int refine_blocks(ka_ctx* c, int mode)
{
        std::vector<int2> tbl;
        c->refine_off.assign(1, 0);
        const int base_mode = mode & 255;
        const int trials3 = (mode >> 16) & 255 ? (mode >> 16) & 255 : 3;
        const int flips = base_mode == 3 ? trials3 - 1 : (base_mode == 4 ? 0 : 4);      // (adaptive budget: up to 7, shared by at most 4 members)
        bool starved = false;
        for (auto& L : c->levels) {
                int nref = 0;
                for (int t : L) nref += (flips > 0 && (base_mode != 2 || c->descs[t].refine)) ? 1 : 0;
                int G = 1;
                if (!c->shared_gpu && !c->env.refine_serial)
                        // (at most KA_REFINE_MAX_G members: the member report of ka_task_body_refine has that many slots in the
                        // task's control block -- n_trials beyond 9 would otherwise ask for 8 or more and overrun it)
                        while (G * 2 <= std::min(flips, KA_REFINE_MAX_G) && (long long)nref * G * 2 + ((long long)L.size() - nref) <= c->n_cus) G *= 2;
                for (int t : L) {
                        const int g = (flips > 0 && (base_mode != 2 || c->descs[t].refine)) ? G : 1;
                        // tests: one member of the first multi-workgroup edge never starts -- its member barrier starves, the
                        // watchdog reports it and ka_tree_sync re-plans with one workgroup per edge (the refine marks must survive)
                        const bool starve = (c->test_hooks & KA_DEBUG_STARVE_REFINE_MEMBER) && g > 1 && !starved;
                        if (starve) starved = true;
                        for (int m = 0; m < g; m++) tbl.push_back(starve && m == g - 1 ? make_int2(-1, 0) : make_int2(t, m | (g << 8)));
                }
                c->refine_off.push_back((int)tbl.size());
        }
        if (c->d_refine_blocks.alloc(tbl.size())) return fail("hipMalloc failed");
        HIPCHK(hipMemcpy(c->d_refine_blocks.p, tbl.data(), sizeof(int2) * tbl.size(), hipMemcpyHostToDevice));
        return KA_OK;
}

static int refine_launch(ka_ctx* c, int mode)
{
        if (!c->plan_active.empty()) { c->plan_active.clear(); if (plan_launches(c)) return KA_FAIL; }
        c->refine_mode = mode;
        if (refine_blocks(c, mode)) return KA_FAIL;
        if (upload_plan(c)) return KA_FAIL;
        c->ran = false; c->synced = false;
        if (tree_launch(c)) return KA_FAIL;
        c->ran = true;
        return KA_OK;
}

extern "C" int ka_tree_refine(ka_ctx* c, int mode_in, const float* conf_in)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        const int mode = mode_in & 255, adaptive = mode_in & KA_REFINE_ADAPTIVE, trials = mode_in & KA_REFINE_TRIALS(255);
        if (mode < 1 || mode > 4 || (mode_in & ~(255 | KA_REFINE_ADAPTIVE | KA_REFINE_TRIALS(255))))
                return fail("ka_tree_refine: mode must be 1 (all), 2 (confident), 3 (inline) or 4 (first pass, exact confidences), optionally | KA_REFINE_ADAPTIVE");
        if (trials && mode != 3) return fail("ka_tree_refine: KA_REFINE_TRIALS goes with mode 3 (create_msa_tree_inline_refine)");
        if (adaptive && mode != 1 && mode != 2) return fail("ka_tree_refine: KA_REFINE_ADAPTIVE goes with modes 1 and 2 (refine_edge)");
        if (c->n_tasks < 1) return fail("ka_tree_refine: no tasks");
        HIPCHK(hipSetDevice(c->device));
        if (c->ran && !c->synced && ka_tree_sync(c)) return KA_FAIL;
        if (!c->have_colof) {                            // the sum-of-pairs score reads every member's column
                if (setup_colof(c)) return KA_FAIL;
        }
        c->flags |= KA_FLAG_DEVICE_GAPS;
        std::vector<float> conf;
        if (mode == 2 && !conf_in) {
                // task->confidence of the first pass is a float sum in depth-first order (aln_controller.c:194-436): the
                // level-synchronous first pass adds the same margins in another order, so its value can differ in the last
                // bits -- and the median rule below compares them.  Run the first pass again depth first and read its sums.
                for (auto& d : c->descs) d.refine = 0;
                if (refine_launch(c, 4) || ka_tree_sync(c)) return KA_FAIL;
                std::vector<ka_task_rec> r(c->n_tasks);
                HIPCHK(hipMemcpy(r.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
                conf.resize(c->n_tasks);
                for (int t = 0; t < c->n_tasks; t++) conf[t] = r[t].confidence;
                conf_in = conf.data();
        }
        float thr = 0.0f;
        if (mode == 2) {                                 // compute_confidence_threshold (aln_refine.c:674-712): the median
                std::vector<float> v(conf_in, conf_in + c->n_tasks);
                std::sort(v.begin(), v.end());
                const int n = c->n_tasks;
                thr = (n % 2 == 0) ? (v[n / 2 - 1] + v[n / 2]) / 2.0F : v[n / 2];
        }
        for (int t = 0; t < c->n_tasks; t++) c->descs[t].refine = mode == 1 ? 1 : (mode == 2 && conf_in[t] <= thr ? 1 : 0);
        return refine_launch(c, mode | adaptive | trials);
}

extern "C" int ka_tree_sync(ka_ctx* c)
{
        if (!c || !c->ran) return fail("nothing running");
        HIPCHK(hipSetDevice(c->device));
        for (int attempt = 0; attempt < 24; attempt++) {
                int err = 0;
                HIPCHK(hipStreamSynchronize(c->stream));
                HIPCHK(hipMemcpy(&err, c->d_error.p, sizeof(int), hipMemcpyDeviceToHost));
                if (!err) {
                        HIPCHK(hipMemcpy(c->h_counters, c->d_counters.p, sizeof(c->h_counters), hipMemcpyDeviceToHost));
                        c->synced = true;
                        if (!c->shared_gpu) c->fallback_streak = 0;        // (a clean run of the fast plan)
                        return KA_OK;
                }
                if (err == 5) return fail("device watchdog: a strip pipeline stopped making progress");
                if (err == 6) {
                        // workgroups that wait for each other were not all resident: somebody else is using the GPU.
                        // Fall back to the plan that needs no co-residency (ka_ctx_set_shared) and run again.
                        if (c->shared_gpu || c->partial) return fail("device watchdog: a wait between workgroups never completed");
                        // (a 2 s stall is never silent: one line per fallback; the queue's head and the helped count say which wait it was)
                        c->shared_gpu = true; c->shared_by_fallback = true; c->fallback_runs++;
                        // the next job tries the fast plan again -- unless that one stalled too: a GPU that stays shared (or a runtime
                        // that keeps the launches apart) would otherwise cost every job its 2 s; then 4, 16, 64 jobs stay on the shared plan
                        c->fallback_streak++;
                        c->fallback_hold = c->fallback_streak >= 2 ? std::min(64, 1 << (2 * (c->fallback_streak - 1))) : 0;
                        {
                                unsigned long long hc[6] = {0, 0, 0, 0, 0, 0};
                                (void)hipMemcpy(hc, c->d_counters.p, sizeof(hc), hipMemcpyDeviceToHost);
                                fprintf(stderr, "[kalign_amd] a wait between workgroups expired (somebody else on the GPU?): queue head %llu of %d, helped %llu, launches %d; "
                                        "the run is repeated on the plan without such waits (fallback %d of this context, the next %d jobs stay on it)\n",
                                        hc[4], c->queue_n, hc[5], c->n_launches, c->fallback_runs, c->fallback_hold);
                        }
                        if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
                        if (c->refine_mode && refine_blocks(c, c->refine_mode)) return KA_FAIL;       // one workgroup per edge from here on
                        if (tree_launch(c)) return KA_FAIL;
                        continue;
                }
                if (c->partial) return fail("a device arena overflowed during a partial run (ka_tree_run_tasks does not re-run)");
                // an arena overflowed: grow it and run again (results are only trusted from a clean run)
                if (err == 1) { c->prof_cap *= 2; c->path_cap *= 2; c->d_prof_arena.release(); c->d_path_arena.release(); }
                else if (err == 2) { c->scratch_cap *= 2; c->d_scratch.release(); }
                else if (err == 3) { c->path_cap *= 2; c->d_path_arena.release(); }
                else { c->dbg_cap *= 2; c->d_dbg_arena.release(); }
                if (c->d_prof_arena.alloc((size_t)c->prof_cap) || c->d_path_arena.alloc((size_t)c->path_cap) ||
                    c->d_scratch.alloc((size_t)c->scratch_cap) || c->d_dbg_arena.alloc((size_t)std::max<long long>(c->dbg_cap, 1)))
                        return fail("hipMalloc failed while growing an arena");
                if (tree_launch(c)) return KA_FAIL;
        }
        return fail("device arenas kept overflowing");
}

extern "C" long long ka_tree_paths_size(ka_ctx* c)
{
        if (!c || !c->synced) return -1;
        return (long long)c->h_counters[2];
}

// make_seq + update_gaps (weave_alignment.c:41-112): fold one task's gap columns into the
// gaps[] arrays of every member sequence.
static void fold_gaps(int len, int* gis, const int* newgaps)
{
        int rel = 0;
        for (int i = 0; i <= len; i++) {
                int add = 0;
                for (int j = rel; j <= rel + gis[i]; j++) add += newgaps[j];
                rel += gis[i] + 1;
                gis[i] += add;
        }
}

// make_seq + update_gaps for every task in tree order (weave_alignment.c:41-112): host-only, needs
// only (a, b, c, path_off) of every task and the coded paths.
extern "C" int ka_weave_gaps(int numseq, const int* lens, int n_tasks, const ka_task_rec* recs, const int* paths, int* gaps_out)
{
        if (numseq < 1 || n_tasks < 0 || n_tasks > numseq - 1 || !lens || !recs || !paths || !gaps_out) return fail("ka_weave_gaps: bad arguments");
        const int nprof = 2 * numseq - 1;
        std::vector<int> goff(numseq);
        long long g = 0;
        for (int i = 0; i < numseq; i++) { goff[i] = (int)g; g += lens[i] + 1; }
        memset(gaps_out, 0, sizeof(int) * (size_t)g);
        std::vector<std::vector<int>> sip(nprof);
        for (int i = 0; i < numseq; i++) sip[i] = {i};
        std::vector<int> ga, gb;
        for (int t = 0; t < n_tasks; t++) {
                const ka_task_rec& r = recs[t];
                if (r.a < 0 || r.b < 0 || r.c < numseq || r.a >= nprof || r.b >= nprof || r.c >= nprof) return fail("ka_weave_gaps: bad task record");
                const int* p = paths + r.path_off;
                ga.assign(p[0] + 1, 0); gb.assign(p[0] + 1, 0);
                int posa = 0, posb = 0;
                for (int k = 1; p[k] != 3; k++) {
                        if (!p[k]) { posa++; posb++; }
                        else if (p[k] & 1) { ga[posa] += 1; posb++; }
                        else if (p[k] & 2) { gb[posb] += 1; posa++; }
                }
                for (int x : sip[r.a]) fold_gaps(lens[x], gaps_out + goff[x], ga.data());
                for (int x : sip[r.b]) fold_gaps(lens[x], gaps_out + goff[x], gb.data());
                sip[r.c].reserve(sip[r.a].size() + sip[r.b].size());
                sip[r.c].insert(sip[r.c].end(), sip[r.a].begin(), sip[r.a].end());
                sip[r.c].insert(sip[r.c].end(), sip[r.b].begin(), sip[r.b].end());
                std::vector<int>().swap(sip[r.a]);
                std::vector<int>().swap(sip[r.b]);
        }
        return KA_OK;
}

// alignment length of the tree each sequence belongs to (a sequence in no task aligns to itself); needs h_recs
static void tree_alnlens(ka_ctx* c, std::vector<int>& alen)
{
        alen.assign(c->lens.begin(), c->lens.end());
        for (int t = 0; t < c->n_tasks; t++) {
                if (!c->descs[t].is_root) continue;
                long long lo, hi;
                node_members(c, c->descs[t].c, &lo, &hi);
                for (long long k = lo; k < hi; k++) alen[c->sip_flat[k]] = c->h_recs[t].plen;
        }
}

extern "C" int ka_tree_download(ka_ctx* c, ka_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out)
{
        if (!c || !c->ran) return fail("nothing to download");
        if (!c->synced && ka_tree_sync(c)) return KA_FAIL;
        HIPCHK(hipSetDevice(c->device));
        const long long used = (long long)c->h_counters[2];
        if (used > paths_cap) { g_err = "paths_out too small"; return KA_ERR_PATHS_CAP; }
        c->h_recs.resize(c->n_tasks);
        HIPCHK(hipMemcpy(c->h_recs.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
        std::vector<int> arena((size_t)used);
        if (copy_to_host(c, arena.data(), c->d_path_arena.p, sizeof(int) * (size_t)used)) return KA_FAIL;
        // repack the paths in task order (arena order depends on workgroup scheduling)
        long long o = 0;
        double cells = 0.0;
        for (int t = 0; t < c->n_tasks; t++) {
                ka_task_rec& r = c->h_recs[t];
                const int n = r.plen + 2;
                memcpy(paths_out + o, arena.data() + r.path_off, sizeof(int) * n);
                r.path_off = (int)o;
                o += n;
                cells += (double)r.len_a * (double)r.len_b;
        }
        c->cells = cells;
        if (c->flags & KA_FLAG_DEBUG_ROWS) {
                // FNV-1a of the top-level rows, to compare with the reference harness
                std::vector<long long> dbg_off(c->n_tasks);
                HIPCHK(hipMemcpy(dbg_off.data(), c->d_dbg_off.p, sizeof(long long) * c->n_tasks, hipMemcpyDeviceToHost));
                std::vector<float> rows;
                for (int t = 0; t < c->n_tasks; t++) {
                        ka_task_rec& r = c->h_recs[t];
                        const int lb = r.swapped ? r.len_a : r.len_b;
                        const size_t n = 3 * (size_t)(lb + 1);
                        if (dbg_off[t] < 0) continue;
                        rows.resize(2 * n);
                        HIPCHK(hipMemcpy(rows.data(), c->d_dbg_arena.p + dbg_off[t], sizeof(float) * 2 * n, hipMemcpyDeviceToHost));
                        auto fnv = [](const void* p, size_t bytes) {
                                const unsigned char* b = (const unsigned char*)p;
                                uint64_t h = 1469598103934665603ULL;
                                for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ULL; }
                                return h;
                        };
                        if (const char* dump = getenv("KA_DUMP_ROWS")) {   // debugging aid: the rows of every task appended to a file (task, n, 2n floats)
                                if (FILE* fh = fopen(dump, "ab")) { const long long hd[2] = { t, (long long)n }; fwrite(hd, sizeof(hd), 1, fh); fwrite(rows.data(), sizeof(float), 2 * n, fh); fclose(fh); }
                        }
                        r.fhash = fnv(rows.data(), sizeof(float) * n);
                        r.bhash = fnv(rows.data() + n, sizeof(float) * n);
                }
        }
        if (recs) memcpy(recs, c->h_recs.data(), sizeof(ka_task_rec) * c->n_tasks);

        if (gaps_out && (c->flags & KA_FLAG_DEVICE_GAPS) && c->have_colof && !c->partial) {
                // the kernels kept every residue's column (make_seq / update_gaps in the device's form): the gap
                // arrays are its first differences, O(sum of lengths) instead of O(N L log N) folding on the host
                std::vector<int> col(c->colof_n);
                if (copy_to_host(c, col.data(), c->d_colof.p, sizeof(int) * c->colof_n)) return KA_FAIL;
                std::vector<int> alen;
                tree_alnlens(c, alen);
                long long g = 0;
                for (int i = 0; i < c->numseq; i++) {
                        const int* cc = col.data() + c->off[i];
                        const int len = c->lens[i];
                        const int alnlen = alen[i];
                        gaps_out[g] = cc[0];
                        for (int p = 1; p < len; p++) gaps_out[g + p] = cc[p] - cc[p - 1] - 1;
                        gaps_out[g + len] = alnlen - 1 - cc[len - 1];
                        g += len + 1;
                }
                return KA_OK;
        }
        if (gaps_out && ka_weave_gaps(c->numseq, c->lens.data(), c->n_tasks, c->h_recs.data(), paths_out, gaps_out)) return KA_FAIL;
        return KA_OK;
}


// After a run made ELSEWHERE -- the sharded tree of ka_dist_* / ka_multi_*: records and coded paths gathered, the gap arrays
// woven on the host -- this context, which holds the same uploaded job, becomes the holder of the finished alignment: the
// records go to HBM, every residue's column follows from gaps[] (make_linear_sequence, msa_op.c:578-598: column of residue p =
// p + gaps[0] + .. + gaps[p]).  ka_tree_aligned_rows, ka_aln_guide_tree and ka_tree_refine then carry on here as after ka_tree_run.
extern "C" int ka_tree_adopt_alignment(ka_ctx* c, const ka_task_rec* recs, const int* gaps)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (!recs || !gaps) return fail("ka_tree_adopt_alignment: null argument");
        HIPCHK(hipSetDevice(c->device));
        if (c->ran && !c->synced) HIPCHK(hipStreamSynchronize(c->stream));
        if (!c->have_colof && setup_colof(c)) return KA_FAIL;
        std::vector<int> col(c->colof_n, 0);
        long long g = 0;
        for (int i = 0; i < c->numseq; i++) {
                int at = 0;
                for (int p = 0; p < c->lens[i]; p++) { at += gaps[g + p]; col[(size_t)c->off[i] + p] = at + p; }
                g += c->lens[i] + 1;
        }
        for (int t = 0; t < c->n_tasks; t++)
                if (recs[t].c != c->abc[3 * t + 2]) return fail("ka_tree_adopt_alignment: the records are not this job's (task order)");
        HIPCHK(hipMemcpy(c->d_colof.p, col.data(), sizeof(int) * col.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_recs.p, recs, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyHostToDevice));
        c->flags |= KA_FLAG_DEVICE_GAPS;
        c->h_counters[2] = 0;                                         // (no coded paths of its own)
        c->ran = true; c->synced = true; c->partial = false; c->state_valid = false;
        c->rows_n = 0;
        return KA_OK;
}

// ---- finalise_alignment (msa_op.c:546-598): the aligned rows, built on the device from the residue->column tables ----
// alignment length per sequence of the finished run (checks included)
static int rows_prepare(ka_ctx* c, const uint8_t* letters, std::vector<int>& alen, int* widest)
{
        if (!c || !c->have_job || !c->ran) return fail("no finished run");
        if (!letters) return fail("null argument");
        if (!(c->flags & KA_FLAG_DEVICE_GAPS) || !c->have_colof) return fail("the job was uploaded without KA_FLAG_DEVICE_GAPS");
        if (c->partial) return fail("aligned rows need a complete run (ka_tree_run), not a partial one");
        HIPCHK(hipSetDevice(c->device));
        if (!c->synced && ka_tree_sync(c)) return KA_FAIL;
        // the records hold the alignment length of every tree
        c->h_recs.resize(c->n_tasks);
        HIPCHK(hipMemcpy(c->h_recs.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
        tree_alnlens(c, alen);
        *widest = 0;
        for (int i = 0; i < c->numseq; i++) *widest = std::max(*widest, alen[i]);
        return KA_OK;
}

// the rows in HBM (c->d_rows, row_stride apart); they stay there for ka_aln_guide_tree
static int rows_build(ka_ctx* c, const uint8_t* letters, uint8_t gap_char, const std::vector<int>& alen, int widest, long long row_stride)
{
        const size_t bytes = (size_t)c->numseq * (size_t)row_stride;
        if (c->d_letters.alloc(c->h_codes.size()) || c->d_alnlen.alloc(c->numseq) || c->d_rows.alloc(bytes)) return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->d_letters.p, letters, c->h_codes.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_alnlen.p, alen.data(), sizeof(int) * c->numseq, hipMemcpyHostToDevice, c->stream));
        ka_launch_rows(c->d_letters.p, c->d_seq_off.p, c->d_node_len.p, c->d_colof.p, c->d_alnlen.p, c->numseq, gap_char,
                       c->d_rows.p, row_stride, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));                      // `alen` and `letters` are the caller's
        // (one alignment only: a forest has no common row length)
        c->rows_n = (c->n_tasks == c->numseq - 1) ? c->numseq : 0;
        c->rows_stride = row_stride; c->rows_alnlen = widest; c->rows_gap = gap_char;
        return KA_OK;
}

extern "C" int ka_tree_aligned_rows(ka_ctx* c, const uint8_t* letters, uint8_t gap_char, uint8_t* rows_out,
                                    long long row_stride, int* alnlen_out)
{
        if (!rows_out && !alnlen_out) return fail("null argument");
        std::vector<int> alen;
        int widest = 0;
        if (rows_prepare(c, letters, alen, &widest)) return KA_FAIL;
        if (!rows_out) { memcpy(alnlen_out, alen.data(), sizeof(int) * c->numseq); return KA_OK; }       // size query
        if (row_stride < (long long)widest + 1) return fail("row_stride is smaller than the longest alignment + terminator");
        if (rows_build(c, letters, gap_char, alen, widest, row_stride)) return KA_FAIL;
        if (copy_to_host(c, rows_out, c->d_rows.p, (size_t)c->numseq * (size_t)row_stride)) return KA_FAIL;
        if (alnlen_out) memcpy(alnlen_out, alen.data(), sizeof(int) * c->numseq);
        return KA_OK;
}

// ---- realignment (kalign_run_realign, aln_wrap.c:449-495): compute_aln_pairwise_dist + build_tree_from_pairwise ----
extern "C" int ka_aln_guide_tree(ka_ctx* c, int numseq, const uint8_t* rows, long long row_stride, int alnlen, uint8_t gap_char,
                                 int* tasks_abc, float* seq_distances, float* dm_out)
{
        if (!c) return fail("null ctx");
        if (!tasks_abc) return fail("null argument");
        HIPCHK(hipSetDevice(c->device));
        const uint8_t* d_rows = nullptr;
        if (rows) {
                if (numseq < 2 || alnlen < 1 || row_stride < alnlen) return fail("ka_aln_guide_tree: bad arguments");
                const size_t bytes = (size_t)numseq * (size_t)row_stride;
                if (c->d_rows.alloc(bytes)) return fail("hipMalloc failed");
                HIPCHK(hipMemcpyAsync(c->d_rows.p, rows, bytes, hipMemcpyHostToDevice, c->stream));
                c->rows_n = 0;                                        // no longer the rows of the last run
        } else {
                if (c->rows_n < 2) return fail("ka_aln_guide_tree: no rows on the device (call ka_tree_aligned_rows on a finished single-tree run first)");
                if (numseq != c->rows_n) return fail("ka_aln_guide_tree: numseq does not match the rows on the device");
                row_stride = c->rows_stride; alnlen = c->rows_alnlen; gap_char = c->rows_gap;
        }
        d_rows = c->d_rows.p;
        if (numseq > 46340) return fail("ka_aln_guide_tree: more than 46340 sequences (pair indices are 32-bit)");
        const size_t nn = (size_t)numseq * (size_t)numseq;
        if (c->d_adm.alloc(nn) || c->d_amean.alloc(numseq) || c->d_uactive.alloc(numseq) || c->d_ucand.alloc(2 * (size_t)numseq) ||
            c->d_umerges.alloc(numseq))
                return fail("hipMalloc failed");
        ka_launch_aln_dist(d_rows, row_stride, alnlen, numseq, gap_char, c->d_adm.p, c->d_amean.p, c->stream);
        HIPCHK(hipGetLastError());
        if (dm_out) HIPCHK(hipMemcpyAsync(dm_out, c->d_adm.p, sizeof(float) * nn, hipMemcpyDeviceToHost, c->stream));
        if (seq_distances) HIPCHK(hipMemcpyAsync(seq_distances, c->d_amean.p, sizeof(float) * numseq, hipMemcpyDeviceToHost, c->stream));
        std::vector<int> ones(numseq, 1);
        HIPCHK(hipMemcpyAsync(c->d_uactive.p, ones.data(), sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        ka_launch_upgma(c->d_adm.p, c->d_uactive.p, c->d_ucand.p, c->d_umerges.p, numseq, c->env.upgma_launches ? 1 : 0, c->stream);
        HIPCHK(hipGetLastError());
        std::vector<int> merges(2 * (size_t)numseq);
        HIPCHK(hipMemcpyAsync(merges.data(), c->d_umerges.p, sizeof(int2) * (numseq - 1), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return ka_tasks_from_merges(numseq, merges.data(), tasks_abc);
}

// ---- kalign_run_seeded / kalign_run_realign between "sequences encoded" and "rows finalised" (aln_wrap.c:144-251,361-527)
//      as one call: the composition of the entry points above, with the intermediate rows of realignment passes
//      never leaving HBM ----
// refine_mode: 0 none; 1 / 2 (| KA_REFINE_ADAPTIVE) refine_alignment after the last alignment (aln_wrap.c:229-232, :506-509);
// 3 KALIGN_REFINE_INLINE: every alignment is create_msa_tree_inline_refine instead of create_msa_tree (:222-226, :498-502)
static int run_encoded(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                       const int* off, const int* lens, const float* subm, const float* scal,
                       int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads, int refine_mode,
                       uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out);

extern "C" int ka_run_encoded(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                              const int* off, const int* lens, const float* subm, const float* scal,
                              int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads,
                              uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out)
{
        return run_encoded(c, numseq, tree_codes, codes, letters, off, lens, subm, scal, n_anchors, weight, realign_iterations, dm_scale,
                           n_threads, 0, gap_char, rows_out, row_stride, alnlen_out);
}

extern "C" int ka_run_encoded_refine(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                                     const int* off, const int* lens, const float* subm, const float* scal,
                                     int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads,
                                     int refine_mode, uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out)
{
        const int base = refine_mode & 255;
        if (refine_mode < 0 || base > 3 || (refine_mode & ~(255 | KA_REFINE_ADAPTIVE)) || ((refine_mode & KA_REFINE_ADAPTIVE) && base != 1 && base != 2))
                return fail("ka_run_encoded_refine: refine_mode must be 0, 1, 2 (optionally | KA_REFINE_ADAPTIVE) or 3");
        return run_encoded(c, numseq, tree_codes, codes, letters, off, lens, subm, scal, n_anchors, weight, realign_iterations, dm_scale,
                           n_threads, refine_mode, gap_char, rows_out, row_stride, alnlen_out);
}

static int run_encoded(ka_ctx* c, int numseq, const uint8_t* tree_codes, const uint8_t* codes, const uint8_t* letters,
                       const int* off, const int* lens, const float* subm, const float* scal,
                       int n_anchors, float weight, int realign_iterations, const float* dm_scale, int n_threads, int refine_mode,
                       uint8_t gap_char, uint8_t* rows_out, long long row_stride, int* alnlen_out)
{
        const bool inline_refine = (refine_mode & 255) == 3;
        auto align = [&]() -> int {
                if (inline_refine ? ka_tree_refine(c, 3, nullptr) : ka_tree_run(c)) return KA_FAIL;
                return ka_tree_sync(c);
        };
        if (!c) return fail("null ctx");
        if (numseq < 2 || !tree_codes || !codes || !letters || !off || !lens || !subm || !scal || (!rows_out && !alnlen_out))
                return fail("ka_run_encoded: bad arguments");
        std::vector<int> tasks(3 * (size_t)(numseq - 1));
        std::vector<float> sd(numseq);
        if (ka_guide_tree(c, numseq, tree_codes, off, lens, n_threads, dm_scale, tasks.data(), sd.data())) return KA_FAIL;
        if (ka_tree_upload(c, numseq, codes, off, lens, sd.data(), numseq - 1, tasks.data(), subm, scal, KA_FLAG_DEVICE_GAPS)) return KA_FAIL;
        if (n_anchors > 0 && ka_tree_build_consistency(c, n_anchors, weight)) return KA_FAIL;
        if (align()) return KA_FAIL;
        std::vector<int> alen;
        int widest = 0;
        for (int it = 0; it < realign_iterations; it++) {
                if (rows_prepare(c, letters, alen, &widest) || rows_build(c, letters, gap_char, alen, widest, (long long)widest + 1)) return KA_FAIL;
                if (ka_aln_guide_tree(c, numseq, nullptr, 0, 0, 0, tasks.data(), sd.data(), nullptr)) return KA_FAIL;
                if (ka_tree_upload(c, numseq, codes, off, lens, sd.data(), numseq - 1, tasks.data(), subm, scal,
                                   KA_FLAG_DEVICE_GAPS | KA_FLAG_KEEP_CONSISTENCY)) return KA_FAIL;
                if (align()) return KA_FAIL;
        }
        if ((refine_mode & 255) == 1 || (refine_mode & 255) == 2) {
                if (ka_tree_refine(c, refine_mode, nullptr) || ka_tree_sync(c)) return KA_FAIL;
        }
        if (rows_prepare(c, letters, alen, &widest)) return KA_FAIL;
        if (alnlen_out) memcpy(alnlen_out, alen.data(), sizeof(int) * numseq);
        if (!rows_out) return KA_OK;                                  // the alignment stays on the device: ka_tree_aligned_rows fetches it
        if (row_stride < (long long)widest + 1) { g_err = "ka_run_encoded: row_stride is smaller than the alignment + terminator (alnlen_out says how long; ka_tree_aligned_rows fetches the rows)"; return KA_ERR_ROWS_STRIDE; }
        if (rows_build(c, letters, gap_char, alen, widest, row_stride)) return KA_FAIL;
        return copy_to_host(c, rows_out, c->d_rows.p, (size_t)numseq * (size_t)row_stride);
}


// ---- partial runs: the pieces single-tree multi-GPU sharding is made of ----
// Run the listed tasks only, level by level.  Their children must already be available on this context:
// leaves, tasks run earlier (ka_tree_run_tasks does not reset the device state), or injected profiles.
extern "C" int ka_tree_run_tasks(ka_ctx* c, const int* task_ids, int n)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        c->synced = false;
        std::vector<std::vector<int>> by_level(c->levels.size());
        std::vector<char> have(2 * c->numseq - 1, 0);
        for (int i = 0; i < c->numseq; i++) have[i] = 1;
        for (int t = 0; t < c->n_tasks; t++) if (c->task_done[t]) have[c->descs[t].c] = 1;
        for (int node : c->injected) have[node] = 1;
        for (int i = 0; i < n; i++) {
                const int t = task_ids[i];
                if (t < 0 || t >= c->n_tasks) return fail("task id out of range");
                if (c->task_done[t]) return fail("task already run");
                by_level[c->task_level[t]].push_back(t);
        }
        for (auto& L : by_level)                                   // dependency check in level order
                for (int t : L) {
                        if (!have[c->descs[t].a] || !have[c->descs[t].b]) return fail("a task's operand is neither computed nor injected on this context");
                        have[c->descs[t].c] = 1;
                }
        const KaTreeDev D = tree_dev(c);
        c->partial = true;
        HIPCHK(hipEventRecord(c->ev0, c->stream));
        c->n_launches = 0;
        for (auto& L : by_level) {
                if (L.empty()) continue;
                std::vector<int2> tbl;
                int lean = 0;
                build_blocks(c, L, tbl, &lean);
                if (c->d_blocks_tmp.alloc(tbl.size())) return fail("hipMalloc failed");
                // the table is consumed by the launch below; stream order makes the reuse of the buffer safe
                HIPCHK(hipMemsetAsync(c->d_counters.p + 1, 0, sizeof(unsigned long long), c->stream));
                HIPCHK(hipMemcpyAsync(c->d_blocks_tmp.p, tbl.data(), sizeof(int2) * tbl.size(), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                ka_launch_task_level(&D, c->d_blocks_tmp.p, (int)tbl.size(), lean, 0, c->stream);
                HIPCHK(hipStreamSynchronize(c->stream));
                c->n_launches++;
                for (int t : L) c->task_done[t] = 1;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        c->ran = true;
        return KA_OK;
}

// The listed tasks as ONE planned run: queued and chained launches where they apply, as for a whole tree -- what a rank
// of a sharded tree does with its subtrees.  The set must be closed under descendants among the tasks not run yet (a
// task's internal children are in the set, already run, or injected).  task_ids == NULL: plan the whole tree again.
extern "C" int ka_tree_plan_tasks(ka_ctx* c, const int* task_ids, int n)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        if (c->ran && !c->synced && ka_tree_sync(c)) return KA_FAIL;
        if (!task_ids) c->plan_active.clear();
        else {
                c->plan_active.assign(c->n_tasks, 0);
                for (int i = 0; i < n; i++) {
                        if (task_ids[i] < 0 || task_ids[i] >= c->n_tasks) return fail("task id out of range");
                        c->plan_active[task_ids[i]] = 1;
                }
        }
        for (auto& d : c->descs) d.refine = 0;
        c->refine_mode = 0;
        return (plan_launches(c) || upload_plan(c)) ? KA_FAIL : KA_OK;
}

// Run the planned tasks on top of what the context holds (leaves, tasks run earlier, injected profiles); no reset.
extern "C" int ka_tree_run_planned(ka_ctx* c)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (c->plan_active.empty()) return fail("ka_tree_run_planned: no task subset planned (ka_tree_plan_tasks)");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        std::vector<char> have(2 * c->numseq - 1, 0);
        for (int i = 0; i < c->numseq; i++) have[i] = 1;
        for (int t = 0; t < c->n_tasks; t++) if (c->task_done[t]) have[c->descs[t].c] = 1;
        for (int node : c->injected) have[node] = 1;
        for (auto& L : c->plan_levels)
                for (int t : L) {
                        if (c->task_done[t]) return fail("task already run");
                        if (!have[c->descs[t].a] || !have[c->descs[t].b]) return fail("a task's operand is neither computed, planned nor injected on this context");
                        have[c->descs[t].c] = 1;
                }
        c->ran = false; c->synced = false;
        if (tree_launch(c, false)) return KA_FAIL;
        c->ran = true;
        return KA_OK;
}

// Forget every computed / injected node: the next ka_tree_run_tasks starts from the leaves again.
extern "C" int ka_tree_reset(ka_ctx* c)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        HIPCHK(hipSetDevice(c->device));
        c->ran = false; c->synced = false;
        return tree_reset(c);
}

extern "C" int ka_tree_node_len(ka_ctx* c, int node)
{
        if (!c || !c->have_job || !c->state_valid) return -1;
        if (node < 0 || node >= 2 * c->numseq - 1) return -1;
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
        int len = 0;
        if (hipMemcpy(&len, c->d_node_len.p + node, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return len;
}

// Make the merged profile of an internal node available on this context without running its task
// (it was computed on another GPU): (plen+2)*64 floats as ka_tree_get_profile returns them.
extern "C" int ka_tree_set_profile(ka_ctx* c, int node, const float* prof, int plen)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (node < c->numseq || node >= 2 * c->numseq - 1 || plen < 1 || !prof) return fail("bad node / profile");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        unsigned long long top = 0;
        HIPCHK(hipMemcpy(&top, c->d_counters.p, sizeof(top), hipMemcpyDeviceToHost));
        const unsigned long long need = (unsigned long long)(plen + 2) * KA_REC;
        if ((long long)(top + need) > c->prof_cap) return fail("profile arena too small for the injected profile");
        const long long po = (long long)top;
        top += need;
        HIPCHK(hipMemcpy(c->d_prof_arena.p + po, prof, sizeof(float) * (size_t)need, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_counters.p, &top, sizeof(top), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_len.p + node, &plen, sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_prof.p + node, &po, sizeof(long long), hipMemcpyHostToDevice));
        { const long long none = -1; HIPCHK(hipMemcpy(c->d_node_vote.p + node, &none, sizeof(long long), hipMemcpyHostToDevice)); }   // (its votes are counted from its members)
        c->injected.push_back(node);
        return KA_OK;
}

// Device-to-device hand-over of a subtree root's profile between the GPUs of a sharded tree: the source exposes
// where the profile lies in its arena, the destination reserves arena space for it; the caller moves the bytes HBM to
// HBM (RCCL send / recv over xGMI, hipMemcpyPeer) -- no host bounce.
extern "C" int ka_tree_profile_dev(ka_ctx* c, int node, void** dev_ptr, int* plen_out)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (node < 0 || node >= 2 * c->numseq - 1 || !dev_ptr || !plen_out) return fail("bad node");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        int len = 0;
        long long po = -1;
        HIPCHK(hipMemcpy(&len, c->d_node_len.p + node, sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&po, c->d_node_prof.p + node, sizeof(long long), hipMemcpyDeviceToHost));
        if (po < 0) return fail("node has no profile (root, or not computed)");
        *dev_ptr = c->d_prof_arena.p + po;
        *plen_out = len;
        return KA_OK;
}

extern "C" int ka_tree_reserve_profile_dev(ka_ctx* c, int node, int plen, void** dev_ptr)
{
        if (!c || !c->have_job) return fail("no uploaded job");
        if (node < c->numseq || node >= 2 * c->numseq - 1 || plen < 1 || !dev_ptr) return fail("bad node / profile");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        unsigned long long top = 0;
        HIPCHK(hipMemcpy(&top, c->d_counters.p, sizeof(top), hipMemcpyDeviceToHost));
        const unsigned long long need = (unsigned long long)(plen + 2) * KA_REC;
        if ((long long)(top + need) > c->prof_cap) return fail("profile arena too small for the incoming profile");
        const long long po = (long long)top;
        top += need;
        HIPCHK(hipMemcpy(c->d_counters.p, &top, sizeof(top), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_len.p + node, &plen, sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_node_prof.p + node, &po, sizeof(long long), hipMemcpyHostToDevice));
        { const long long none = -1; HIPCHK(hipMemcpy(c->d_node_vote.p + node, &none, sizeof(long long), hipMemcpyHostToDevice)); }   // (its votes are counted from its members)
        c->injected.push_back(node);
        *dev_ptr = c->d_prof_arena.p + po;
        return KA_OK;
}

// Consistency state of a node for partial runs: the residue -> column table of its member sequences, concatenated
// in the node's member order (sum of their lengths ints).  Moves with the profile when a node changes GPUs.
void node_members(const ka_ctx* c, int node, long long* lo, long long* hi)
{
        // sip_flat holds the leaves first (one entry each), then every internal node in task order
        *lo = c->sip_off[node];
        if (node < c->numseq) { *hi = *lo + 1; return; }
        long long best = (long long)c->sip_flat.size();
        for (size_t k = 0; k < c->sip_off.size(); k++) if (c->sip_off[k] > *lo && c->sip_off[k] < best) best = c->sip_off[k];
        *hi = best;
}

extern "C" long long ka_tree_node_cols_size(ka_ctx* c, int node)
{
        if (!c || !c->have_job || node < 0 || node >= 2 * c->numseq - 1) return -1;
        long long lo, hi, n = 0;
        node_members(c, node, &lo, &hi);
        for (long long k = lo; k < hi; k++) n += c->lens[c->sip_flat[k]];
        return n;
}

extern "C" int ka_tree_get_node_cols(ka_ctx* c, int node, int* out)
{
        if (!c || !c->have_job || c->cons_K <= 0) return fail("no consistency table on this context");
        if (node < 0 || node >= 2 * c->numseq - 1) return fail("bad node");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        long long lo, hi, o = 0;
        node_members(c, node, &lo, &hi);
        for (long long k = lo; k < hi; k++) {
                const int si = c->sip_flat[k];
                HIPCHK(hipMemcpy(out + o, c->d_colof.p + c->off[si], sizeof(int) * c->lens[si], hipMemcpyDeviceToHost));
                o += c->lens[si];
        }
        return KA_OK;
}

extern "C" int ka_tree_set_node_cols(ka_ctx* c, int node, const int* cols)
{
        if (!c || !c->have_job || c->cons_K <= 0) return fail("no consistency table on this context");
        if (node < 0 || node >= 2 * c->numseq - 1) return fail("bad node");
        HIPCHK(hipSetDevice(c->device));
        if (!c->state_valid && tree_reset(c)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        long long lo, hi, o = 0;
        node_members(c, node, &lo, &hi);
        for (long long k = lo; k < hi; k++) {
                const int si = c->sip_flat[k];
                HIPCHK(hipMemcpy(c->d_colof.p + c->off[si], cols + o, sizeof(int) * c->lens[si], hipMemcpyHostToDevice));
                o += c->lens[si];
        }
        return KA_OK;
}

// Records and coded paths of the listed tasks (they must have been run on this context), paths packed
// in the order of the list; *used receives the number of ints written.
extern "C" int ka_tree_download_tasks(ka_ctx* c, const int* task_ids, int n, ka_task_rec* recs, int* paths_out, long long paths_cap, long long* used_out)
{
        if (!c || !c->ran) return fail("nothing to download");
        if (!c->synced && ka_tree_sync(c)) return KA_FAIL;
        HIPCHK(hipSetDevice(c->device));
        const long long used = (long long)c->h_counters[2];
        std::vector<ka_task_rec> all(c->n_tasks);
        HIPCHK(hipMemcpy(all.data(), c->d_recs.p, sizeof(ka_task_rec) * c->n_tasks, hipMemcpyDeviceToHost));
        std::vector<int> arena((size_t)std::max<long long>(used, 1));
        if (copy_to_host(c, arena.data(), c->d_path_arena.p, sizeof(int) * (size_t)used)) return KA_FAIL;
        long long o = 0;
        for (int i = 0; i < n; i++) {
                const int t = task_ids[i];
                if (t < 0 || t >= c->n_tasks || !c->task_done[t]) return fail("task was not run on this context");
                ka_task_rec r = all[t];
                const int cnt = r.plen + 2;
                if (o + cnt > paths_cap) { g_err = "paths_out too small"; return KA_ERR_PATHS_CAP; }
                memcpy(paths_out + o, arena.data() + r.path_off, sizeof(int) * cnt);
                r.path_off = (int)o;
                o += cnt;
                recs[i] = r;
        }
        if (used_out) *used_out = o;
        return KA_OK;
}

extern "C" int ka_tree_get_profile(ka_ctx* c, int node, float* out, long long cap_floats)
{
        if (!c || !c->synced) return fail("run + sync first");
        HIPCHK(hipSetDevice(c->device));
        const int nprof = 2 * c->numseq - 1;
        if (node < 0 || node >= nprof) return fail("bad node");
        int len = 0;
        long long po = -1;
        HIPCHK(hipMemcpy(&len, c->d_node_len.p + node, sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&po, c->d_node_prof.p + node, sizeof(long long), hipMemcpyDeviceToHost));
        if (po < 0) return fail("node has no profile (root, or not computed)");
        const long long n = (long long)(len + 2) * KA_REC;
        if (n > cap_floats) return fail("profile buffer too small");
        HIPCHK(hipMemcpy(out, c->d_prof_arena.p + po, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
        return KA_OK;
}

extern "C" int ka_tree_get_timing(ka_ctx* c, long long* out)
{
        if (!c || !c->synced) return fail("run + sync first");
        if (!(c->flags & KA_FLAG_TIMING)) return fail("KA_FLAG_TIMING was not set");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemcpy(out, c->d_timing.p, sizeof(long long) * (8 * c->n_tasks + 48 + 512), hipMemcpyDeviceToHost));
        return KA_OK;
}

extern "C" float ka_pairwise_kernel_ms(ka_ctx* c) { return c ? c->pair_ms : 0.0f; }

extern "C" int ka_debug_trace(ka_ctx* c, int* out64)
{
        if (!c || !c->h_trace) return fail("KA_TRACE was not set when the context was created");
        memcpy(out64, c->h_trace, 64 * sizeof(int));
        return KA_OK;
}

extern "C" double ka_tree_cells(ka_ctx* c) { return c ? c->cells : 0.0; }

extern "C" int ka_tree_kernel_ms(ka_ctx* c, float* ms, int* n_launches)
{
        if (!c || !c->synced) return fail("run + sync first");
        HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
        if (n_launches) *n_launches = c->n_launches;
        return KA_OK;
}

// Measurements: the duration of every launch of the last run (context created with KA_LAUNCH_EV=1 in the environment);
// returns the number of launches written, -1 on error.
extern "C" int ka_tree_launch_ms(ka_ctx* c, float* ms, int cap)
{
        if (!c || !c->synced) { fail("run + sync first"); return -1; }
        if (!c->env.launch_ev || (int)c->launch_ev.size() < c->n_launches) { fail("ka_tree_launch_ms: no launch events (KA_LAUNCH_EV=1)"); return -1; }
        const int n = std::min(cap, c->n_launches);
        for (int i = 0; i < n; i++)
                if (hipEventElapsedTime(ms + i, i ? c->launch_ev[i - 1] : c->ev0, c->launch_ev[i]) != hipSuccess) { fail("hipEventElapsedTime"); return -1; }
        return n;
}


