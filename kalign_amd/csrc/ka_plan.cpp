// ka_plan.cpp -- the launch planner of the task tree (round 5: out of ka_api.cpp): which guide-tree levels run as leaf launches, which as
// ONE queued launch, from where the rest of the tree is ONE chained launch; the workgroup table of every launch; the spare workgroups of
// the chained launch by a greedy pass over a simulated schedule (DESIGN.md sections 4, 4d, 4h).  Replaces the task order of
// create_msa_tree / recursive_aln (lib/src/aln_run.c:43-124) -- the reference walks the tree with OpenMP tasks.
#include "ka_ctx.h"

// mean seq_distance over both clusters in sip order (aln_run.c:126-203)
static float mean_distance(const float* dist, const std::vector<int>& ma, const std::vector<int>& mb, int numseq, int* count)
{
        float sum = 0.0f;
        int n = 0;
        for (int x : ma) if (x < numseq) { sum += dist[x]; n++; }
        for (int x : mb) if (x < numseq) { sum += dist[x]; n++; }
        *count = n;
        return n ? sum / (float)n : 0.0f;
}

// Launch plan of the uploaded job: parents and join counts of the chained launch, workgroup tables per level.
// Depends on c->shared_gpu (no clusters, no chain), so ka_tree_sync can re-plan after a residency failure.
int plan_launches(ka_ctx* c)
{
        const int numseq = c->numseq, n_tasks = c->n_tasks;
        const int* abc = c->abc.data();
        const int max_level = (int)c->levels.size();
        // the tasks this plan covers: all of them, or the subset of ka_tree_plan_tasks (a rank's subtrees of a sharded
        // tree: closed under descendants).  A task outside the plan is neither a parent nor a producer in it.
        const bool subset = !c->plan_active.empty();
        auto act = [&](int t) { return !subset || c->plan_active[t] != 0; };
        c->plan_levels.assign(max_level, std::vector<int>());
        for (int L = 0; L < max_level; L++) for (int t : c->levels[L]) if (act(t)) c->plan_levels[L].push_back(t);
        const std::vector<std::vector<int>>& levels = c->plan_levels;
        // ---- parents, and the level from which the rest of the tree runs as ONE chained launch: the first
        // non-leaf level with at most one task per CU (all its workgroups resident at once; levels only get
        // narrower above it).  KA_NO_CHAIN=1 keeps one launch per level.
        {
                std::vector<int> task_of((2 * numseq - 1), -1);
                for (int t = 0; t < n_tasks; t++) task_of[abc[3 * t + 2]] = t;
                for (int t = 0; t < n_tasks; t++) { c->descs[t].parent = -1; c->descs[t].chain_need = 0; }
                for (int t = 0; t < n_tasks; t++) c->descs[t].is_root = 1;
                for (int t = 0; t < n_tasks; t++) {
                        const int a = abc[3 * t], b = abc[3 * t + 1];
                        // (is_root is a property of the tree: the root's task builds no profile.  parent is one of the plan.)
                        if (a >= numseq) { c->descs[task_of[a]].is_root = 0; if (act(t) && act(task_of[a])) c->descs[task_of[a]].parent = t; }
                        if (b >= numseq) { c->descs[task_of[b]].is_root = 0; if (act(t) && act(task_of[b])) c->descs[task_of[b]].parent = t; }
                }
                {
                        // join watchdog of the chained launch: ~2 s per 4e9 estimated DP cells below the task (a healthy
                        // sibling subtree of a huge job may legitimately take longer than the base bound)
                        std::vector<double> len(2 * numseq - 1, 0.0), cells(2 * numseq - 1, 0.0);
                        for (int i = 0; i < numseq; i++) len[i] = c->lens[i];
                        for (int t = 0; t < n_tasks; t++) {
                                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                                len[cc] = 1.1 * std::max(len[a], len[b]);
                                cells[cc] = cells[a] + cells[b] + len[a] * len[b];
                                c->descs[t].wait_mult = 1 + (int)std::min(63.0, cells[cc] / 4e9);
                                // (descs[t].refine -- the edges a KALIGN_REFINE_CONFIDENT pass refines -- is not part of the plan: it is
                                // set by ka_tree_refine and must survive the re-plan of a watchdog fallback, ka_tree_sync)
                        }
                }
                c->n_trees = numseq - n_tasks;
                c->chain_level = -1;
                if (!c->env.no_chain && !c->shared_gpu) {
                        for (int L = 0; L + 1 < max_level; L++) {
                                bool all_ss = true;
                                for (int t : levels[L]) if (c->descs[t].nsip_a != 1 || c->descs[t].nsip_b != 1) all_ss = false;
                                // (one tree: the chain starts where a level has at most 200 tasks -- the ~50 workgroups that leaves free go to the
                                // entries under the critical path, and since the chain overlaps the queue (round 5) a later start costs little:
                                // 4096 x 400 aa 14.70 -> 14.61 ms, default mode 21.4 -> 20.6, 4096 x 2000 nt 71.8 -> 70.2, 1024 x 2000 nt 33.3 -> 31.9;
                                // a forest keeps every CU it can get: four 4096 x 2000 nt trees 183 -> 191 with 200; profiles/r05_chain_start.log)
                                int chain_tasks = (c->n_trees <= 1) ? std::min(c->n_cus - 8, 200) : c->n_cus - 8;
                                if (c->env.chain_tasks > 0) chain_tasks = std::min(chain_tasks, c->env.chain_tasks);   // experiments
                                if (!all_ss && (int)levels[L].size() <= chain_tasks) { c->chain_level = L; break; }   // one workgroup per CU, all resident
                        }
                }
                // SPINES IN THE CHAIN (round 6; KA_SPINE = how many; built, bit-identical, measured, OFF: DESIGN 4j).  What the single tree waits for is not
                // a CU or an operand but the LATENCY of the tasks on its longest dependency chains (13.3 of 14.7 ms are the run times of 17
                // tasks, tools/levels_real.py) -- and the first four or five of those run in the queued launch: one four-wave workgroup,
                // ka_strip with its event steps, 0.54-0.72 ms for a 430 x 420 task that takes 0.33-0.45 ms in the chained launch (helper
                // strips, two workgroups).  So the chain reaches DOWN along the most critical entries: from each of the KA_SPINE entries of
                // the chain's first level with the longest estimated path through them (leaves .. entry .. root), the child with the later
                // estimated finish, and its child, ... down to the queue's first level are tasks of the chained launch too (c->spine):
                // the lowest one is an entry of its own (both children come from the leaf levels / the queue: done flags), the others
                // have ONE child inside the launch (chain_need 1) and one from the queue (qa / qb).  Nothing in the queue consumes a spine
                // task (its parent is the spine task above it), so the queue's order stays topological.
                c->spine.assign(n_tasks, 0);
                {
                        int L0 = 0;
                        while (c->chain_level >= 1 && L0 < c->chain_level) {
                                bool all_ss = true;
                                for (int t : levels[L0]) if (c->descs[t].nsip_a != 1 || c->descs[t].nsip_b != 1) all_ss = false;
                                if (!all_ss) break;
                                L0++;
                        }
                        const bool queue_ok = c->chain_level >= 1 && !c->env.no_queue && !c->env.no_half && c->chain_level - L0 >= 2 && (int)levels[L0].size() > c->n_cus;
                        const bool overlap_ok = c->env.overlap > 0 && !subset && !c->env.no_lean && !c->shared_gpu;
                        const int K = env_int("KA_SPINE", 0);             // (measured: 1 % at best with KA_RESERVE -- DESIGN 4j; off)
                        if (K > 0 && queue_ok && overlap_ok && !c->env.no_crit) {
                                std::vector<double> lmax(2 * numseq - 1, 0.0), nmem(2 * numseq - 1, 1.0), qlen(2 * numseq - 1, 0.0), fin(2 * numseq - 1, 0.0), dur(n_tasks, 0.0), upw(n_tasks, 0.0);
                                for (int i = 0; i < numseq; i++) { lmax[i] = c->lens[i]; qlen[i] = c->lens[i]; }
                                for (int t = 0; t < n_tasks; t++) {
                                        const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                                        lmax[cc] = std::max(lmax[a], lmax[b]); nmem[cc] = nmem[a] + nmem[b];
                                        qlen[cc] = lmax[cc] * (1.0 + 0.1 * std::sqrt(nmem[cc]));
                                        dur[t] = 2.0 * std::max(qlen[a], qlen[b]) + std::min(qlen[a], qlen[b]);
                                        fin[cc] = std::max(fin[a], fin[b]) + dur[t];
                                }
                                for (int t = n_tasks - 1; t >= 0; t--) upw[t] = dur[t] + ((act(t) && c->descs[t].parent >= 0) ? upw[c->descs[t].parent] : 0.0);
                                std::vector<int> ent(levels[c->chain_level].begin(), levels[c->chain_level].end());
                                std::stable_sort(ent.begin(), ent.end(), [&](int x, int y) { return fin[abc[3 * x + 2]] + upw[x] > fin[abc[3 * y + 2]] + upw[y]; });
                                for (int e = 0; e < (int)ent.size() && e < K; e++) {
                                        int cur = ent[e];
                                        while (true) {
                                                int best = -1;
                                                for (int k = 0; k < 2; k++) {
                                                        const int ch = abc[3 * cur + k];
                                                        if (ch < numseq) continue;
                                                        const int tc = task_of[ch];
                                                        if (!act(tc) || c->task_level[tc] < L0 || c->task_level[tc] >= c->chain_level) continue;
                                                        if (best < 0 || fin[ch] > fin[abc[3 * best + 2]]) best = tc;
                                                }
                                                if (best < 0) break;
                                                c->spine[best] = 1;
                                                cur = best;
                                        }
                                }
                        }
                }
                auto inchain = [&](int t) { return t >= 0 && act(t) && (c->task_level[t] >= c->chain_level || c->spine[t]); };
                if (c->chain_level >= 0) {
                        for (int t = 0; t < n_tasks; t++) {
                                if (!inchain(t)) continue;
                                int need = 0;
                                for (int k = 0; k < 2; k++) {
                                        const int ch = abc[3 * t + k];
                                        if (ch >= numseq && inchain(task_of[ch])) need++;
                                }
                                c->descs[t].chain_need = need;
                        }
                        // tests: make the last join wait for a workgroup that never comes (a residency failure as seen
                        // from the device) -- the bounded wait must report it and ka_tree_sync must re-plan and re-run
                        if (c->test_hooks & KA_DEBUG_STARVE_ROOT_JOIN) c->descs[n_tasks - 1].chain_need += 1;
                }
                // ---- the queued launch: every level between the seq-seq leaves and the chained launch (each holds more
                // tasks than the GPU has workgroup slots) as ONE launch of the half kernel; see ka_task_queue_entry.
                // KA_NO_QUEUE=1 keeps one launch per level.
                for (int t = 0; t < n_tasks; t++) { c->descs[t].qa = -1; c->descs[t].qb = -1; }
                c->queue_first = -1;
                c->overlap_plan = 0;
                if (c->chain_level >= 1 && !c->env.no_queue && !c->env.no_half) {
                        int L0 = 0;
                        while (L0 < c->chain_level) {                       // skip the leading seq-seq levels (lean kernel)
                                bool all_ss = true;
                                for (int t : levels[L0]) if (c->descs[t].nsip_a != 1 || c->descs[t].nsip_b != 1) all_ss = false;
                                if (!all_ss) break;
                                L0++;
                        }
                        bool ok = c->chain_level - L0 >= 2;                 // one level alone gains nothing
                        if ((int)levels[L0].size() <= c->n_cus) ok = false;   // (the queue's first level must fill the GPU; later ones need not)
                        if (ok) {
                                c->queue_first = L0;
                                // Overlapping launches (KA_OVERLAP; whole-tree plans only): the chained launch goes out beside the queued one, and
                                // its tasks wait for the done flags of what the QUEUE makes for them (qa / qb of a chain task: its operands'
                                // producers in the queued launch; inside the chain the join points order things as before).
                                c->overlap_plan = (c->env.overlap > 0 && c->plan_active.empty() && !c->env.no_lean && !c->shared_gpu) ? c->env.overlap : 0;
                                const int lo = L0;
                                for (int t = 0; t < n_tasks; t++) {
                                        if (c->task_level[t] < L0 || !act(t)) continue;
                                        // (a task of the chained launch: only what OTHER launches make -- inside the chain the join points order things)
                                        const bool in_chain = inchain(t);
                                        if (in_chain && !c->overlap_plan) continue;
                                        const int a = abc[3 * t], b = abc[3 * t + 1];
                                        auto dep = [&](int node) -> int {
                                                if (node < numseq || !act(task_of[node])) return -1;
                                                const int lv = c->task_level[task_of[node]];
                                                if (lv < lo || (in_chain && inchain(task_of[node]))) return -1;
                                                return task_of[node];
                                        };
                                        c->descs[t].qa = dep(a);
                                        c->descs[t].qb = dep(b);
                                }
                        }
                }
        }

        // (the spines only exist beside an overlapping queued launch: the conditions above are the queue block's own)
        {
                bool any = false;
                for (int t = 0; t < n_tasks; t++) any = any || c->spine[t];
                if (any && !(c->queue_first >= 0 && c->overlap_plan)) return fail("plan_launches: spine tasks without an overlapping queued launch");
        }
        // ---- workgroup tables, one per dependency level (build_blocks) ----
        // Workgroups one task may use: 16, or 32 for jobs whose top tasks are big enough to be work-bound at 16 (round 4: a
        // 9000 x 9700 task of C3 takes 5.8 ms on 16 workgroups, of which ~1.6 ms are the wavefront's dependent steps) -- by the
        // estimated root (longest sequence x (1 + 0.1 sqrt(sequences)), squared): >= 6e7 cells.  Measured, limit 16 -> 32
        // (profiles/r04_max_cluster.log): C3 81.9 -> 74.8 ms, 1024 x 2000 nt 34.9 -> 33.0, 512 x 3000 nt 43.6 -> 41.9; 16384 x 500 aa
        // and 2048 x 1000 aa unchanged; 4096 x 400 aa and 8192 x 300 aa 1-2 % slower (surplus members waiting at the joins).
        {
                double lmax = 0.0;
                for (int i = 0; i < numseq; i++) lmax = std::max(lmax, (double)c->lens[i]);
                const double root = lmax * (1.0 + 0.1 * std::sqrt((double)numseq));
                // (... and for big jobs with a consistency table: the votes of their top tasks share by member ranges from 20 workgroups on)
                const bool big_cons = c->cons_K > 0 && numseq >= 2048;
                c->max_cluster = c->env.max_cluster > 0 ? std::min(32, c->env.max_cluster) : ((root * root >= 6e7 || big_cons) ? 32 : 16);
        }
        if (c->shared_gpu) c->max_cluster = 1;
        c->blocks_flat.clear(); c->blocks_off.assign(1, 0); c->level_lean.clear();
        for (auto& L : levels) {
                std::vector<int2> tbl;
                int lean = 0;
                build_blocks(c, L, tbl, &lean);
                c->level_lean.push_back(lean);
                c->blocks_flat.insert(c->blocks_flat.end(), tbl.begin(), tbl.end());
                c->blocks_off.push_back((int)c->blocks_flat.size());
        }

        c->queue_off = (int)c->blocks_flat.size(); c->queue_n = 0;
        if (c->queue_first >= 0) {
                // Within a level the tasks with the longest way to the root go first (estimated wavefront steps of the task and of
                // everything above it, as for the chain's spare workgroups below): what the chained launch waits for longest -- the
                // spine of a caterpillar tree -- then leaves the queue early instead of wherever the task list put it.  KA_QORDER=0: list order.
                std::vector<double> qlen(2 * numseq - 1, 0.0), qup(n_tasks, 0.0);
                if (env_int("KA_QORDER", 1)) {
                        std::vector<double> lmax(2 * numseq - 1, 0.0), nmem(2 * numseq - 1, 1.0);
                        for (int i = 0; i < numseq; i++) { lmax[i] = c->lens[i]; qlen[i] = c->lens[i]; }
                        for (int t = 0; t < n_tasks; t++) {
                                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                                lmax[cc] = std::max(lmax[a], lmax[b]); nmem[cc] = nmem[a] + nmem[b];
                                qlen[cc] = lmax[cc] * (1.0 + 0.1 * std::sqrt(nmem[cc]));
                        }
                        for (int t = n_tasks - 1; t >= 0; t--) {               // parents come after their children in the task list
                                const double la = qlen[abc[3 * t]], lb = qlen[abc[3 * t + 1]];
                                qup[t] = 2.0 * std::max(la, lb) + std::min(la, lb) + ((act(t) && c->descs[t].parent >= 0) ? qup[c->descs[t].parent] : 0.0);
                        }
                }
                if (env_int("KA_QORDER", 1) == 2) {
                        // experiment (round 6; measured and not kept): ONE order over all the queue's levels, by the way to the root alone.  A
                        // child's way is longer than its parent's, so the order is still topological (a producer is pulled before its consumer: no
                        // deadlock) -- the subtrees under the spine go first whatever their level, at the price of consumers pulled right behind
                        // their producers, whose workgroups then hold a slot and wait: the chain gains 0.3 ms, the queue loses 1.2 (headline
                        // 14.70 -> 15.45 ms, C3 70.0 -> 77.6, 16384 x 500 41.4 -> 45.7, sixteen trees unchanged; profiles/r06_queue_order.log).
                        std::vector<int> all;
                        for (int L = c->queue_first; L < c->chain_level; L++) for (int t : levels[L]) if (!c->spine[t]) all.push_back(t);
                        std::stable_sort(all.begin(), all.end(), [&](int x, int y) { return qup[x] > qup[y]; });
                        for (int t : all) { c->blocks_flat.push_back(make_int2(t, 1 << 8)); c->queue_n++; }
                } else
                for (int L = c->queue_first; L < c->chain_level; L++) {
                        std::vector<int> lv;
                        for (int t : levels[L]) if (!c->spine[t]) lv.push_back(t);       // (the spines' tasks run in the chained launch)
                        std::stable_sort(lv.begin(), lv.end(), [&](int x, int y) { return qup[x] > qup[y]; });
                        for (int t : lv) { c->blocks_flat.push_back(make_int2(t, 1 << 8)); c->queue_n++; }
                }
        }
        if (c->chain_level >= 0) {
                // Every task of the chain's first level starts on a single workgroup; clusters form on the way up.
                // Entries are laid out in depth-first order of the upper tree, one contiguous run per XCD
                // (block b runs on XCD b % 8 -- observed, not contractual): subtrees that merge early share an
                // L2, only the top three levels cross XCDs.
                std::vector<int> task_of((2 * numseq - 1), -1), order;
                for (int t = 0; t < n_tasks; t++) task_of[abc[3 * t + 2]] = t;
                std::vector<int> stack;
                auto inchain2 = [&](int t) { return t >= 0 && act(t) && (c->task_level[t] >= c->chain_level || c->spine[t]); };
                for (int t = n_tasks - 1; t >= 0; t--) if (act(t) && c->descs[t].parent < 0 && c->task_level[t] >= c->chain_level) stack.push_back(t);   // every root above the cut
                while (!stack.empty()) {
                        const int t = stack.back(); stack.pop_back();
                        // an entry of the chain: no child of it runs inside the launch (the chain's first level -- unless a spine hangs below
                        // it --, the lowest task of a spine; in a plan over a subset also a task whose children were all run before)
                        if (c->descs[t].chain_need == 0) { order.push_back(t); continue; }
                        for (int k = 1; k >= 0; k--) {
                                const int ch = abc[3 * t + k];
                                if (ch >= numseq && inchain2(task_of[ch])) stack.push_back(task_of[ch]);
                        }
                }
                const int m = ((int)order.size() + 7) / 8;
                // A narrow upper tree (the chain-like UPGMA trees of a realignment pass) never merges clusters: its
                // tasks would all run on the one workgroup they started with.  Start with as many workgroups per
                // task as a separate launch of this level would get (build_blocks); members of one cluster sit in
                // one column = one XCD.
                int G0 = 1;
                while (G0 * 2 <= c->max_cluster && 8 * m * G0 * 2 <= c->n_cus) G0 *= 2;
                if (c->env.chain_g1) G0 = 1;
                // The CUs this leaves idle go to the entries whose way to the root is the longest (estimated wavefront steps
                // of the tasks above them): clusters only grow where subtrees of the SAME launch meet, and the critical path
                // of a k-means tree is a caterpillar that absorbs small subtrees finished by earlier launches -- its tasks
                // would run on the one workgroup their entry started with while most of the GPU waits at join points.  A
                // cluster keeps its workgroups all the way up (surplus members climb with it), so a workgroup given to an
                // entry serves every task on that entry's path.  Extra members sit behind the regular table, in the
                // entry's XCD column.
                std::vector<int> extra(order.size(), 0);
                int spare = (c->n_cus - 8 * m * G0) / 8 * 8;
                if (!c->env.no_crit && spare > 0 && !order.empty()) {
                        std::vector<double> len(2 * numseq - 1, 0.0), up(n_tasks, 0.0);
                        for (int i = 0; i < numseq; i++) len[i] = c->lens[i];
                        if (c->env.crit_greedy) {
                                // profile lengths are only known on the device; the estimate: the longest member sequence times
                                // (1 + 0.1 sqrt(members)) -- the growth of the DSSim sets with their indel-rich tails (13151 columns for
                                // 4096 x 2000 nt, 2965 for 4096 x 400 aa), harmless where alignments stay shorter
                                std::vector<double> lmax(2 * numseq - 1, 0.0), nmem(2 * numseq - 1, 1.0);
                                for (int i = 0; i < numseq; i++) lmax[i] = c->lens[i];
                                for (int t = 0; t < n_tasks; t++) {
                                        const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                                        lmax[cc] = std::max(lmax[a], lmax[b]); nmem[cc] = nmem[a] + nmem[b];
                                        len[cc] = lmax[cc] * (1.0 + 0.1 * std::sqrt(nmem[cc]));
                                }
                        } else
                        for (int t = 0; t < n_tasks; t++) len[abc[3 * t + 2]] = 1.1 * std::max(len[abc[3 * t]], len[abc[3 * t + 1]]);
                        for (int t = n_tasks - 1; t >= 0; t--) {               // parents come after their children in the task list
                                const double la = len[abc[3 * t]], lb = len[abc[3 * t + 1]];
                                up[t] = 2.0 * std::max(la, lb) + std::min(la, lb) + (c->descs[t].parent >= 0 ? up[c->descs[t].parent] : 0.0);
                        }
                        // Round 4: first a GREEDY pass on a simulated schedule.  The ranking below only knows how LONG an entry's way
                        // to the root is, not how it will be staffed: a caterpillar spine that absorbs siblings finished by earlier
                        // launches stays on the one workgroup of its entry through level after level of 2400 x 2300 tasks (C3: six
                        // of them at 3.8 ms, a third of the launch, next to ~200 idle CUs) while a spine fed by subtrees of THIS
                        // launch collects their workgroups at every join.  Model: a task on G workgroups takes
                        // a * (2 max + min) + b * la * lb / G (fitted on C3's and the headline's task times: the first term the
                        // wavefront's dependent steps, the second the cells shared by the cluster; b / a = 0.02 from the fit, 0.01 in use), a parent has the
                        // workgroups of its children in this launch (up to the limit) and starts when the later one ends.  One spare
                        // workgroup at a time goes to the entry under the simulated critical path, until it stops paying; what is
                        // left goes out by the ranking.  KA_CRIT_GREEDY=0: the ranking alone (round 3).
                        if (c->env.crit_greedy) {
                                std::vector<int> entry_of(n_tasks, -1);
                                for (size_t r = 0; r < order.size(); r++) entry_of[order[r]] = (int)r;
                                auto in_chain = [&](int t) { return inchain2(t); };
                                std::vector<double> fin(n_tasks, 0.0);
                                std::vector<int> Gt(n_tasks, 0), crit_child(n_tasks, -1);
                                const double ba = 1e-3 * (double)env_int("KA_CRIT_BA", 10);   // (b / a of the model, per mille; 10 from a sweep over five job shapes, profiles/r04_crit_ba.log)
                                auto simulate = [&]() -> int {
                                        int last = -1;
                                        for (int t = 0; t < n_tasks; t++) {                  // children come before their parents
                                                if (!in_chain(t)) continue;
                                                double start = 0.0; int cc = -1, G = 0;
                                                if (entry_of[t] >= 0) G = G0 + extra[entry_of[t]];
                                                else {
                                                        for (int k = 0; k < 2; k++) {
                                                                const int ch = abc[3 * t + k];
                                                                const int tc = ch >= numseq ? task_of[ch] : -1;
                                                                if (!in_chain(tc)) continue;
                                                                G += Gt[tc];
                                                                if (fin[tc] >= start) { start = fin[tc]; cc = tc; }
                                                        }
                                                        G = std::max(1, std::min(G, c->max_cluster));
                                                }
                                                const double la = len[abc[3 * t]], lb = len[abc[3 * t + 1]];
                                                fin[t] = start + 2.0 * std::max(la, lb) + std::min(la, lb) + ba * la * lb / G;
                                                Gt[t] = G; crit_child[t] = cc;
                                                if (last < 0 || fin[t] > fin[last]) last = t;
                                        }
                                        return last;                                       // the task that ends last (a root)
                                };
                                // (several paths can be critical at once: a workgroup that shortens ONE of them leaves the end where it was.
                                // Keep going -- the next round takes the next path -- and fall back to the best state seen when a
                                // stretch of eight additions has not moved the end.)
                                int given = 0, since_best = 0;
                                std::vector<int> best_extra = extra;
                                int best_spare = spare;
                                double best_end = -1.0;
                                { const int t = simulate(); if (t >= 0) best_end = fin[t]; }
                                while (spare > 0 && best_end > 0.0 && since_best < 8) {
                                        int t = simulate();
                                        if (t < 0) break;
                                        while (crit_child[t] >= 0) t = crit_child[t];        // down the critical path to its entry
                                        const int r = entry_of[t];
                                        if (r < 0 || G0 + extra[r] >= c->max_cluster) break;
                                        extra[r] += 1; spare -= 1;
                                        const int t2 = simulate();
                                        if (fin[t2] < best_end * (1.0 - 1e-4)) { best_end = fin[t2]; best_extra = extra; best_spare = spare; since_best = 0; }
                                        else since_best += 1;
                                }
                                extra = best_extra; spare = best_spare;
                                for (size_t r = 0; r < order.size(); r++) given += extra[r];
                                if (getenv("KA_PLAN_VERBOSE")) {
                                        const int t = simulate();
                                        fprintf(stderr, "chain plan: greedy pass gave %d workgroups, %d left for the ranking; simulated end %.0f\n", given, spare, t >= 0 ? fin[t] : 0.0);
                                        for (size_t r = 0; r < order.size(); r++) if (extra[r] > 0)
                                                fprintf(stderr, "  entry task %d (node %d) level %d: +%d\n", order[r], abc[3 * order[r] + 2], c->task_level[order[r]], extra[r]);
                                }
                        }
                        std::vector<int> by_up(order.size());
                        for (size_t r = 0; r < order.size(); r++) by_up[r] = (int)r;
                        std::stable_sort(by_up.begin(), by_up.end(), [&](int x, int y) { return up[order[x]] > up[order[y]]; });
                        int top_g = 4;
                        if (c->env.crit_top > 0) top_g = std::min(c->max_cluster, c->env.crit_top);   // experiments
                        for (size_t i = 0; i < by_up.size() && spare > 0; i++) {
                                // (never beyond the cluster limit: surplus workgroups would only spin at a join and leave)
                                const int want = std::min(spare, std::max(0, std::min(c->max_cluster, i == 0 ? top_g : 2 * G0) - G0 - extra[by_up[i]]));
                                extra[by_up[i]] += want; spare -= want;
                        }
                        if (getenv("KA_PLAN_VERBOSE")) {
                                fprintf(stderr, "chain plan: level %d, %zu entries, G0 %d, spare after extras %d, top_g %d\n", c->chain_level, order.size(), G0, spare, top_g);
                                for (size_t i = 0; i < by_up.size() && i < 12; i++) {
                                        const int t = order[by_up[i]];
                                        fprintf(stderr, "  rank %zu: task %d (node %d) level %d lens %.0f x %.0f up %.0f extra %d\n", i, t, abc[3 * t + 2], c->task_level[t],
                                                len[abc[3 * t]], len[abc[3 * t + 1]], up[t], extra[by_up[i]]);
                                }
                        }
                }
                // CUs KEPT FOR THE HEAD OF THE CHAIN (round 6; KA_RESERVE = 8 / 16 / 24; built, measured, OFF: the spine then starts 1 ms earlier and the run is as long -- other chains of the same length take over).  The chained launch goes out beside the
                // queued one, but a workgroup of it wants a CU's whole LDS and the queue's 512 persistent workgroups hold two to a CU until
                // their list is empty: the chain only ever started when the queue was over (the spine's first chained task of the headline
                // tree waited 1.0 ms for a CU for its cluster's second workgroup; tools/levels_real.py, `prep`).  Now the queue's workgroups
                // that find themselves on the first R CUs of XCC 0 (shader engines 0 .. R/8 - 1; ka_task_queue_entry reads HW_ID / XCC_ID)
                // leave at once, and the chain's most critical entries -- by their estimated way to the root, all their workgroups, as
                // many as fit R -- sit at the head of column 0 of the block table: block b goes to XCC b % 8 (tools/microbench/cu_map.hip:
                // 0 exceptions in 512), XCC 0 dispatches its share of the chain in order onto the CUs the queue left, and those
                // workgroups wait for their operands' done flags instead of for a CU (they do not help the queue: KA_BLK_NOHELP).
                c->reserve_cus = 0;
                int front_n = 0;
                {
                        int R = env_int("KA_RESERVE", 0);              // (measured: no gain -- DESIGN 4j; off)
                        R = std::max(0, std::min(24, R / 8 * 8));
                        if (R > 0 && c->overlap_plan && c->queue_first >= 0 && c->n_trees <= 1 && !subset && (int)order.size() > 8 && !c->env.no_crit) {
                                std::vector<double> lmax(2 * numseq - 1, 0.0), nmem(2 * numseq - 1, 1.0), qlen(2 * numseq - 1, 0.0), crit(n_tasks, 0.0);
                                for (int i = 0; i < numseq; i++) { lmax[i] = c->lens[i]; qlen[i] = c->lens[i]; }
                                for (int t = 0; t < n_tasks; t++) {
                                        const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                                        lmax[cc] = std::max(lmax[a], lmax[b]); nmem[cc] = nmem[a] + nmem[b];
                                        qlen[cc] = lmax[cc] * (1.0 + 0.1 * std::sqrt(nmem[cc]));
                                }
                                for (int t = n_tasks - 1; t >= 0; t--) {
                                        const double la = qlen[abc[3 * t]], lb = qlen[abc[3 * t + 1]];
                                        crit[t] = 2.0 * std::max(la, lb) + std::min(la, lb) + ((act(t) && c->descs[t].parent >= 0) ? crit[c->descs[t].parent] : 0.0);
                                }
                                std::vector<int> idx(order.size());
                                for (size_t r = 0; r < order.size(); r++) idx[r] = (int)r;
                                // (the lowest tasks of the spines first: they are what the kept CUs are for; a 430-row task uses two workgroups)
                                for (size_t r = 0; r < order.size(); r++) if (c->task_level[order[r]] < c->chain_level) { crit[order[r]] += 1e12; extra[r] = std::min(extra[r], 1); }
                                std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return crit[order[x]] > crit[order[y]]; });
                                std::vector<char> is_front(order.size(), 0);
                                std::vector<int> front;
                                int used = 0;
                                for (int r : idx) {
                                        const int Gr = G0 + extra[r];
                                        if (used + Gr > R || (int)front.size() + 1 > m) break;
                                        front.push_back(r); is_front[r] = 1; used += Gr;
                                }
                                if (!front.empty()) {
                                        std::vector<int> o2, e2;
                                        for (int r : front) { o2.push_back(order[r]); e2.push_back(extra[r]); }
                                        for (size_t r = 0; r < order.size(); r++) if (!is_front[r]) { o2.push_back(order[r]); e2.push_back(extra[r]); }
                                        order.swap(o2); extra.swap(e2);
                                        front_n = (int)front.size();
                                        c->reserve_cus = R;
                                }
                        }
                }
                int n_extra = 0;
                std::vector<int> col_need(8, 0);
                for (size_t r = 0; r < order.size(); r++) { n_extra += extra[r]; col_need[r / m] += extra[r]; }
                int extra_rows = *std::max_element(col_need.begin(), col_need.end());
                const bool by_column = 8 * m * G0 + 8 * extra_rows <= c->n_cus;       // else: packed densely, any XCD
                if (!by_column) extra_rows = (n_extra + 7) / 8;
                c->chain_blocks.assign((size_t)8 * m * G0 + (size_t)8 * extra_rows, make_int2(-1, 0));
                std::vector<int> col_fill(8, 0);
                int dense = 0;
                for (int r = 0; r < (int)order.size(); r++) {
                        const int Gr = G0 + extra[r];
                        for (int g = 0; g < G0; g++)
                                c->chain_blocks[((size_t)(r % m) * G0 + g) * 8 + (r / m)] = make_int2(order[r], g | (Gr << 8));
                        for (int g = G0; g < Gr; g++) {
                                const size_t pos = (size_t)8 * m * G0 + (by_column ? (size_t)8 * col_fill[r / m]++ + (r / m) : (size_t)dense++);
                                c->chain_blocks[pos] = make_int2(order[r], g | (Gr << 8));
                        }
                }
                if (front_n > 0) {
                        // The front entries' workgroups -- ALL of them: their extra members sit behind the regular table, in whatever column --
                        // trade places with what the head of column 0 holds (a swap: every column keeps its number of workgroups, and with it
                        // the guarantee that the whole launch is resident at once).
                        std::vector<size_t> col0;
                        for (size_t p = 0; p < c->chain_blocks.size(); p += 8) if (c->chain_blocks[p].x >= 0) col0.push_back(p);
                        size_t i = 0;
                        bool ok = true;
                        for (int r = 0; r < front_n && ok; r++)
                                for (int g = 0; g < G0 + extra[r] && ok; g++, i++) {
                                        size_t src = c->chain_blocks.size();
                                        for (size_t p = 0; p < c->chain_blocks.size(); p++)
                                                if (c->chain_blocks[p].x == order[r] && (c->chain_blocks[p].y & 0xff) == g) { src = p; break; }
                                        if (src == c->chain_blocks.size() || i >= col0.size()) { ok = false; break; }
                                        std::swap(c->chain_blocks[col0[i]], c->chain_blocks[src]);
                                        c->chain_blocks[col0[i]].y |= KA_BLK_NOHELP;
                                }
                        if (!ok) { c->reserve_cus = 0; for (auto& bb : c->chain_blocks) if (bb.x >= 0) bb.y &= ~KA_BLK_NOHELP; }
                } else c->reserve_cus = 0;
                if (getenv("KA_PLAN_VERBOSE")) fprintf(stderr, "chain plan: %d CUs kept for the head of the chain, %d front entries, m %d, G0 %d, by_column %d, overlap %d, queue_first %d\n",
                                                       c->reserve_cus, front_n, m, G0, (int)by_column, c->overlap_plan, c->queue_first);
                c->chain_blocks_off = (int)c->blocks_flat.size();
                c->blocks_flat.insert(c->blocks_flat.end(), c->chain_blocks.begin(), c->chain_blocks.end());
        }

        return KA_OK;
}

extern "C" int ka_tree_upload(ka_ctx* c, int numseq, const uint8_t* codes, const int* off, const int* lens,
                              const float* seq_distances, int n_tasks, const int* abc,
                              const float* subm, const float* scal, int flags)
{
        if (!c) return fail("null ctx");
        // n_tasks == numseq-1: one guide tree.  Fewer tasks: a FOREST -- several independent alignments (a batch of
        // families, ensemble members) scheduled together; every task with no consumer is the root of its tree.
        if (numseq < 2 || n_tasks < 1 || n_tasks > numseq - 1) return fail("need numseq >= 2 and 1 <= n_tasks <= numseq-1");
        HIPCHK(hipSetDevice(c->device));
        const int nprof = 2 * numseq - 1;
        // kalign_run_realign aligns a second time on a new tree with the consistency table of the first pass
        // (aln_wrap.c:424-431,497-502): same sequences, new task list.  Anything else starts without a table.
        bool keep_cons = false;
        if ((flags & KA_FLAG_KEEP_CONSISTENCY) && c->have_job && c->cons_K > 0) {
                bool same = numseq == c->numseq;
                for (int i = 0; same && i < numseq; i++)
                        same = lens[i] == c->lens[i] && off[i] == c->off[i] && memcmp(codes + off[i], c->h_codes.data() + off[i], lens[i]) == 0;
                if (!same) return fail("KA_FLAG_KEEP_CONSISTENCY: the sequences differ from those the consistency table was built on");
                keep_cons = true;
        }
        c->have_job = false; c->ran = false; c->synced = false; c->state_valid = false;
        // a join watchdog of an earlier job forced the no-cluster plan: a new job gets the fast plan again (the
        // fallback is counted, ka_ctx_fallback_runs); a caller's own ka_ctx_set_shared stays
        // -- unless the fallback before it was a fast-plan job's too: then fallback_hold jobs stay on the shared plan first (ka_tree_sync)
        if (c->shared_by_fallback) {
                if (c->fallback_hold > 0) c->fallback_hold--;
                else { c->shared_gpu = false; c->shared_by_fallback = false; }
        }
        if (!keep_cons) c->cons_K = 0;           // a new job starts without a consistency table
        c->have_colof = false;
        c->rows_n = 0;
        c->numseq = numseq; c->n_tasks = n_tasks; c->flags = flags;
        c->lens.assign(lens, lens + numseq);
        c->off.assign(off, off + numseq);
        c->abc.assign(abc, abc + 3 * n_tasks);
        memcpy(c->subm, subm, sizeof(c->subm));
        memcpy(c->scal, scal, sizeof(c->scal));
        c->sum_len = 0; c->max_len = 0;
        long long codes_bytes = 0;
        int max_code = 0;
        for (int i = 0; i < numseq; i++)
                for (int j = 0; j < lens[i]; j++) max_code = std::max<int>(max_code, codes[off[i] + j]);
        if (max_code > 22) return fail("sequence code out of range (alphabet is 0..22)");
        // nucleotide alphabets use codes 0..4 (alphabet.c:206-245); proteins without B / Z / X only codes 0..19
        c->nres = (max_code < 5) ? 5 : (max_code < 20 ? 20 : 23);
        for (int i = 0; i < numseq; i++) {
                if (lens[i] < 1) return fail("zero-length sequence (the reference removes them before the dispatcher, msa_check.c:66)");
                c->sum_len += lens[i];
                c->max_len = std::max(c->max_len, lens[i]);
                codes_bytes = std::max<long long>(codes_bytes, (long long)off[i] + lens[i]);
        }

        c->h_codes.assign(codes, codes + codes_bytes);
        if (seq_distances) c->seq_dist.assign(seq_distances, seq_distances + numseq); else c->seq_dist.clear();
        c->sip_flat.clear(); c->sip_off.assign(nprof, 0);
        for (int i = 0; i < numseq; i++) { c->sip_off[i] = (long long)c->sip_flat.size(); c->sip_flat.push_back(i); }

        // ---- host-side task preparation: nsip, sip order, gap_scale / subm_offset, levels ----
        std::vector<int> nsip(nprof, 0), level(nprof, 0);
        std::vector<std::vector<int>> sip(nprof);
        std::vector<char> made(nprof, 0);
        for (int i = 0; i < numseq; i++) { nsip[i] = 1; sip[i] = {i}; made[i] = 1; }
        c->descs.assign(n_tasks, KaTaskDesc());
        const float gpo0 = scal[0], gpe0 = scal[1], tgpe0 = scal[2], dist_scale = scal[3], vsm_amax = scal[4];
        int max_level = 0;
        for (int t = 0; t < n_tasks; t++) {
                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                if (a < 0 || b < 0 || cc < numseq || a >= nprof || b >= nprof || cc >= nprof || !made[a] || !made[b] || made[cc])
                        return fail("task list is not in TASK_ORDER_TREE order (children before parents)");
                // a node is the operand of at most one task, and never both operands of it (its member list is
                // handed to the parent below)
                if (a == b || made[a] == 2 || made[b] == 2)
                        return fail("task list is not in TASK_ORDER_TREE order (a node is consumed twice)");
                made[a] = 2; made[b] = 2;
                KaTaskDesc& d = c->descs[t];
                float gap_scale = 1.0f, soff = 0.0f;
                int cnt = 0;
                if (dist_scale > 0.0f && seq_distances) {
                        const float avg = mean_distance(seq_distances, sip[a], sip[b], numseq, &cnt);
                        if (cnt) {
                                gap_scale = 1.0f - dist_scale * avg;
                                if (gap_scale < 0.3f) gap_scale = 0.3f;
                                if (gap_scale > 1.0f) gap_scale = 1.0f;
                        }
                }
                if (vsm_amax > 0.0f && seq_distances) {
                        const float avg = mean_distance(seq_distances, sip[a], sip[b], numseq, &cnt);
                        if (cnt) {
                                soff = vsm_amax - avg;
                                if (soff < 0.0f) soff = 0.0f;
                        }
                }
                d.a = a; d.b = b; d.c = cc;
                d.nsip_a = nsip[a]; d.nsip_b = nsip[b];
                d.is_root = 0;                                   // set below: tasks nobody consumes
                d.gpo = gpo0; d.gpe = gpe0; d.tgpe = tgpe0;
                if (gap_scale < 1.0f || soff > 0.0f) { d.gpo *= gap_scale; d.gpe *= gap_scale; d.tgpe *= gap_scale; }
                else soff = 0.0f;
                d.soff = soff; d.gap_scale = gap_scale; d.parent = -1; d.chain_need = 0;
                nsip[cc] = nsip[a] + nsip[b];
                sip[cc].reserve(nsip[cc]);
                for (int j = nsip[a]; j--;) sip[cc].push_back(sip[a][j]);        // aln_run.c:428-436
                for (int j = nsip[b]; j--;) sip[cc].push_back(sip[b][j]);
                c->sip_off[cc] = (long long)c->sip_flat.size();
                c->sip_flat.insert(c->sip_flat.end(), sip[cc].begin(), sip[cc].end());
                std::vector<int>().swap(sip[a]);
                std::vector<int>().swap(sip[b]);
                made[cc] = 1;
                level[cc] = 1 + std::max(level[a], level[b]);
                max_level = std::max(max_level, level[cc]);
        }
        c->levels.assign(max_level, std::vector<int>());
        for (int t = 0; t < n_tasks; t++) c->levels[level[abc[3 * t + 2]] - 1].push_back(t);
        c->level_ids_flat.clear(); c->level_off.assign(1, 0);
        for (auto& L : c->levels) {
                c->level_ids_flat.insert(c->level_ids_flat.end(), L.begin(), L.end());
                c->level_off.push_back((int)c->level_ids_flat.size());
        }

        c->task_level.assign(n_tasks, 0);
        for (int t = 0; t < n_tasks; t++) c->task_level[t] = level[abc[3 * t + 2]] - 1;
        c->plan_active.clear();
        if (plan_launches(c)) return KA_FAIL;

        // ---- arenas ----
        c->leaf_prof_off.assign(numseq, 0);
        long long top = 0;
        for (int i = 0; i < numseq; i++) { c->leaf_prof_off[i] = top; top += (long long)(lens[i] + 2) * KA_REC; }
        c->leaf_prof_total = top;
        // merged profiles: alignment lengths are only known on the device; start with a generous
        // estimate and let ka_tree_sync grow + re-run on overflow.
        const long long worst_cols = c->sum_len * (long long)std::max(1, max_level) + 2LL * n_tasks;
        const long long est_cols = 3LL * (long long)n_tasks * (c->max_len + 2) + 1024;
        const long long cols = std::min(worst_cols, est_cols);
        c->prof_cap = std::max(c->prof_cap, top + cols * KA_REC);
        c->path_cap = std::max(c->path_cap, cols + c->sum_len + 2LL * numseq + 1024);
        long long scr = 0;
        // per level every sequence is a member of at most one task; profile lengths never exceed
        // the sum of their members' lengths
        const long long scr_level = ka_scratch_bytes_host(c->sum_len, c->sum_len, c->max_len) / 2 + (long long)numseq * (2048 + 12LL * c->max_len) + 65536;
        scr = scr_level;
        // the chained launch and the queued launch never reset the scratch counter: several levels' worth; grows on demand
        if (c->chain_level >= 0) scr = scr_level * (long long)std::min(max_level - c->chain_level, 8);
        if (c->max_cluster > 1 && !c->shared_gpu) scr *= 2;          // clusters: every member's private queues and row buffers
        if (c->queue_first >= 0) {
                // The chained launch that overlaps the queue shares its arena (no reset between the two).  Where their sum is affordable --
                // single trees: a few GB of 288 -- the arena holds both, so that a one-shot caller (the CLI, the drop-in: one alignment
                // per process) does not pay an overflow, a doubled arena and a second run of the tree on its only job.  Big
                // forests keep the larger of the two estimates (the sum asked for more than the GPU holds on a 16-tree forest) and grow on
                // demand: ka_tree_sync doubles the arena and repeats the run, once per job shape.
                const long long scr_queue = scr_level * (long long)(c->chain_level - c->queue_first);
                const long long both = scr + scr_queue;
                scr = (c->env.overlap && both <= (24LL << 30)) ? both : std::max(scr, scr_queue);
        }
        c->scratch_cap = std::max(c->scratch_cap, scr);
        if (c->test_hooks & KA_DEBUG_SMALL_ARENAS) {
                // tests: start with arenas that are certainly too small, so that the overflow -> grow -> re-run
                // path of ka_tree_sync is exercised (also across the join points of the chained launch)
                c->prof_cap = top + 64LL * KA_REC; c->path_cap = 64; c->scratch_cap = 1 << 16;
                c->d_prof_arena.release(); c->d_path_arena.release(); c->d_scratch.release();
        }
        c->dbg_cap = (flags & KA_FLAG_DEBUG_ROWS) ? std::max<long long>(c->dbg_cap, 6LL * (cols + 2LL * n_tasks + c->sum_len)) : c->dbg_cap;

        if (c->d_codes.alloc((size_t)codes_bytes) || c->d_seq_off.alloc(numseq) || c->d_node_len.alloc(nprof) ||
            c->d_node_prof.alloc(nprof) || c->d_node_vote.alloc(nprof) || c->d_level_ids.alloc(c->level_ids_flat.size()) ||
            c->d_tasks.alloc(n_tasks) || c->d_recs.alloc(n_tasks) || c->d_subm.alloc(23 * 23) ||
            c->d_counters.alloc(8) || c->d_timing.alloc(8 * (size_t)n_tasks + 48 + 512) ||
            c->d_ctl.alloc((size_t)ka_ctl_bytes_host() * n_tasks) || c->d_join.alloc(n_tasks) || c->d_blocks.alloc(c->blocks_flat.size()) || c->d_error.alloc(1) || c->d_dbg_off.alloc(n_tasks) ||
            c->d_prof_arena.alloc((size_t)c->prof_cap) || c->d_path_arena.alloc((size_t)c->path_cap) ||
            c->d_scratch.alloc((size_t)c->scratch_cap) || c->d_dbg_arena.alloc((size_t)std::max<long long>(c->dbg_cap, 1)))
                return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->d_codes.p, codes, (size_t)codes_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_seq_off.p, off, sizeof(int) * numseq, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_level_ids.p, c->level_ids_flat.data(), sizeof(int) * c->level_ids_flat.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_tasks.p, c->descs.data(), sizeof(KaTaskDesc) * n_tasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_blocks.p, c->blocks_flat.data(), sizeof(int2) * c->blocks_flat.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_subm.p, subm, sizeof(float) * 23 * 23, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->have_job = true;
        if (((flags & KA_FLAG_DEVICE_GAPS) || keep_cons) && setup_colof(c)) { c->have_job = false; return KA_FAIL; }
        return KA_OK;
}

// (re)upload the launch plan: task descriptors (parents, join counts) and workgroup tables
int upload_plan(ka_ctx* c)
{
        if (c->d_blocks.alloc(c->blocks_flat.size())) return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(c->d_tasks.p, c->descs.data(), sizeof(KaTaskDesc) * c->n_tasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_blocks.p, c->blocks_flat.data(), sizeof(int2) * c->blocks_flat.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return KA_OK;
}

