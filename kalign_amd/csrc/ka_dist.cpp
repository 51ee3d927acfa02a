// ka_dist.cpp -- one alignment over the GPUs of a node, driven from C (round 5: out of ka_api.cpp).
#include "ka_ctx.h"

// =================================================================================================================
// One alignment over the GPUs of a node: one process per GPU, RCCL over xGMI, driven from C.
//
//   * anchor_consistency_build: every rank aligns its share of the N x K seq-seq batch and fills their position maps
//     in its copy of the table; every rank's range is then broadcast IN PLACE, HBM to HBM (ncclBroadcast, all ranges in
//     one group);
//   * the guide tree is cut ONCE per job into one subtree per rank (balanced by estimated DP cells); a rank's subtrees
//     run as ONE planned run (queued / chained launches, like a whole tree: ka_tree_plan_tasks); above the cut the
//     profile of the smaller child moves device to device (ncclSend / ncclRecv: a two-int header, the records from
//     where they lie in the source's arena into room reserved in the destination's, and -- default mode -- the
//     residue -> column table of the moved subtree, packed and unpacked on the device) to the rank that holds the other
//     child, which runs the parent on up to 16 CUs;
//   * records and coded paths: every rank scatters its own into the job-wide layout on the device and ONE all-reduce
//     each (disjoint ranges, zeros elsewhere: the sum of integers words is exact) leaves every rank with everything.
// Results do not depend on the number of ranks: tasks are position-addressed and a task's DP is the same code wherever
// it runs -- the reference's thread-count invariance (lib/src/aln_run.c:95-109, independent subtrees).
// RCCL is loaded at run time (dlopen): the single-GPU library has no link-time dependency on it.
// =================================================================================================================
#include <dlfcn.h>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#if __has_include(<rccl/rccl.h>) && !defined(KA_NO_RCCL_HEADERS)
#include <rccl/rccl.h>
#else
// (hosts without the RCCL development headers: the few names of NCCL's public, stable ABI this file uses -- the library
// itself is only ever looked for at run time, rccl_load)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclInt32 = 2 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
}
#endif

extern "C" void ka_launch_cols_pack(int* colof, const int* seq_off, const int* seq_len, const int* members, const long long* moff, int nmem,
                                    int* buf, int unpack, hipStream_t stream);
extern "C" void ka_launch_path_counts(const ka_task_rec* recs, const char* mine, int n_tasks, int* counts, hipStream_t stream);
extern "C" void ka_launch_path_scatter(const ka_task_rec* recs, const char* mine, int n_tasks, const int* arena, const long long* goff, int* out, hipStream_t stream);

namespace {
struct Rccl {
        void* lib = nullptr;
        ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
        ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
        ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
        ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*GroupStart)() = nullptr;
        ncclResult_t (*GroupEnd)() = nullptr;
        const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load()
{
        if (g_rccl.lib) return KA_OK;
        // One process must never hold two RCCLs: a caller that already carries one (PyTorch maps its own librccl.so; torch.distributed's
        // "nccl" backend IS that library) gets exactly that one.  First the symbols already in the process (the global scope: whatever
        // was linked or loaded RTLD_GLOBAL), then the library under each of its names if it is mapped already (RTLD_NOLOAD finds a copy that
        // was loaded RTLD_LOCAL), and only then a fresh load.
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        void* h = nullptr;
        {
                void* self = dlopen(nullptr, RTLD_NOW);
                if (self && dlsym(self, "ncclCommInitRank") && dlsym(self, "ncclSend")) h = self;
                else if (self) dlclose(self);
        }
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return fail(std::string("ka_dist: cannot load RCCL (librccl.so.1): ") + dlerror());
#define KA_SYM(field_, name_) *(void**)(&g_rccl.field_) = dlsym(h, name_); if (!g_rccl.field_) return fail(std::string("ka_dist: RCCL lacks ") + name_)
        KA_SYM(GetUniqueId, "ncclGetUniqueId"); KA_SYM(CommInitRank, "ncclCommInitRank"); KA_SYM(CommDestroy, "ncclCommDestroy");
        KA_SYM(Broadcast, "ncclBroadcast"); KA_SYM(AllReduce, "ncclAllReduce"); KA_SYM(Send, "ncclSend"); KA_SYM(Recv, "ncclRecv");
        KA_SYM(GroupStart, "ncclGroupStart"); KA_SYM(GroupEnd, "ncclGroupEnd"); KA_SYM(GetErrorString, "ncclGetErrorString");
#undef KA_SYM
        g_rccl.lib = h;
        return KA_OK;
}
#define NCCLCHK(x)                                                                                         \
        do {                                                                                               \
                ncclResult_t r_ = (x);                                                                     \
                if (r_ != ncclSuccess) return fail(std::string(#x) + ": " + g_rccl.GetErrorString(r_));     \
        } while (0)

// An in-process stand-in for the communicator (tests): the ranks are threads of ONE process, each with its own context on
// the SAME GPU -- RCCL refuses two ranks on one device, and the pool's GPU boxes have one.  Host-synchronous, FIFO per
// (source, destination) pair; collectives meet at a barrier and reduce through the host.  Same call sequence as RCCL.
struct KaLoopback {
        int world = 1;
        std::mutex m;
        std::condition_variable cv;
        struct Msg { const void* ptr; size_t bytes; bool taken; };
        std::map<std::pair<int, int>, std::deque<Msg*>> box;
        std::vector<void*> bufs;
        int arrived = 0;
        long long gen = 0;
        void barrier(std::unique_lock<std::mutex>& lk)
        {
                const long long g = gen;
                if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
                else cv.wait(lk, [&] { return gen != g; });
        }
};
}  // namespace

// A child profile that changes GPUs above the cut
struct KaMove {
        int task, child, src, dst;
        int nmem = 0;                  // default mode: the child's member sequences ...
        long long ncols = 0;           // ... and the ints of their residue -> column tables
        DevBuf<int> d_members; DevBuf<long long> d_moff;
};
struct ka_dist {
        ka_ctx* c = nullptr;
        int rank = 0, world = 1;
        ncclComm_t comm = nullptr;
        KaLoopback* loop = nullptr;                      // tests: threads of one process instead of RCCL
        bool planned = false;
        // ---- the five transport operations (RCCL on the context's stream, or the loopback) ----
        int all_reduce_i32(int* buf, size_t count, bool take_max);
        int broadcast_i32(int* buf, size_t count, int root);
        int send(const void* buf, size_t bytes, int peer);
        int recv(void* buf, size_t bytes, int peer);
        int group_start() { if (loop || world == 1) return KA_OK; ncclResult_t r = g_rccl.GroupStart(); return r == ncclSuccess ? KA_OK : fail(std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r)); }
        int group_end() { if (loop || world == 1) return KA_OK; ncclResult_t r = g_rccl.GroupEnd(); return r == ncclSuccess ? KA_OK : fail(std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r)); }
        std::vector<int> run_rank, top, mine_sub;        // rank of every task; the tasks above the cut (tree order); this rank's subtree tasks
        std::vector<KaMove> moves;                       // in the order the top tasks need them
        std::vector<std::vector<int>> top_moves;         // per top task: indices into moves
        std::vector<DevBuf<int2>> top_blocks;            // per top task this rank runs: its workgroup table
        DevBuf<char> d_mine; DevBuf<int> d_counts, d_gpaths, d_colbuf, d_hdr, d_status; DevBuf<long long> d_goff;
        DevBuf<float> d_sink;                            // an incoming profile the arena has no room for (the step is then repeated)
        int retries = 0;                                 // steps repeated after an arena overflow on some rank
        std::vector<char> mine;
        std::vector<ka_task_rec> h_recs;
        std::vector<int> h_paths;
        std::vector<long long> goff;
        int* h_head = nullptr;                           // pinned: the two-int header of an incoming profile
        double last_ms = 0.0, last_kernel_wait_ms = 0.0;
};

int ka_dist::all_reduce_i32(int* buf, size_t count, bool take_max)
{
        if (world == 1 && !comm) return KA_OK;
        if (!loop) { NCCLCHK(g_rccl.AllReduce(buf, buf, count, ncclInt32, take_max ? ncclMax : ncclSum, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        std::unique_lock<std::mutex> lk(loop->m);
        loop->bufs[rank] = buf;
        loop->barrier(lk);
        std::vector<int> acc(count, take_max ? INT_MIN : 0), tmp(count);
        for (int r = 0; r < world; r++) {
                HIPCHK(hipMemcpy(tmp.data(), loop->bufs[r], sizeof(int) * count, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < count; i++) acc[i] = take_max ? std::max(acc[i], tmp[i]) : acc[i] + tmp[i];
        }
        loop->barrier(lk);                                          // everybody has read every buffer
        HIPCHK(hipMemcpyAsync(buf, acc.data(), sizeof(int) * count, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return KA_OK;
}
int ka_dist::broadcast_i32(int* buf, size_t count, int root)
{
        if (world == 1 && !comm) return KA_OK;
        if (!loop) { NCCLCHK(g_rccl.Broadcast(buf, buf, count, ncclInt32, root, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        std::unique_lock<std::mutex> lk(loop->m);
        loop->bufs[rank] = buf;
        loop->barrier(lk);
        // (on this rank's stream and complete before anybody moves on: a plain hipMemcpy between device buffers is ordered in the null
        // stream only and need not have happened when it returns -- the rank's own, non-blocking stream would read stale words)
        if (rank != root) {
                HIPCHK(hipMemcpyAsync(buf, loop->bufs[root], sizeof(int) * count, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
        }
        loop->barrier(lk);
        return KA_OK;
}
int ka_dist::send(const void* buf, size_t bytes, int peer)
{
        if (!loop) { NCCLCHK(g_rccl.Send(buf, bytes, ncclInt8, peer, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        KaLoopback::Msg msg = { buf, bytes, false };
        std::unique_lock<std::mutex> lk(loop->m);
        loop->box[std::make_pair(rank, peer)].push_back(&msg);
        loop->cv.notify_all();
        loop->cv.wait(lk, [&] { return msg.taken; });
        return KA_OK;
}
int ka_dist::recv(void* buf, size_t bytes, int peer)
{
        if (!loop) { NCCLCHK(g_rccl.Recv(buf, bytes, ncclInt8, peer, comm, c->stream)); return KA_OK; }
        HIPCHK(hipStreamSynchronize(c->stream));
        std::unique_lock<std::mutex> lk(loop->m);
        auto& q = loop->box[std::make_pair(peer, rank)];
        loop->cv.wait(lk, [&] { return !q.empty(); });
        KaLoopback::Msg* msg = q.front();
        q.pop_front();
        if (msg->bytes != bytes) { msg->taken = true; loop->cv.notify_all(); return fail("ka_dist loopback: message size mismatch"); }
        hipError_t e = hipMemcpyAsync(buf, msg->ptr, bytes, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // (the sender reuses its buffer once the message is taken)
        msg->taken = true;
        loop->cv.notify_all();
        if (e != hipSuccess) return fail(std::string("ka_dist loopback: ") + hipGetErrorString(e));
        return KA_OK;
}

// Tests: a loopback "communicator" for `world` ranks living in one process (threads), each with its own context.
extern "C" void* ka_dist_loopback_new(int world)
{
        if (world < 1) return nullptr;
        KaLoopback* l = new KaLoopback();
        l->world = world;
        l->bufs.assign(world, nullptr);
        return l;
}
extern "C" void ka_dist_loopback_free(void* l) { delete (KaLoopback*)l; }
extern "C" int ka_dist_create_loopback(ka_ctx* c, int rank, int world, void* loopback, ka_dist** out)
{
        if (!c || !out || !loopback || world < 1 || rank < 0 || rank >= world || ((KaLoopback*)loopback)->world != world) return fail("ka_dist_create_loopback: bad arguments");
        HIPCHK(hipSetDevice(c->device));
        ka_dist* d = new ka_dist();
        d->c = c; d->rank = rank; d->world = world; d->loop = (KaLoopback*)loopback;
        if (hipHostMalloc((void**)&d->h_head, 64, hipHostMallocDefault) != hipSuccess) { delete d; return fail("hipHostMalloc failed"); }
        *out = d;
        return KA_OK;
}

// Pure planning (no device, no communicator): cut the tree into at most `world` subtrees balanced by estimated DP cells;
// run_rank[t] = the rank that runs task t, top[0 .. *n_top) = the tasks above the cut in tree order.  Every rank derives
// the same plan from the same inputs.
extern "C" int ka_dist_plan_subtrees(int numseq, const int* lens, int n_tasks, const int* abc, int world, int* run_rank, int* top, int* n_top)
{
        if (numseq < 2 || n_tasks < 1 || world < 1 || !lens || !abc || !run_rank || !top || !n_top) return fail("ka_dist_plan_subtrees: bad arguments");
        const int nprof = 2 * numseq - 1;
        std::vector<int> task_of(nprof, -1), members(nprof, 0);
        std::vector<double> est(nprof, 0.0), work(nprof, 0.0);
        for (int i = 0; i < numseq; i++) { est[i] = lens[i]; members[i] = 1; }
        for (int t = 0; t < n_tasks; t++) {
                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                if (a < 0 || b < 0 || cc < numseq || a >= nprof || b >= nprof || cc >= nprof) return fail("ka_dist_plan_subtrees: bad task list");
                task_of[cc] = t;
                est[cc] = 1.05 * std::max(est[a], est[b]);
                work[cc] = work[a] + work[b] + est[a] * est[b];
                members[cc] = members[a] + members[b];
        }
        const int root = abc[3 * (n_tasks - 1) + 2];
        std::vector<int> frontier(1, root), tops;
        auto internal = [&](const std::vector<int>& f) { int n = 0; for (int x : f) n += x >= numseq; return n; };
        while (internal(frontier) < world) {
                int best = -1;
                for (int x : frontier) if (x >= numseq && (best < 0 || work[x] > work[best] || (work[x] == work[best] && x < best))) best = x;
                if (best < 0) break;
                const int t = task_of[best];
                const int kids = (abc[3 * t] >= numseq) + (abc[3 * t + 1] >= numseq);
                if (kids == 0) break;                            // splitting would not add a subtree (both children are leaves)
                frontier.erase(std::find(frontier.begin(), frontier.end(), best));
                frontier.push_back(abc[3 * t]); frontier.push_back(abc[3 * t + 1]);
                tops.push_back(t);
        }
        std::sort(tops.begin(), tops.end());
        std::vector<int> roots;
        for (int x : frontier) if (x >= numseq) roots.push_back(x);
        std::sort(roots.begin(), roots.end(), [&](int x, int y) { return work[x] > work[y] || (work[x] == work[y] && x < y); });
        for (int t = 0; t < n_tasks; t++) run_rank[t] = -1;
        std::vector<int> holder(nprof, -1);
        for (size_t r = 0; r < roots.size(); r++) {
                std::vector<int> stack(1, roots[r]);
                while (!stack.empty()) {
                        const int v = stack.back(); stack.pop_back();
                        if (v < numseq) continue;
                        const int t = task_of[v];
                        run_rank[t] = (int)(r % world);
                        stack.push_back(abc[3 * t]); stack.push_back(abc[3 * t + 1]);
                }
                holder[roots[r]] = (int)(r % world);
        }
        for (int t : tops) {                                     // tree order: children first
                const int a = abc[3 * t], b = abc[3 * t + 1], cc = abc[3 * t + 2];
                const int ha = holder[a], hb = holder[b];
                int r;
                if (ha < 0 && hb < 0) r = 0;
                else if (ha < 0 || (hb >= 0 && members[b] > members[a])) r = hb;
                else r = ha;
                run_rank[t] = r;
                holder[cc] = r;
        }
        for (size_t i = 0; i < tops.size(); i++) top[i] = tops[i];
        *n_top = (int)tops.size();
        return KA_OK;
}

extern "C" int ka_dist_unique_id(void* id128)
{
        if (!id128) return fail("null id");
        if (rccl_load()) return KA_FAIL;
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        ncclUniqueId id;
        NCCLCHK(g_rccl.GetUniqueId(&id));
        memcpy(id128, &id, sizeof(id));
        return KA_OK;
}

extern "C" void ka_dist_destroy(ka_dist* d)
{
        if (!d) return;
        if (d->c) (void)hipSetDevice(d->c->device);
        if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
        for (auto& m : d->moves) { m.d_members.release(); m.d_moff.release(); }
        for (auto& b : d->top_blocks) b.release();
        d->d_mine.release(); d->d_counts.release(); d->d_gpaths.release(); d->d_colbuf.release(); d->d_goff.release();
        if (d->h_head) (void)hipHostFree(d->h_head);
        delete d;
}

// id128: the 128 bytes rank 0 got from ka_dist_unique_id, handed to every rank by the launcher (a file, MPI, torch...).
// world == 1: no communicator is made (every step degenerates to the local run) -- the code path is the same.
extern "C" int ka_dist_create(ka_ctx* c, int rank, int world, const void* id128, ka_dist** out)
{
        if (!c || !out || world < 1 || rank < 0 || rank >= world) return fail("ka_dist_create: bad arguments");
        HIPCHK(hipSetDevice(c->device));
        ka_dist* d = new ka_dist();
        d->c = c; d->rank = rank; d->world = world;
        if (hipHostMalloc((void**)&d->h_head, 64, hipHostMallocDefault) != hipSuccess) { delete d; return fail("hipHostMalloc failed"); }
        if (world > 1 || id128) {
                if (!id128 || rccl_load()) { ka_dist_destroy(d); return id128 ? KA_FAIL : fail("ka_dist_create: a world of several ranks needs the unique id"); }
                ncclUniqueId id;
                memcpy(&id, id128, sizeof(id));
                ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank);
                if (r != ncclSuccess) { const std::string m = g_rccl.GetErrorString(r); ka_dist_destroy(d); return fail("ncclCommInitRank: " + m); }
        }
        *out = d;
        return KA_OK;
}

// Once per uploaded job: the cut, who runs what, the hand-overs above the cut, this rank's subtrees planned as one run.
extern "C" int ka_dist_plan(ka_dist* d)
{
        if (!d || !d->c || !d->c->have_job) return fail("ka_dist_plan: no uploaded job");
        ka_ctx* c = d->c;
        HIPCHK(hipSetDevice(c->device));
        if (c->n_tasks != c->numseq - 1) return fail("ka_dist_plan: one guide tree per job");
        const int n_tasks = c->n_tasks, numseq = c->numseq;
        d->run_rank.assign(n_tasks, -1);
        d->top.assign(n_tasks, 0);
        int n_top = 0;
        if (ka_dist_plan_subtrees(numseq, c->lens.data(), n_tasks, c->abc.data(), d->world, d->run_rank.data(), d->top.data(), &n_top)) return KA_FAIL;
        d->top.resize(n_top);
        std::vector<char> is_top(n_tasks, 0);
        for (int t : d->top) is_top[t] = 1;
        d->mine_sub.clear();
        d->mine.assign(n_tasks, 0);
        for (int t = 0; t < n_tasks; t++) {
                if (d->run_rank[t] == d->rank) d->mine[t] = 1;
                if (d->run_rank[t] == d->rank && !is_top[t]) d->mine_sub.push_back(t);
        }
        // the hand-overs: a child of a top task that sits on another rank than the one running the parent
        for (auto& m : d->moves) { m.d_members.release(); m.d_moff.release(); }
        for (auto& b : d->top_blocks) b.release();
        d->moves.clear(); d->top_moves.assign(n_top, std::vector<int>()); d->top_blocks.clear(); d->top_blocks.resize(n_top);
        std::vector<int> holder(2 * numseq - 1, -1);
        for (int t = 0; t < n_tasks; t++) if (!is_top[t]) holder[c->abc[3 * t + 2]] = d->run_rank[t];
        long long max_cols = 0;
        for (int i = 0; i < n_top; i++) {
                const int t = d->top[i], dst = d->run_rank[t];
                for (int k = 0; k < 2; k++) {
                        const int child = c->abc[3 * t + k];
                        const int src = child >= numseq ? holder[child] : -1;
                        if (child < numseq || src < 0 || src == dst) continue;
                        d->moves.emplace_back();
                        KaMove& m = d->moves.back();
                        m.task = t; m.child = child; m.src = src; m.dst = dst;
                        if (d->rank == src || d->rank == dst) {
                                long long lo, hi;
                                node_members(c, child, &lo, &hi);
                                std::vector<int> mem(c->sip_flat.begin() + lo, c->sip_flat.begin() + hi);
                                std::vector<long long> off(mem.size());
                                long long o = 0;
                                for (size_t q = 0; q < mem.size(); q++) { off[q] = o; o += c->lens[mem[q]]; }
                                m.nmem = (int)mem.size(); m.ncols = o;
                                max_cols = std::max(max_cols, o);
                                if (m.d_members.alloc(mem.size()) || m.d_moff.alloc(off.size())) return fail("hipMalloc failed");
                                HIPCHK(hipMemcpy(m.d_members.p, mem.data(), sizeof(int) * mem.size(), hipMemcpyHostToDevice));
                                HIPCHK(hipMemcpy(m.d_moff.p, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice));
                        }
                        d->top_moves[i].push_back((int)d->moves.size() - 1);
                }
                holder[c->abc[3 * t + 2]] = dst;
                if (dst == d->rank) {
                        std::vector<int2> tbl;
                        int lean = 0;
                        build_blocks(c, std::vector<int>(1, t), tbl, &lean);
                        if (d->top_blocks[i].alloc(tbl.size())) return fail("hipMalloc failed");
                        HIPCHK(hipMemcpy(d->top_blocks[i].p, tbl.data(), sizeof(int2) * tbl.size(), hipMemcpyHostToDevice));
                }
        }
        if (d->d_mine.alloc(n_tasks) || d->d_counts.alloc(n_tasks) || d->d_goff.alloc(n_tasks) || d->d_colbuf.alloc((size_t)std::max<long long>(max_cols, 1)))
                return fail("hipMalloc failed");
        HIPCHK(hipMemcpy(d->d_mine.p, d->mine.data(), n_tasks, hipMemcpyHostToDevice));
        // this rank's subtrees as ONE planned run (queued / chained launches where they apply)
        c->plan_active.assign(n_tasks, 0);
        for (int t : d->mine_sub) c->plan_active[t] = 1;
        if (d->mine_sub.empty()) c->plan_active.assign(n_tasks, 0);
        // (an all-zero mask is a plan over nothing: every level empty)
        if (plan_launches(c) || upload_plan(c)) return KA_FAIL;
        d->planned = true;
        return KA_OK;
}

extern "C" int ka_dist_get_plan(ka_dist* d, int* run_rank, int* top, int* n_top, int* n_moves)
{
        if (!d || !d->planned) return fail("ka_dist_get_plan: plan first");
        if (run_rank) memcpy(run_rank, d->run_rank.data(), sizeof(int) * d->run_rank.size());
        if (top) memcpy(top, d->top.data(), sizeof(int) * d->top.size());
        if (n_top) *n_top = (int)d->top.size();
        if (n_moves) *n_moves = (int)d->moves.size();
        return KA_OK;
}

// anchor_consistency_build over the ranks: this rank's share of the N x K batch, then every share broadcast in place.
extern "C" int ka_dist_consistency(ka_dist* d, int n_anchors, float weight)
{
        if (!d || !d->c) return fail("ka_dist_consistency: null");
        ka_ctx* c = d->c;
        HIPCHK(hipSetDevice(c->device));
        // a part that cannot be built (e.g. it holds only anchors) must not leave the other ranks waiting in a collective:
        // every rank reduces the outcome first
        int rc = ka_tree_build_consistency_part(c, n_anchors, weight, d->rank, d->world);
        const std::string why = rc ? std::string(ka_last_error()) : std::string();
        if (d->world > 1 || d->comm) {
                int* flag = (int*)d->d_counts.p;
                if (!flag && d->d_counts.alloc(std::max(c->n_tasks, 1))) return fail("hipMalloc failed");
                flag = d->d_counts.p;
                const int mine = rc ? 1 : 0;
                HIPCHK(hipMemcpyAsync(flag, &mine, sizeof(int), hipMemcpyHostToDevice, c->stream));
                if (d->all_reduce_i32(flag, 1, true)) return KA_FAIL;
                int any = 0;
                HIPCHK(hipMemcpyAsync(&any, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if (any) return fail(rc ? why : std::string("ka_dist_consistency: another rank could not build its part"));
                if (c->cons_K <= 0) return KA_OK;                    // the job declined on every rank alike (fewer than 3 sequences ...)
                if (d->group_start()) return KA_FAIL;
                for (int r = 0; r < d->world; r++) {
                        long long lo = 0, hi = 0;
                        if (ka_tree_consistency_part_range(c, r, d->world, &lo, &hi)) { (void)d->group_end(); return KA_FAIL; }
                        if (hi > lo && d->broadcast_i32(c->d_cons_maps.p + lo, (size_t)(hi - lo), r)) { (void)d->group_end(); return KA_FAIL; }
                }
                if (d->group_end()) return KA_FAIL;
        } else if (rc) return KA_FAIL;
        return KA_OK;
}

// One attempt at a step of the sharded tree, up to the point where this rank knows how ITS part went.  Conditions a repeat can
// cure -- a device arena overflowed in one of this rank's kernels, the profile arena has no room for an incoming profile, a
// profile this rank should send was never made because an earlier task failed -- do NOT leave the protocol: the rank keeps
// matching every send / receive of the walk (an unusable profile travels as a header of zero and nothing else; a profile
// without room lands in a sink buffer), stops launching, and reports through *status (0 clean, 1 repeat after growing, 2 fatal).
// Only HIP / RCCL API failures return KA_FAIL from inside (nothing sensible can be agreed on a broken device).
static int dist_tree_attempt(ka_dist* d, int* status, int* grow)
{
        ka_ctx* c = d->c;
        const int n_tasks = c->n_tasks;
        *status = 0; *grow = 0;
        if (c->plan_active.empty()) return fail("ka_dist_tree_run: the context's plan was replaced by a whole-tree run; call ka_dist_plan again");
        c->ran = false; c->synced = false;
        if (tree_reset(c)) return KA_FAIL;                          // (keeps the consistency table; residue -> column tables back to the leaves)
        if (tree_launch(c, false)) return KA_FAIL;                  // this rank's subtrees
        const KaTreeDev D = tree_dev(c);
        if (d->d_hdr.alloc(4)) return fail("hipMalloc failed");
        bool stop = false;                                           // something went wrong on this rank: no more launches
        for (size_t i = 0; i < d->top.size(); i++) {
                const int t = d->top[i], dst = d->run_rank[t];
                for (int mi : d->top_moves[i]) {
                        KaMove& m = d->moves[mi];
                        if (d->rank != m.src && d->rank != m.dst) continue;
                        const bool cols = c->have_colof && m.ncols > 0 && (c->cons_K > 0 || (c->flags & KA_FLAG_DEVICE_GAPS));
                        if (d->rank == m.src) {
                                // header (plen) straight from the node table, the records from where they lie in the arena
                                HIPCHK(hipMemcpyAsync(d->h_head, c->d_node_len.p + m.child, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipMemcpyAsync(d->h_head + 2, c->d_node_prof.p + m.child, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipMemcpyAsync(d->h_head + 8, c->d_error.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipStreamSynchronize(c->stream));
                                int plen = d->h_head[0];
                                long long po; memcpy(&po, d->h_head + 2, sizeof(po));
                                const bool missing = d->h_head[8] == 0 && !stop && (plen < 1 || po < 0);
                                if (stop || d->h_head[8] != 0 || plen < 1 || po < 0) { plen = 0; stop = true; }   // (its status comes from the device error word below)
                                // a profile that was never made although no kernel of this rank reported anything (a bug, not an arena to grow):
                                // this rank says so -- the receiver reports "repeat", and repeating cannot cure it
                                if (missing) { *status = 2; fail("sharded tree: the profile of node " + std::to_string(m.child) + " was never made on the rank that owns it"); }
                                d->h_head[12] = plen;
                                HIPCHK(hipMemcpyAsync(d->d_hdr.p, d->h_head + 12, sizeof(int), hipMemcpyHostToDevice, c->stream));
                                // (plain stream-ordered point-to-point operations, matched in order with the receiver's: every rank
                                // walks the hand-overs in the same order, so no two ranks ever wait for each other crosswise)
                                if (d->send(d->d_hdr.p, sizeof(int), m.dst)) return KA_FAIL;
                                if (plen > 0) {
                                        if (cols) ka_launch_cols_pack(c->d_colof.p, c->d_seq_off.p, c->d_node_len.p, m.d_members.p, m.d_moff.p, m.nmem, d->d_colbuf.p, 0, c->stream);
                                        if (d->send(c->d_prof_arena.p + po, sizeof(float) * (size_t)(plen + 2) * KA_REC, m.dst)) return KA_FAIL;
                                        if (cols && d->send(d->d_colbuf.p, sizeof(int) * (size_t)m.ncols, m.dst)) return KA_FAIL;
                                }
                                HIPCHK(hipStreamSynchronize(c->stream));          // (h_head and d_hdr are reused by the next hand-over)
                        } else {
                                // the header first: it sizes the room the records get in this rank's arena
                                if (d->recv(d->d_hdr.p, sizeof(int), m.src)) return KA_FAIL;
                                HIPCHK(hipMemcpyAsync(d->h_head, d->d_hdr.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipMemcpyAsync(d->h_head + 2, c->d_counters.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
                                HIPCHK(hipStreamSynchronize(c->stream));
                                const int plen = d->h_head[0];
                                if (plen < 1) { stop = true; if (*status < 1) *status = 1; continue; }   // the sender has nothing to send: its own status says why
                                unsigned long long top_; memcpy(&top_, d->h_head + 2, sizeof(top_));
                                const unsigned long long need = (unsigned long long)(plen + 2) * KA_REC;
                                if ((long long)(top_ + need) > c->prof_cap) {
                                        // no room: take the payload off the wire all the same, then ask for a repeat with a bigger arena
                                        if (d->d_sink.alloc((size_t)need)) return fail("hipMalloc failed");
                                        if (d->recv(d->d_sink.p, sizeof(float) * (size_t)need, m.src)) return KA_FAIL;
                                        if (cols && d->recv(d->d_colbuf.p, sizeof(int) * (size_t)m.ncols, m.src)) return KA_FAIL;
                                        HIPCHK(hipStreamSynchronize(c->stream));
                                        stop = true; *status = std::max(*status, 1); *grow |= 1;
                                        continue;
                                }
                                const long long po = (long long)top_;
                                top_ += need;
                                memcpy(d->h_head + 4, &top_, sizeof(top_)); memcpy(d->h_head + 6, &po, sizeof(po));
                                HIPCHK(hipMemcpyAsync(c->d_counters.p, d->h_head + 4, sizeof(top_), hipMemcpyHostToDevice, c->stream));
                                HIPCHK(hipMemcpyAsync(c->d_node_prof.p + m.child, d->h_head + 6, sizeof(po), hipMemcpyHostToDevice, c->stream));
                                HIPCHK(hipMemsetAsync(c->d_node_vote.p + m.child, 0xff, sizeof(long long), c->stream));   // (no carried vote table comes with it)
                                HIPCHK(hipMemcpyAsync(c->d_node_len.p + m.child, d->h_head, sizeof(int), hipMemcpyHostToDevice, c->stream));
                                if (d->recv(c->d_prof_arena.p + po, sizeof(float) * (size_t)need, m.src)) return KA_FAIL;
                                if (cols && d->recv(d->d_colbuf.p, sizeof(int) * (size_t)m.ncols, m.src)) return KA_FAIL;
                                if (cols) ka_launch_cols_pack(c->d_colof.p, c->d_seq_off.p, c->d_node_len.p, m.d_members.p, m.d_moff.p, m.nmem, d->d_colbuf.p, 1, c->stream);
                                HIPCHK(hipStreamSynchronize(c->stream));          // (h_head is reused by the next hand-over)
                                c->injected.push_back(m.child);
                        }
                }
                if (dst == d->rank && !stop) {
                        HIPCHK(hipMemsetAsync(c->d_counters.p + 1, 0, sizeof(unsigned long long), c->stream));
                        ka_launch_task_level(&D, d->top_blocks[i].p, (int)d->top_blocks[i].n, 0, 0, c->stream);
                        c->n_launches++;
                        c->task_done[t] = 1;
                }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        c->ran = true; c->partial = true;
        // how this rank's kernels went (ka_tree_sync would turn an overflow of a partial run into a failure of this rank alone)
        int err = 0;
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(&err, c->d_error.p, sizeof(int), hipMemcpyDeviceToHost));
        if (err >= 1 && err <= 4) { *status = std::max(*status, 1); *grow |= (err == 1) ? 1 : (err == 2 ? 2 : (err == 3 ? 4 : 8)); }
        else if (err != 0) {
                // (the device's error words, ka_device.h: 5 / 6 the two watchdogs, 7 the vote table of a profile that outgrew its LDS slot)
                *status = 2;
                fail(err == 5 ? "device watchdog: a strip pipeline stopped making progress"
                   : err == 6 ? "device watchdog: a wait between workgroups never completed"
                   : err == 7 ? "anchor consistency: a vote table did not fit its LDS slot"
                   : "device error " + std::to_string(err));
        }
        (void)n_tasks;
        return KA_OK;
}

// One step of the sharded tree: from the leaves to every rank holding every record and coded path.  Every rank learns how
// every other rank's part went BEFORE the collectives of the gather (all-reduce of the status: a rank that failed alone would
// leave the others waiting in RCCL); an arena overflow anywhere makes every rank repeat the step, the ranks that overflowed
// with bigger arenas -- what ka_tree_sync does for a single GPU.
extern "C" int ka_dist_tree_run(ka_dist* d)
{
        if (!d || !d->planned) return fail("ka_dist_tree_run: plan first");
        ka_ctx* c = d->c;
        HIPCHK(hipSetDevice(c->device));
        const auto t_begin = std::chrono::steady_clock::now();
        const int n_tasks = c->n_tasks;
        if (d->d_status.alloc(1)) return fail("hipMalloc failed");
        for (int attempt = 0; ; attempt++) {
                int status = 0, grow = 0;
                if (dist_tree_attempt(d, &status, &grow)) return KA_FAIL;
                const std::string why = ka_last_error();
                int agreed = status;
                if (d->world > 1 || d->comm) {
                        HIPCHK(hipMemcpyAsync(d->d_status.p, &status, sizeof(int), hipMemcpyHostToDevice, c->stream));
                        if (d->all_reduce_i32(d->d_status.p, 1, true)) return KA_FAIL;
                        HIPCHK(hipMemcpyAsync(&agreed, d->d_status.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                        HIPCHK(hipStreamSynchronize(c->stream));
                }
                if (agreed == 0) break;
                if (agreed >= 2) return fail(status >= 2 ? why : std::string("ka_dist_tree_run: another rank's part of the step failed"));
                if (attempt >= 24) return fail("ka_dist_tree_run: device arenas kept overflowing");
                d->retries++;
                if (grow & 1) { c->prof_cap *= 2; c->path_cap *= 2; c->d_prof_arena.release(); c->d_path_arena.release(); }
                if (grow & 2) { c->scratch_cap *= 2; c->d_scratch.release(); }
                if (grow & 4) { c->path_cap *= 2; c->d_path_arena.release(); }
                if (grow & 8) { c->dbg_cap *= 2; c->d_dbg_arena.release(); }
                if (c->d_prof_arena.alloc((size_t)c->prof_cap) || c->d_path_arena.alloc((size_t)c->path_cap) ||
                    c->d_scratch.alloc((size_t)c->scratch_cap) || c->d_dbg_arena.alloc((size_t)std::max<long long>(c->dbg_cap, 1)))
                        return fail("hipMalloc failed while growing an arena");
        }
        // ---- every rank ends with every record and every coded path ----
        if (ka_tree_sync(c)) return KA_FAIL;                        // (clean on every rank: reads the counters)
        ka_launch_path_counts(c->d_recs.p, d->d_mine.p, n_tasks, d->d_counts.p, c->stream);
        if (d->all_reduce_i32(d->d_counts.p, (size_t)n_tasks, false)) return KA_FAIL;
        std::vector<int> counts(n_tasks);
        HIPCHK(hipMemcpyAsync(counts.data(), d->d_counts.p, sizeof(int) * n_tasks, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        d->goff.assign(n_tasks, 0);
        long long total = 0;
        for (int t = 0; t < n_tasks; t++) { d->goff[t] = total; total += counts[t]; }
        if (d->d_gpaths.alloc((size_t)std::max<long long>(total, 1))) return fail("hipMalloc failed");
        HIPCHK(hipMemcpyAsync(d->d_goff.p, d->goff.data(), sizeof(long long) * n_tasks, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(d->d_gpaths.p, 0, sizeof(int) * (size_t)total, c->stream));
        ka_launch_path_scatter(c->d_recs.p, d->d_mine.p, n_tasks, c->d_path_arena.p, d->d_goff.p, d->d_gpaths.p, c->stream);
        static_assert(sizeof(ka_task_rec) % 4 == 0, "records are reduced as 32-bit words");
        if (d->group_start()) return KA_FAIL;
        if (d->all_reduce_i32(d->d_gpaths.p, (size_t)total, false)) return KA_FAIL;
        if (d->all_reduce_i32((int*)c->d_recs.p, (size_t)n_tasks * (sizeof(ka_task_rec) / 4), false)) return KA_FAIL;
        if (d->group_end()) return KA_FAIL;
        d->h_recs.resize(n_tasks);
        d->h_paths.resize((size_t)total);
        HIPCHK(hipMemcpyAsync(d->h_recs.data(), c->d_recs.p, sizeof(ka_task_rec) * n_tasks, hipMemcpyDeviceToHost, c->stream));
        if (copy_to_host(c, d->h_paths.data(), d->d_gpaths.p, sizeof(int) * (size_t)total)) return KA_FAIL;
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int t = 0; t < n_tasks; t++) d->h_recs[t].path_off = (int)d->goff[t];
        d->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return KA_OK;
}

// How many times ka_dist_tree_run repeated a step on this rank because an arena overflowed somewhere (tests, reports)
extern "C" int ka_dist_retries(ka_dist* d) { return d ? d->retries : -1; }

// Records (task order, path_off into paths) and coded paths of the last ka_dist_tree_run; *used = ints written.
extern "C" int ka_dist_download(ka_dist* d, ka_task_rec* recs, int* paths, long long paths_cap, long long* used)
{
        if (!d || d->h_recs.empty()) return fail("ka_dist_download: run first");
        if ((long long)d->h_paths.size() > paths_cap) { fail("paths_out too small"); return KA_ERR_PATHS_CAP; }
        memcpy(recs, d->h_recs.data(), sizeof(ka_task_rec) * d->h_recs.size());
        memcpy(paths, d->h_paths.data(), sizeof(int) * d->h_paths.size());
        if (used) *used = (long long)d->h_paths.size();
        return KA_OK;
}

extern "C" long long ka_dist_paths_size(ka_dist* d) { return d ? (long long)d->h_paths.size() : -1; }
extern "C" double ka_dist_last_ms(ka_dist* d) { return d ? d->last_ms : -1.0; }

