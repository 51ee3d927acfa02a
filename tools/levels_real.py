"""Per-level task timing (KA_FLAG_TIMING) of the task tree on the reference's own k-means guide tree;
arguments: [anchors (default 5, 0 = --fast)] [nseq 1024] [len 400] [dna 0] [bal = balanced synthetic tree].  Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd, torch
torch.cuda.init()
from kalign_amd import api, guide, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NSEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
LEN = int(sys.argv[3]) if len(sys.argv) > 3 else 400
DNA = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
seqs = synth.dssim(NSEQ, LEN, dna=DNA, seed=1)
order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
seqs = [seqs[i] for i in order]
tcodes = guide.encode_tree(seqs, dna=DNA); codes = guide.encode(seqs, dna=DNA)
subm, scal = bench.scoring(DNA)
ctx = kalign_amd.Context(0)
tasks, sd = ctx.guide_tree(tcodes, n_threads=16)
if len(sys.argv) > 5 and sys.argv[5] == 'bal':          # balanced synthetic tree instead (profile-profile at every level)
    tasks = guide.bisecting_tree(NSEQ, seed=1, jitter=0.0)
ctx.tree_upload(codes, tasks, subm, scal, sd, flags=api.FLAG_TIMING)
if K: ctx.tree_build_consistency(K, 2.0)
for _ in range(3): ctx.tree_run(); ctx.tree_sync()
recs, paths, _ = ctx.tree_download(want_gaps=False)
tm = ctx.tree_timing()
print('ms', ctx.tree_kernel_ms())
n = len(seqs)
lvl = {i: 0 for i in range(n)}
tl = []
for r in recs:
    l = 1 + max(lvl[r.a], lvl[r.b]); lvl[r.c] = l; tl.append(l)
tl = np.array(tl); kind = np.array([r.kind for r in recs])
GHZ = 2.4
tot = tm[:, :4].sum(1)
for l in range(1, tl.max()+1):
    m = tl == l
    i = np.argmax(np.where(m, tot, -1))
    print('L%2d n=%4d kinds=%s  max task %.0f us (prep %.0f hirsch %.0f [pass %.0f meet %.0f] code %.0f merge %.0f) lens %dx%d nsip %d+%d  mean %.0f us' % (
        l, m.sum(), np.bincount(kind[m], minlength=3), tot[i]/GHZ/1e3, tm[i,0]/GHZ/1e3, tm[i,1]/GHZ/1e3, tm[i,4]/GHZ/1e3, tm[i,5]/GHZ/1e3, tm[i,2]/GHZ/1e3, tm[i,3]/GHZ/1e3,
        recs[i].len_a, recs[i].len_b, recs[i].nsip_a, recs[i].nsip_b, tot[m].mean()/GHZ/1e3))
done = {i: 0.0 for i in range(n)}
for r, t in zip(recs, tot):
    done[r.c] = max(done[r.a], done[r.b]) + t/GHZ/1e3
lvmax = sum(max(tot[tl==l]) for l in range(1, tl.max()+1))/GHZ/1e3
print('sum of per-level max %.0f us; dependency-driven critical path %.0f us' % (lvmax, done[recs[-1].c]))

# the tasks on the dependency-driven critical path, root first
start = {}
for r, t in zip(recs, tot):
    start[r.c] = max(done[r.a], done[r.b])
by_c = {r.c: (r, t, tmr) for r, t, tmr in zip(recs, tot, tm)}
node = recs[-1].c
crit = []
print('critical path (root first): node lens nsip kind  start_us  total_us  prep hirsch[pass meet levels] code merge')
while node in by_c:
    r, t, x = by_c[node]
    print('  %6d %5dx%-5d %4d+%-4d k%d  %8.0f %7.0f   %5.0f %6.0f [%6.0f %5.0f %2d] %4.0f %5.0f  G %d of %d  L%d' % (
        node, r.len_a, r.len_b, r.nsip_a, r.nsip_b, r.kind, start[node], t/GHZ/1e3, x[0]/GHZ/1e3, x[1]/GHZ/1e3, x[4]/GHZ/1e3, x[5]/GHZ/1e3, int(x[6]) & 255, x[2]/GHZ/1e3, x[3]/GHZ/1e3,
        (int(x[6]) >> 8) & 255, int(x[6]) >> 16, lvl[node]))
    crit.append(node)
    node = r.a if done[r.a] >= done[r.b] else r.b
print('%s task, per Hirschberg level: sub-problems, pass us, meetup us' % (os.environ.get('KA_PROF_TASK', 'root')))
for l, (nsub, cp, cm) in enumerate(ctx.root_levels):
    if nsub and l < 13: print('  level %2d  n=%5d  pass %7.1f  meet %6.1f' % (l, nsub, cp/GHZ/1e3, cm/GHZ/1e3))   # (slots 13.. hold the subtree statistics)

# per-level times of more tasks on the critical path (KA_PROF_TASK): tree levels given as the 7th argument, e.g. 8,12,16
if len(sys.argv) > 6:
    idx_of = {r.c: i for i, r in enumerate(recs)}
    for want in [int(v) for v in sys.argv[6].split(',')]:
        nodes = [nd for nd in crit if lvl[nd] == want]
        if not nodes: continue
        os.environ['KA_PROF_TASK'] = str(idx_of[nodes[0]])
        ctx.reload_env()
        ctx.tree_run(); ctx.tree_sync(); ctx.tree_timing()
        r = recs[idx_of[nodes[0]]]
        print('task of node %d (tree level %d, %dx%d, nsip %d+%d, kind %d), per Hirschberg level:' % (nodes[0], want, r.len_a, r.len_b, r.nsip_a, r.nsip_b, r.kind))
        for l, (nsub, cp, cm) in enumerate(ctx.root_levels):
            if nsub and l < 13: print('  level %2d  n=%5d  pass %7.1f  meet %6.1f' % (l, nsub, cp/GHZ/1e3, cm/GHZ/1e3))
        st = ctx.root_levels.reshape(-1)[41:48]
        if st[0]:
            print('  wave-local subtrees of the leading workgroup: %d, mean us: staging %.1f passes %.1f meetups %.1f total %.1f; longest %.1f; mean levels/rows/cols code %.0f' % (
                st[0], st[1]/st[0]/GHZ/1e3, st[2]/st[0]/GHZ/1e3, st[3]/st[0]/GHZ/1e3, st[4]/st[0]/GHZ/1e3, st[5]/GHZ/1e3, st[6]/st[0]))
