"""The second tree of a `--precise` member (kalign_run_realign: UPGMA on the identity distances of the first alignment's rows,
aln_wrap.c:449-504) -- an unbalanced tree, whose dependency chain decides its time.  Per-task timing (KA_FLAG_TIMING) along that
chain, by task kind and phase.  python tools/realign_levels.py [NSEQ 2048] [LEN 300] [ANCHORS 5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd
from kalign_amd import api, guide
NSEQ = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
LEN = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
inp = bench.workload_letters(NSEQ, LEN, False, 5)
order = sorted(range(len(inp)), key=lambda i: (-len(inp[i]), i))
seqs = [inp[i] for i in order]
tcodes, codes = guide.encode_tree(seqs, dna=False), guide.encode(seqs, dna=False)
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
tasks, sd = ctx.guide_tree(tcodes, n_threads=16)
ctx.msa_tree(codes, tasks, subm, scal, sd, n_anchors=K, weight=2.0)
print('first tree: kernel ms', ctx.tree_kernel_ms())
ctx.tree_aligned_rows(seqs)
tasks2, sd2 = ctx.aln_guide_tree()
ctx.tree_upload(codes, tasks2, subm, scal, sd2, flags=api.FLAG_DEVICE_GAPS | api.FLAG_KEEP_CONSISTENCY | api.FLAG_TIMING)
for _ in range(2):
    ctx.tree_run(); ctx.tree_sync()
print('second tree: kernel ms', ctx.tree_kernel_ms(), 'launches', ctx.tree_launches() if hasattr(ctx, 'tree_launches') else '')
recs, _, _ = ctx.tree_download(want_gaps=False)
tm = ctx.tree_timing()
GHZ = 2.4
tot = tm[:, :4].sum(1) / GHZ / 1e3
n = len(seqs)
done, lvl = {i: 0.0 for i in range(n)}, {i: 0 for i in range(n)}
for r, t in zip(recs, tot):
    done[r.c] = max(done[r.a], done[r.b]) + t
    lvl[r.c] = 1 + max(lvl[r.a], lvl[r.b])
by_c = {r.c: (r, t, x) for r, t, x in zip(recs, tot, tm)}
print('tree depth %d, tasks %d by kind %s, summed task time %.1f ms, dependency chain %.1f ms' % (
    lvl[recs[-1].c], len(recs), np.bincount([r.kind for r in recs], minlength=3), tot.sum() / 1e3, done[recs[-1].c] / 1e3))
node, crit = recs[-1].c, []
while node in by_c:
    crit.append(node)
    r = by_c[node][0]
    node = r.a if done[r.a] >= done[r.b] else r.b
ph = np.zeros((3, 6)); cnt = np.zeros(3, int)
for nd in crit:
    r, t, x = by_c[nd]
    ph[r.kind, :4] += x[:4] / GHZ / 1e3; ph[r.kind, 4] += x[4] / GHZ / 1e3; ph[r.kind, 5] += x[5] / GHZ / 1e3
    cnt[r.kind] += 1
print('chain by kind: tasks, us total, (prep, hirschberg [passes, meetups], code, merge)')
for k, name in enumerate(('seq-seq', 'seq-profile', 'profile-profile')):
    if cnt[k]:
        print('  %-16s %3d  %8.0f   prep %7.0f  hirsch %7.0f [%7.0f %6.0f]  code %6.0f  merge %7.0f' % (name, cnt[k], ph[k, :4].sum(), ph[k, 0], ph[k, 1], ph[k, 4], ph[k, 5], ph[k, 2], ph[k, 3]))
print('chain, root first (every 4th): node lens nsip kind total_us prep hirsch code merge  G')
for nd in crit[::4]:
    r, t, x = by_c[nd]
    print('  %6d %5dx%-5d %4d+%-4d k%d %7.0f  %5.0f %6.0f %4.0f %5.0f  G %d of %d' % (nd, r.len_a, r.len_b, r.nsip_a, r.nsip_b, r.kind, t, x[0]/GHZ/1e3, x[1]/GHZ/1e3, x[2]/GHZ/1e3, x[3]/GHZ/1e3,
                                                                                     (int(x[6]) >> 8) & 255, int(x[6]) >> 16))
