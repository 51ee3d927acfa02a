"""Per-level task timing of the second alignment of a realignment pass (UPGMA tree), 2048 x 300 protein, --fast."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd, torch
torch.cuda.init()
from kalign_amd import api, guide, synth
n, L = 2048, 300
seqs = synth.dssim(n, L, seed=1)
order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
seqs = [seqs[i] for i in order]
tcodes = guide.encode_tree(seqs); codes = guide.encode(seqs)
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
tasks, sd = ctx.guide_tree(tcodes, n_threads=16)
ctx.msa_tree(codes, tasks, subm, scal, sd)
ctx.tree_aligned_rows(seqs)
tasks2, sd2 = ctx.aln_guide_tree()
ctx.tree_upload(codes, tasks2, subm, scal, sd2, flags=api.FLAG_TIMING)
for _ in range(2): ctx.tree_run(); ctx.tree_sync()
recs, paths, _ = ctx.tree_download(want_gaps=False)
tm = ctx.tree_timing()
print('ms', ctx.tree_kernel_ms())
lvl = {i: 0 for i in range(n)}
tl = []
for r in recs:
    l = 1 + max(lvl[r.a], lvl[r.b]); lvl[r.c] = l; tl.append(l)
tl = np.array(tl); kind = np.array([r.kind for r in recs])
GHZ = 2.4
tot = tm[:, :4].sum(1)
print('levels', tl.max(), 'tasks per level (first 12):', [int((tl == l).sum()) for l in range(1, 13)])
for l in list(range(1, 8)) + list(range(8, tl.max() + 1, max(1, tl.max() // 12))):
    m = tl == l
    if not m.any(): continue
    i = np.argmax(np.where(m, tot, -1))
    print('L%3d n=%4d kinds=%s  max task %.0f us (prep %.0f hirsch %.0f [pass %.0f meet %.0f] code %.0f merge %.0f) lens %dx%d nsip %d+%d' % (
        l, m.sum(), np.bincount(kind[m], minlength=3), tot[i]/GHZ/1e3, tm[i,0]/GHZ/1e3, tm[i,1]/GHZ/1e3, tm[i,4]/GHZ/1e3, tm[i,5]/GHZ/1e3, tm[i,2]/GHZ/1e3, tm[i,3]/GHZ/1e3,
        recs[i].len_a, recs[i].len_b, recs[i].nsip_a, recs[i].nsip_b))
done = {i: 0.0 for i in range(n)}
for r, t in zip(recs, tot):
    done[r.c] = max(done[r.a], done[r.b]) + t/GHZ/1e3
print('dependency-driven critical path %.0f us' % done[recs[-1].c])
