"""How sparse are the count vectors of the profile-profile tasks' operands (aln_profileprofile.c:70-77 builds a list of a row's non-zero
counts and adds only those products)?  Per tree level of the headline tree: non-zero counts per profile column -- mean, and the maximum
over every 64 / 128 consecutive columns (what a strip's wave-uniform term count would be), for both operands.
arguments: [nseq 4096] [len 400].  Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd
NSEQ = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
LEN = int(sys.argv[2]) if len(sys.argv) > 2 else 400
codes, tasks, dist = bench.make_workload(NSEQ, LEN, False, 1)
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
recs, paths, _ = ctx.msa_tree(codes, tasks, subm, scal, dist)
n = len(codes)
lvl = {i: 0 for i in range(n)}; plen = {i: len(codes[i]) for i in range(n)}
nnz = {}
def node_nnz(node):
    if node not in nnz:
        p = ctx.tree_profile(node, plen[node]).reshape(-1, 64)[1:-1, :23]
        nnz[node] = (p != 0).sum(1)
    return nnz[node]
rows = []
for r in recs:
    lvl[r.c] = 1 + max(lvl[r.a], lvl[r.b]); plen[r.c] = r.plen
for r in recs[:-1]:
    pass
stats = {}
for r in recs:
    if r.kind != 2:
        continue
    out = []
    for node in (r.a, r.b):
        z = node_nnz(node)
        m64 = np.array([z[i:i + 64].max() for i in range(0, len(z), 64)])
        m128 = np.array([z[i:i + 128].max() for i in range(0, len(z), 128)])
        out.append((z.mean(), m64.mean(), m128.mean(), z.max()))
    stats.setdefault(lvl[r.c], []).append((r.len_a * r.len_b, r.swapped, r.nsip_a, r.nsip_b, out))
print("level tasks  cells(M)  swapped  nsip a/b (mean)   operand a: nnz mean / strip64 max / strip128 max / max    operand b: the same")
tot = 0
for l in sorted(stats):
    S = stats[l]
    cells = sum(s[0] for s in S) / 1e6; tot += cells
    a = np.array([s[4][0] for s in S]); b = np.array([s[4][1] for s in S])
    print("L%-3d %5d  %8.1f  %5.2f   %7.1f %7.1f     %5.1f %5.1f %5.1f %3d      %5.1f %5.1f %5.1f %3d" % (
        l, len(S), cells, np.mean([s[1] for s in S]), np.mean([s[2] for s in S]), np.mean([s[3] for s in S]),
        a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].max(), b[:, 0].mean(), b[:, 1].mean(), b[:, 2].mean(), b[:, 3].max()))
print("total profile-profile cells %.1f M" % tot)
ctx.close()
