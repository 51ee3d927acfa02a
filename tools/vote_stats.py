"""How often a carried vote cell needs a count, and how many different positions the voters of a cell hold (oracle, CPU):
python tools/vote_stats.py [NSEQ 256] [LEN 300]  -- the consistency goldens, then a synthetic family on a random guide tree."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracledrv as o
from util import Golden, cons_cases
L = o.lib()
L.ko_carried_votes_distinct.argtypes = [C.c_void_p]


def report(name, run):
    L.ko_set_carried_votes(2)
    run()
    cells, counted = o.carried_votes_cells()
    d = (C.c_longlong * 8)()
    L.ko_carried_votes_distinct(d)
    both = sum(d)
    L.ko_set_carried_votes(0)
    print("%-22s cells %8d, needed a count %6.2f %%; cells where both operands vote %8d, different positions among their voters: %s" % (
        name, cells, 100.0 * counted / max(cells, 1), both, "  ".join("%d: %.1f %%" % (i, 100.0 * d[i] / max(both, 1)) for i in range(1, 8))))


for name in cons_cases():
    g = Golden(name)
    report(name, lambda: o.msa_tree_cons(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, int(g.n_anchors), float(g.weight)))
from kalign_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ln = int(sys.argv[2]) if len(sys.argv) > 2 else 300
seqs = synth.dssim(n, ln, dna=False, seed=5)
alpha = "ARNDCQEGHILKMFPSTWYV"
codes = [np.array([alpha.index(ch) for ch in s], np.uint8) for s in seqs]
rng = np.random.RandomState(3)
tasks, nodes, nxt = [], list(range(n)), n
while len(nodes) > 1:                                   # balanced-ish: pair neighbours level by level
    new = []
    for k in range(0, len(nodes) - 1, 2):
        tasks.append((nodes[k], nodes[k + 1], nxt)); new.append(nxt); nxt += 1
    if len(nodes) & 1:
        new.append(nodes[-1])
    nodes = new
tasks = np.array(tasks, np.int32)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "param_tables.npz"))
subm, scal = z["subm_0_3"], z["scal_0_3"].copy()
dist = rng.uniform(0.2, 1.2, size=n).astype(np.float32)
report("dssim %d x %d" % (n, ln), lambda: o.msa_tree_cons(codes, tasks, subm, scal, dist, 5, 2.0))
