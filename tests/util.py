"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tree_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "tree_*.npz")))


def guide_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("guide_") and f.endswith(".npz"))


def cons_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "cons_*.npz")))


def refine_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "refine_*.npz")))


def pair_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "pairs_*.npz")))


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.lens = z["lens"]
        off = np.concatenate([[0], np.cumsum(self.lens)])
        if "codes" in z.files:
            self.codes = [z["codes"][off[i]:off[i + 1]] for i in range(len(self.lens))]
        if "tree_codes" in z.files:          # the alphabet the reference built its guide tree in
            self.tree_seqs = [z["tree_codes"][off[i]:off[i + 1]] for i in range(len(self.lens))]
        for k in z.files:
            if k not in ("lens", "codes"):
                setattr(self, k, z[k])
        if hasattr(self, "seq_distances") and len(self.seq_distances) == 0:
            self.seq_distances = None

    def rec(self, field):
        return self.z["rec_" + field]

    def path(self, t):
        o, n = int(self.rec("path_off")[t]), int(self.rec("plen")[t])
        return self.paths[o:o + n + 2]

    def gaps_list(self):
        out, o = [], 0
        for n in self.lens:
            out.append(self.gaps[o:o + int(n) + 1])
            o += int(n) + 1
        return out

    def maps_list(self):
        """consistency goldens: maps[i][k] = position map of sequence i against anchor k"""
        K = len(self.anchor_ids)
        out, o = [], 0
        for n in self.lens:
            row = []
            for _ in range(K):
                row.append(self.maps[o:o + int(n)])
                o += int(n)
            out.append(row)
        return out

    def sorted_seqs(self):
        """input strings in the reference's sorted order (rank -> input index)"""
        return [str(self.seqs[r]) for r in self.ranks]


def compare_recs(g, recs, paths, exact_fields, tol_fields=("confidence",), rtol=1e-5):
    """Compare a run's task records + coded paths against a Golden case."""
    problems = []
    for t, r in enumerate(recs):
        for f in exact_fields:
            if getattr(r, f) != g.rec(f)[t]:
                problems.append((t, f, getattr(r, f), g.rec(f)[t]))
        for f in tol_fields:
            want = float(g.rec(f)[t])
            if abs(getattr(r, f) - want) > rtol * max(1.0, abs(want)):
                problems.append((t, f, getattr(r, f), want))
        got = paths[r.path_off:r.path_off + r.plen + 2]
        if not np.array_equal(got, g.path(t)):
            problems.append((t, "path", None, None))
    return problems
