"""Synthetic guide trees, alphabets and forests for benchmarks and tests.

The reference's own guide tree (bisecting k-means, lib/src/bisectingKmeans.c) is built by the library
(ka_guide_tree, kalign_amd/csrc/ka_guide.cpp).  The headline benchmark fixes the tree instead, so that the timed step
is the dispatcher and nothing else: a task list with the contract create_tasks() produces
(bisectingKmeans.c:1067-1110): post-order, task t merges (a, b) into node c = numseq + t, children before parents,
root last.
"""
import numpy as np


def bisecting_tree(n, seed=1, jitter=0.15):
    """Recursive bisection of the leaf range with a randomised split point: depth ~ log2 n,
    like the k-means tree on a well-mixed family."""
    rng = np.random.RandomState(seed)
    order = rng.permutation(n)
    tasks = []
    counter = [n]

    def build(lo, hi):
        if hi - lo == 1:
            return int(order[lo])
        m = hi - lo
        split = lo + int(np.clip(round(m * (0.5 + rng.uniform(-jitter, jitter))), 1, m - 1))
        a = build(lo, split)
        b = build(split, hi)
        c = counter[0]
        counter[0] += 1
        tasks.append((a, b, c))
        return c

    import sys
    sys.setrecursionlimit(max(10000, 4 * n))
    build(0, n)
    return np.array(tasks, np.int32)


def encode(seqs, dna=False):
    """Kalign's internal codes (alphabet.c:179-245) for the unambiguous letters."""
    alpha = "ACGT" if dna else "ARNDCQEGHILKMFPSTWYV"
    lut = np.full(256, 255, np.uint8)
    for i, ch in enumerate(alpha):
        lut[ord(ch)] = i
    return [lut[np.frombuffer(s.encode(), np.uint8)] for s in seqs]


def encode_tree(seqs, dna=False):
    """The alphabet the reference builds its guide tree in (aln_wrap.c:155-160): for proteins the 13 classes of
    ALPHA_redPROTEIN (alphabet.c:236-290: (L,M) (I,V) (K,R) (E,Q,Z) (A,S,T) (N,D,B) (F,Y) C(,U) G H P W X), for
    nucleotides A C G T(,U) and one class for every ambiguity code (:203-234).  The edit distances built on these
    codes depend on the classes only, not on how they are numbered."""
    classes = ["A", "C", "G", "TU", "NRYSWKMBDHV"] if dna else \
        ["AST", "CU", "DNB", "EQZ", "FY", "G", "H", "IV", "KR", "LM", "P", "W", "X"]
    lut = np.full(256, 255, np.uint8)
    for i, cl in enumerate(classes):
        for ch in cl:
            lut[ord(ch)] = i
            lut[ord(ch.lower())] = i
    out = [lut[np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), np.uint8)] for s in seqs]
    if any((c == 255).any() for c in out):
        raise ValueError("letter outside the alphabet")
    return out


def forest(jobs):
    """Several independent alignment jobs as ONE forest job (kalign_amd.h: n_tasks < numseq-1).

    jobs: list of (codes, tasks[, seq_distances]) with tasks in TASK_ORDER_TREE order over that job's own
    node ids (leaves 0..n-1, internal nodes n..2n-2).  Returns (codes, tasks, seq_distances or None, spans) where
    spans[j] = (first sequence, first task, n sequences, n tasks) of job j in the combined job."""
    import numpy as np
    total = sum(len(j[0]) for j in jobs)
    codes, tasks, dists, spans = [], [], [], []
    seq0, node0, task0 = 0, total, 0
    have_dist = all(len(j) > 2 and j[2] is not None for j in jobs)
    for j in jobs:
        cj, tj = j[0], np.asarray(j[1])
        n = len(cj)

        def m(x):
            return seq0 + int(x) if x < n else node0 + (int(x) - n)

        codes += list(cj)
        tasks += [(m(a), m(b), m(c)) for a, b, c in tj]
        if have_dist:
            dists += list(j[2])
        spans.append((seq0, task0, n, len(tj)))
        seq0 += n
        node0 += len(tj)
        task0 += len(tj)
    return codes, np.array(tasks, np.int32), (np.array(dists, np.float32) if have_dist else None), spans
