"""Seeded synthetic sequence families (stand-in for the reference's DSSim
simulator, tests/dssim.c:68-168: related sequences of length ~L with
substitutions and indels).  Pure numpy; used by tests and bench.py."""
import numpy as np

PROTEIN = "ARNDCQEGHILKMFPSTWYV"
DNA = "ACGT"


def family(n_seq, length, dna=False, seed=1, sub_rate=0.12, indel_rate=0.02):
    """n_seq sequences evolved from one root along a random binary tree."""
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer((DNA if dna else PROTEIN).encode(), np.uint8)
    root = alpha[rng.randint(0, len(alpha), size=length)]

    def mutate(s):
        s = s.copy()
        m = rng.random_sample(len(s)) < sub_rate
        s[m] = alpha[rng.randint(0, len(alpha), size=int(m.sum()))]
        n_ev = rng.poisson(indel_rate * len(s))
        for _ in range(n_ev):
            pos = rng.randint(0, max(1, len(s)))
            k = 1 + rng.geometric(0.45)
            if rng.random_sample() < 0.5 and len(s) > 4 * k:
                s = np.concatenate([s[:pos], s[pos + k:]])
            else:
                s = np.concatenate([s[:pos], alpha[rng.randint(0, len(alpha), size=k)], s[pos:]])
        return s

    pool = [root]
    while len(pool) < n_seq:
        parent = pool[rng.randint(0, len(pool))]
        pool.append(mutate(parent))
        pool[rng.randint(0, len(pool) - 1)] = mutate(pool[rng.randint(0, len(pool) - 1)])
    out = [mutate(s) for s in pool[:n_seq]]
    return [bytes(s.tobytes()).decode() for s in out]


def read_fasta(path):
    names, seqs = [], []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:])
                seqs.append([])
            elif line and seqs:
                seqs[-1].append(line)
    return names, ["".join(s) for s in seqs]
