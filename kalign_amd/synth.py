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


# background frequencies used as the emission prior (Robinson & Robinson order ACDEFGHIKLMNPQRSTVWY)
_PRIOR_AA = np.array([0.0755, 0.0170, 0.0530, 0.0632, 0.0407, 0.0685, 0.0224, 0.0573, 0.0594, 0.0934,
                      0.0236, 0.0453, 0.0493, 0.0402, 0.0516, 0.0722, 0.0574, 0.0652, 0.0125, 0.0322])
_AA_ORDER = "ACDEFGHIKLMNPQRSTVWY"


def dssim(n_seq, length, dna=False, seed=1, n_obs=30, match_err=0.05, insert_err=0.25):
    """Restatement of the reference's DSSim generator (tests/dssim.c:68-168, 333-431): n_seq
    independent samples from ONE random profile HMM of `length` match states.  Every match /
    insert state has a dominant residue observed n_obs times with an error rate, plus the
    background prior; transitions M->M 1-p, M->I = M->D = p/2 with p = 0.02 (n_seq > 100) or
    0.04, I->I = I->M = D->D = D->M = 0.5; uniform start state.  Output lengths ~ length +- 5 %."""
    rng = np.random.RandomState(seed)
    if dna:
        alpha, prior = "ACGT", np.full(4, 0.25)
    else:
        alpha, prior = _AA_ORDER, _PRIOR_AA / _PRIOR_AA.sum()
    L = len(alpha)
    letters = np.frombuffer(alpha.encode(), np.uint8)

    def emissions(err):
        pick = rng.choice(L, size=length, p=prior)
        e = np.tile(prior, (length, 1))
        wrong = rng.random_sample((length, n_obs)) < err
        rnd = rng.randint(0, L, size=(length, n_obs))
        obs = np.where(wrong, rnd, pick[:, None])
        for c in range(L):
            e[:, c] += (obs == c).sum(1)
        return np.cumsum(e / e.sum(1, keepdims=True), axis=1)

    cm, ci = emissions(match_err), emissions(insert_err)
    p = 0.02 if n_seq > 100 else 0.04
    out = []
    for _ in range(n_seq):
        seq = []
        i, state = 0, rng.randint(0, 3)          # 0 M, 1 I, 2 D
        while i < length:
            r = rng.random_sample()
            if state == 0:
                seq.append(letters[min(int(np.searchsorted(cm[i], rng.random_sample())), L - 1)])
                i += 1
                state = 0 if r < 1.0 - p else (1 if r < 1.0 - p / 2 else 2)
            elif state == 1:
                seq.append(letters[min(int(np.searchsorted(ci[i], rng.random_sample())), L - 1)])
                state = 1 if r < 0.5 else 0
            else:
                i += 1
                state = 2 if r < 0.5 else 0
        if not seq:
            seq.append(letters[0])
        out.append(bytes(seq).decode())
    return out


def dssim_fast(n_seq, length, dna=False, seed=1, n_obs=30, match_err=0.05, insert_err=0.25):
    """The same profile-HMM family as dssim(), sampled for all sequences in lock-step (numpy over the sequences instead
    of a Python loop per residue): same distribution, a different random stream.  For the big benchmark sets
    (16384 x 500 takes half a minute with dssim())."""
    rng = np.random.RandomState(seed)
    if dna:
        alpha, prior = "ACGT", np.full(4, 0.25)
    else:
        alpha, prior = _AA_ORDER, _PRIOR_AA / _PRIOR_AA.sum()
    L = len(alpha)
    letters = np.frombuffer(alpha.encode(), np.uint8)

    def emissions(err):
        pick = rng.choice(L, size=length, p=prior)
        e = np.tile(prior, (length, 1))
        wrong = rng.random_sample((length, n_obs)) < err
        rnd = rng.randint(0, L, size=(length, n_obs))
        obs = np.where(wrong, rnd, pick[:, None])
        for c in range(L):
            e[:, c] += (obs == c).sum(1)
        return np.cumsum(e / e.sum(1, keepdims=True), axis=1)

    cm, ci = emissions(match_err), emissions(insert_err)
    p = 0.02 if n_seq > 100 else 0.04
    pos = np.zeros(n_seq, np.int64)
    state = rng.randint(0, 3, n_seq)                 # 0 M, 1 I, 2 D
    cap = int(length * 1.5) + 64
    out = np.zeros((n_seq, cap), np.uint8)
    olen = np.zeros(n_seq, np.int64)
    idx = np.arange(n_seq)
    while True:
        live = pos < length
        if not live.any():
            break
        r = rng.random_sample(n_seq)
        u = rng.random_sample(n_seq)
        pc = np.minimum(pos, length - 1)
        emit_m = live & (state == 0)
        emit_i = live & (state == 1)
        sym_m = np.minimum((cm[pc] < u[:, None]).sum(1), L - 1)
        sym_i = np.minimum((ci[pc] < u[:, None]).sum(1), L - 1)
        emit = emit_m | emit_i
        room = emit & (olen < cap)
        out[idx[room], olen[room]] = letters[np.where(emit_m, sym_m, sym_i)[room]]
        olen += room
        adv = live & (state != 1)
        nstate = np.where(state == 0, np.where(r < 1.0 - p, 0, np.where(r < 1.0 - p / 2, 1, 2)),
                          np.where(state == 1, np.where(r < 0.5, 1, 0), np.where(r < 0.5, 2, 0)))
        state = np.where(live, nstate, state)
        pos += adv
    res = []
    for i in range(n_seq):
        row = out[i, :max(int(olen[i]), 1)]
        if olen[i] == 0:
            row = letters[:1]
        res.append(row.tobytes().decode())
    return res
