"""GPU parity for the distance-estimation kernel (ka_bpm_batch): integer-exact against golden values of the
reference's bpm_block and against the oracle on an N x 32 batch like d_estimation's."""
import numpy as np
import pytest

from util import Golden

pytestmark = pytest.mark.gpu


def test_bpm_matches_reference_golden():
    import kalign_amd
    g = Golden("bpm_mixed")
    ctx = kalign_amd.Context(0)
    d = ctx.bpm_batch(g.codes, g.ia, g.ib)
    ctx.close()
    assert np.array_equal(d, g.dist)


def test_bpm_n_by_32_matches_oracle(oracle):
    """the shape d_estimation produces: every sequence against 32 samples (sequence_distance.c:98-121)"""
    import kalign_amd
    rng = np.random.RandomState(5)
    n = 600
    base = rng.randint(0, 13, 420).astype(np.uint8)
    codes = []
    for i in range(n):
        s = base.copy()
        idx = rng.rand(len(s)) < rng.uniform(0.02, 0.6)
        s[idx] = rng.randint(0, 13, int(idx.sum()))
        lo, hi = rng.randint(0, 30), len(s) - rng.randint(0, 30)
        codes.append(s[lo:hi])
    samples = rng.choice(n, 32, replace=False)
    ia = np.repeat(np.arange(n), 32).astype(np.int32)
    ib = np.tile(samples, n).astype(np.int32)
    ctx = kalign_amd.Context(0)
    d = ctx.bpm_batch(codes, ia, ib)
    ms = ctx.pairwise_kernel_ms()
    ctx.close()
    assert np.array_equal(d, oracle.bpm_batch(codes, ia, ib))
    assert ms > 0.0


def test_bpm_rejects_codes_outside_the_distance_alphabet():
    import kalign_amd
    ctx = kalign_amd.Context(0)
    with pytest.raises(kalign_amd.KalignAmdError):
        ctx.bpm_batch([np.array([1, 2, 20], np.uint8), np.array([1, 2, 3], np.uint8)], [0], [1])
    ctx.close()
