"""Linear sum assignment: the C restatement in oracle/lsap.c against the third-party routine
the reference actually calls (scipy.optimize.linear_sum_assignment, matcher.py:116). scipy is
part of this image (1.15.3; the reference pins 1.15.1, same algorithm), so it is used directly
as the known-answer source; no reference test pins this path."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment as scipy_lsap

from oracle.lsap import linear_sum_assignment as oracle_lsap


@pytest.mark.parametrize("seed", range(6))
def test_oracle_lsap_equals_scipy_index_for_index(seed):
    rng = np.random.default_rng(seed)
    for trial in range(60):
        kind = trial % 4
        if kind == 0:  # heavy ties: small integer costs (tie-breaking must agree too)
            c = rng.integers(0, 4, size=(rng.integers(1, 40), rng.integers(1, 40))).astype(float)
        elif kind == 1:
            c = rng.standard_normal((rng.integers(1, 40), rng.integers(1, 40)))
        elif kind == 2:  # the matcher's shape: 300 queries x n_gt, float32 costs
            c = rng.random((300, int(rng.integers(1, 40)))).astype(np.float32)
        else:  # constant matrix (all ties)
            c = np.full((rng.integers(1, 20), rng.integers(1, 20)), 1.0)
        r0, c0 = scipy_lsap(c)
        r1, c1 = oracle_lsap(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1)


def test_empty_and_infeasible():
    r, c = oracle_lsap(np.zeros((300, 0)))
    assert len(r) == 0 and len(c) == 0
    with pytest.raises(ValueError):
        oracle_lsap(np.array([[np.nan, 1.0], [1.0, 2.0]]))
