"""oracle/knn_oracle.py (float32 restatement of simple-knn's distCUDA2) against an independent exact search."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import util  # noqa: F401
from oracle import knn_oracle


@pytest.mark.parametrize("P,seed", [(5, 0), (257, 1), (3000, 2)])
def test_knn_oracle_vs_kdtree(P, seed):
    rng = np.random.default_rng(seed)
    pts = (rng.standard_normal((P, 3)) * np.array([3.0, 1.0, 0.2])).astype(np.float32)
    got = knn_oracle.dist2_knn3(pts)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    want = (d[:, 1:4] ** 2).mean(1)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-9)


def test_knn_oracle_degenerate_inputs():
    FLT_MAX = np.float32(3.4028234663852886e38)
    one = knn_oracle.dist2_knn3(np.zeros((1, 3), np.float32))
    assert one.shape == (1,) and (np.isinf(one[0]) or one[0] >= FLT_MAX / 3)         # three missing neighbours
    three = knn_oracle.dist2_knn3(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32))
    assert (three >= FLT_MAX / 3).all() or np.isinf(three).all()                      # only two neighbours each
    dup = np.array([[1, 2, 3]] * 5 + [[4, 4, 4]], np.float32)
    out = knn_oracle.dist2_knn3(dup)
    assert (out[:5] == 0).all() and out[5] == np.float32(9 + 4 + 1)                  # duplicates count with distance 0
