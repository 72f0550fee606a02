"""oracle/knn_oracle.py (float32 restatement of simple-knn's distCUDA2) against an independent exact search."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import util  # noqa: F401
from oracle import knn_oracle, pyoracle

needs_ref = pytest.mark.skipif(not pyoracle.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


def _cloud(P, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        return (rng.standard_normal((P, 3)) * np.array([3.0, 1.0, 0.2])).astype(np.float32)
    if kind == "shifted":   # far from the origin: the origin takes part in the Morton bounds (simple_knn.cu:197, init = {0,0,0})
        return (rng.random((P, 3)) + np.array([50.0, -20.0, 7.0])).astype(np.float32)
    if kind == "dups":
        base = rng.standard_normal((max(P // 3, 1), 3)).astype(np.float32)
        return base[rng.integers(0, base.shape[0], P)]
    if kind == "plane":     # one degenerate axis: every Morton z code is the same
        a = rng.standard_normal((P, 3)).astype(np.float32)
        a[:, 2] = 0.25
        return a
    raise ValueError(kind)


@needs_ref
@pytest.mark.parametrize("P,kind", [(1, "gauss"), (2, "gauss"), (3, "gauss"), (4, "gauss"), (7, "gauss"), (255, "gauss"),
                                    (1024, "gauss"), (1025, "shifted"), (3000, "plane"), (4097, "dups"), (20000, "gauss")])
def test_knn_oracle_pinned_to_reference_source(P, kind):
    """The numpy restatement against the reference's OWN device code (simple_knn.cu:28-191 compiled verbatim, host sequence
    :193-220 restated with std::sort / a fold; oracle/refbuild/ref_knn.cpp): bit-exact, contraction off on both sides."""
    pts = _cloud(P, 100 + P, kind)
    want = pyoracle.ref_dist2_knn3(pts)
    got = knn_oracle.dist2_knn3(pts)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


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
