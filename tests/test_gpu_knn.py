"""distCUDA2 on the GPU (csrc/knn.hip) against the oracle: bit-exact (FP contraction off, same operation order)."""
import time

import numpy as np
import pytest
import torch

import util  # noqa: F401
from oracle import knn_oracle, pyoracle

pytestmark = pytest.mark.gpu


def _pts(P, seed, kind="gauss"):
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        return (rng.standard_normal((P, 3)) * np.array([3.0, 1.0, 0.2])).astype(np.float32)
    if kind == "shifted":   # far from the origin: the origin takes part in the Morton bounds (reference quirk)
        return (rng.random((P, 3)) + np.array([50.0, -20.0, 7.0])).astype(np.float32)
    if kind == "dups":
        base = rng.standard_normal((max(P // 3, 1), 3)).astype(np.float32)
        return base[rng.integers(0, base.shape[0], P)]
    raise ValueError(kind)


@pytest.mark.parametrize("P,kind", [(1, "gauss"), (3, "gauss"), (4, "gauss"), (7, "gauss"), (255, "gauss"), (1025, "shifted"),
                                    (5000, "gauss"), (4097, "dups"), (20000, "gauss")])
def test_dist2_bit_exact_vs_oracle(P, kind, gpu_device):
    from fdgs.knn import distCUDA2
    pts = _pts(P, P, kind)
    got = distCUDA2(torch.from_numpy(pts).to(gpu_device)).cpu().numpy()
    want = knn_oracle.dist2_knn3(pts)
    assert got.shape == want.shape == (P,)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.skipif(not pyoracle.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("P,kind", [(1, "gauss"), (3, "gauss"), (4, "gauss"), (255, "gauss"), (1025, "shifted"), (4097, "dups"),
                                    (20000, "gauss"), (60000, "gauss")])
def test_dist2_bit_exact_vs_reference_source(P, kind, gpu_device):
    """knn.hip against the reference's own simple_knn.cu device code compiled for the CPU (oracle/_ref, ref_knn.cpp)."""
    from fdgs.knn import distCUDA2
    pts = _pts(P, 7 * P + 1, kind)
    got = distCUDA2(torch.from_numpy(pts).to(gpu_device)).cpu().numpy()
    want = pyoracle.ref_dist2_knn3(pts)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_dist2_full_size_vs_kdtree(gpu_device):
    """300 k points (the workload's initial point count): exact neighbours, checked against an independent search."""
    from scipy.spatial import cKDTree
    from fdgs.knn import distCUDA2
    pts = _pts(300000, 11, "gauss")
    t = torch.from_numpy(pts).to(gpu_device)
    distCUDA2(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = distCUDA2(t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4, workers=-1)
    np.testing.assert_allclose(got.cpu().numpy(), (d[:, 1:4] ** 2).mean(1), rtol=2e-5, atol=1e-10)
    print("distCUDA2 300k points: %.2f ms" % (dt * 1e3))
    assert dt < 1.0


def test_dist2_rejects_cpu_tensor():
    from fdgs.knn import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(4, 3))
