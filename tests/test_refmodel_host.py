"""CPU: fdgs.train_host.ReferenceStyleModel's Python-side covariance functions (what render() calls on the model with
``pipe.compute_cov3D_python``, gaussian_renderer/__init__.py:73-81) against golden vectors produced by the reference's OWN
scene/gaussian_model.py + utils/general_utils.py (tests/golden/make_golden_pycov.py ran them on the CPU)."""
import os

import numpy as np
import pytest
import torch

from util import synth  # noqa: F401  (sys.path)
from fdgs import train_host

HERE = os.path.dirname(os.path.abspath(__file__))


def _model(d):
    m = train_host.ReferenceStyleModel.__new__(train_host.ReferenceStyleModel)
    m.rot_4d, m.gaussian_dim, m.prefilter_var = bool(d["rot_4d"]), int(d["gaussian_dim"]), float(d["prefilter_var"])
    for k in ("scaling", "scaling_t", "rotation", "rotation_r", "t"):
        setattr(m, "_" + k, torch.from_numpy(d[k].copy()))
    return m


@pytest.mark.parametrize("name", ["rot4d", "dim4", "dim3"])
def test_python_covariance_matches_the_reference(name):
    d = np.load(os.path.join(HERE, "golden", "pycov_%s.npz" % name))
    m = _model(d)
    mod, ts = float(d["mod"]), float(d["timestamp"])
    if m.rot_4d:
        cov, off = m.get_current_covariance_and_mean_offset(mod, ts)
        np.testing.assert_allclose(off.numpy(), d["mean_offset"], rtol=2e-5, atol=1e-7)
    else:
        cov = m.get_covariance(mod)
    scale = np.abs(d["cov"]).max(axis=1, keepdims=True)   # per Gaussian: the conditional covariance is a difference of O(scale) terms
    assert np.abs(cov.numpy() - d["cov"]).max() <= 1e-5 * scale.max()
    assert (np.abs(cov.numpy() - d["cov"]) <= 2e-5 * scale + 1e-12).all()
    if m.gaussian_dim == 4:
        np.testing.assert_allclose(m.get_cov_t(mod).numpy(), d["cov_t"], rtol=1e-5)
        np.testing.assert_allclose(m.get_marginal_t(ts).numpy(), d["marginal_t"], rtol=2e-5, atol=1e-9)
