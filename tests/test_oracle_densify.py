"""The densification oracle (oracle/densify_oracle.py) against fixtures produced by the reference's own
GaussianModel.densify_and_prune run on CPU (tests/golden/make_golden_densify.py)."""
import os

import numpy as np
import pytest

import util  # noqa: F401
from oracle import densify_oracle as do

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["rot4d", "rot4d_pruneonly", "dim4_norot", "dim3"]
REF2OURS = {"xyz": "_xyz", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation", "t": "_t",
            "scaling_t": "_scaling_t", "rotation_r": "_rotation_r"}


def load_state(d, prefix):
    """golden (reference names, f_dc / f_rest) -> oracle state (our names, one _features tensor)"""
    st = {"params": {}, "exp_avg": {}, "exp_avg_sq": {}}
    for suffix, dst in (("", "params"), (".exp_avg", "exp_avg"), (".exp_avg_sq", "exp_avg_sq")):
        for r, o in REF2OURS.items():
            if prefix + r + suffix in d.files:
                st[dst][o] = d[prefix + r + suffix]
        st[dst]["_features"] = np.concatenate([d[prefix + "f_dc" + suffix], d[prefix + "f_rest" + suffix]], 1)
    for k in ("xyz_gradient_accum", "t_gradient_accum", "denom", "max_radii2D"):
        if prefix + k in d.files:
            st[k] = d[prefix + k]
    return st


def golden_call(d):
    max_grad, min_opacity, extent, mss, mgt, prune_only, percent_dense = d["args"]
    sh_degree, sh_degree_t, gaussian_dim, rot_4d = [int(x) for x in d["cfg"]]
    normals = [d[k] for k in sorted((k for k in d.files if k.startswith("normal.")), key=lambda s: int(s.split(".")[1]))]
    kw = dict(max_grad=float(max_grad), min_opacity=float(min_opacity), extent=float(extent),
              max_screen_size=None if mss < 0 else float(mss), max_grad_t=None if mgt < 0 else float(mgt),
              prune_only=bool(prune_only), percent_dense=float(percent_dense), N=2, rot_4d=bool(rot_4d), gaussian_dim=gaussian_dim,
              samples=normals[0] if normals else None, samples_t=normals[1] if len(normals) > 1 else None)
    return kw


def assert_state_equal(got, want, rtol=0.0, atol=0.0):
    for grp in ("params", "exp_avg", "exp_avg_sq"):
        assert set(got[grp]) == set(want[grp])
        for k in want[grp]:
            assert got[grp][k].shape == want[grp][k].shape, (grp, k, got[grp][k].shape, want[grp][k].shape)
            np.testing.assert_allclose(got[grp][k], want[grp][k], rtol=rtol, atol=atol, err_msg="%s %s" % (grp, k))
    for k in ("xyz_gradient_accum", "t_gradient_accum", "denom", "max_radii2D"):
        if k in want:
            np.testing.assert_allclose(np.asarray(got[k]).reshape(want[k].shape), want[k], rtol=rtol, atol=atol, err_msg=k)


@pytest.mark.parametrize("case", CASES)
def test_densify_oracle_matches_reference_run(case):
    d = np.load(os.path.join(GOLD, "densify_%s.npz" % case))
    got = do.densify_and_prune(load_state(d, "in."), **golden_call(d))
    want = load_state(d, "out.")
    assert got["params"]["_xyz"].shape[0] == want["params"]["_xyz"].shape[0]
    # copies are exact; the split children's xyz / t / scaling go through exp / log / a 4x4 matmul: 1e-6
    assert_state_equal(got, want, rtol=2e-6, atol=2e-6)
