"""GPU parity tests: the HIP product (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.md / north_star): tile / key indexing bit-exact (radii, tiles_touched, depth-key bits,
point_list, ranges, n_contrib), pixels within 1e-4 abs, gradients within 1e-4 (of max(1, max|ref|)).
Pixels flagged by the oracle as sitting on a threshold cliff (alpha ~ 1/255, T ~ 1e-4 within 1e-5 relative)
are excluded from the pixel / n_contrib comparison and their fraction is bounded.
"""
import numpy as np
import pytest
import torch

from util import GRAD_SCALE, check_backward, check_forward, run_hip, run_oracle, synth

pytestmark = pytest.mark.gpu

SC = synth.SceneConfig


def _variants():
    v = {}
    v["C1_rot4d_sh0"] = dict(cfg=synth.CONFIGS["C1"], kw=dict(random_flow=True, bg=(0.3, 0.5, 0.7)))
    v["rot4d_sh3_t2"] = dict(cfg=SC("v", 30000, 400, 304, 3, 2, 0.015, 10.0, True, 4, False), kw=dict(random_flow=True))
    v["rot4d_sh3_t1"] = dict(cfg=SC("v", 8000, 200, 120, 3, 1, 0.03, 2.0, True, 4, False), kw=dict(bg=(1.0, 1.0, 1.0)))
    v["rot4d_sh2_4d"] = dict(cfg=SC("v", 8000, 200, 120, 2, 2, 0.03, 2.0, True, 4, False), kw=dict())
    v["dim3_sh2"] = dict(cfg=SC("v", 8000, 256, 256, 2, 0, 0.03, 1.0, False, 3, False), kw=dict(random_flow=True))
    v["dim4_norot_sh1"] = dict(cfg=SC("v", 8000, 250, 130, 1, 0, 0.03, 1.0, False, 4, True), kw=dict(bg=(0.1, 0.2, 0.3)))
    v["ragged_33x17"] = dict(cfg=SC("v", 500, 33, 17, 3, 0, 0.05, 1.0, True, 4, True), kw=dict())
    return v


VARIANTS = _variants()


def _scene(name):
    spec = VARIANTS[name]
    return synth.make_scene(spec["cfg"], seed=3, **spec["kw"])


@pytest.mark.parametrize("name", list(VARIANTS))
def test_forward_backward_vs_oracle(name, gpu_device):
    scene = _scene(name)
    grads = synth.make_upstream_grads(scene["W"], scene["H"], seed=1, scale=GRAD_SCALE)
    hip, hipg = run_hip(scene, gpu_device, grads)
    ref, refg = run_oracle(scene, grads, kind="port")
    rep = check_forward(hip, ref, name)
    repg = check_backward(hipg, refg, name)
    print(name, "R", ref["R"], {k: ("%.2e" % v if isinstance(v, float) else v) for k, v in rep.items()})
    print(name, {k: "%.2e/%.1e" % v for k, v in repg.items()})


def test_precomputed_cov_and_colors(gpu_device):
    """cov3D_precomp + colors_precomp branch (forward.cu:411-414, 476): feed the oracle's own cov3D / rgb back in."""
    base = synth.make_scene(SC("v", 6000, 200, 160, 1, 0, 0.03, 1.0, True, 4, True), seed=5)
    ref0, _ = run_oracle(base, None, kind="port")
    scene = dict(base)
    scene["means3D"] = torch.from_numpy(ref0["out_means3D"].copy())
    scene["cov3D_precomp"] = torch.from_numpy(ref0["cov3D"].copy())
    scene["colors_precomp"] = torch.from_numpy(np.random.default_rng(0).random((6000, 3)).astype(np.float32))
    for k in ("shs", "scales", "rotations", "scales_t", "rotations_r", "ts"):
        scene[k] = None
    scene["rot_4d"], scene["gaussian_dim"] = False, 3
    grads = synth.make_upstream_grads(scene["W"], scene["H"], seed=2, scale=GRAD_SCALE)
    hip, hipg = run_hip(scene, gpu_device, grads)
    ref, refg = run_oracle(scene, grads, kind="port")
    check_forward(hip, ref, "precomp", precomp_cov=True, precomp_colors=True)
    check_backward(hipg, refg, "precomp")


def test_all_culled_and_empty(gpu_device):
    """Edge cases: every Gaussian behind the camera (R == 0) and P == 0: background image, T == 1."""
    scene = synth.make_scene(SC("v", 300, 64, 48, 0, 0, 0.03, 1.0, True, 4, True), seed=1, bg=(0.2, 0.4, 0.6))
    scene["means3D"] = scene["means3D"].clone()
    scene["means3D"][:, 2] = -10.0
    hip, _ = run_hip(scene, gpu_device, None)
    assert hip["R"] == 0 and (hip["radii"] == 0).all()
    assert np.allclose(hip["out_T"], 1.0) and (hip["n_contrib"] == 0).all()
    for c, v in enumerate((0.2, 0.4, 0.6)):
        assert np.allclose(hip["out_color"][c], v)
    empty = dict(scene)
    for k in ("means3D", "ts", "scales", "scales_t", "rotations", "rotations_r", "opacities", "shs", "flow_2d"):
        empty[k] = scene[k][:0].clone()
    hip, _ = run_hip(empty, gpu_device, None)
    assert hip["R"] == 0 and np.allclose(hip["out_T"], 1.0)


def test_c2_full_size(gpu_device):
    """BASELINE configs[1]: 100k Gaussians, 800x800, SH degree 3, forward + backward against the oracle."""
    scene = synth.make_scene(synth.CONFIGS["C2"], seed=0)
    grads = synth.make_upstream_grads(scene["W"], scene["H"], seed=1, scale=GRAD_SCALE)
    hip, hipg = run_hip(scene, gpu_device, grads)
    ref, refg = run_oracle(scene, grads, kind="port")
    rep = check_forward(hip, ref, "C2")
    repg = check_backward(hipg, refg, "C2")
    print("C2 R", ref["R"], rep)
    print("C2", {k: "%.2e/%.1e" % v for k, v in repg.items()})


@pytest.mark.parametrize("name", ["rot4d_sh3_t2", "C1_rot4d_sh0", "ragged_33x17"])
def test_colour_only_backward_vs_oracle(name, gpu_device):
    """Only the colour image has an upstream gradient (depth / alpha / flow gradients None at the binding, NULL at
    the C ABI): the colour-only blend-backward variant must equal the oracle fed with explicit zeros."""
    scene = _scene(name)
    grads = synth.make_upstream_grads(scene["W"], scene["H"], seed=3, scale=GRAD_SCALE)
    zeros = {k: (v if k == "grad_color" else torch.zeros_like(v)) for k, v in grads.items()}
    nones = {k: (v if k == "grad_color" else None) for k, v in grads.items()}
    _, hipg = run_hip(scene, gpu_device, nones)
    _, refg = run_oracle(scene, zeros, kind="port")
    check_backward(hipg, refg, name + " colour-only")


def test_depth_only_backward_vs_oracle(gpu_device):
    """No upstream gradient for the colour image (NULL at the C ABI), only for depth and alpha: the general backward
    variant with zeros substituted in the kernel equals the oracle fed with an explicit zero colour gradient."""
    name = "rot4d_sh3_t1"
    scene = _scene(name)
    grads = synth.make_upstream_grads(scene["W"], scene["H"], seed=4, scale=GRAD_SCALE)
    zeros = {k: (torch.zeros_like(v) if k in ("grad_color", "grad_flow") else v) for k, v in grads.items()}
    nones = {k: (None if k in ("grad_color", "grad_flow") else v) for k, v in grads.items()}
    _, hipg = run_hip(scene, gpu_device, nones)
    _, refg = run_oracle(scene, zeros, kind="port")
    check_backward(hipg, refg, name + " depth+alpha only")
