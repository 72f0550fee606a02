"""CPU: the differentiable PyTorch statement of the forward (oracle/torch_oracle.py; BASELINE.json configs[0]: "10 k random 4D
Gaussians, 400 x 400, SH degree 0, forward-only via PyTorch CPU -- plumbing, no GPU") against the C oracle, and autograd through
it -- the analytic gradient of the reference's forward -- against the C oracle's backward: which of the reference's backward
formulas are the forward's derivative, and which are not (Q1-Q3 in the default mode; Q4 / Q5 under rot_4d)."""
import numpy as np
import pytest
import torch

from oracle import pyoracle, torch_oracle
from util import synth

SC = synth.SceneConfig


def _oracle(scene, up=None, analytic=False):
    sc = dict(scene, analytic_sh_grad=analytic)
    o = pyoracle.Oracle(sc, kind="port")
    out = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in o.forward().items()}
    out["R"] = o.R
    g = None
    if up is not None:
        g = {k: v.copy() for k, v in o.backward(up["grad_color"], up["grad_depth"], up["grad_alpha"], up["grad_flow"]).items()}
    o.close()
    return out, g


def test_c1_forward_on_the_cpu_in_pytorch():
    """BASELINE configs[0] at full size: C1 through PyTorch on the CPU (float64: in float32 torch's own operation order moves
    1e-5 of the pixels across an alpha = 1/255 threshold the C oracle does not flag) against the C oracle."""
    scene = synth.make_scene(synth.CONFIGS["C1"], seed=0, random_flow=True, bg=(0.3, 0.5, 0.7))
    ref, _ = _oracle(scene)
    with torch.no_grad():
        out, _p = torch_oracle.render(scene, ref, dtype=torch.float64)
    ok = ~ref["border"].astype(bool)
    assert ref["R"] > 50_000 and (~ok).mean() < 1e-3
    assert np.abs(out["out_color"].numpy() - ref["out_color"])[:, ok].max() <= 2e-5
    assert np.abs(out["out_depth"].numpy() - ref["out_depth"])[ok].max() <= 1e-4
    assert np.abs(out["out_T"].numpy() - ref["out_T"])[ok].max() <= 2e-5
    assert np.abs(out["out_flow"].numpy() - ref["out_flow"])[:, ok].max() <= 2e-5
    vis = ref["radii"] > 0
    pre = out["pre"]
    assert np.abs(pre["pix"].numpy() - ref["means2D"][vis]).max() <= 1e-3
    assert np.abs(pre["conic"].numpy() - ref["conic_opacity"][vis][:, :3]).max() <= 1e-4 * np.abs(ref["conic_opacity"][vis][:, :3]).max()
    assert np.abs(pre["rgb"].numpy() - ref["rgb"][vis]).max() <= 1e-5


CASES = {
    # name: (config, which gradients autograd must reproduce)
    "dim3_sh3": (SC("t", 1500, 96, 80, 3, 0, 0.04, 1.0, False, 3, False),
                 ("dL_dmean3D", "dL_dopacity", "dL_dsh", "dL_dscale", "dL_drot")),
    # 4D without rot_4d: the reference's backward has no marginal-opacity term (Q6: a TODO at backward.cu:917-919) -> dL_dopacity is
    # not rescaled by the marginal and ts / scales_t get nothing: only the tensors the marginal does not touch are the forward's derivative
    "dim4_norot_sh2": (SC("t", 1500, 96, 80, 2, 0, 0.04, 1.0, False, 4, True), ("dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot")),
    # rot_4d: the forward takes the SH direction from the un-shifted mean, the backward evaluates the basis at the shifted one (Q4:
    # dL_dsh is basis(shifted direction) x dL_dRGB -- off by the shift for every coefficient but the constant one), and the
    # conditional-covariance backward folds the SH part of dL_dmean into d(delta mean) (Q5): only the opacity is clean
    "rot4d_sh3t2": (SC("t", 1500, 96, 80, 3, 2, 0.04, 6.0, True, 4, False), ("dL_dopacity",)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_autograd_through_the_forward_against_the_backward_formulas(name):
    cfg, clean = CASES[name]
    scene = synth.make_scene(cfg, seed=17, random_flow=True, bg=(0.2, 0.1, 0.4))
    up = synth.make_upstream_grads(scene["W"], scene["H"], seed=3, scale=1e-2)
    ref, refg = _oracle(scene, up, analytic=True)
    keep = torch.from_numpy(~ref["border"].astype(bool))
    out, p = torch_oracle.render(scene, ref, dtype=torch.float64,
                                 requires_grad=("means3D", "opacities", "shs", "ts", "scales", "scales_t", "rotations", "rotations_r"))
    img = out["out_color"]
    assert float((img.detach().float() - torch.from_numpy(ref["out_color"])).abs()[:, keep].max()) <= 2e-5
    # the kernels receive d loss / d alpha as dL_dmask (alpha = 1 - T); the flow image has no clean counterpart here (Q9)
    loss = ((img * up["grad_color"].double()).sum() + (out["out_depth"] * up["grad_depth"][0].double()).sum()
            + ((1 - out["out_T"]) * up["grad_alpha"][0].double()).sum() + (out["out_flow"] * up["grad_flow"].double()).sum())
    loss.backward()
    got = {"dL_dmean3D": p["means3D"].grad, "dL_dopacity": p["opacities"].grad.reshape(-1), "dL_dsh": p["shs"].grad,
           "dL_dscale": p["scales"].grad, "dL_drot": p["rotations"].grad}
    report = {}
    for k, g in got.items():
        want = refg[k].reshape(g.shape)
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(g.numpy() - want).max())
        report[k] = "%.1e/%.1e%s" % (err, scale, "" if k in clean else " (not the forward's derivative: quirk)")
        if k in clean:
            assert err <= 2e-4 * scale, "%s %s: autograd %g away from the reference's backward (scale %g)" % (name, k, err, scale)
    print(name, report)
    if name == "rot4d_sh3t2":
        auto = p["shs"].grad.numpy()
        # the constant coefficient has no direction: clean in every mode
        assert np.abs(auto[:, 0] - refg["dL_dsh"][:, 0]).max() <= 2e-4
        # analytic mode: what is left at the other coefficients is Q4's shift of the direction, small ...
        q4 = np.abs(auto - refg["dL_dsh"]).max()
        assert 1e-4 < q4 < 5e-2, q4
        # ... and in the DEFAULT (bug-compatible) mode the 4D-SH backward is far from the forward's derivative (Q1: dL_dsh[1] is
        # written with the degree-0 basis value; Q2 / Q3 sit in dL_dts)
        _, refq = _oracle(scene, up, analytic=False)
        q1 = np.abs(auto[:, 1] - refq["dL_dsh"][:, 1]).max()
        assert q1 > 5 * np.abs(auto[:, 1] - refg["dL_dsh"][:, 1]).max() and q1 > 5e-3, q1
