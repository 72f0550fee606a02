"""GPU: the opt-in analytic 4D-SH gradient (fdgs_scene.analytic_sh_grad).

Default: the backward reproduces the reference's three deviations (backward.cu:190, 303 / 384, 403).  Opt-in: the gradient
of the forward pass.  Checked (a) against the port oracle run with the same switch and (b) against finite differences of
the forward's own per-Gaussian colour.  Scene: gaussian_dim 4 WITHOUT rot_4d -- there the reference's backward has no
other path to ts (Q6), so dL/dts is the SH path alone and the per-Gaussian colour depends on (sh, ts) through SH only."""
import numpy as np
import pytest
import torch

from util import GRAD_SCALE, check_backward, run_hip, run_oracle, synth

pytestmark = pytest.mark.gpu


def test_analytic_sh_gradient_vs_oracle_and_finite_differences(gpu_device):
    from fdgs.gaussian_renderer import diff_gaussian_rasterization as dgr
    cfg = synth.SceneConfig("an", 4000, 176, 144, 3, 2, 0.04, 4.0, False, 4, False)
    scene = synth.make_scene(cfg, seed=31)
    assert scene["M"] == 48 and not scene["rot_4d"] and scene["gaussian_dim"] == 4
    up = synth.make_upstream_grads(scene["W"], scene["H"], seed=2, scale=GRAD_SCALE)
    _, compat = run_hip(scene, gpu_device, up)
    try:
        dgr.set_analytic_sh_gradients(True)
        fwd, ana = run_hip(scene, gpu_device, up)
    finally:
        dgr.set_analytic_sh_gradients(False)
    # (a) the oracle with the same switch; and the default mode still equals the default oracle
    _, ref_ana = run_oracle(dict(scene, analytic_sh_grad=True), up, kind="port")
    _, ref_compat = run_oracle(scene, up, kind="port")
    print("analytic:", {k: "%.2e/%.1e" % v for k, v in check_backward(ana, ref_ana, "analytic").items()})
    check_backward(compat, ref_compat, "compat")
    # the switch changes exactly dL_dsh[:, 1] (Q1) and dL_dts (Q2, Q3), nothing else
    vis = fwd["radii"] > 0
    d_sh = np.abs(ana["dL_dsh"] - compat["dL_dsh"]).reshape(-1, 48, 3).max(axis=(0, 2))
    assert d_sh[1] > 1e-3 and np.delete(d_sh, 1).max() <= 1e-6 * max(1.0, np.abs(ana["dL_dsh"]).max())
    assert np.abs(ana["dL_dts"] - compat["dL_dts"]).max() > 1e-3
    for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dmean2D"):
        assert np.abs(ana[k] - compat[k]).max() <= 1e-5 * max(1.0, np.abs(ana[k]).max()), k

    # (b) finite differences of L(sh, ts) = sum_g dL/dcolor_g . rgb_g(sh, ts), rgb from the forward's own records
    w = ana["dL_dcolor"].astype(np.float64)      # per-Gaussian colour gradient of the blend backward

    smooth = [vis & (fwd["clamped"].sum(1) == 0)]   # Gaussians visible and un-clamped in every evaluation (FD needs a smooth function)

    def colours(sc):
        out, _ = run_hip(sc, gpu_device, None)
        smooth[0] = smooth[0] & (out["radii"] > 0) & (out["clamped"].sum(1) == 0)
        return out["rgb"].astype(np.float64)

    eps = 2e-3 * scene["time_duration"]
    sp, sm = dict(scene), dict(scene)
    sp["ts"], sm["ts"] = scene["ts"] + eps, scene["ts"] - eps
    fd_ts = (w * (colours(sp) - colours(sm))).sum(1) / (2 * eps)
    sel = smooth[0].copy()
    assert sel.sum() > 1000
    err = np.abs(ana["dL_dts"][sel] - fd_ts[sel]).max()
    scale = np.abs(fd_ts[sel]).max()
    print("dL/dts: analytic vs FD max abs err %.2e (max|FD| %.2e); reference-compatible mode is off by %.2e" % (
        err, scale, np.abs(compat["dL_dts"][sel] - fd_ts[sel]).max()))
    assert err <= 2e-3 * scale
    assert np.abs(compat["dL_dts"][sel] - fd_ts[sel]).max() > 0.1 * scale      # Q2 / Q3 are real
    # Q1: coefficient 1 (l = 1, m = -1)
    e = 1e-2
    sp, sm = dict(scene), dict(scene)
    sp["shs"], sm["shs"] = scene["shs"].clone(), scene["shs"].clone()
    sp["shs"][:, 1, :] += e
    sm["shs"][:, 1, :] -= e
    fd_sh1 = w * (colours(sp) - colours(sm)) / (2 * e)   # rgb_c depends on sh[1, c] only -> per channel
    got = ana["dL_dsh"].reshape(-1, 48, 3)[:, 1, :]
    sel = smooth[0]
    err = np.abs(got[sel] - fd_sh1[sel]).max()
    scale = np.abs(fd_sh1[sel]).max()
    print("dL/dsh[1]: analytic vs FD max abs err %.2e (max|FD| %.2e)" % (err, scale))
    assert err <= 2e-3 * scale
    assert np.abs(compat["dL_dsh"].reshape(-1, 48, 3)[:, 1, :][sel] - fd_sh1[sel]).max() > 0.1 * scale   # Q1 is real
