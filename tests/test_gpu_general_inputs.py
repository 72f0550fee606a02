"""GPU parity on the two input families EVERY real training run has and the synthetic default scene does not:

1. **General cameras** (fdgs.synth.POSES): rotated, off-axis, one with the centre-shift projection, one on a steep slant.  With the
   default camera (R = I, centre on the z axis) the off-diagonal terms of ``viewmatrix`` / ``projmatrix`` are zero, so ``W`` in
   computeCov2D (forward.cu:218-224) and its backward (backward.cu:525-531), transformVec4x3Transpose (backward.cu:611,
   auxiliary.h:90-98), 8 of the 12 ``proj[]`` terms of the projection Jacobian (backward.cu:886-891) and viewmatrix[12,13] were
   multiplied by zero in every comparison.  Real cameras (scene/cameras.py:65-71) always carry a rotation.
2. **Active SH degree below the allocated coefficient count**: the reference allocates M = 48 coefficients from iteration 0
   (scene/gaussian_model.py:65,92) and raises (active_sh_degree, active_sh_degree_t) one step every 1000 iterations
   (:253-257, train.py:93-94): (0,0) -> (1,0) -> (2,0) -> (3,0) -> (3,1) -> (3,2).  computeColorFromSH_4D (forward.cu:73-195) and
   its backward (backward.cu:144-481) then run with a coefficient stride larger than what they use; the gradient of every unused
   coefficient stays at the zero ``torch::zeros`` put there.

Both against the port oracle, which tests/test_oracle_pin.py pins to the reference's own source on the same poses and degrees.
Bar as everywhere: tile / key indexing bit-exact, pixels 1e-4 abs, gradients 1e-4 of max(1, max|ref|)."""
import numpy as np
import pytest
import torch

from util import GRAD_SCALE, check_backward, check_forward, run_hip, run_oracle, synth

pytestmark = pytest.mark.gpu
SC = synth.SceneConfig
RIG = [p for p in synth.POSES if p != "axis"]

SCENES = {
    "C1_rot4d_sh0": (synth.CONFIGS["C1"], dict(random_flow=True, bg=(0.3, 0.5, 0.7))),
    "rot4d_sh3_t2": (SC("v", 30000, 400, 304, 3, 2, 0.015, 10.0, True, 4, False), dict(random_flow=True)),
    "rot4d_sh3_t1": (SC("v", 8000, 200, 120, 3, 1, 0.03, 2.0, True, 4, False), dict(bg=(1.0, 1.0, 1.0))),
    "dim3_sh2": (SC("v", 8000, 256, 256, 2, 0, 0.03, 1.0, False, 3, False), dict(random_flow=True)),
    "dim4_norot_sh1": (SC("v", 8000, 250, 130, 1, 0, 0.03, 1.0, False, 4, True), dict(bg=(0.1, 0.2, 0.3))),
    "ragged_33x17": (SC("v", 500, 33, 17, 3, 0, 0.05, 1.0, True, 4, True), dict()),
}


def _masked_grads(scene, ref, seed=1):
    """Upstream gradients with the oracle-flagged cliff pixels zeroed (both sides see the same tensors): the comparison measures
    arithmetic, not which side of alpha >= 1/255 / T >= 1e-4 a borderline pixel fell on (as tests/test_gpu_tile_cull.py)."""
    W, H = scene["W"], scene["H"]
    keep = torch.from_numpy(~ref["border"].astype(bool)).to(torch.float32)
    grads = synth.make_upstream_grads(W, H, seed=seed, scale=GRAD_SCALE)
    return {k: v * keep.reshape((1,) * (v.dim() - 2) + (H, W)) for k, v in grads.items()}


def _matrix_is_general(scene):
    """Every entry of the 3x4 part of the view matrix and the 12 entries of the full projection the kernels read are non-zero."""
    v, p = scene["world_view_transform"].numpy(), scene["full_proj_transform"].numpy()
    return bool((np.abs(v[:, :3]) > 1e-3).all() and (np.abs(p) > 1e-3).all())


@pytest.mark.parametrize("tile_cull", [False, True], ids=["reference-lists", "tile-cull"])
@pytest.mark.parametrize("pose", RIG)
@pytest.mark.parametrize("name", list(SCENES))
def test_forward_backward_vs_oracle_on_general_cameras(name, pose, tile_cull, gpu_device):
    cfg, kw = SCENES[name]
    scene = synth.make_scene(cfg, seed=3, pose=pose, **kw)
    assert _matrix_is_general(scene)
    ref, _ = run_oracle(scene, None, kind="port")
    grads = _masked_grads(scene, ref)
    ref, refg = run_oracle(scene, grads, kind="port")
    hip, hipg = run_hip(scene, gpu_device, grads, tile_cull=tile_cull)
    label = "%s @ %s" % (name, pose)
    rep = check_forward(hip, ref, label, tile_cull=tile_cull, WH=(scene["W"], scene["H"]), max_border=5e-3)
    repg = check_backward(hipg, refg, label)
    assert ref["R"] > 0 and (ref["radii"] > 0).any()
    print(label, "R", ref["R"], {k: ("%.2e" % v if isinstance(v, float) else v) for k, v in rep.items()})
    print(label, {k: "%.2e/%.1e" % v for k, v in repg.items()})


def test_slanted_camera_exercises_the_clamp_and_the_near_plane():
    """(CPU part of the case above, stated once: what the 'slant' pose is for.)  Gaussians on BOTH sides of the z <= 0.2 cull
    (auxiliary.h:153) and beyond the 1.3 tanfov clamp of the EWA Jacobian (forward.cu:206-211) in x and in y, all on a rotated view."""
    cfg, kw = SCENES["rot4d_sh3_t2"]
    scene = synth.make_scene(cfg, seed=3, pose="slant", **kw)
    V = scene["world_view_transform"].numpy().astype(np.float64)
    p = scene["means3D"].numpy().astype(np.float64)
    pv = p @ V[:3, :3] + V[3, :3]
    z = pv[:, 2]
    assert (z <= 0.2).sum() > 100 and (z > 0.2).sum() > 1000
    front = z > 0.2
    tx, ty = pv[front, 0] / z[front], pv[front, 1] / z[front]
    assert (np.abs(tx) > 1.3 * scene["tanfovx"]).sum() > 50 and (np.abs(ty) > 1.3 * scene["tanfovy"]).sum() > 50


@pytest.mark.parametrize("pose", list(synth.POSES))
def test_mark_visible_on_general_cameras(pose, gpu_device):
    """markVisible / checkFrustum (rasterizer_impl.cu:45-67, auxiliary.h:139-166) through the drop-in class."""
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import pyoracle
    scene = synth.make_scene(SC("api", 4999, 64, 64, 0, 0, 0.03, 1.0, True, 4, True), seed=2, pose=pose)
    scene["means3D"][::4, 2] = -4.5
    rs = GaussianRasterizationSettings(64, 64, scene["tanfovx"], scene["tanfovy"], scene["bg"].to(gpu_device), 1.0,
                                       scene["world_view_transform"].to(gpu_device), scene["full_proj_transform"].to(gpu_device), 0, 0,
                                       scene["camera_center"].to(gpu_device), 0.5, 1.0, True, 4, True, False, False)
    vis = GaussianRasterizer(rs).markVisible(scene["means3D"].to(gpu_device))
    ref = pyoracle.mark_visible(scene["means3D"], scene["world_view_transform"], scene["full_proj_transform"])
    assert vis.dtype == torch.bool and np.array_equal(vis.cpu().numpy(), ref) and 0 < ref.sum() < ref.size


# ---------------------------------------------------------------------------------------------------------------------------
# active degree below the allocated one
# ---------------------------------------------------------------------------------------------------------------------------

RAMP = [(0, 0), (1, 0), (2, 0), (3, 0), (3, 1)]     # the first 5000 iterations of every run (gaussian_model.py:253-257)


@pytest.mark.parametrize("tile_cull", [False, True], ids=["reference-lists", "tile-cull"])
@pytest.mark.parametrize("pose", ["axis", "rig1"])
@pytest.mark.parametrize("deg", RAMP)
def test_below_allocated_degree_vs_oracle(deg, pose, tile_cull, gpu_device):
    """shs is [P, 48, 3] (allocated for (3, 2)), the active degrees are ``deg``; the coefficients beyond the active ones hold random
    non-zero numbers the kernels must not read; their gradient must be exactly zero."""
    cfg = SC("v", 8000, 200, 120, deg[0], deg[1], 0.03, 2.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=3, pose=pose, alloc=(3, 2), random_flow=True)
    assert scene["shs"].shape == (8000, 48, 3) and scene["sh_degree"] == deg[0] and scene["sh_degree_t"] == deg[1]
    ref, _ = run_oracle(scene, None, kind="port")
    grads = _masked_grads(scene, ref)
    ref, refg = run_oracle(scene, grads, kind="port")
    hip, hipg = run_hip(scene, gpu_device, grads, tile_cull=tile_cull)
    label = "alloc (3,2) active %r @ %s" % (deg, pose)
    check_forward(hip, ref, label, tile_cull=tile_cull, WH=(scene["W"], scene["H"]), max_border=5e-3)
    repg = check_backward(hipg, refg, label)
    n = synth.active_sh_coeffs(deg[0], deg[1], False, 4)
    assert hipg["dL_dsh"].shape == (8000, 48, 3)
    assert float(np.abs(refg["dL_dsh"][:, n:]).max()) == 0.0 and float(np.abs(hipg["dL_dsh"][:, n:]).max()) == 0.0, label
    assert float(np.abs(hipg["dL_dsh"][:, n - 1]).max()) > 0.0
    # and the image does not depend on what the unused coefficients hold
    other = dict(scene)
    other["shs"] = scene["shs"].clone()
    other["shs"][:, n:] = 7.0
    hip2, _ = run_hip(other, gpu_device, None, tile_cull=tile_cull)
    for k in ("out_color", "rgb", "clamped_bits"):
        np.testing.assert_array_equal(hip[k], hip2[k], err_msg=label + ": %s reads an inactive coefficient" % k)
    print(label, {k: "%.2e/%.1e" % v for k, v in repg.items()})


@pytest.mark.parametrize("cfg,alloc", [(SC("v", 6000, 192, 144, 1, 0, 0.03, 1.0, False, 3, False), (3, 0)),
                                       (SC("v", 6000, 192, 144, 0, 0, 0.03, 1.0, False, 3, False), (2, 0)),
                                       (SC("v", 6000, 192, 144, 2, 0, 0.03, 1.0, True, 4, True), (3, 0)),
                                       (SC("v", 6000, 192, 144, 1, 0, 0.03, 1.0, False, 4, True), (3, 0))],
                         ids=["dim3-1of3", "dim3-0of2", "rot4d-sh3d-2of3", "dim4norot-sh3d-1of3"])
def test_below_allocated_degree_3d_sh_vs_oracle(cfg, alloc, gpu_device):
    """computeColorFromSH (the 3D statement, forward.cu:22-70 / backward.cu:20-141) with unused coefficients, on a rotated camera."""
    scene = synth.make_scene(cfg, seed=5, pose="rig3", alloc=alloc)
    ref, _ = run_oracle(scene, None, kind="port")
    grads = _masked_grads(scene, ref)
    ref, refg = run_oracle(scene, grads, kind="port")
    hip, hipg = run_hip(scene, gpu_device, grads)
    check_forward(hip, ref, cfg.name, max_border=5e-3)
    check_backward(hipg, refg, cfg.name)
    n = (cfg.sh_degree + 1) ** 2
    assert scene["shs"].shape[1] == (alloc[0] + 1) ** 2 > n
    assert float(np.abs(hipg["dL_dsh"][:, n:]).max()) == 0.0 and float(np.abs(hipg["dL_dsh"][:, :n]).max()) > 0.0


# ---------------------------------------------------------------------------------------------------------------------------
# the timed path (raw parameters, fused loss, accumulation over views, deferred SH gradient) on both families
# ---------------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("tile_cull", [False, True], ids=["reference-lists", "tile-cull"])
def test_timed_path_on_a_camera_rig_vs_oracle(tile_cull, gpu_device):
    """What StepPipeline does for one optimizer step, three views from three DIFFERENT rotated cameras (one with the centre-shift
    projection, one slanted), against the oracle view by view (tests/test_gpu_parity.py::_timed_path_vs_oracle)."""
    from test_gpu_parity import _timed_path_vs_oracle
    _timed_path_vs_oracle(SC("v", 30000, 400, 304, 3, 2, 0.015, 10.0, True, 4, False), gpu_device, 3, "timed-rig", 2e-3, tile_cull=tile_cull,
                          poses=["rig0", "rig2", "slant"])


@pytest.mark.parametrize("deg", [(0, 0), (2, 0), (3, 1)])
def test_timed_path_below_allocated_degree_vs_oracle(deg, gpu_device):
    """The timed path with M = 48 allocated and ``deg`` active, two rotated cameras: accumulated raw-parameter gradients against the
    oracle, the gradient of every inactive coefficient exactly zero (asserted inside)."""
    from test_gpu_parity import _timed_path_vs_oracle
    _timed_path_vs_oracle(SC("v", 12000, 320, 240, deg[0], deg[1], 0.015, 10.0, True, 4, False), gpu_device, 2, "timed-alloc48-%d%d" % deg, 2e-3,
                          tile_cull=True, poses=["rig1", "rig3"], alloc=(3, 2))


@pytest.mark.parametrize("fuse", [True, False], ids=["fused-sh-adam", "flush+adam"])
def test_degree_ramp_trains_like_the_reference_loop(fuse, gpu_device):
    """The schedule of the first iterations -- M = 48 allocated, degrees (0,0) -> (1,0) -> ... one step at a time (oneupSHdegree,
    gaussian_model.py:253-257) -- through StepPipeline (deferred SH stages -> sh_flush / fused SH-Adam) against the reference's own
    loop on a reference-style model (render() + autograd + torch.optim.Adam, train.py:104-170, 247-249), rotated cameras.
    A coefficient block that has never been active keeps its value bit for bit (g = 0, m = v = 0: Adam's update is 0 / (0 + eps));
    a block that WAS active and whose gradient is zero afterwards keeps moving on its decaying moments, exactly as torch's does."""
    from fdgs import train_host
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    from fdgs.pipeline import StepPipeline
    cfg = SC("ramp", 6000, 208, 160, 0, 0, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=4, alloc=(3, 2))
    B = 2
    dev = gpu_device
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    pipe = train_host.PipelineFlags()
    cams = [train_host.SyntheticCamera(dict(scene, **synth.camera_for(p, scene["W"], scene["H"])), dev, timestamp=(b + 0.5) / B * 10.0)
            for b, p in enumerate(["rig0", "rig1"])]
    gen = torch.Generator(device="cpu").manual_seed(7)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(dev) for _ in range(B)]

    ref = train_host.ReferenceStyleModel(scene, dev, optimizer="torch")
    mp = train_host.GaussianParams(scene, dev)
    assert (mp.max_sh_degree, mp.max_sh_degree_t, mp.active_sh_degree, mp.active_sh_degree_t) == (3, 2, 0, 0)
    sp = StepPipeline(mp, train_host.make_optimizer(mp), world_size=1, lambda_dssim=0.2, fuse_sh_adam=fuse)
    init = scene["shs"].to(dev)
    schedule = [(0, 0), (0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 2), (3, 2)]
    seen = 0
    for it, deg in enumerate(schedule):
        ref.active_sh_degree, ref.active_sh_degree_t = deg
        while (mp.active_sh_degree, mp.active_sh_degree_t) != deg:
            mp.oneupSHdegree()
        ref_losses = []
        for b in range(B):
            loss = fused_l1_ssim(render(cams[b], ref, pipe, bg)["render"], gts[b], 0.2)
            (loss / B).backward()
            ref_losses.append(float(loss))
        ref.optimizer.step()
        ref.optimizer.zero_grad(set_to_none=True)
        _, losses = sp.step(cams, gts, pipe, bg)
        torch.cuda.synchronize()
        np.testing.assert_allclose([float(l) for l in losses], ref_losses, rtol=3e-5, atol=3e-6, err_msg="iteration %d degrees %r" % (it, deg))
        n = synth.active_sh_coeffs(deg[0], deg[1], False, 4)
        seen = max(seen, n)
        feats = mp.params["_features"].detach()
        want = torch.cat((ref._features_dc.detach(), ref._features_rest.detach()), dim=1)
        # never active so far: untouched, bit for bit, on both sides
        assert torch.equal(feats[:, seen:], init[:, seen:]) and torch.equal(want[:, seen:], init[:, seen:]), (it, deg, seen)
        assert not torch.equal(feats[:, :n], init[:, :n])
        # what has been active: the same trajectory (up to Adam's sign-of-noise steps, tests/test_gpu_api.py)
        perr = (feats[:, :seen] - want[:, :seen]).abs()
        assert (perr > 2e-3).float().mean().item() <= 5e-3 and perr.max().item() <= 0.25, (it, deg, (perr > 2e-3).float().mean().item(), perr.max().item())
    assert seen == 48
    for n_, a in (("_xyz", ref._xyz), ("_opacity", ref._opacity), ("_scaling", ref._scaling), ("_rotation", ref._rotation)):
        perr = (mp.params[n_].detach() - a.detach()).abs()
        assert (perr > 2e-3).float().mean().item() <= 5e-3, (n_, (perr > 2e-3).float().mean().item())


def test_dropin_adam_below_allocated_degree(gpu_device):
    """fdgs.optim.Adam (the one-line swap for torch.optim.Adam) in the reference's loop with M = 48 allocated and (1, 0) active, then
    (2, 0): the deferred SH stages only ever write the active blocks; ``_features_rest`` beyond them stays bit-identical to its
    initial value, as with torch.optim.Adam."""
    from fdgs import train_host
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    cfg = SC("opt", 6000, 208, 160, 1, 0, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=4, alloc=(3, 2), pose="rig2")
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    pipe = train_host.PipelineFlags()
    B = 2
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / B * 10.0) for b in range(B)]
    gen = torch.Generator(device="cpu").manual_seed(7)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in range(B)]
    out = {}
    for which in ("torch", "fdgs"):
        model = train_host.ReferenceStyleModel(scene, gpu_device, optimizer=which)
        losses = []
        for it in range(4):
            if it == 2:
                model.active_sh_degree = 2
            for b in range(B):
                loss = fused_l1_ssim(render(cams[b], model, pipe, bg)["render"], gts[b], 0.2)
                (loss / B).backward()
                losses.append(float(loss))
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        out[which] = (losses, torch.cat((model._features_dc.detach(), model._features_rest.detach()), dim=1).clone())
    np.testing.assert_allclose(out["fdgs"][0], out["torch"][0], rtol=3e-5, atol=3e-6)
    init = scene["shs"].to(gpu_device)
    for which in out:
        assert torch.equal(out[which][1][:, 9:], init[:, 9:]), which
        assert not torch.equal(out[which][1][:, 4:9], init[:, 4:9]), which
    perr = (out["fdgs"][1] - out["torch"][1]).abs()
    assert (perr > 2e-3).float().mean().item() <= 5e-3 and perr.max().item() <= 0.25
