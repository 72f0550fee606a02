"""GPU: the branches of render() no other test takes (gaussian_renderer/__init__.py:73-96, 122-141, 165-186 of the reference):
``pipe.compute_cov3D_python`` (3D, 4D without rot_4d, rot_4d incl. the marginal_t mask and its scatter-back into radii) against
the in-kernel covariance -- the reference-vs-reference cross-check its two code paths imply -- and the
environment-map compositing against a PyTorch statement of the same formulas."""
import math

import numpy as np
import pytest
import torch

from util import synth

pytestmark = pytest.mark.gpu
SC = synth.SceneConfig


class _Pipe:
    compute_cov3D_python = False
    convert_SHs_python = False
    debug = False
    env_map_res = 0


def _setup(cfg, dev, seed=6, prefilter_var=-1.0, pose="axis"):
    from fdgs import train_host
    scene = synth.make_scene(cfg, seed=seed, bg=(0.1, 0.3, 0.2), pose=pose)
    model = train_host.ReferenceStyleModel(scene, dev)
    model.prefilter_var = prefilter_var
    cam = train_host.SyntheticCamera(scene, dev)
    return scene, model, cam


# SH degree 0 where rot_4d: the forward SH direction of the kernel uses the mean it is HANDED (Q4, forward.cu:480-482) -- the
# un-shifted one with the in-kernel covariance, the shifted one on the Python path -- so view-dependent colours differ between the
# reference's own two paths; degree 0 has no direction.
CASES = {
    "dim3_sh2": (SC("b", 6000, 208, 160, 2, 0, 0.03, 1.0, False, 3, False), 1.0, -1.0),
    "dim3_sh1_mod": (SC("b", 6000, 208, 160, 1, 0, 0.03, 1.0, False, 3, False), 1.4, -1.0),
    "dim4_norot_sh1": (SC("b", 6000, 208, 160, 1, 0, 0.03, 1.0, False, 4, True), 1.0, -1.0),
    "dim4_norot_sh1_pf": (SC("b", 6000, 208, 160, 1, 0, 0.03, 1.0, False, 4, True), 1.0, 0.02),
    "rot4d_sh0": (SC("b", 6000, 208, 160, 0, 0, 0.03, 4.0, True, 4, False), 1.0, -1.0),
    # (scaling_modifier stays 1 where a temporal marginal exists: render() calls get_marginal_t(timestamp) WITHOUT the modifier
    # (gaussian_renderer/__init__.py:80) while the kernel scales the temporal axis too (forward.cu:333, 434) -- the reference's two
    # paths disagree there by construction)
    "rot4d_sh0_pf": (SC("b", 6000, 208, 160, 0, 0, 0.03, 4.0, True, 4, False), 1.0, 0.1),
}


def _conjugated(model):
    """The same model with the spatial rotation inverted (quaternion conjugate)."""
    import copy
    other = copy.copy(model)
    with torch.no_grad():
        other._rotation = torch.nn.Parameter(model._rotation.detach() * torch.tensor([1.0, -1.0, -1.0, -1.0], device=model._rotation.device))
    return other


@pytest.mark.parametrize("pose", ["axis", "rig3"])
@pytest.mark.parametrize("name", list(CASES))
def test_python_covariance_matches_kernel_covariance(name, pose, gpu_device):
    """render() with pipe.compute_cov3D_python against render() with the in-kernel covariance.

    rot_4d: the model's get_current_covariance_and_mean_offset (L L^T with L = R4 S, gaussian_model.py:34-47) is the kernel's
    conditional covariance (forward.cu:279-352): same image.
    3D / 4D without rot_4d: the reference's Python covariance is (S R)^T (S R) = R^T S^2 R (gaussian_model.py:28-32 with
    general_utils.py:103-111 ``L = L @ R``) while its kernel computes R S^2 R^T (forward.cu:242-276: glm::mat3's constructor is
    column-major, so its R is the transpose of build_rotation's): the reference's two paths render DIFFERENT images -- the Python
    one rotates every splat with the inverse quaternion.  The branch is therefore checked against the kernel path of the model
    with conjugated rotations, which must give the Python path's image."""
    from fdgs.gaussian_renderer import render
    cfg, mod, pv = CASES[name]
    from fdgs import train_host
    scene = synth.make_scene(cfg, seed=6, bg=(0.1, 0.3, 0.2), pose=pose)   # (rig3: a rotated, off-axis camera: scene/cameras.py:65-71)
    if cfg.gaussian_dim == 4 and not cfg.rot_4d:
        scene["scales_t"] = scene["scales_t"] * 0.03   # (a variance here, forward.cu:431-437) small enough for the 0.05 mask to remove some
    model = train_host.ReferenceStyleModel(scene, gpu_device)
    model.prefilter_var = pv
    cam = train_host.SyntheticCamera(scene, gpu_device)
    bg = scene["bg"].to(gpu_device)
    P = cfg.P
    kernel_model = model if cfg.rot_4d else _conjugated(model)
    a = render(cam, kernel_model, _Pipe(), bg, scaling_modifier=mod)
    pp = _Pipe()
    pp.compute_cov3D_python = True
    b = render(cam, model, pp, bg, scaling_modifier=mod)
    if not cfg.rot_4d:
        c = render(cam, model, _Pipe(), bg, scaling_modifier=mod)
        assert float((c["render"] - b["render"]).abs().max()) > 1e-2, "un-conjugated kernel path equals the Python path: the docstring is wrong"
    # same keys / shapes; radii scattered back to all P Gaussians through the marginal_t mask (gaussian_renderer/__init__.py:178-182)
    assert set(a) == set(b)
    assert b["radii"].shape == (P,) and b["visibility_filter"].shape == (P,) and b["viewspace_points"].shape == (P, 3)
    ra, rb = a["radii"].cpu().numpy(), b["radii"].cpu().numpy()
    if cfg.gaussian_dim == 4:
        m = model.get_marginal_t(cam.timestamp)[:, 0].detach().cpu().numpy()
        assert (rb[m <= 0.05] == 0).all(), "a Gaussian the marginal_t mask removed has a radius"
        assert 0 < (m > 0.05).sum() < P, "the mask removed nothing (or everything): the case does not test the scatter-back"
    # the two covariances differ in the last bits (torch.bmm vs the kernel's GLM-ordered products): a radius = ceil(3 sigma) may
    # flip for a handful of Gaussians, a marginal within rounding of 0.05 may be culled on one side only
    flips = int((ra != rb).sum())
    assert flips <= max(2, P // 1000), "%s: %d radii differ between the two covariance paths" % (name, flips)
    worst = {}
    for k in ("render", "depth", "alpha"):
        tol = 1e-5 * max(1.0, float(a[k].abs().max()))
        d = (a[k] - b[k]).abs()
        frac = float((d > tol).float().mean())
        worst[k] = (float(d.max()), frac)
        # <= 1e-5 (of the output's scale) everywhere except where an alpha >= 1/255 / T >= 1e-4 decision fell the other way (a 1e-6
        # relative change of the conic moves ~1e-4 of the pixels across a threshold; each such pixel moves by <= 1/255 of a colour)
        assert frac <= 2e-3 and worst[k][0] <= 2e3 * tol, "%s: %s differs: max %g, %g of the pixels beyond %g" % (name, k, worst[k][0], frac, tol)
    print(name, "radii flips", flips, {k: "max %.1e, frac beyond 1e-5 of scale %.1e" % v for k, v in worst.items()})
    # gradients flow through the Python covariance (autograd) into every parameter and the screen-space means
    up = torch.from_numpy(np.random.default_rng(0).standard_normal((3, scene["H"], scene["W"])).astype(np.float32)).to(gpu_device) * 1e-2
    params = ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"] + (["_t", "_scaling_t"] if cfg.gaussian_dim == 4 else []) + (
        ["_rotation_r"] if cfg.rot_4d else [])
    grads = {}
    for tag, pkg, mdl in (("kernel", a, kernel_model), ("python", b, model)):
        for n in params:
            getattr(mdl, n).grad = None
        (pkg["render"] * up).sum().backward()
        grads[tag] = {n: (getattr(mdl, n).grad.detach().clone() if getattr(mdl, n).grad is not None else None) for n in params}
        assert pkg["viewspace_points"].grad is not None and torch.isfinite(pkg["viewspace_points"].grad).all()
    for n in params:
        g = grads["python"][n]
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, "python path: no gradient reached %s" % n
    if cfg.gaussian_dim == 3:
        # 3D: the in-kernel covariance backward is the analytic one (computeCov3D, backward.cu:619-700; no quirk on this path), so
        # autograd through the Python covariance must agree with it (rotation: through the conjugation)
        conj = torch.tensor([1.0, -1.0, -1.0, -1.0], device=gpu_device)
        for n in params:
            ga, gb = grads["kernel"][n], grads["python"][n]
            if n == "_rotation":
                ga = ga * conj
            if n == "_scaling":
                # the kernel returns dL/d(mod * scale) as dL/dscale (backward.cu:638-668: S is built from mod * scale, dL_dscale =
                # dot(Rt, dL_dMt) without the modifier) -- the reference's own gradient is short of the factor scale_modifier
                ga = ga * mod
            scale = max(1.0, float(ga.abs().max()))
            d = (ga - gb).abs()
            assert float((d > 1e-4 * scale).float().mean()) <= 2e-3 and float(d.max()) <= 5e-2 * scale, "%s: d/d%s differs between the paths: %g (scale %g)" % (
                name, n, float(d.max()), scale)


@pytest.mark.parametrize("pose", ["axis", "rig1"])
def test_environment_map_compositing(pose, gpu_device):
    """pipe.env_map_res != 0: black background inside the rasterizer, then render + (1 - alpha) * env(ray), the environment
    looked up on a sphere of radius 60 around the origin (gaussian_renderer/__init__.py:41, 165-177)."""
    from fdgs.gaussian_renderer import render
    cfg = SC("e", 3000, 176, 128, 1, 0, 0.03, 1.0, True, 4, True)
    scene, model, cam = _setup(cfg, gpu_device, pose=pose)
    bg = torch.tensor([0.9, 0.8, 0.7], device=gpu_device)   # must be ignored: the rasterizer composites over black
    g = torch.Generator().manual_seed(3)
    model.env_map = torch.rand(3, 32, 64, generator=g).to(gpu_device).requires_grad_(True)
    pe = _Pipe()
    pe.env_map_res = 32
    out = render(cam, model, pe, bg)
    base = render(cam, model, _Pipe(), torch.zeros(3, device=gpu_device))
    assert torch.equal(out["alpha"], base["alpha"]) and torch.equal(out["depth"], base["depth"])
    # PyTorch statement of the lookup, written from the formulas: ray-sphere intersection, spherical coordinates, bilinear sample
    o, d = cam.get_rays()
    assert d.shape == (scene["H"], scene["W"], 3) and torch.allclose(d.norm(dim=-1), torch.ones_like(d[..., 0]), atol=1e-5)
    # the central ray looks down the camera's +z axis: the third column of the camera-to-world rotation (= the third row of the
    # transposed world-to-view matrix the Camera stores, scene/cameras.py:65); (0, 0, 1) for the unrotated camera
    zaxis = scene["world_view_transform"][:3, 2].to(gpu_device)
    assert torch.allclose(d[scene["H"] // 2, scene["W"] // 2], zaxis / zaxis.norm(), atol=2e-2)
    if pose == "axis":
        assert torch.allclose(zaxis, torch.tensor([0.0, 0.0, 1.0], device=gpu_device))
    od, dd, oo = (o * d).sum(-1), (d * d).sum(-1), (o * o).sum(-1)
    t = -od + torch.sqrt(od ** 2 - dd * (oo - 60.0 ** 2)) / dd
    x = o + d * t.unsqueeze(-1)
    u = torch.atan2(x[..., 1], x[..., 0]) / (2 * math.pi) + 0.5
    v = torch.acos(x[..., 2] / 60.0) / math.pi
    grid = torch.stack([u, v], dim=-1) * 2 - 1
    env = torch.nn.functional.grid_sample(model.env_map[None], grid[None])[0]
    want = base["render"] + (1 - base["alpha"]) * env
    assert float((out["render"] - want).abs().max()) <= 1e-6
    assert float(((1 - base["alpha"]) * env).abs().max()) > 0.05, "the environment is invisible: the case does not test it"
    out["render"].sum().backward()
    assert model.env_map.grad is not None and float(model.env_map.grad.abs().sum()) > 0
