"""GPU: the drop-in Python surface (render(), GaussianRasterizer, markVisible) and size-independent
properties of the full-size C3 workload."""
import numpy as np
import pytest
import torch

from util import run_hip, run_oracle, synth

pytestmark = pytest.mark.gpu


def _model_cam(scene, dev):
    from fdgs import train_host
    return train_host.GaussianParams(scene, dev), train_host.SyntheticCamera(scene, dev), train_host.PipelineFlags()


def test_render_dict_and_gradients(gpu_device):
    """render() returns the reference's 7 keys with the reference's shapes / dtypes; gradients reach every
    parameter and viewspace_points (the densification statistic, train.py:164)."""
    from fdgs.gaussian_renderer import render
    cfg = synth.SceneConfig("api", 5000, 200, 152, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=4)
    model, cam, pipe = _model_cam(scene, gpu_device)
    pkg = render(cam, model, pipe, scene["bg"].to(gpu_device))
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "depth", "alpha", "flow"}
    H, W, P = scene["H"], scene["W"], 5000
    assert pkg["render"].shape == (3, H, W) and pkg["depth"].shape == (1, H, W) and pkg["alpha"].shape == (1, H, W)
    assert pkg["flow"].shape == (2, H, W) and pkg["radii"].shape == (P,) and pkg["radii"].dtype == torch.int32
    assert pkg["visibility_filter"].dtype == torch.bool and pkg["viewspace_points"].shape == (P, 3)
    # the oracle must see exactly what the rasterizer saw: the model's post-activation tensors
    osc = dict(scene)
    osc.update(means3D=model.get_xyz.detach().cpu(), opacities=model.get_opacity.detach().cpu(),
               scales=model.get_scaling.detach().cpu(), rotations=model.get_rotation.detach().cpu(),
               scales_t=model.get_scaling_t.detach().cpu(), ts=model.get_t.detach().cpu(),
               rotations_r=model.get_rotation_r.detach().cpu(), shs=model.get_features.detach().cpu())
    ref, _ = run_oracle(osc, None, kind="port")
    assert ref["border_g"].sum() == 0
    ok = ~ref["border"].astype(bool)
    assert np.array_equal(pkg["radii"].cpu().numpy(), ref["radii"])
    assert np.abs(pkg["render"].detach().cpu().numpy() - ref["out_color"])[:, ok].max() <= 1e-4
    assert np.abs(pkg["alpha"].detach().cpu().numpy()[0] - (1.0 - ref["out_T"]))[ok].max() <= 1e-4
    (pkg["render"].mean() + 0.1 * pkg["depth"].mean() + 0.1 * pkg["alpha"].mean()).backward()
    for name, p in model.params.items():
        assert torch.isfinite(p.grad).all(), name
        assert p.grad.abs().sum() > 0, "no gradient reached %s" % name
    vs = pkg["viewspace_points"].grad
    assert vs is not None and vs.shape == (P, 3) and vs[:, :2].abs().sum() > 0


def test_render_python_sh_branch_matches_kernel_sh(gpu_device):
    """pipe.convert_SHs_python must give the same image as in-kernel SH for 3D SH (the cross-check the reference's
    two code paths imply); uses a gaussian_dim == 3 scene so both use the same view direction."""
    from fdgs.gaussian_renderer import render
    cfg = synth.SceneConfig("api", 3000, 160, 120, 3, 0, 0.03, 1.0, False, 3, False)
    scene = synth.make_scene(cfg, seed=9)
    model, cam, pipe = _model_cam(scene, gpu_device)
    model.get_current_covariance_and_mean_offset = lambda s, t: (None, torch.zeros_like(model.get_xyz))
    bg = scene["bg"].to(gpu_device)
    a = render(cam, model, pipe, bg)["render"]
    pipe2 = type(pipe)()
    pipe2.convert_SHs_python = True
    b = render(cam, model, pipe2, bg)["render"]
    assert (a - b).abs().max().item() <= 1e-4


def test_mark_visible(gpu_device):
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import pyoracle
    scene = synth.make_scene(synth.SceneConfig("api", 999, 64, 64, 0, 0, 0.03, 1.0, True, 4, True), seed=2)
    scene["means3D"][::4, 2] = -4.5
    rs = GaussianRasterizationSettings(64, 64, scene["tanfovx"], scene["tanfovy"], scene["bg"].to(gpu_device), 1.0,
                                       scene["world_view_transform"].to(gpu_device),
                                       scene["full_proj_transform"].to(gpu_device), 0, 0,
                                       scene["camera_center"].to(gpu_device), 0.5, 1.0, True, 4, True, False, False)
    vis = GaussianRasterizer(rs).markVisible(scene["means3D"].to(gpu_device))
    ref = pyoracle.mark_visible(scene["means3D"], scene["world_view_transform"], scene["full_proj_transform"])
    assert vis.dtype == torch.bool and np.array_equal(vis.cpu().numpy(), ref)


def test_c3_full_size_properties(gpu_device):
    """BASELINE configs[2] at full size (300 k Gaussians, 1352x1014, M = 48), checked through size-independent
    properties: sum(tiles_touched) == R, ranges partition [0, R), every tile's slice is sorted by
    (depth bits, Gaussian id), the tile of every instance lies inside its Gaussian's rectangle count, T in [0, 1],
    alpha == 1 - T, forward integer outputs are run-to-run deterministic, and gradients are finite."""
    scene = synth.make_scene(synth.CONFIGS["C3"], seed=0)
    up = synth.make_upstream_grads(scene["W"], scene["H"], seed=1, scale=1e-2)
    a, ag = run_hip(scene, gpu_device, up)
    b, _ = run_hip(scene, gpu_device, None)
    R = a["R"]
    assert R == int(a["tiles_touched"].sum()) and R > 2_000_000
    for k in ("radii", "tiles_touched", "point_list", "ranges", "n_contrib"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k + " not deterministic")
    rg = a["ranges"].astype(np.int64)
    nonempty = rg[:, 1] > rg[:, 0]
    starts = rg[nonempty]
    order = np.argsort(starts[:, 0])
    starts = starts[order]
    assert starts[0, 0] == 0 and starts[-1, 1] == R and np.array_equal(starts[1:, 0], starts[:-1, 1])
    tile_of = a["tile_keys"].astype(np.int64)
    assert np.all(np.diff(tile_of) >= 0)
    depth_bits = a["depths"].view(np.uint32)[a["point_list"]].astype(np.int64)
    key2 = (depth_bits << 32) | a["point_list"].astype(np.int64)
    same_tile = np.diff(tile_of) == 0
    assert np.all(np.diff(key2)[same_tile] > 0), "a tile slice is not sorted by (depth, id)"
    counts = np.bincount(a["point_list"], minlength=scene["means3D"].shape[0])
    np.testing.assert_array_equal(counts.astype(np.uint32), a["tiles_touched"])
    assert a["out_T"].min() >= 0.0 and a["out_T"].max() <= 1.0
    assert (a["n_contrib"].astype(np.int64) <= (rg[:, 1] - rg[:, 0]).max()).all()
    for k, v in ag.items():
        assert np.isfinite(v).all(), k
    # culled Gaussians get exactly zero gradient
    culled = a["radii"] <= 0
    for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dsh", "dL_dopacity"):
        assert np.abs(ag[k][culled]).max() == 0.0, k


class _ActivatedModel:
    """Duck-typed GaussianModel whose getters return fixed post-activation leaf tensors."""

    def __init__(self, model, act):
        op, sc, sct, rot, rotr = act
        self._t = {"xyz": model._xyz.detach().clone(), "feat": model._features.detach().clone(), "t": model._t.detach().clone(),
                   "op": op.clone(), "sc": sc.clone(), "sct": sct.clone(), "rot": rot.clone(), "rotr": rotr.clone()}
        for t in self._t.values():
            t.requires_grad_(True)
        for k in ("active_sh_degree", "active_sh_degree_t", "time_duration", "rot_4d", "gaussian_dim", "force_sh_3d",
                  "prefilter_var", "env_map", "get_max_sh_channels"):
            setattr(self, k, getattr(model, k))

    get_xyz = property(lambda s: s._t["xyz"])
    get_features = property(lambda s: s._t["feat"])
    get_t = property(lambda s: s._t["t"])
    get_opacity = property(lambda s: s._t["op"])
    get_scaling = property(lambda s: s._t["sc"])
    get_scaling_t = property(lambda s: s._t["sct"])
    get_rotation = property(lambda s: s._t["rot"])
    get_rotation_r = property(lambda s: s._t["rotr"])


def test_render_raw_matches_render(gpu_device):
    """Fused activations (fdgs_scene.raw_params): render_raw on RAW parameters against render() on the ACTIVATED
    tensors the kernels derive themselves (fdgs_debug_activations, bit-identical to the in-flight values): the forward
    must then be bit-identical, and the raw-parameter gradients must equal the float64 chain rule of the
    reference's activations (scene/gaussian_model.py:55-66) applied to render()'s gradients to 1e-4 of the tensor
    scale -- the same bar as every other gradient test -- with and without a gradient sink."""
    from fdgs import _capi, train_host
    from fdgs.fused import render_raw
    from fdgs.gaussian_renderer import render
    cfg = synth.SceneConfig("raw", 6000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=6)
    bg = torch.tensor([0.2, 0.1, 0.3], device=gpu_device)
    up = synth.make_upstream_grads(scene["W"], scene["H"], seed=3, scale=1e-2)
    wc, wd, wa = up["grad_color"].to(gpu_device), up["grad_depth"].to(gpu_device), up["grad_alpha"].to(gpu_device)

    def loss_of(pkg):
        return (pkg["render"] * wc).sum() + (pkg["depth"] * wd).sum() + (pkg["alpha"] * wa).sum()

    base, cam, pipe = _model_cam(scene, gpu_device)
    act = _capi.debug_activations(base._opacity.detach(), base._scaling.detach(), base._scaling_t.detach(),
                                  base._rotation.detach(), base._rotation_r.detach())
    # the kernel's activations agree with PyTorch's to an ulp or two
    assert (act[0] - torch.sigmoid(base._opacity.detach())).abs().max().item() <= 2e-7
    assert ((act[1] - torch.exp(base._scaling.detach())).abs() / act[1]).max().item() <= 3e-7
    assert (act[3] - torch.nn.functional.normalize(base._rotation.detach())).abs().max().item() <= 2e-7
    am = _ActivatedModel(base, act)
    ref = render(cam, am, pipe, bg)
    loss_of(ref).backward()
    f64 = lambda t: t.detach().double()  # noqa: E731
    o = f64(am._t["op"])
    want = {"_xyz": f64(am._t["xyz"].grad), "_features": f64(am._t["feat"].grad), "_t": f64(am._t["t"].grad),
            "_opacity": f64(am._t["op"].grad) * o * (1 - o), "_scaling": f64(am._t["sc"].grad) * f64(am._t["sc"]),
            "_scaling_t": f64(am._t["sct"].grad) * f64(am._t["sct"])}
    for name, key in (("_rotation", "rot"), ("_rotation_r", "rotr")):
        q, g = f64(am._t[key]), f64(am._t[key].grad)
        inv = 1.0 / f64(base.params[name]).norm(dim=1, keepdim=True).clamp_min(1e-12)
        want[name] = (g - q * (q * g).sum(1, keepdim=True)) * inv

    for use_sink in (False, True):
        model = train_host.GaussianParams(scene, gpu_device)
        model.flat_grad.fill_(float("nan") if use_sink else 0.0)  # a sink must be fully overwritten
        pkg = render_raw(cam, model, pipe, bg, grad_sink=model.grad_sink() if use_sink else None)
        assert torch.equal(pkg["radii"], ref["radii"])
        for k in ("render", "depth", "alpha", "flow"):
            assert torch.equal(pkg[k], ref[k]), k   # same kernels, bit-identical inputs
        loss_of(pkg).backward()
        assert torch.isfinite(model.flat_grad).all()
        for name in model.NAMES:
            g, r = model.params[name].grad.double(), want[name].reshape(model.params[name].shape)
            scale = max(1.0, r.abs().max().item())
            err = (g - r).abs().max().item()
            print("raw sink=%s %s: max abs err %.2e (max|ref| %.1e)" % (use_sink, name, err, scale))
            assert err <= 1e-4 * scale, (name, use_sink, err, scale)
        assert pkg["viewspace_points"].grad is not None


def test_c5_stress_size_properties(gpu_device):
    """BASELINE configs[4]: 2 M Gaussians at 2704x2028 (R ~ 16 M instances, 21 463 tiles): the maximum-size case,
    checked through the same size-independent properties as C3 (forward only) plus an oracle spot check of the
    per-Gaussian integers on a 20 k subset rendered on its own."""
    scene = synth.make_scene(synth.CONFIGS["C5"], seed=0)
    a, _ = run_hip(scene, gpu_device, None)
    R = a["R"]
    assert R == int(a["tiles_touched"].sum()) and R > 10_000_000
    rg = a["ranges"].astype(np.int64)
    starts = rg[rg[:, 1] > rg[:, 0]]
    starts = starts[np.argsort(starts[:, 0])]
    assert starts[0, 0] == 0 and starts[-1, 1] == R and np.array_equal(starts[1:, 0], starts[:-1, 1])
    tile_of = a["tile_keys"].astype(np.int64)
    assert np.all(np.diff(tile_of) >= 0) and tile_of.max() < rg.shape[0]
    depth_bits = a["depths"].view(np.uint32)[a["point_list"]].astype(np.int64)
    key2 = (depth_bits << 32) | a["point_list"].astype(np.int64)
    assert np.all(np.diff(key2)[np.diff(tile_of) == 0] > 0)
    np.testing.assert_array_equal(np.bincount(a["point_list"], minlength=2_000_000).astype(np.uint32), a["tiles_touched"])
    assert a["out_T"].min() >= 0.0 and a["out_T"].max() <= 1.0 and np.isfinite(a["out_color"]).all()
    # per-Gaussian integers do not depend on the other Gaussians: the oracle on a subset must agree exactly
    sub = {k: (v[:20000].clone() if isinstance(v, torch.Tensor) and v.shape[:1] == (2_000_000,) else v) for k, v in scene.items()}
    ref, _ = run_oracle(sub, None, kind="port")
    assert ref["border_g"].sum() == 0
    np.testing.assert_array_equal(a["radii"][:20000], ref["radii"])
    np.testing.assert_array_equal(a["tiles_touched"][:20000], ref["tiles_touched"])
    vis = ref["radii"] > 0
    np.testing.assert_array_equal(a["depths"][:20000][vis].view(np.uint32), ref["depths"][vis].view(np.uint32))


def test_gradient_accumulation_over_views(gpu_device):
    """accumulate=True (fdgs_backward_out.accumulate): two views accumulated into the sink equal the sum of the two
    views' separately computed gradients (the reference sums loss / batch_size over the batch, train.py:104-166)."""
    from fdgs import train_host
    from fdgs.fused import render_raw
    cfg = synth.SceneConfig("acc", 5000, 176, 144, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=8)
    bg = torch.zeros(3, device=gpu_device)
    pipe = train_host.PipelineFlags()
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=t) for t in (3.0, 7.0)]
    w = synth.make_upstream_grads(scene["W"], scene["H"], seed=5, scale=1e-2)["grad_color"].to(gpu_device)

    singles = []
    for cam in cams:
        m = train_host.GaussianParams(scene, gpu_device)
        (render_raw(cam, m, pipe, bg, grad_sink=m.grad_sink())["render"] * w).sum().backward()
        singles.append(m.flat_grad.clone())
    m = train_host.GaussianParams(scene, gpu_device)
    m.flat_grad.fill_(float("nan"))
    sink = m.grad_sink()
    for v, cam in enumerate(cams):
        (render_raw(cam, m, pipe, bg, grad_sink=sink, accumulate=v > 0)["render"] * w).sum().backward()
    want = singles[0] + singles[1]
    assert torch.isfinite(m.flat_grad).all()
    err = (m.flat_grad - want).abs()
    scale = max(1.0, want.abs().max().item())
    assert (err > 1e-4 * scale).float().mean().item() <= 1e-4 and err.max().item() <= 1e-2 * scale, err.max().item()


@pytest.mark.parametrize("D,D_t,sh3d", [(3, 2, False), (3, 1, False), (3, 0, False), (2, 0, True), (0, 0, True)])
def test_deferred_sh_gradient_matches_accumulation(gpu_device, D, D_t, sh3d):
    """fdgs_backward_out.sh_stage + fdgs_sh_flush (one write of dL_dsh per optimizer step) against backward calls that
    accumulate dL_dsh view after view.  The flush performs the same additions in the same order; what differs between the
    two runs compared here is only the run-to-run noise of the blend backward's float atomics in dL_dRGB (1e-7 relative),
    the same for every other gradient, which the mode does not touch."""
    from fdgs import _capi, train_host
    from fdgs.fused import raw_backward, raw_forward, raw_settings
    cfg = synth.SceneConfig("defer", 7000, 208, 160, D, D_t, 0.03, 10.0, True, 4, sh3d)
    scene = synth.make_scene(cfg, seed=21)
    bg = torch.zeros(3, device=gpu_device)
    pipe = train_host.PipelineFlags()
    B = 3
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
    ups = [synth.make_upstream_grads(scene["W"], scene["H"], seed=30 + b, scale=1e-2)["grad_color"].to(gpu_device) for b in range(B)]

    def run(deferred):
        m = train_host.GaussianParams(scene, gpu_device)
        m.flat_grad.fill_(float("nan"))
        sink = m.grad_sink()
        stage = torch.full((B, m.P, 8), float("nan"), device=gpu_device) if deferred else None
        for b, cam in enumerate(cams):
            rs, (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = raw_settings(cam, m, pipe, bg)
            (R, color, flow, depth, T, radii, geom, binb, img, _c, om) = raw_forward(rs, xyz, feats, opacity, ts, scaling, scaling_t,
                                                                                  rotation, rotation_r, pv)
            raw_backward(rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img,
                         ups[b], None, None, None, sink, b > 0, sh_stage=stage[b] if deferred else None)
        if deferred:
            _capi.sh_flush(stage, sink["dL_dsh"], rs.sh_degree, rs.sh_degree_t, rs.gaussian_dim, rs.force_sh_3d)
        torch.cuda.synchronize()
        return m
    a, d = run(False), run(True)
    assert torch.isfinite(d.flat_grad).all()
    ga, gd = a.params["_features"].grad, d.params["_features"].grad
    assert ga.abs().sum() > 0 and (ga != 0).float().mean().item() > 0.05
    assert torch.equal(ga == 0, gd == 0)        # the same Gaussians / coefficients receive a gradient
    for n in a.NAMES:
        s = max(1.0, a.params[n].grad.abs().max().item())
        err = (a.params[n].grad - d.params[n].grad).abs().max().item()
        assert err <= (1e-5 if n == "_features" else 1e-4) * s, (n, err, s)   # the others: atomics noise through the covariance chain


@pytest.mark.parametrize("D,D_t,M,sh3d,analytic", [(3, 2, 48, False, False), (3, 1, 48, False, True), (3, 0, 16, False, False),
                                                   (2, 0, 16, True, False), (0, 0, 16, True, False), (3, 2, 64, False, False),
                                                   (1, 0, 4, True, False), (3, 2, 16, False, False)])
def test_sh_adam_fused_is_flush_plus_adam(gpu_device, D, D_t, M, sh3d, analytic):
    """fdgs_adam_step_sh (SH gradient built from the staged views and consumed by Adam in one kernel) against fdgs_sh_flush
    followed by fdgs_adam_step on the same stages: gradient, parameters and both moments bit-identical -- over two steps, with
    Gaussians no view touched (g = 0: the moments still decay), inactive coefficient blocks and the DC learning rate."""
    from fdgs import _capi
    P, B = 4 * 1237, 3
    gen = torch.Generator(device="cpu").manual_seed(100 + D + 7 * D_t + M)
    stages = torch.randn(B, P, 8, generator=gen)
    stages[:, :, 3] = torch.rand(B, P, generator=gen) * 2 - 1    # cosine factors of the two time blocks
    stages[:, :, 7] = torch.rand(B, P, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(stages[:, :, 4:7], dim=-1)
    stages[:, :, 4:7] = d
    dead = torch.rand(B, P, generator=gen) < 0.4                 # views that did not touch the Gaussian
    stages[:, :, 0:3][dead] = 0.0
    stages = stages.to(gpu_device).contiguous()
    lr, lr_dc, b1, b2, eps = 1.25e-4, 2.5e-3, 0.9, 0.999, 1e-15

    def fresh():
        g = torch.Generator(device="cpu").manual_seed(5)
        p = torch.randn(P, M, 3, generator=g).to(gpu_device)
        m = (0.01 * torch.randn(P, M, 3, generator=g)).to(gpu_device)
        v = (1e-4 * torch.rand(P, M, 3, generator=g)).to(gpu_device)
        return p, m, v
    pa, ma, va = fresh()
    pf, mf, vf = fresh()
    ga = torch.full((P, M, 3), float("nan"), device=gpu_device)
    gf = torch.full((P, M, 3), float("nan"), device=gpu_device)
    seg = (_capi.FdgsAdamSegment * 1)(_capi.FdgsAdamSegment(0, P * M * 3, lr, lr_dc, M * 3, 3))
    for step in (1, 2):
        _capi.sh_flush(stages, ga, D, D_t, 3 if sh3d else 4, sh3d, analytic)
        rc = _capi.lib.fdgs_adam_step(pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), P * M * 3, seg, 1, b1, b2, eps, step,
                                      _capi.current_stream_handle(gpu_device))
        assert rc == 0
        assert _capi.adam_step_sh(pf, mf, vf, stages, D, D_t, 3 if sh3d else 4, sh3d, analytic, lr, lr_dc, b1, b2, eps, step, dL_dsh=gf)
        torch.cuda.synchronize()
        assert torch.isfinite(ga).all() and torch.equal(ga, gf)
        assert (ga != 0).float().mean().item() > 0.01
        assert torch.equal(ma, mf) and torch.equal(va, vf) and torch.equal(pa, pf)
    # without the gradient output; and a layout it refuses (row not a multiple of 16 coefficients) is reported, not half done
    pg, mg, vg = fresh()
    for step in (1, 2):
        assert _capi.adam_step_sh(pg, mg, vg, stages, D, D_t, 3 if sh3d else 4, sh3d, analytic, lr, lr_dc, b1, b2, eps, step)
    torch.cuda.synchronize()
    assert torch.equal(pg, pf) and torch.equal(mg, mf) and torch.equal(vg, vf)
    podd = torch.zeros(P, 1, 3, device=gpu_device)
    assert not _capi.adam_step_sh(podd, podd.clone(), podd.clone(), stages, 0, 0, 3, True, False, lr, lr_dc, b1, b2, eps, 1)


@pytest.mark.parametrize("overlap,fuse,B", [(True, True, 3), (False, True, 3), (True, False, 3), (True, True, 1), (True, False, 1)])
@pytest.mark.parametrize("batch,group", [(True, 1), (True, 4), (False, 2), (False, 1)], ids=["batched-colours", "all-batched", "sh-pairs", "per-view"])
def test_step_pipeline_matches_autograd_step(gpu_device, overlap, fuse, B, batch, group):
    """fdgs.pipeline.StepPipeline (explicit forward / fused loss / backward on two HIP streams, no autograd) performs
    the same optimizer step as render_raw + fused_l1_ssim + backward() + Adam on one stream."""
    from fdgs import train_host
    from fdgs.fused import render_raw
    from fdgs.loss import fused_l1_ssim
    from fdgs.pipeline import StepPipeline
    cfg = synth.SceneConfig("pipe", 6000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=4)
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    pipe = train_host.PipelineFlags()
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
    gen = torch.Generator(device="cpu").manual_seed(7)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in range(B)]

    ma = train_host.GaussianParams(scene, gpu_device)
    oa = train_host.make_optimizer(ma)
    sink = ma.grad_sink()
    ref_losses = []
    for _ in range(2):  # two optimizer steps
        for b in range(B):
            loss = fused_l1_ssim(render_raw(cams[b], ma, pipe, bg, grad_sink=sink, accumulate=b > 0)["render"], gts[b], 0.2)
            (loss / B).backward()
            ref_losses.append(float(loss))
        oa.step()

    mp = train_host.GaussianParams(scene, gpu_device)
    sp = StepPipeline(mp, train_host.make_optimizer(mp), world_size=1, lambda_dssim=0.2, overlap=overlap, fuse_sh_adam=fuse, batch_views=batch, sh_group=group)
    got_losses = []
    for _ in range(2):
        results, losses = sp.step(cams, gts, pipe, bg)
        got_losses += [float(l) for l in losses]
        assert len(results) == B and results[0]["render"].shape == (3, scene["H"], scene["W"])
    torch.cuda.synchronize()
    np.testing.assert_allclose(got_losses, ref_losses, rtol=3e-5, atol=1e-6)   # (second step: behind one Adam step's float-atomics noise)
    # the gradients of the last step (float atomics: summation order differs run to run) and the parameters after two steps
    # (fuse_sh_adam: the SH gradient goes from the views' stages straight into Adam and is not materialised in the bucket)
    n_cmp = mp.offsets["_features"][0] if fuse else mp.flat.numel()
    gerr = (mp.flat_grad[:n_cmp] - ma.flat_grad[:n_cmp]).abs()
    gscale = max(1e-6, ma.flat_grad.abs().max().item())
    assert gerr.max().item() <= 1e-3 * gscale, (gerr.max().item(), gscale)
    # Adam's first steps move every parameter by ~lr whatever the gradient's magnitude: where a gradient is float-atomics noise
    # around zero (more of them with a single view) its sign, hence 2 lr per step, differs between two runs
    perr = (mp.flat - ma.flat).abs()
    assert (perr > 2e-3).float().mean().item() <= 2e-3 and perr.max().item() <= 0.25, ((perr > 2e-3).float().mean().item(), perr.max().item())


@pytest.mark.parametrize("cfg", [synth.SceneConfig("b4", 20000, 320, 240, 3, 2, 0.03, 10.0, True, 4, False),
                                 synth.SceneConfig("b3", 9000, 208, 160, 2, 0, 0.03, 1.0, False, 3, True),
                                 synth.SceneConfig("b0", 5000, 160, 128, 0, 0, 0.03, 1.0, True, 4, True)], ids=["4d-sh", "3d-sh", "deg0"])
def test_preprocess_batch_forward_is_bit_identical(cfg, gpu_device):
    """fdgs_preprocess_batch: the geometry of every view per view, the SH colours of all views in ONE pass over the coefficients;
    every forward output and every introspected buffer of every view equals the per-view forward bit for bit (same arithmetic in
    the same order).  9 views: two launches of the colour kernel (8 views per launch)."""
    from fdgs import train_host
    from fdgs.fused import raw_forward, raw_preprocess_batch, raw_settings
    from util import collect_forward
    scene = synth.make_scene(cfg, seed=31)
    model = train_host.GaussianParams(scene, gpu_device)
    pipe = train_host.PipelineFlags()
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    nv = 9
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / nv * scene["time_duration"]) for b in range(nv)]
    for b, c in enumerate(cams):   # different cameras too: shift the view and camera centre a little per view
        wv = c.world_view_transform.clone()
        wv[3, 0] += 0.03 * b
        wv[3, 1] -= 0.02 * b
        c.world_view_transform = wv
        c.full_proj_transform = wv @ (torch.linalg.inv(scene["world_view_transform"].to(gpu_device)) @ scene["full_proj_transform"].to(gpu_device))
        c.camera_center = torch.linalg.inv(wv)[3, :3].contiguous()
    P, W, H = model.P, scene["W"], scene["H"]
    sets = [raw_settings(c, model, pipe, bg) for c in cams]
    tens = sets[0][1]
    single = [collect_forward(raw_forward(rs, *tens), P, W, H) for rs, _ in sets]
    handles = raw_preprocess_batch([rs for rs, _ in sets], *tens)
    batched = [collect_forward(raw_forward(rs, *tens, preprocessed=handles[b]), P, W, H) for b, (rs, _) in enumerate(sets)]
    for b in range(nv):
        for key in ("R", "out_color", "out_flow", "out_depth", "out_T", "radii", "out_means3D", "covs_com", "n_contrib", "final_T", "point_list",
                    "ranges", "rgb", "clamped_bits", "conic_opacity", "means2D", "rec_depth", "rec_flow", "depths", "cov3D", "tiles_touched"):
            np.testing.assert_array_equal(single[b][key], batched[b][key], err_msg="view %d %s" % (b, key))
    assert single[0]["R"] > 0 and (single[0]["rgb"] != 0).any()
    assert not np.array_equal(single[0]["out_color"], single[nv - 1]["out_color"])   # the views do differ


@pytest.mark.parametrize("cfg", [synth.SceneConfig("s4", 20000, 320, 240, 3, 2, 0.03, 10.0, True, 4, False),
                                 synth.SceneConfig("s3", 9000, 208, 160, 2, 0, 0.03, 1.0, False, 3, True),
                                 synth.SceneConfig("s1", 6000, 160, 128, 3, 1, 0.03, 2.0, True, 4, False)], ids=["4d-sh-t2", "3d-sh", "4d-sh-t1"])
def test_sh_backward_batch_matches_per_view(cfg, gpu_device):
    """fdgs_sh_backward_batch: blend backward per view (stage_mask 5), ONE SH backward pass for all views, geometry backward per
    view -- against the per-view backward with the deferred SH gradient.  The two runs differ only in the order of the blend
    backward's float atomics, so: stage records and every gradient equal to 1e-5 of their scale; where a view's dL_dRGB is exactly
    zero (Gaussian not live) both leave a zero stage record."""
    from fdgs import _capi, train_host
    from fdgs.fused import raw_backward, raw_forward, raw_settings
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import _C, analytic_sh_gradients
    scene = synth.make_scene(cfg, seed=33)
    model = train_host.GaussianParams(scene, gpu_device)
    pipe = train_host.PipelineFlags()
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    nv = 5   # odd: the two view parities of the kernel get 3 and 2 views
    P, W, H = model.P, scene["W"], scene["H"]
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / nv * scene["time_duration"]) for b in range(nv)]
    gen = torch.Generator(device="cpu").manual_seed(5)
    ups = [(torch.randn(3, H, W, generator=gen) * 1e-2).to(gpu_device) for _ in range(nv)]
    sets = [raw_settings(c, model, pipe, bg) for c in cams]
    tens = sets[0][1]
    (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = tens
    fwd = [raw_forward(rs, *tens) for rs, _ in sets]

    def run(batched):
        sink = {k: torch.zeros_like(v) for k, v in model.grad_sink().items()}
        stage = torch.full((nv, P, 8), float("nan"), device=gpu_device)
        gacc = torch.zeros((nv, P, 16), device=gpu_device)
        per_view, pend = [], []
        for b, (rs, _) in enumerate(sets):
            (R, color, flow, depth, T, radii, geom, binb, img, _covs, om) = fwd[b]
            args = (rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img, ups[b], None, None, None,
                    sink, b > 0)
            if batched:
                pend.append(raw_backward(*args, grad_accum=gacc[b], sh_stage=stage[b], begin_only=True))
            else:
                per_view.append(raw_backward(*args, grad_accum=gacc[b], sh_stage=stage[b]))
        if batched:
            _C.sh_backward_batch(pend)
            per_view = [_C.backward_finish(p) for p in pend]
        torch.cuda.synchronize()
        assert float(gacc.abs().max()) == 0.0   # every view's accumulator is left all zero
        dsh = torch.empty((P, model.M, 3), device=gpu_device)
        _capi.sh_flush(stage, dsh, sets[0][0].sh_degree, sets[0][0].sh_degree_t, sets[0][0].gaussian_dim, sets[0][0].force_sh_3d, analytic_sh_gradients())
        torch.cuda.synchronize()
        return stage, sink, per_view, dsh

    st_a, sink_a, pv_a, dsh_a = run(False)
    st_b, sink_b, pv_b, dsh_b = run(True)
    live_a, live_b = (st_a[:, :, :3] != 0).any(2), (st_b[:, :, :3] != 0).any(2)
    assert torch.equal(live_a, live_b) and live_a.any() and not live_a.all()
    # a live record: 8 numbers; a dead one: only its first four are written (zeros) -- compare what is defined
    a = torch.where(live_a[..., None], st_a, torch.zeros_like(st_a))
    b = torch.where(live_b[..., None], st_b, torch.zeros_like(st_b))
    sc = a.abs().max().item()
    assert (a - b).abs().max().item() <= 1e-5 * sc, ((a - b).abs().max().item(), sc)
    assert float(st_a[:, :, :4][~live_a].abs().max()) == 0.0 and float(st_b[:, :, :4][~live_b].abs().max()) == 0.0
    for k in sink_a:
        if k == "dL_dsh":
            continue   # deferred: built by the flush below
        # the two runs sum the blend backward's float atomics in different orders; the covariance-chain gradients amplify that
        # rounding noise (cancelling sums, see test_gpu_parity._timed_path_vs_oracle): 3e-4 of the tensor scale
        sc = max(1e-6, sink_a[k].abs().max().item())
        err = (sink_a[k] - sink_b[k]).abs().max().item()
        # (observed up to 3.2e-4 on dL_dscales_t -- run-to-run noise, not a property of either path: 1e-3 for the four covariance-chain tensors)
        chain = k in ("dL_dscales", "dL_dscales_t", "dL_drotations", "dL_drotations_r")
        assert err <= (1e-3 if chain else 3e-4) * sc, (k, err, sc)
    sc = dsh_a.abs().max().item()
    assert (dsh_a - dsh_b).abs().max().item() <= 1e-5 * sc
    for va, vb in zip(pv_a, pv_b):   # per-view outputs: viewspace gradient, colour gradient, covariance gradient
        for i in (0, 1, 4):
            sc = max(1e-6, va[i].abs().max().item())
            assert (va[i] - vb[i]).abs().max().item() <= 1e-4 * sc, i


def test_spatial_sort_renders_the_same_images(gpu_device):
    """The order in which the model stores its Gaussians is not part of its semantics: after train_host.spatial_sort the forward
    gives the same image (contributions are blended in depth order; only exact depth ties could change their order), the per-Gaussian
    outputs are the permuted ones, and one optimizer step moves every Gaussian as before."""
    from fdgs import train_host
    from fdgs.fused import raw_forward, raw_settings
    from fdgs.pipeline import StepPipeline
    scene = synth.make_scene(synth.SceneConfig("so", 20000, 320, 240, 3, 2, 0.03, 10.0, True, 4, False), seed=8)
    pipe = train_host.PipelineFlags()
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / 2 * scene["time_duration"]) for b in range(2)]
    gen = torch.Generator(device="cpu").manual_seed(3)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in range(2)]
    ma, mb = train_host.GaussianParams(scene, gpu_device), train_host.GaussianParams(scene, gpu_device)
    oa, ob = train_host.make_optimizer(ma), train_host.make_optimizer(mb)
    perm = train_host.spatial_sort(mb, ob)
    outs = []
    for m in (ma, mb):
        rs, tens = raw_settings(cams[0], m, pipe, bg)
        outs.append(raw_forward(rs, *tens))
    (Ra, ca, fa, da, Ta, ra, *_), (Rb, cb, fb, db, Tb, rb, *_) = outs
    assert Ra == Rb and torch.equal(rb, ra[perm])
    assert (ca - cb).abs().max().item() <= 1e-6 and (Ta - Tb).abs().max().item() <= 1e-6 and (da - db).abs().max().item() <= 1e-5
    for m, o in ((ma, oa), (mb, ob)):
        StepPipeline(m, o, world_size=1, lambda_dssim=0.2).step(cams, gts, pipe, bg)
    torch.cuda.synchronize()
    for n in ma.NAMES:
        a, b = ma.params[n].detach()[perm], mb.params[n].detach()
        # Adam's first step moves a parameter by ~lr whatever the gradient's size: where a gradient is float-atomics noise around
        # zero its sign may differ between two runs (2 lr)
        assert ((a - b).abs() > 2e-3).float().mean().item() <= 2e-3, n


def test_step_pipeline_lazy_equals_waiting_and_redoes_an_overflowing_step(gpu_device):
    """StepPipeline(lazy=True) -- no forward of the step waits for num_rendered, one look at the reports before the last view's
    backward -- takes the same optimizer steps as lazy=False; a step whose views outgrow the run-ahead buffers is redone through the
    waiting path before anything of the optimizer step has been enqueued (lazy_redone counts it) and still equals the waiting run."""
    from fdgs import train_host
    from fdgs.pipeline import StepPipeline
    cfg = synth.SceneConfig("lzp", 6007, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=4)
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    pipe = train_host.PipelineFlags()
    B = 3
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
    gen = torch.Generator(device="cpu").manual_seed(7)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in range(B)]
    runs = {}
    for lazy in (False, True):
        m = train_host.GaussianParams(scene, gpu_device)
        sp = StepPipeline(m, train_host.make_optimizer(m), world_size=1, lambda_dssim=0.2, lazy=lazy)
        losses, Rs = [], []
        # steps 0-1 at scaling_modifier 1, step 2 at 2.6 (2.7 x the instances, 2.4 x the longest list: beyond the 1.5 x + 64 the sort
        # instances of a lazy forward are launched for), step 3 at 2.6 again (the guess has learnt it)
        for mod in (1.0, 1.0, 2.6, 2.6):
            results, ls = sp.step(cams, gts, pipe, bg, scaling_modifier=mod)
            losses += [float(l) for l in ls]
            Rs += [r["num_rendered"] for r in results]
        torch.cuda.synchronize()
        runs[lazy] = (m.flat.detach().clone(), losses, Rs, sp.lazy_redone)
    assert runs[False][3] == 0 and runs[True][3] == 1, (runs[False][3], runs[True][3])
    # the first step starts from identical parameters: identical lists; later steps follow an Adam update whose float-atomics noise
    # differs from run to run (a parameter may move by 2 lr): a handful of the ~14-37 k instances per view may come or go
    assert runs[True][2][:B] == runs[False][2][:B] and min(runs[True][2]) > 0, (runs[True][2], runs[False][2])
    assert all(abs(a - b) <= 1e-3 * b for a, b in zip(runs[True][2], runs[False][2])), (runs[True][2], runs[False][2])
    # (losses behind an Adam step carry its float-atomics noise, which grows step by step: 1e-5 of round 5 was met in every run so far,
    # 2.6e-5 was seen once at step 8 of the overlap_steps test below; a forward that reads wrong lists is off by 1e-3 and more)
    np.testing.assert_allclose(runs[True][1], runs[False][1], rtol=3e-5, atol=1e-6)
    perr = (runs[True][0] - runs[False][0]).abs()
    assert (perr > 2e-3).float().mean().item() <= 2e-3 and perr.max().item() <= 0.25   # (Adam on float-atomics noise: see above)


def test_step_pipeline_overlap_steps_equals_plain(gpu_device):
    """StepPipeline(overlap_steps=True): the SH coefficients' update of step k on a third stream, geometry + binning + sort of step
    k + 1's first view next to it, that view's colours behind it (fdgs_forward_out.colour_stream).  Same kernels on the same numbers:
    losses and parameters follow the plain pipeline's (to the float-atomics noise two runs of the SAME pipeline differ by).  A model
    modified through torch between two steps is noticed (the version counter), and so is another pipeline that stepped the same model
    in between; one modified behind torch's back needs barrier()."""
    from fdgs import train_host
    from fdgs.pipeline import StepPipeline
    cfg = synth.SceneConfig("ovs", 90012, 320, 240, 3, 2, 0.012, 10.0, True, 4, False)     # 13 M SH coefficients: an update of ~30 us
    scene = synth.make_scene(cfg, seed=6, pose="rig1")
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    pipe = train_host.PipelineFlags()
    B, steps = 3, 8
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
    gen = torch.Generator(device="cpu").manual_seed(9)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in range(B)]
    runs = {}
    for mode in ("plain", "overlap", "plain again"):
        m = train_host.GaussianParams(scene, gpu_device)
        opt = train_host.make_optimizer(m)
        sp = StepPipeline(m, opt, world_size=1, lambda_dssim=0.2, overlap_steps=mode == "overlap")
        assert sp.overlap_steps == (mode == "overlap")
        other = StepPipeline(m, opt, world_size=1, lambda_dssim=0.2)     # a second pipeline on the same model (an evaluation loop, a bench leg)
        losses = []
        for k in range(steps):
            if k == 3:     # through torch: seen without being told
                with torch.no_grad():
                    m.params["_opacity"].mul_(0.98)
            if k == 5:     # behind torch's back (what a bench's restore does): the caller says so
                m.flat.data[: m.P * 3].add_(1e-4)
                sp.barrier()
            if k == 7:     # the other pipeline takes a step: seen without being told
                other.step(cams, gts, pipe, bg)
            _res, ls = sp.step(cams, gts, pipe, bg)
            losses += [float(l) for l in ls]
        torch.cuda.synchronize()
        runs[mode] = (m.flat.detach().clone(), losses, sp.steps_carried, opt.exp_avg.clone())
    # steps 1, 2, 4, 6 start under the previous step's SH update; 0 (nothing before it), 3, 5 and 7 wait for the caller's stream
    assert runs["overlap"][2] == steps - 4 and runs["plain"][2] == 0, (runs["overlap"][2], runs["plain"][2])
    np.testing.assert_allclose(runs["overlap"][1][:B], runs["plain"][1][:B], rtol=1e-6, atol=1e-7)      # first step: identical inputs
    noise = (runs["plain again"][0] - runs["plain"][0]).abs()
    # the losses of the later steps carry the float-atomics noise of the Adam steps before them (it grows step by step: 2.6e-5 relative at
    # step 8 in one run of ten, against a fixed bar of 2e-5): the bar is what two runs of the PLAIN pipeline differ by, with room, and
    # never more than 1e-4 (a colour pass that does not wait for the SH update is off by ~1e-3)
    la, lb = np.array(runs["plain"][1]), np.array(runs["plain again"][1])
    loss_rtol = float(min(1e-4, max(2e-5, 4.0 * np.max(np.abs(la - lb) / np.abs(la)))))
    for other in ("overlap",):
        np.testing.assert_allclose(runs[other][1], runs["plain"][1], rtol=loss_rtol, atol=1e-6)
        perr = (runs[other][0] - runs["plain"][0]).abs()
        # (Adam on float-atomics noise: a parameter whose gradient is noise around zero may move by 2 lr per step either way; the bar is
        # what two runs of the plain pipeline differ by, with room)
        assert (perr > 2e-3).float().mean().item() <= max(2e-3, 4.0 * (noise > 2e-3).float().mean().item()), (perr > 2e-3).float().mean().item()
        assert perr.max().item() <= max(0.25, 2.0 * noise.max().item())
        merr = (runs[other][3] - runs["plain"][3]).abs().max().item()
        assert merr <= 10.0 * max((runs["plain again"][3] - runs["plain"][3]).abs().max().item(), 1e-7), merr


@pytest.mark.parametrize("variant", ["batch_views", "sh_group"])
def test_step_pipeline_overlap_steps_with_view_batching_waits_for_the_sh_update(variant, gpu_device):
    """overlap_steps leaves step k's SH update running on stream A; a step whose first SH colours are NOT a split_colour launch on that
    stream -- the view-batched colour pass (batch_views) or the batched step (sh_group > 1) on stream F -- must wait for it (round-5
    advisor finding: it did not, and read coefficients the update was still writing).  Steps with 1 view (carried, split colour) and
    3 views (batched) alternate, against the same sequence without overlap_steps.  The race is made certain instead of likely: a
    10 ms sleep kernel is put in front of every SH update (on its stream), so a colour pass that does not wait for stream A
    reads the coefficients of BEFORE the update (lr 0.02 per step: its loss is off by ~1e-3 relative; two runs of the plain pipeline
    differ by 1e-5)."""
    from fdgs import train_host
    from fdgs.pipeline import StepPipeline
    cfg = synth.SceneConfig("ovb", 60012, 256, 192, 3, 2, 0.012, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=6, pose="rig1")
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    pipe = train_host.PipelineFlags()
    B, steps = 3, 6
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
    gen = torch.Generator(device="cpu").manual_seed(9)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in range(B)]
    kw = dict(batch_views=True) if variant == "batch_views" else dict(sh_group=3, lazy=False)
    runs = {}
    for mode in ("plain", "overlap", "plain again"):
        m = train_host.GaussianParams(scene, gpu_device)
        opt = train_host.make_optimizer(m)
        opt.set_lr("_features", 0.02, 0.02)
        sp = StepPipeline(m, opt, world_size=1, lambda_dssim=0.2, overlap_steps=mode == "overlap", **kw)
        update = opt.step_sh_staged

        def slow_update(*a, _update=update, **k):      # ~10 ms in front of every SH update, on the stream the pipeline puts it on
            torch.cuda._sleep(20_000_000)
            return _update(*a, **k)
        opt.step_sh_staged = slow_update
        losses = []
        for k in range(steps):
            n = 1 if k % 2 == 0 else B
            _res, ls = sp.step(cams[:n], gts[:n], pipe, bg)
            losses += [l.clone() for l in ls]      # (no float() here: reading a loss would make the HOST wait for the SH update, as bench.py does not)
        torch.cuda.synchronize()
        runs[mode] = (m.flat.detach().clone(), [float(l) for l in losses], sp.steps_carried)
    np.testing.assert_allclose(runs["plain again"][1], runs["plain"][1], rtol=1e-4, atol=1e-6)     # (what two runs of one pipeline differ by)
    np.testing.assert_allclose(runs["overlap"][1], runs["plain"][1], rtol=1e-4, atol=1e-6)
    b, e = m.offsets["_features"]
    perr = (runs["overlap"][0][b:e] - runs["plain"][0][b:e]).abs()
    noise = (runs["plain again"][0][b:e] - runs["plain"][0][b:e]).abs()
    assert (perr > 1e-2).float().mean().item() <= max(1e-3, 4.0 * (noise > 1e-2).float().mean().item()), (perr > 1e-2).float().mean().item()
    # batch_views: the 1-view steps 2 and 4 are carried (they start under the previous step's SH update); the batched steps 1, 3, 5 must
    # not be.  sh_group: the batched step puts its SH update on stream F, so nothing is left to carry into the 1-view steps either
    assert runs["plain"][2] == 0 and runs["overlap"][2] == (2 if variant == "batch_views" else 0), (runs["plain"][2], runs["overlap"][2])


def test_backward_without_the_per_view_outputs(gpu_device):
    """fdgs_backward_out.dL_dcolors / dL_dcov3D / dL_dflows = NULL (``per_view_outputs=False``, what StepPipeline passes): the three
    slots come back as None, everything else -- dL_dmeans2D and every parameter gradient -- is what the full call writes (the two
    calls differ only in the order of the blend backward's float atomics), and the persistent accumulator is left all zero by a
    geometry backward that now only re-zeroes the records that hold something."""
    from fdgs import train_host
    from fdgs.fused import raw_backward, raw_forward, raw_settings
    cfg = synth.SceneConfig("pvo", 7000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=21, pose="rig1")
    model = train_host.GaussianParams(scene, gpu_device)
    pipe = train_host.PipelineFlags()
    bg = torch.zeros(3, device=gpu_device)
    cam = train_host.SyntheticCamera(scene, gpu_device)
    up = synth.make_upstream_grads(scene["W"], scene["H"], seed=30, scale=1e-2)["grad_color"].to(gpu_device)
    rs, tens = raw_settings(cam, model, pipe, bg)
    (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = tens
    (R, color, flow, depth, T, radii, geom, binb, img, _c, om) = raw_forward(rs, *tens)
    outs = {}
    for full in (True, False):
        sink = {k: torch.full_like(v, float("nan")) for k, v in model.grad_sink().items()}
        gacc = torch.zeros((model.P, 16), device=gpu_device)
        res = raw_backward(rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img, up, None, None, None,
                           sink, False, grad_accum=gacc, per_view_outputs=full)
        torch.cuda.synchronize()
        assert float(gacc.abs().max()) == 0.0
        outs[full] = (res, sink)
    assert all(outs[True][0][i] is not None for i in (1, 4, 6)) and all(outs[False][0][i] is None for i in (1, 4, 6))
    assert float(outs[True][0][4].abs().max()) > 0.0
    a, b = outs[True][0][0], outs[False][0][0]
    assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
    for k, want in outs[True][1].items():
        got = outs[False][1][k]
        assert torch.isfinite(got).all(), k
        sc = max(1e-6, float(want.abs().max()))
        assert float((got - want).abs().max()) <= 1e-3 * sc, k   # (atomics order through the covariance chain: as test_sh_backward_batch_matches_per_view)


@pytest.mark.parametrize("cfg", [synth.SceneConfig("ga4", 20012, 320, 240, 3, 2, 0.015, 10.0, True, 4, False),
                                 synth.SceneConfig("ga3", 9004, 200, 160, 2, 0, 0.03, 1.0, False, 3, False)], ids=["rot4d", "dim3"])
def test_geometry_adam_inside_the_last_views_backward_is_the_separate_step(cfg, gpu_device):
    """fdgs_backward_out.adam: the last view's geometry backward takes the Adam step of the 17 geometry parameters per Gaussian with the
    gradient it has just completed (here: accumulated over two views).  The gradient bucket still receives the sums, so the separate
    launch can be replayed on a copy of the state before: parameters and both moments must come out bit for bit the same -- for the
    Gaussians the last view did not see too (non-zero moments move a parameter whose gradient is zero)."""
    from fdgs import train_host
    from fdgs.fused import raw_backward, raw_forward, raw_settings
    from fdgs.loss import l1_ssim_grad
    scene = synth.make_scene(cfg, seed=41, pose="rig3")
    scene["means3D"][::7, 2] = -9.0                       # every seventh Gaussian behind the camera: never visible
    needs_all_seven = not (cfg.rot_4d and cfg.gaussian_dim == 4)   # a 3D scene leaves _t / _scaling_t / _rotation_r out of the call: refused
    model = train_host.GaussianParams(scene, gpu_device)
    opt = train_host.make_optimizer(model)
    g = torch.Generator(device="cpu").manual_seed(5)
    opt.exp_avg.copy_((1e-3 * torch.randn(opt.exp_avg.shape, generator=g)).to(gpu_device))
    opt.exp_avg_sq.copy_((1e-6 * torch.rand(opt.exp_avg_sq.shape, generator=g)).to(gpu_device))
    opt.step_count = 6
    before = (model.flat.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())
    pipe, bg = train_host.PipelineFlags(), scene["bg"].to(gpu_device)
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(b + 0.5) / 2 * scene["time_duration"]) for b in range(2)]
    up = torch.full((1,), 0.5, device=gpu_device)
    sink = model.grad_sink()
    feat = model.offsets["_features"][0]
    gacc = torch.zeros((model.P, 16), device=gpu_device)
    for b, cam in enumerate(cams):
        rs, (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = raw_settings(cam, model, pipe, bg)
        (R, color, flow, depth, T, radii, geom, binb, img, _c, om) = raw_forward(rs, xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv)
        g_color, _h = l1_ssim_grad(color, torch.rand(3, cfg.H, cfg.W, generator=g).to(gpu_device), 0.2, up)

        def geo_adam():
            opt.step_count += 1
            lr = {s_["name"]: s_["lr"] for s_ in opt.named_segments()}
            return dict(flat=model.flat, exp_avg=opt.exp_avg, exp_avg_sq=opt.exp_avg_sq, betas=opt.betas, eps=opt.eps, step=opt.step_count,
                        lr=dict(means3D=lr["_xyz"], opacities=lr["_opacity"], ts=lr["_t"], scales=lr["_scaling"], scales_t=lr["_scaling_t"],
                                rotations=lr["_rotation"], rotations_r=lr["_rotation_r"]))
        if b == 1 and needs_all_seven:
            with pytest.raises(Exception, match="all seven geometry tensors"):
                raw_backward(rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img, g_color, None, None, None,
                             sink, True, grad_accum=gacc, geometry_adam=geo_adam)
            assert float(gacc.abs().max()) == 0.0
            return
        raw_backward(rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img, g_color, None, None, None,
                     sink, b > 0, grad_accum=gacc, geometry_adam=geo_adam if b == 1 else None)
    torch.cuda.synchronize()
    got = (model.flat.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())
    invisible = (radii == 0)
    assert int(invisible.sum()) >= model.P // 8
    # replay: the state before + the gradient sums the kernels left in the bucket -> the separate Adam launch over the geometry segment
    with torch.no_grad():
        model.flat.copy_(before[0])
    opt.exp_avg.copy_(before[1]); opt.exp_avg_sq.copy_(before[2])
    opt.step_range(0, feat)
    torch.cuda.synchronize()
    for name, a, b_ in (("parameters", got[0][:feat], model.flat.detach()[:feat]), ("exp_avg", got[1][:feat], opt.exp_avg[:feat]), ("exp_avg_sq", got[2][:feat], opt.exp_avg_sq[:feat])):
        assert torch.equal(a, b_), "%s: the fused geometry Adam differs from the separate launch (max %g)" % (name, float((a - b_).abs().max()))
    assert torch.equal(got[0][feat:], before[0][feat:])              # the SH coefficients are not this call's business
    moved = (got[0][:feat] != before[0][:feat]).float().mean().item()
    assert moved > 0.95, moved

