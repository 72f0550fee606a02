"""GPU: the fused path at the reference's own boundary -- a reference-style model (separate nn.Parameters, ``_features_dc`` /
``_features_rest``, PyTorch getters; fdgs.train_host.ReferenceStyleModel) through this package's ``render()``:
* the fast path of render() (raw parameters -> fused activations) against the reference's own sequence (getters -> rasterizer);
* ``fdgs.optim.Adam`` in place of ``torch.optim.Adam`` (scene/gaussian_model.py:353): same gradients, same training trajectory,
  and it survives what the reference's densification does to the optimizer (:376-452), ``state_dict`` round trips, foreign groups."""
import numpy as np
import pytest
import torch

from util import synth

pytestmark = pytest.mark.gpu
SC = synth.SceneConfig
NAMES9 = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_t", "_scaling_t", "_rotation_r")


def _setup(dev, optimizer, cfg=None, seed=4, B=2):
    from fdgs import train_host
    cfg = cfg or SC("opt", 6000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=seed)
    model = train_host.ReferenceStyleModel(scene, dev, optimizer=optimizer)
    cams = [train_host.SyntheticCamera(scene, dev, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
    gen = torch.Generator(device="cpu").manual_seed(7)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(dev) for _ in range(B)]
    return scene, model, cams, gts, train_host.PipelineFlags(), torch.tensor([0.1, 0.2, 0.3], device=dev)


def _names(model):
    return [n for n in NAMES9 if isinstance(getattr(model, n, None), torch.nn.Parameter)]


@pytest.mark.parametrize("cfg", [SC("f4", 6000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False), SC("f4n", 5000, 192, 144, 1, 0, 0.03, 1.0, False, 4, True),
                                 SC("f3", 5000, 192, 144, 2, 0, 0.03, 1.0, False, 3, False)], ids=["rot4d", "dim4-norot", "dim3"])
def test_render_fast_path_equals_the_reference_sequence(cfg, gpu_device):
    """render() on a reference-style model: raw parameters + in-kernel activations (fast path) against getters + PyTorch activations
    (what the reference does, gaussian_renderer/__init__.py:57-121): same image, radii, and parameter gradients."""
    import fdgs.gaussian_renderer as gr
    scene, model, cams, gts, pipe, bg = _setup(gpu_device, "torch", cfg)
    up = torch.from_numpy(np.random.default_rng(1).standard_normal((3, scene["H"], scene["W"])).astype(np.float32)).to(gpu_device) * 1e-2
    out, grads = {}, {}
    names = [n for n in _names(model) if cfg.gaussian_dim == 4 or n not in ("_t", "_scaling_t", "_rotation_r")]
    if not cfg.rot_4d:
        names = [n for n in names if n != "_rotation_r"]
    for fast in (False, True):
        gr.render_options["fast_path"] = fast
        try:
            for n in names:
                getattr(model, n).grad = None
            pkg = gr.render(cams[0], model, pipe, bg)
            ((pkg["render"] * up).sum() + 0.1 * (pkg["depth"] * up[:1]).sum()).backward()
        finally:
            gr.render_options["fast_path"] = True
        out[fast] = pkg
        grads[fast] = {n: getattr(model, n).grad.detach().clone() for n in names}
        grads[fast]["viewspace"] = pkg["viewspace_points"].grad.detach().clone()
    # The kernels' activations agree with PyTorch's to an ulp or two (tests/test_gpu_api.py::test_render_raw_matches_render holds
    # the raw-parameter path to the oracle with bit-identical activations); an ulp in a scale or an opacity moves a handful of
    # alpha >= 1/255 / radius = ceil(3 sigma) decisions: <= 1e-4 of the output's scale on all but 1e-3 of the pixels, a flipped
    # pixel by at most 1/255 of a colour.
    flips = int((out[True]["radii"] != out[False]["radii"]).sum())
    assert flips <= max(2, cfg.P // 2000), flips
    for k in ("render", "depth", "alpha"):
        # (rot_4d: the conditional covariance is a difference of O(scale^2) terms -- it amplifies the ulp, as it amplifies rounding
        # in the backward, tests/test_gpu_parity.py -- so the bar is the pixel bar of the parity tests, 1e-4, not 1e-5)
        tol = 1e-4 * max(1.0, float(out[False][k].abs().max()))
        d = (out[True][k] - out[False][k]).abs()
        assert float((d > tol).float().mean()) <= 1e-3 and float(d.max()) <= 2e2 * tol, (k, float((d > tol).float().mean()), float(d.max()))
    for n, g in grads[False].items():
        scale = max(1.0, float(g.abs().max()))
        d = (grads[True][n] - g).abs()
        assert float((d > 1e-4 * scale).float().mean()) <= 1e-3 and float(d.max()) <= 2e-2 * scale, "%s: %g (scale %g), %g beyond 1e-4" % (
            n, float(d.max()), scale, float((d > 1e-4 * scale).float().mean()))


@pytest.mark.parametrize("defer_sh", [True, False])
def test_fdgs_adam_trains_like_torch_adam(defer_sh, gpu_device):
    """The reference's loop (train.py:104-170, 247-249: B views, loss / B, backward each, step, zero_grad(set_to_none=True)) with
    torch.optim.Adam against the same loop with fdgs.optim.Adam: same losses step after step, same parameters (up to Adam's
    sign-of-noise steps, see tests/test_gpu_api.py), gradients of the first batch equal."""
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    from fdgs.optim import Adam as FdgsAdam
    runs = {}
    for which in ("torch", "fdgs"):
        scene, model, cams, gts, pipe, bg = _setup(gpu_device, which)
        if which == "fdgs":
            model.optimizer.defer_sh = defer_sh
            assert isinstance(model.optimizer, FdgsAdam)
        losses, first_grads = [], None
        for it in range(3):
            for b in range(len(cams)):
                pkg = render(cams[b], model, pipe, bg)
                loss = fused_l1_ssim(pkg["render"], gts[b], 0.2)
                (loss / len(cams)).backward()
                losses.append(float(loss))
                assert pkg["viewspace_points"].grad is not None
            if it == 0:
                if which == "fdgs" and defer_sh:
                    assert model._features_dc.grad is None and model._features_rest.grad is None   # staged, not materialised
                first_grads = {n: (getattr(model, n).grad.detach().clone() if getattr(model, n).grad is not None else None) for n in _names(model)}
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        runs[which] = (losses, first_grads, {n: getattr(model, n).detach().clone() for n in _names(model)}, model)
    np.testing.assert_allclose(runs["fdgs"][0], runs["torch"][0], rtol=5e-5, atol=2e-6)   # (three steps of float-atomics noise through Adam)
    for n, g in runs["torch"][1].items():
        got = runs["fdgs"][1][n]
        if got is None:
            assert defer_sh and n in ("_features_dc", "_features_rest")
            continue
        scale = max(1e-6, float(g.abs().max()))
        assert float((got - g).abs().max()) <= 1e-3 * scale, n
    for n, want in runs["torch"][2].items():
        perr = (runs["fdgs"][2][n] - want).abs()
        assert (perr > 2e-3).float().mean().item() <= 5e-3 and perr.max().item() <= 0.25, (n, (perr > 2e-3).float().mean().item(), perr.max().item())
    opt = runs["fdgs"][3].optimizer
    assert opt.homings == 1 and opt.is_homed()
    # the homed layout: f_dc / f_rest are the two strided views of one contiguous [P, M, 3] array
    m = runs["fdgs"][3]
    assert m._features_dc.data_ptr() == opt.features().data_ptr() and m._features_rest.data_ptr() == opt.features().data_ptr() + 12
    assert torch.equal(torch.cat((m._features_dc, m._features_rest), dim=1), opt.features())
    st = opt.state[m._xyz]
    assert int(float(st["step"])) == 3 and st["exp_avg"].shape == m._xyz.shape


def test_fdgs_adam_survives_the_reference_densification_and_state_dict(gpu_device):
    """What scene/gaussian_model.py:376-452 does to the optimizer -- parameters replaced by fresh nn.Parameters, state dicts moved
    over, moments gathered with a mask / extended with zeros -- then training goes on: fdgs.optim.Adam homes again and keeps the
    moments; torch.optim.Adam put through the very same manipulations ends up with the same parameters."""
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    results = {}
    for which in ("torch", "fdgs"):
        scene, model, cams, gts, pipe, bg = _setup(gpu_device, which)

        def iteration():
            for b in range(len(cams)):
                (fused_l1_ssim(render(cams[b], model, pipe, bg)["render"], gts[b], 0.2) / len(cams)).backward()
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)

        iteration()
        P = model._xyz.shape[0]
        g = torch.Generator(device="cpu").manual_seed(3)
        # prune a fifth, clone 300 of the survivors (new rows: zero moments), reset the opacities
        mask = (torch.rand(P, generator=g) < 0.2).to(gpu_device)
        exp_avg_before = model.optimizer.state[model._scaling]["exp_avg"].detach().clone()
        model.prune_points(mask)
        assert torch.equal(model.optimizer.state[model._scaling]["exp_avg"], exp_avg_before[~mask])
        sel = torch.arange(0, 300, device=gpu_device)
        model.densification_postfix({name: getattr(model, attr).detach()[sel].clone() for name, attr in model._ATTR.items()
                                     if any(grp["name"] == name for grp in model.optimizer.param_groups)})
        model.replace_tensor_to_optimizer(torch.full_like(model._opacity.detach(), -2.0), "opacity")
        P2 = model._xyz.shape[0]
        assert P2 == int((~mask).sum()) + 300
        iteration()
        # a checkpoint round trip in the middle (capture / restore, scene/gaussian_model.py:82-177)
        import copy
        sd = copy.deepcopy(model.optimizer.state_dict())
        model.optimizer.load_state_dict(sd)
        iteration()
        torch.cuda.synchronize()
        if which == "fdgs":
            assert model.optimizer.is_homed() and model.optimizer.homings == 3, model.optimizer.homings   # start, re-layout, load_state_dict
            assert int(float(model.optimizer.state[model._xyz]["step"])) == 3
            assert float(model.optimizer.state[model._opacity]["exp_avg"].abs().max()) > 0
        results[which] = {n: getattr(model, n).detach().clone() for n in _names(model)}
    for n, want in results["torch"].items():
        perr = (results["fdgs"][n] - want).abs()
        assert perr.shape == want.shape
        assert (perr > 2e-3).float().mean().item() <= 5e-3 and perr.max().item() <= 0.25, (n, (perr > 2e-3).float().mean().item(), perr.max().item())


def test_fdgs_adam_foreign_groups_and_autograd_gradients(gpu_device):
    """A group that is not one of the reference's nine (e.g. the environment map, optimised alongside the Gaussians) is stepped
    tensor by tensor with torch.optim.Adam's arithmetic; a gradient that reaches a homed parameter through plain autograd (a
    regulariser on the opacities) is absorbed into the bucket and applied together with the rasterizer's."""
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    vals = {}
    for which in ("torch", "fdgs"):
        scene, model, cams, gts, pipe, bg = _setup(gpu_device, which)
        env = torch.nn.Parameter(torch.full((3, 8, 16), 0.5, device=gpu_device))
        model.optimizer.add_param_group({"params": [env], "lr": 1e-2, "name": "env_map"})
        for it in range(2):
            reg = 1e-3 * torch.sigmoid(model._opacity).sum() + (env ** 2).sum()   # autograd-only terms, BEFORE the render's backward
            reg.backward()
            (fused_l1_ssim(render(cams[0], model, pipe, bg)["render"], gts[0], 0.2)).backward()
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        vals[which] = (env.detach().clone(), model._opacity.detach().clone(), model._xyz.detach().clone())
    assert float((vals["fdgs"][0] - vals["torch"][0]).abs().max()) <= 1e-6
    for a, b in zip(vals["fdgs"][1:], vals["torch"][1:]):
        perr = (a - b).abs()
        assert (perr > 2e-3).float().mean().item() <= 5e-3 and perr.max().item() <= 0.25


def test_fdgs_adam_when_the_loop_clears_gradients_by_hand(gpu_device):
    """Gradients cleared WITHOUT optimizer.zero_grad() -- ``p.grad = None`` per parameter (what ``zero_grad(set_to_none=True)`` does,
    done by a model-level helper instead): the next backward must OVERWRITE the gradient bucket and drop the staged SH views of the
    last iteration, not accumulate on top of them.  Same trajectory as torch.optim.Adam in the same loop."""
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    runs = {}
    for which, defer in (("torch", None), ("fdgs", True), ("fdgs", False)):
        scene, model, cams, gts, pipe, bg = _setup(gpu_device, which)
        if which == "fdgs":
            model.optimizer.defer_sh = defer
        losses, gnorm = [], []
        for it in range(4):
            for b in range(len(cams)):
                loss = fused_l1_ssim(render(cams[b], model, pipe, bg)["render"], gts[b], 0.2)
                (loss / len(cams)).backward()
                losses.append(float(loss))
            gnorm.append(float(model._xyz.grad.abs().sum()))
            model.optimizer.step()
            for n in _names(model):
                getattr(model, n).grad = None
        torch.cuda.synchronize()
        runs[(which, defer)] = (losses, gnorm, {n: getattr(model, n).detach().clone() for n in _names(model)})
    want = runs[("torch", None)]
    for key in (("fdgs", True), ("fdgs", False)):
        np.testing.assert_allclose(runs[key][0], want[0], rtol=3e-5, atol=3e-6, err_msg=str(key))
        # (a bucket that kept accumulating would show a gradient norm growing iteration after iteration)
        np.testing.assert_allclose(runs[key][1], want[1], rtol=2e-3, err_msg=str(key))
        for n, w in want[2].items():
            perr = (runs[key][2][n] - w).abs()
            assert (perr > 2e-3).float().mean().item() <= 5e-3 and perr.max().item() <= 0.25, (key, n, (perr > 2e-3).float().mean().item())


def test_optimizer_edits_between_backward_and_step(gpu_device):
    """The reference densifies, prunes and resets the opacities BETWEEN backward() and optimizer.step() (train.py:160-249).  The
    replaced nn.Parameters have no gradient, so torch.optim.Adam skips them in that step: all nine after a prune / densification, the
    opacity alone after reset_opacity.  fdgs.optim.Adam re-homes inside step() with live gradients and staged SH views: it must end
    up with the same parameters."""
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    results = {}
    for which in ("torch", "fdgs"):
        scene, model, cams, gts, pipe, bg = _setup(gpu_device, which)

        def backward_views():
            for b in range(len(cams)):
                (fused_l1_ssim(render(cams[b], model, pipe, bg)["render"], gts[b], 0.2) / len(cams)).backward()

        def finish():
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)

        backward_views(); finish()
        # The reference resets the opacities every opacity_reset_interval = 3000 iterations (arguments/__init__.py:98), never earlier:
        # put both optimizers at step 3000.  (torch.optim.Adam counts steps PER PARAMETER and does not count a skipped one, so from
        # here on its opacity is one step behind the other tensors; fdgs.optim.Adam has one count for the bucket.  The bias
        # corrections 1 - beta^t differ by 5e-5 relative between t = 3000 and 3001 -- nothing; in the first few steps of a run,
        # where they would differ visibly, the reference never skips a single tensor.  See fdgs/optim.py.)
        for st in model.optimizer.state.values():
            st["step"] = torch.tensor(3000.0)
        if which == "fdgs":
            model.optimizer._fa.step_count = 3000
        # (1) opacity reset only (train.py:243-245 at opacity_reset_interval): every other tensor keeps its gradient and is stepped
        backward_views()
        before = {n: getattr(model, n).detach().clone() for n in _names(model)}
        model.replace_tensor_to_optimizer(torch.full_like(model._opacity.detach(), -2.0), "opacity")
        finish()
        torch.cuda.synchronize()
        assert float((model._opacity.detach() + 2.0).abs().max()) == 0.0, which            # no gradient, zero moments: not moved
        assert float((model._xyz.detach() - before["_xyz"]).abs().max()) > 0.0, which       # the others were stepped
        assert float((model._features_rest.detach() - before["_features_rest"]).abs().max()) > 0.0, which
        # (2) prune + densify (train.py:239-241): every parameter is replaced, nothing has a gradient, nothing moves in this step
        backward_views()
        P = model._xyz.shape[0]
        g = torch.Generator(device="cpu").manual_seed(3)
        mask = (torch.rand(P, generator=g) < 0.2).to(gpu_device)
        model.prune_points(mask)
        sel = torch.arange(0, 300, device=gpu_device)
        model.densification_postfix({name: getattr(model, attr).detach()[sel].clone() for name, attr in model._ATTR.items()
                                     if any(grp["name"] == name for grp in model.optimizer.param_groups)})
        edited = {n: getattr(model, n).detach().clone() for n in _names(model)}
        finish()
        torch.cuda.synchronize()
        for n in _names(model):
            assert torch.equal(getattr(model, n).detach(), edited[n]), (which, n)
        # (3) training goes on
        backward_views(); finish()
        backward_views(); finish()
        torch.cuda.synchronize()
        results[which] = {n: getattr(model, n).detach().clone() for n in _names(model)}
    for n, want in results["torch"].items():
        perr = (results["fdgs"][n] - want).abs()
        assert perr.shape == want.shape
        assert (perr > 2e-3).float().mean().item() <= 5e-3 and perr.max().item() <= 0.25, (n, (perr > 2e-3).float().mean().item(), perr.max().item())
