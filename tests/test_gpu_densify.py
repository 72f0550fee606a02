"""Densification / pruning on the GPU (csrc/densify.hip, fdgs/densify.py) against the fixtures produced by the
reference's own GaussianModel.densify_and_prune (tests/golden/make_golden_densify.py) and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from util import synth
from test_oracle_densify import CASES, GOLD, assert_state_equal, golden_call, load_state

pytestmark = pytest.mark.gpu
NAMES = ("_xyz", "_features", "_opacity", "_scaling", "_rotation", "_t", "_scaling_t", "_rotation_r")


def _model_from_state(st, cfgv, device):
    """GaussianParams / FlatAdam / DensificationStats holding exactly the arrays of an oracle state."""
    from fdgs import harness, train_host
    sh_degree, sh_degree_t, gaussian_dim, rot_4d = cfgv
    P, M = st["params"]["_xyz"].shape[0], st["params"]["_features"].shape[1]
    cfg = synth.SceneConfig("d", P, 64, 48, sh_degree, sh_degree_t, 0.05, 10.0, bool(rot_4d), gaussian_dim, False)
    scene = synth.make_scene(cfg, seed=0)
    assert scene["M"] == M
    model = train_host.GaussianParams(scene, device)
    opt = train_host.make_optimizer(model)
    with torch.no_grad():
        for n in NAMES:
            if n in st["params"]:
                model.params[n].copy_(torch.from_numpy(st["params"][n]).reshape(model.params[n].shape))
                b, e = model.offsets[n]
                opt.exp_avg[b:e].copy_(torch.from_numpy(st["exp_avg"][n]).flatten())
                opt.exp_avg_sq[b:e].copy_(torch.from_numpy(st["exp_avg_sq"][n]).flatten())
    stats = harness.DensificationStats(P, device, 1)
    stats.xyz_gradient_accum.copy_(torch.from_numpy(st["xyz_gradient_accum"]))
    stats.denom.copy_(torch.from_numpy(st["denom"]))
    stats.max_radii2D.copy_(torch.from_numpy(st["max_radii2D"]))
    if "t_gradient_accum" in st:
        stats.t_gradient_accum.copy_(torch.from_numpy(st["t_gradient_accum"]))
    return model, opt, stats


def _state_of(model, opt, stats, like):
    out = {"params": {}, "exp_avg": {}, "exp_avg_sq": {}}
    for n in like["params"]:
        b, e = model.offsets[n]
        out["params"][n] = model.params[n].detach().cpu().numpy().reshape((model.P,) + like["params"][n].shape[1:])
        out["exp_avg"][n] = opt.exp_avg[b:e].cpu().numpy().reshape(out["params"][n].shape)
        out["exp_avg_sq"][n] = opt.exp_avg_sq[b:e].cpu().numpy().reshape(out["params"][n].shape)
    for k in ("xyz_gradient_accum", "t_gradient_accum", "denom", "max_radii2D"):
        if k in like:
            out[k] = getattr(stats, k).cpu().numpy()
    return out


@pytest.mark.parametrize("case", CASES)
def test_densify_matches_reference_run(case, gpu_device):
    from fdgs.densify import densify_and_prune
    d = np.load(os.path.join(GOLD, "densify_%s.npz" % case))
    kw = golden_call(d)
    st_in, want = load_state(d, "in."), load_state(d, "out.")
    model, opt, stats = _model_from_state(st_in, [int(x) for x in d["cfg"]], gpu_device)
    s = None if kw["samples"] is None else torch.from_numpy(kw["samples"]).to(gpu_device)
    s_t = None if kw["samples_t"] is None else torch.from_numpy(kw["samples_t"]).to(gpu_device)
    rep = densify_and_prune(model, opt, stats, kw["max_grad"], kw["min_opacity"], kw["extent"], kw["max_screen_size"], kw["max_grad_t"],
                            prune_only=kw["prune_only"], percent_dense=kw["percent_dense"], N=kw["N"], samples=s, samples_t=s_t)
    torch.cuda.synchronize()
    assert rep["P_new"] == model.P == want["params"]["_xyz"].shape[0], rep
    assert model.flat_grad.abs().max().item() == 0.0 and model.params["_xyz"].grad.shape == model.params["_xyz"].shape
    # copies bit-exact; split children's xyz / t / scaling: exp / log / rotation in fp32 on two different machines
    assert_state_equal(_state_of(model, opt, stats, want), want, rtol=3e-6, atol=3e-6)


def test_densify_larger_scene_vs_oracle_and_training_continues(gpu_device):
    """A bigger random case against the numpy oracle, then two optimizer steps on the re-laid-out model."""
    from fdgs import harness, train_host
    from fdgs.densify import densify_and_prune, reset_opacity
    from fdgs.pipeline import StepPipeline
    from oracle import densify_oracle as do
    cfg = synth.SceneConfig("d", 20000, 160, 128, 3, 2, 0.04, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=9)
    model = train_host.GaussianParams(scene, gpu_device)
    opt = train_host.make_optimizer(model)
    g = torch.Generator(device="cpu").manual_seed(5)
    opt.exp_avg.copy_(torch.randn(opt.exp_avg.shape, generator=g) * 1e-3)
    opt.exp_avg_sq.copy_(torch.rand(opt.exp_avg_sq.shape, generator=g) * 1e-6)
    stats = harness.DensificationStats(model.P, gpu_device, 1)
    stats.denom.copy_(torch.randint(0, 4, (model.P, 1), generator=g).float())
    stats.xyz_gradient_accum.copy_(torch.rand(model.P, 1, generator=g) * stats.denom.cpu() * 8e-4)
    stats.max_radii2D.copy_(torch.randint(0, 50, (model.P,), generator=g).float())
    st = {"params": {n: model.params[n].detach().cpu().numpy().copy() for n in NAMES},
          "exp_avg": {n: opt.exp_avg[slice(*model.offsets[n])].cpu().numpy().reshape(model.params[n].shape).copy() for n in NAMES},
          "exp_avg_sq": {n: opt.exp_avg_sq[slice(*model.offsets[n])].cpu().numpy().reshape(model.params[n].shape).copy() for n in NAMES},
          "xyz_gradient_accum": stats.xyz_gradient_accum.cpu().numpy().copy(), "t_gradient_accum": stats.t_gradient_accum.cpu().numpy().copy(),
          "denom": stats.denom.cpu().numpy().copy(), "max_radii2D": stats.max_radii2D.cpu().numpy().copy()}
    extent = float(np.median(np.exp(st["params"]["_scaling"]).max(1)) / 0.01)   # half of the Gaussians are 'small'
    # the oracle needs the samples of ALL split parents: count them the oracle's way, draw once, hand both the same draws
    grads = np.nan_to_num(st["xyz_gradient_accum"] / st["denom"], nan=0.0, posinf=np.inf)
    sel = (grads[:, 0] >= 2e-4) & (np.exp(st["params"]["_scaling"]).max(1) > 0.01 * extent)
    k = int(sel.sum())
    assert k > 50
    stds = np.exp(np.concatenate([st["params"]["_scaling"], st["params"]["_scaling_t"]], 1))[sel]
    samples = (torch.randn(2 * k, 4, generator=g).numpy() * np.concatenate([stds, stds], 0)).astype(np.float32)
    want = do.densify_and_prune(st, 2e-4, 0.02, extent, 20, 2e-4 / 40, percent_dense=0.01, N=2, rot_4d=True, gaussian_dim=4, samples=samples)
    rep = densify_and_prune(model, opt, stats, 2e-4, 0.02, extent, 20, 2e-4 / 40, samples=torch.from_numpy(samples).to(gpu_device))
    assert rep["split_parents"] == k and rep["cloned"] > 0 and rep["P_new"] == want["params"]["_xyz"].shape[0], rep
    assert_state_equal(_state_of(model, opt, stats, want), want, rtol=3e-6, atol=3e-6)
    # the re-laid-out model trains on
    reset_opacity(model, opt)
    assert torch.sigmoid(model._opacity).max().item() <= 0.0100001
    pipe, bg = train_host.PipelineFlags(), torch.zeros(3, device=gpu_device)
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=t) for t in (2.0, 6.0)]
    gts = [torch.rand(3, scene["H"], scene["W"], device=gpu_device) for _ in cams]
    sp = StepPipeline(model, opt)
    before = model.flat.clone()
    for _ in range(2):
        results, losses = sp.step(cams, gts, pipe, bg)
    torch.cuda.synchronize()
    assert results[0]["radii"].shape[0] == model.P and torch.isfinite(model.flat).all() and not torch.equal(before, model.flat)


def test_stats_update_gpu_matches_host_statement(gpu_device):
    """DensificationStats.update on the GPU (two kernels) equals the PyTorch statement of train.py:164-184 run on the CPU."""
    from fdgs import harness
    P, B = 5000, 4
    g = torch.Generator().manual_seed(21)
    gpu = harness.DensificationStats(P, gpu_device, 1)
    cpu = harness.DensificationStats(P, "cpu", 1)
    for _ in range(3):
        radii = [torch.randint(-3, 12, (P,), generator=g).clamp(min=0).to(torch.int32) for _ in range(B)]
        grads = [torch.randn(P, 3, generator=g) for _ in range(B)]
        t_grad = torch.randn(P, 1, generator=g)
        cpu.update([{"radii": r, "viewspace_grad": x} for r, x in zip(radii, grads)], t_grad, B)
        gpu.update([{"radii": r.to(gpu_device), "viewspace_grad": x.to(gpu_device)} for r, x in zip(radii, grads)], t_grad.to(gpu_device), B)
    for name in ("xyz_gradient_accum", "t_gradient_accum", "denom", "max_radii2D"):
        np.testing.assert_allclose(getattr(gpu, name).cpu().numpy(), getattr(cpu, name).numpy(), rtol=2e-6, atol=1e-7, err_msg=name)
