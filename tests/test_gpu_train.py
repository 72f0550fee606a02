"""End-to-end: the whole stack (forward, fused loss, backward, accumulation, Adam, step pipeline, harness) actually
trains -- ground-truth images are rendered from a target model, a perturbed copy is optimised back towards them."""
import numpy as np
import pytest
import torch

from util import synth

pytestmark = pytest.mark.gpu


def test_training_recovers_perturbed_scene(gpu_device):
    from fdgs import harness, train_host
    from fdgs.fused import render_raw
    cfg = synth.SceneConfig("fit", 4000, 160, 128, 3, 2, 0.04, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=2)
    pipe = train_host.PipelineFlags()
    bg = torch.zeros(3, device=gpu_device)
    target = train_host.GaussianParams(scene, gpu_device)
    V = 12
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(v + 0.5) / V * scene["time_duration"]) for v in range(V)]
    with torch.no_grad():
        gts = [render_raw(c, target, pipe, bg)["render"].clone() for c in cams]

    student = train_host.GaussianParams(scene, gpu_device)
    g = torch.Generator(device="cpu").manual_seed(0)
    with torch.no_grad():
        student.params["_features"].add_(0.3 * torch.randn(student.params["_features"].shape, generator=g).to(gpu_device))
        student.params["_opacity"].add_(0.5 * torch.randn(student.params["_opacity"].shape, generator=g).to(gpu_device))
        student.params["_scaling"].add_(0.1 * torch.randn(student.params["_scaling"].shape, generator=g).to(gpu_device))
    opt = train_host.make_optimizer(student)
    lines = []
    hist = harness.train(student, opt, cams, gts, pipe, bg, iterations=120, batch_size=4, log_every=20, log=lines.append)
    torch.cuda.synchronize()
    assert len(lines) == len(hist["loss"]) >= 6
    assert np.isfinite(hist["loss"]).all()
    assert hist["loss"][-1] < 0.5 * hist["loss"][0], hist["loss"]
    assert hist["psnr"][-1] > hist["psnr"][0] + 3.0, hist["psnr"]
    assert torch.isfinite(student.flat).all()


def test_training_with_densification(gpu_device):
    """The harness with the default densification schedule compressed into 150 iterations: the model grows, keeps
    training, and the loss still goes down."""
    from fdgs import harness, train_host
    from fdgs.fused import render_raw
    cfg = synth.SceneConfig("fitd", 3000, 160, 128, 3, 2, 0.04, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=6)
    pipe = train_host.PipelineFlags()
    bg = torch.zeros(3, device=gpu_device)
    target = train_host.GaussianParams(scene, gpu_device)
    V = 12
    cams = [train_host.SyntheticCamera(scene, gpu_device, timestamp=(v + 0.5) / V * scene["time_duration"]) for v in range(V)]
    with torch.no_grad():
        gts = [render_raw(c, target, pipe, bg)["render"].clone() for c in cams]
    student = train_host.GaussianParams(scene, gpu_device)
    g = torch.Generator(device="cpu").manual_seed(0)
    with torch.no_grad():
        student.params["_features"].add_(0.3 * torch.randn(student.params["_features"].shape, generator=g).to(gpu_device))
        student.params["_xyz"].add_(0.01 * torch.randn(student.params["_xyz"].shape, generator=g).to(gpu_device))
    opt = train_host.make_optimizer(student)
    lines = []
    hist = harness.train(student, opt, cams, gts, pipe, bg, iterations=150, batch_size=4, log_every=15, log=lines.append,
                         densify_from_iter=20, densification_interval=25, densify_until_iter=80, opacity_reset_interval=10 ** 6,
                         densify_grad_threshold=5e-4, cameras_extent=2.0)
    torch.cuda.synchronize()
    assert any("densify" in l for l in lines), lines
    assert student.P > 3000 and opt.exp_avg.numel() == student.flat.numel() == student.P * 161
    # splitting nearly every (large, synthetic) Gaussian perturbs the scene: the loss jumps after each densification
    # and has to come back down below where it started once densification stops
    assert np.isfinite(hist["loss"]).all() and hist["loss"][-1] < 0.8 * hist["loss"][0] and hist["loss"][-1] < 0.6 * max(hist["loss"])
