"""Standalone time of the geometry-bucket Adam step (17 floats per Gaussian, 7 segments) at C3 / C5 size, both kernels of fdgs_adam_step:
python tools/adam_time.py          (FDGS_ADAM_GENERAL=1 in the environment: the general kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import synth, train_host
dev = torch.device("cuda:0")
for name in ("C3", "C5"):
    cfg = synth.CONFIGS[name]
    scene = synth.make_scene(cfg, seed=0, P=cfg.P)
    m = train_host.GaussianParams(scene, dev)
    o = train_host.make_optimizer(m)
    m.flat_grad.normal_()
    feat = m.offsets["_features"][0]
    o.step_count = 1
    for _ in range(5):
        o.step_range(0, feat)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        o.step_range(0, feat)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print("%s geometry Adam (%d floats, %s kernel): %.1f us = %.2f TB/s" % (name, feat, "general" if os.environ.get("FDGS_ADAM_GENERAL") == "1" else "segment", us, feat * 28 / us / 1e6))
