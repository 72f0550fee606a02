import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from fdgs import synth, train_host, _capi
from fdgs.pipeline import StepPipeline
dev = torch.device("cuda:0")
B = 4
tiny = synth.make_scene(synth.SceneConfig("tiny", 2000, 64, 48, 3, 2, 0.05, 10.0, True, 4, False), seed=0)
tm = train_host.GaussianParams(tiny, dev)
pipe = train_host.PipelineFlags()
for overlap in (True, False):
    tp = StepPipeline(tm, train_host.make_optimizer(tm), world_size=1, lambda_dssim=0.2, overlap=overlap)
    tcams = [train_host.SyntheticCamera(tiny, dev, timestamp=(b + 0.5) / B * tiny["time_duration"]) for b in range(B)]
    tgts = [torch.rand(3, tiny["H"], tiny["W"], device=dev) for _ in range(B)]
    tbg = tiny["bg"].to(dev)
    for _ in range(5): tp.step(tcams, tgts, pipe, tbg)
    torch.cuda.synchronize(dev)
    s0 = _capi.run_ahead_stats()
    th = time.perf_counter()
    for _ in range(50): tp.step(tcams, tgts, pipe, tbg)
    torch.cuda.synchronize(dev)
    print("overlap", overlap, "host ms/view %.4f" % ((time.perf_counter() - th) / 200 * 1e3), "paths", tuple(b - a for a, b in zip(s0, _capi.run_ahead_stats())))
