"""Throughput of fdgs.harness.train against the bare StepPipeline loop on the same scene / views (dev tool): what do the
learning-rate schedule, the densification statistics and the shard iterator cost per step?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import harness, synth, train_host
from fdgs.pipeline import StepPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
scene = synth.make_scene(synth.CONFIGS[name], seed=0)
pipe = train_host.PipelineFlags()
bg = scene["bg"].to(dev)
B = 4
cams = [train_host.SyntheticCamera(scene, dev, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
gts = [torch.rand(3, scene["H"], scene["W"], device=dev) for _ in range(B)]
for mode in ("pipeline", "harness", "harness-nostats"):
    model = train_host.GaussianParams(scene, dev)
    opt = train_host.make_optimizer(model)
    if mode == "pipeline":
        sp = StepPipeline(model, opt)
        run = lambda n: [sp.step(cams, gts, pipe, bg) for _ in range(n)]
    else:
        kw = dict(densify_until_iter=0) if mode == "harness-nostats" else {}
        run = lambda n: harness.train(model, opt, cams, gts, pipe, bg, iterations=n, batch_size=B, **kw)
    run(10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(iters)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-16s %.3f ms/step  %.0f images/s" % (mode, dt / iters * 1e3, iters * B / dt))
