"""Device allocations of a long StepPipeline run on C3 whose scene drifts (Adam on noise targets shrinks it step by step): how many
hipMallocs happen after the first 30 steps, and the slowest step.  Round 6: 2-3 allocations in 400 steps, slowest step 2.33 ms against
a median of 1.95 -- a slowly drifting scene is no problem for torch's caching allocator; what is one is a state that JUMPS BACK (bench.py's
restores: the rehearsals there).  A pool of whole binning buffers kept by the pipeline was built and measured here: 10 allocations, same
steps -- dropped.  usage: python tools/alloc_probe.py [steps=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import synth, train_host
from fdgs.pipeline import StepPipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
scene = synth.make_scene(synth.CONFIGS["C3"], seed=0)
model = train_host.GaussianParams(scene, dev)
opt = train_host.make_optimizer(model)
train_host.spatial_sort(model, opt)
B = 4
cams = [train_host.SyntheticCamera(dict(scene, **synth.camera_for("rig%d" % b, scene["W"], scene["H"])), dev, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
gts = [torch.rand(3, scene["H"], scene["W"], generator=torch.Generator(device="cpu").manual_seed(1234 + b)).to(dev) for b in range(B)]
pipe, bg = train_host.PipelineFlags(), scene["bg"].to(dev)
sp = StepPipeline(model, opt, world_size=1, lambda_dssim=0.2, overlap_steps=True)
for _ in range(30):
    sp.step(cams, gts, pipe, bg)
torch.cuda.synchronize()
n0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
t0 = time.perf_counter()
for i in range(steps):
    ev[i][0].record()
    res, _ = sp.step(cams, gts, pipe, bg)
    ev[i][1].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms = sorted(a.elapsed_time(b) for a, b in ev)
print("%s%d steps, %.1f images/s, device allocations after warm-up %d, step ms median %.3f p99 %.3f max %.3f, num_rendered now %d, redone %d" % (
    "", steps, B * steps / dt, torch.cuda.memory_stats(dev)["num_device_alloc"] - n0,
    ms[len(ms) // 2], ms[int(0.99 * len(ms))], ms[-1], res[-1]["num_rendered"], sp.lazy_redone))
