"""The training step of bench.py as a bare loop (for rocprofv3 PMC passes: the SAME kernel variants the timed step runs --
raw parameters, tile_cull, colour-only blend backward, deferred SH backward, blend forward without flow accumulators, fused SH
flush + Adam).  usage: python tools/step_loop.py [workload=C3] [steps=3] [storage order: morton|random] [cameras: rig|axis]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import synth, train_host
from fdgs.pipeline import StepPipeline

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
order = sys.argv[3] if len(sys.argv) > 3 else "morton"
cameras = sys.argv[4] if len(sys.argv) > 4 else "rig"
dev = torch.device("cuda:0")
scene = synth.make_scene(synth.CONFIGS[name], seed=0)
model = train_host.GaussianParams(scene, dev)
opt = train_host.make_optimizer(model)
if order == "morton":
    train_host.spatial_sort(model, opt)
B = 4
cams = [train_host.SyntheticCamera(scene if cameras == "axis" else dict(scene, **synth.camera_for("rig%d" % (b % 4), scene["W"], scene["H"])), dev,
                                   timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]   # bench.py's views
gts = [torch.rand(3, scene["H"], scene["W"], generator=torch.Generator(device="cpu").manual_seed(1234 + b)).to(dev) for b in range(B)]
pipe, bg = train_host.PipelineFlags(), scene["bg"].to(dev)
sp = StepPipeline(model, opt, world_size=1, lambda_dssim=0.2, overlap=False)   # one stream: counters per kernel, not per overlap
for _ in range(steps):
    sp.step(cams, gts, pipe, bg)
torch.cuda.synchronize()
print("step_loop", name, steps, order, "done")
