"""Forward-only loop on C3 (dev tool for kernel traces): python tools/fwd_loop.py [iters] [split]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import synth, train_host
from fdgs.fused import raw_forward, raw_settings
dev = torch.device("cuda:0")
scene = synth.make_scene(synth.CONFIGS["C3"], seed=0)
model = train_host.GaussianParams(scene, dev)
pipe = train_host.PipelineFlags()
bg = scene["bg"].to(dev)
cams = [train_host.SyntheticCamera(scene, dev, timestamp=(b + 0.5) / 4 * scene["time_duration"]) for b in range(4)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
split = len(sys.argv) > 2 and sys.argv[2] == "split"
def fwd(c):
    rs, (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = raw_settings(c, model, pipe, bg)
    return raw_forward(rs, xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, split_colour=split)
for i in range(20): fwd(cams[i % 4])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n): fwd(cams[i % 4])
torch.cuda.synchronize(); print("forward %.4f ms" % ((time.perf_counter() - t0) / n * 1e3))
