"""What does the two-stream step PAY for each kernel?  Times bench.py's step (C3, 4 rig views, two streams) with one stage at a time
NOT launched (FDGS_TIMING_PROBE_SKIP, csrc/capi.hip: results are garbage, only the time is looked at) or, for the stages outside
the rasterizer, replaced by a no-op in Python: the difference to the unmodified step is the upper bound of what optimising that
stage can return in images/s -- as opposed to its single-stream duration, most of which may be hidden under another stream's kernel.
The skipping exists only in a copy of the library built with -DFDGS_TIMING_PROBE:
    FDGS_EXTRA_FLAGS=-DFDGS_TIMING_PROBE tools/ab_build.sh HEAD probe        (here, before gpurun)
Run on the GPU box:  python tools/sensitivity_probe.py            (one child process per configuration; uses tools/ab/libfdgs_probe.so)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = {"preprocess_fwd": 0, "tile_count": 1, "tile_scan": 2, "tile_scatter": 3, "tile_sort": 4, "blend_fwd": 6, "blend_bwd": 7, "preprocess_bwd": 8,
          "sh_bwd": 10}

CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from fdgs import synth, train_host, loss as fl
from fdgs.pipeline import StepPipeline
import fdgs.pipeline as fp
what = sys.argv[1]
dev = torch.device("cuda:0")
scene = synth.make_scene(synth.CONFIGS["C3"], seed=0)
model = train_host.GaussianParams(scene, dev)
opt = train_host.make_optimizer(model)
train_host.spatial_sort(model, opt)
B = 4
cams = [train_host.SyntheticCamera(dict(scene, **synth.camera_for("rig%%d" %% b, scene["W"], scene["H"])), dev, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
gts = [torch.rand(3, scene["H"], scene["W"], generator=torch.Generator(device="cpu").manual_seed(1234 + b)).to(dev) for b in range(B)]
pipe, bg = train_host.PipelineFlags(), scene["bg"].to(dev)
if what == "ssim":
    g_fixed = torch.rand(3, scene["H"], scene["W"], device=dev) * 1e-6
    fp.l1_ssim_grad = lambda color, gt, lam, up: (g_fixed, None)
    fp.l1_ssim_loss = lambda h: torch.zeros((), device=dev)
if what == "adam":
    opt.step_sh_staged = lambda *a, **k: True
    opt.step_range = lambda *a, **k: None
sp = StepPipeline(model, opt, world_size=1, lambda_dssim=0.2, lazy=(what == "none"))   # a skipped stage leaves garbage counts: the waiting forward
snap = model.flat.detach().clone()
for _ in range(30):
    sp.step(cams, gts, pipe, bg)
model.flat.data.copy_(snap)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    model.flat.data.copy_(snap); opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_count = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        sp.step(cams, gts, pipe, bg)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 40 * 1e3)
ts.sort()
print("RESULT %%s median %%.4f ms/step min %%.4f" %% (what, ts[2], ts[0]))
''' % ROOT


def run(what, mask):
    env = dict(os.environ)
    env.pop("FDGS_TIMING_PROBE_SKIP", None)
    if mask:
        env["FDGS_TIMING_PROBE_SKIP"] = str(mask)
        env["FDGS_LIB"] = os.path.join(ROOT, "tools", "ab", "libfdgs_probe.so")   # the in-tree library cannot skip stages
    out = subprocess.run([sys.executable, "-c", CHILD, what], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(what, "FAILED", out.stderr[-500:])
        return None
    return float(line[0].split()[3])


if __name__ == "__main__":
    base = run("none", 0)
    print("unmodified step: %.4f ms" % base)
    rows = [(k, 1 << v) for k, v in STAGES.items()] + [("sh_bwd+preprocess_bwd", (1 << 10) | (1 << 8)), ("binning (count..sort)", 0b11110),
                                                        ("front end (preprocess..sort)", 0b11111), ("ssim", 0), ("adam", 0)]
    for what, mask in rows:
        key = what if mask else what
        t = run(what if not mask else what.split()[0], mask)
        if t is not None:
            print("without %-32s %.4f ms/step   pays %6.1f us per step = %5.1f us per view  (%4.1f %%)" % (what, t, (base - t) * 1e3, (base - t) * 250, 100 * (base - t) / base))
