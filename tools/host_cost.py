import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import synth, train_host
from fdgs.fused import render_raw
from fdgs.loss import fused_l1_ssim
dev = torch.device("cuda:0")
cfg = synth.SceneConfig("tiny", 2000, 64, 48, 3, 2, 0.05, 10.0, True, 4, False)
scene = synth.make_scene(cfg, seed=0)
model = train_host.GaussianParams(scene, dev)
opt = train_host.make_optimizer(model)
pipe = train_host.PipelineFlags()
cam = train_host.SyntheticCamera(scene, dev, timestamp=0.5)
bg = scene["bg"].to(dev)
gt = torch.rand(3, scene["H"], scene["W"], device=dev)
sink = model.grad_sink()
def fwd():
    return render_raw(cam, model, pipe, bg, grad_sink=sink, accumulate=False)
for _ in range(20):
    pkg = fwd(); loss = fused_l1_ssim(pkg["render"], gt, 0.2); loss.backward()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(n): fwd()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(n):
    pkg = fwd()
torch.cuda.synchronize(); t2 = time.perf_counter()
for _ in range(n):
    pkg = fwd(); loss = fused_l1_ssim(pkg["render"], gt, 0.2)
torch.cuda.synchronize(); t3 = time.perf_counter()
for _ in range(n):
    pkg = fwd(); loss = fused_l1_ssim(pkg["render"], gt, 0.2); loss.backward()
torch.cuda.synchronize(); t4 = time.perf_counter()
for _ in range(n):
    opt.step()
torch.cuda.synchronize(); t5 = time.perf_counter()
print("tiny scene, host-bound costs per call (us): fwd(no_grad) %.0f  fwd(grad) %.0f  +loss %.0f  +backward %.0f  adam %.0f" % (
    (t1-t0)/n*1e6, (t2-t1)/n*1e6, (t3-t2)/n*1e6, (t4-t3)/n*1e6, (t5-t4)/n*1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    pkg = fwd(); loss = fused_l1_ssim(pkg["render"], gt, 0.2); loss.backward()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
