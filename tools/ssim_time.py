#!/usr/bin/env python3
"""Times the fused L1 + SSIM kernels on their own (HIP events, current stream): forward+backward pair per call.
   python tools/ssim_time.py [H W]"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
loss = importlib.import_module("4d-gaussian-splatting_amd.loss")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1014, 1352)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((3, H, W), device=dev, generator=g)
gt = torch.rand((3, H, W), device=dev, generator=g)
up = torch.ones(1, device=dev)
for _ in range(20):
    loss.l1_ssim_grad(img, gt, 0.2, up)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    loss.l1_ssim_grad(img, gt, 0.2, up)
e1.record()
torch.cuda.synchronize()
print("l1_ssim forward + backward: %.1f us per image (%dx%d)" % (e0.elapsed_time(e1) / n * 1e3, W, H))
