import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from fdgs import _capi
from fdgs.loss import l1_ssim_grad, l1_ssim_loss
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
img = torch.rand(3, 1014, 1352, generator=g).to(dev); gt = torch.rand(3, 1014, 1352, generator=g).to(dev)
up = torch.ones(1, device=dev)
for _ in range(20): gr, h = l1_ssim_grad(img, gt, 0.2, up); l = l1_ssim_loss(h)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200): gr, h = l1_ssim_grad(img, gt, 0.2, up)
b.record(); torch.cuda.synchronize()
print(os.environ.get("FDGS_LIB", "in-tree"), "fwd+bwd %.1f us" % (a.elapsed_time(b) / 200 * 1e3), "loss", l1_ssim_loss(h))
