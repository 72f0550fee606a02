"""Timing of the fused L1+SSIM kernels alone (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs.loss import l1_ssim_value_and_grad
dev = torch.device("cuda:0")
H, W = (int(sys.argv[2]), int(sys.argv[1])) if len(sys.argv) > 2 else (1014, 1352)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.rand(3, H, W, generator=g).to(dev); b = torch.rand(3, H, W, generator=g).to(dev)
up = torch.ones(1, device=dev)
for _ in range(5):
    l1_ssim_value_and_grad(a, b, 0.2, up)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(n):
    l1_ssim_value_and_grad(a, b, 0.2, up)
torch.cuda.synchronize()
print("l1+ssim fwd+bwd %dx%d: %.1f us per call" % (W, H, (time.time() - t0) / n * 1e6))
