"""Stand-alone time of the SH flush (fdgs_sh_flush) and the fused SH flush + Adam (fdgs_adam_step_sh) at C3 size (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import _capi
dev = torch.device("cuda:0")
P, M, B = 300000, 48, 4
g = torch.Generator().manual_seed(0)
st = torch.randn(B, P, 8, generator=g)
st[:, :, 0:3][torch.rand(B, P, generator=g) < 0.6] = 0
st = st.to(dev)
grad = torch.empty(P, M, 3, device=dev)
p = torch.randn(P, M, 3, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
for name, fn in (("sh_flush<0>", lambda: _capi.sh_flush(st, grad, 3, 2, 4, False, False)),
                 ("sh_adam<1>", lambda: _capi.adam_step_sh(p, m, v, st, 3, 2, 4, False, False, 1e-4, 2.5e-3, 0.9, 0.999, 1e-15, 1))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    print(name, "%.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
