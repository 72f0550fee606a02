"""Golden fixtures for the photometric loss, produced by the REFERENCE's own ``utils/loss_utils.py`` (``l1_loss``,
``ssim``: :17-64) evaluated in float64 on the CPU of the build container; its only missing import (torchmetrics, used by
``msssim`` only) is stubbed.  Loss = (1 - lambda) * l1_loss + lambda * (1 - ssim), train.py:115-117.

    python tests/golden/make_golden_ssim.py   ->   tests/golden/ssim_*.npz   (img, gt, lambda, l1, ssim, loss, dloss_dimg)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

tm = types.ModuleType("torchmetrics")
tm.MultiScaleStructuralSimilarityIndexMeasure = lambda **k: None
sys.modules["torchmetrics"] = tm
spec = importlib.util.spec_from_file_location("ref_loss_utils", "/root/reference/utils/loss_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def images(shape, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(shape[0], shape[1] // 4 + 2, shape[2] // 4 + 2, generator=g)
    up = torch.nn.functional.interpolate(base[None], size=shape[1:], mode="bilinear", align_corners=False)[0]
    img = (up + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    gt = (up.flip(-1) * 0.5 + 0.5 * torch.rand(shape, generator=g)).clamp(0, 1)
    return img, gt


CASES = {"ssim_64x64": ((3, 64, 64), 0.2, 1), "ssim_77x131": ((3, 77, 131), 0.2, 2), "ssim_96x160_l07": ((3, 96, 160), 0.7, 3),
         "ssim_1ch_40x33": ((1, 40, 33), 0.2, 4)}

if __name__ == "__main__":
    for name, (shape, lam, seed) in CASES.items():
        img, gt = images(shape, seed)
        x = img.double().requires_grad_(True)
        l1 = ref.l1_loss(x, gt.double())
        ss = ref.ssim(x, gt.double())
        loss = (1.0 - lam) * l1 + lam * (1.0 - ss)
        loss.backward()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), img=img.numpy(), gt=gt.numpy(), lam=np.float64(lam),
                            l1=np.float64(l1.item()), ssim=np.float64(ss.item()), loss=np.float64(loss.item()),
                            dloss_dimg=x.grad.numpy())
        print(name, "l1 %.6f ssim %.6f loss %.6f" % (l1.item(), ss.item(), loss.item()))
