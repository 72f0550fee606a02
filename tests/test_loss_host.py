"""CPU: the PyTorch statement of the photometric loss in fdgs.train_host (bench.py --torch-loss, and the second reference of
tests/test_gpu_loss.py) against fixtures produced by the reference's own utils/loss_utils.py (tests/golden/make_golden_ssim.py)."""
import glob
import os

import numpy as np
import pytest
import torch

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_*.npz")))


def test_fixtures_present():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_train_host_loss_matches_reference_fixtures(path):
    from fdgs import train_host
    f = np.load(path)
    x = torch.from_numpy(f["img"]).double().requires_grad_(True)
    gt = torch.from_numpy(f["gt"]).double()
    assert abs(train_host.l1_loss(x, gt).item() - float(f["l1"])) <= 1e-12
    assert abs(train_host.ssim(x, gt).item() - float(f["ssim"])) <= 1e-12
    loss = train_host.photometric_loss(x, gt, float(f["lam"]))
    loss.backward()
    assert abs(loss.item() - float(f["loss"])) <= 1e-12
    assert float(np.abs(x.grad.numpy() - f["dloss_dimg"]).max()) <= 1e-12
