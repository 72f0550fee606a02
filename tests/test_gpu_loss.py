"""GPU: the fused L1 + SSIM kernel (csrc/ssim.hip) against the PyTorch statement of the reference loss
(utils/loss_utils.py:17-64 via fdgs.train_host.photometric_loss, evaluated on the CPU in float64 and on the GPU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 64, 64), (3, 77, 131), (3, 338, 450), (1, 40, 33)])
@pytest.mark.parametrize("lam", [0.2, 0.7])
def test_fused_l1_ssim_matches_reference_loss(shape, lam, gpu_device):
    from fdgs import train_host
    from fdgs.loss import fused_l1_ssim
    g = torch.Generator().manual_seed(5)
    # smooth-ish images in [0, 1] plus noise, like a render vs a photo
    base = torch.rand(shape[0], shape[1] // 4 + 2, shape[2] // 4 + 2, generator=g)
    up = torch.nn.functional.interpolate(base[None], size=shape[1:], mode="bilinear", align_corners=False)[0]
    img = (up + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    gt = (up.flip(-1) * 0.5 + 0.5 * torch.rand(shape, generator=g)).clamp(0, 1)

    x64 = img.double().requires_grad_(True)
    ref = train_host.photometric_loss(x64, gt.double(), lam)
    ref.backward()

    x = img.to(gpu_device).requires_grad_(True)
    out = fused_l1_ssim(x, gt.to(gpu_device), lam)
    (out * 3.0).backward()  # non-unit upstream gradient
    assert abs(out.item() - ref.item()) <= 2e-6, (out.item(), ref.item())
    gref = 3.0 * x64.grad.float()
    err = (x.grad.cpu() - gref).abs().max().item()
    scale = gref.abs().max().item()
    assert err <= 2e-4 * scale, (err, scale)


def test_fused_loss_rejects_cpu_tensors():
    from fdgs.loss import fused_l1_ssim
    with pytest.raises(RuntimeError, match="no CPU path"):
        fused_l1_ssim(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))
