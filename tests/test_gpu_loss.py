"""GPU: the fused L1 + SSIM kernel (csrc/ssim.hip) against (a) fixtures produced by the reference's own
utils/loss_utils.py in float64 (tests/golden/make_golden_ssim.py) and (b) the PyTorch statement of the same loss in
fdgs.train_host (the --torch-loss A/B path of bench.py; itself pinned to the fixtures in tests/test_loss_host.py)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_fused_l1_ssim_matches_reference_fixtures(path, gpu_device):
    """Loss value and image gradient of the HIP kernels against the reference's l1_loss / ssim (float64 fixtures)."""
    from fdgs.loss import fused_l1_ssim, l1_ssim_value_and_grad
    f = np.load(path)
    lam = float(f["lam"])
    img = torch.from_numpy(f["img"]).to(gpu_device).requires_grad_(True)
    gt = torch.from_numpy(f["gt"]).to(gpu_device)
    out = fused_l1_ssim(img, gt, lam)
    out.backward()
    assert abs(out.item() - float(f["loss"])) <= 2e-6, (out.item(), float(f["loss"]))
    want = f["dloss_dimg"]
    scale = float(np.abs(want).max())
    err = float(np.abs(img.grad.cpu().numpy().astype(np.float64) - want).max())
    print(os.path.basename(path), "loss err %.2e, gradient max abs err %.2e (max|ref| %.2e)" % (abs(out.item() - float(f["loss"])), err, scale))
    assert err <= 1e-4 * scale, (err, scale)
    # the no-autograd entry the step pipeline uses gives the same numbers
    up = torch.full((1,), 2.5, dtype=torch.float32, device=gpu_device)
    val, g = l1_ssim_value_and_grad(img.detach(), gt, lam, up)
    assert abs(val.item() - float(f["loss"])) <= 2e-6
    assert float(np.abs(g.cpu().numpy().astype(np.float64) - 2.5 * want).max()) <= 2.5e-4 * scale


# (3, 1014, 1352): the size bench.py runs the loss at (BASELINE configs[2], H x W = 1014 x 1352)
@pytest.mark.parametrize("shape", [(3, 64, 64), (3, 77, 131), (3, 338, 450), (1, 40, 33), (3, 1014, 1352)])
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


def test_fused_adam_matches_torch_adam(gpu_device):
    """csrc/adam.hip against torch.optim.Adam(eps=1e-15) with the reference's per-group learning rates
    (the SH DC / rest split is a per-coefficient rate inside one tensor here)."""
    from fdgs import synth, train_host
    cfg = synth.SceneConfig("adam", 333, 32, 32, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=0)
    model = train_host.GaussianParams(scene, gpu_device)
    init = model.flat.detach().clone().cpu()
    opt = train_host.FlatAdam(model)
    lr = opt.lr_vector().cpu()
    groups, ref_views = [], []
    for val in sorted(set(lr.tolist())):  # one torch parameter group per distinct learning rate
        idx = (lr == val).nonzero().flatten()
        q = init[idx].clone().requires_grad_(True)
        groups.append({"params": [q], "lr": val})
        ref_views.append((idx, q))
    ref_opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(0)
    for step in range(5):
        grad = torch.randn(model.flat.numel(), generator=g) * (10.0 ** torch.randint(-6, 1, (1,), generator=g).item())
        grad[::7] = 0.0  # culled Gaussians have exactly zero gradients
        model.flat_grad.copy_(grad.to(gpu_device))
        opt.step()
        for idx, q in ref_views:
            q.grad = grad[idx].clone()
        ref_opt.step()
    out = model.flat.detach().cpu()
    ref = torch.empty_like(out)
    for idx, q in ref_views:
        ref[idx] = q.detach()
    assert (out - ref).abs().max().item() <= 1e-6
    assert (out - init).abs().max().item() > 1e-4  # it did move


def test_adam_step_range_equals_whole_step(gpu_device):
    """FlatAdam.step_range over pieces of the bucket (segment table shifted, phases kept: negative begins) updates
    exactly like one step over the whole bucket -- what allreduce_and_step relies on."""
    from fdgs import synth, train_host
    cfg = synth.SceneConfig("ad", 777, 64, 48, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=1)
    a = train_host.GaussianParams(scene, gpu_device)
    b = train_host.GaussianParams(scene, gpu_device)
    oa, ob = train_host.make_optimizer(a), train_host.make_optimizer(b)
    g = torch.Generator(device="cpu").manual_seed(9)
    for _ in range(3):
        grad = torch.randn(a.flat.shape, generator=g).to(gpu_device) * 1e-3
        a.flat_grad.copy_(grad); b.flat_grad.copy_(grad)
        oa.step()
        ob.step_count += 1
        n = b.flat.numel()
        cuts = [0, 1000, 1000 + 4 * 12345, n // 2 // 4 * 4, n]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            ob.step_range(lo, hi)
    torch.cuda.synchronize()
    assert torch.equal(a.flat, b.flat) and torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)


def test_adam_segment_kernel_equals_general_kernel(gpu_device):
    """fdgs_adam_step takes the per-segment kernel (one segment per blockIdx.y: the learning rate is a workgroup constant) when the
    segment table tiles [0, n), and the general kernel (per-element search) otherwise.  Same arithmetic: bit-identical parameters and
    moments -- with segment boundaries that are not multiples of 4, a segment shorter than a float4, a periodic head (the SH DC
    rate; also periods of 1, 2 and 3 elements, shorter than a float4), a segment that starts before the chunk (negative begin: step_range) and one that ends behind it."""
    import ctypes as C
    from fdgs import _capi
    n = 4 * 2503 + 3
    g = torch.Generator(device="cpu").manual_seed(3)
    grad = (torch.randn(n, generator=g) * 1e-2).to(gpu_device)
    grad[::5] = 0.0
    bounds = [-77, 3, 5, 1000, 1001, 4099, 7001, n + 50]     # segments [b_i, b_i+1): clipped to [0, n) they tile it
    segs = []
    for i, (b, e) in enumerate(zip(bounds[:-1], bounds[1:])):
        # (periods below 4: a float4 spans more than one period -- round-5 advisor finding: the phase was reduced once only)
        period, head = {3: (21, 3), 5: (21, 3), 1: (2, 1), 4: (1, 1), 2: (3, 2), 6: (2, 1)}.get(i, (0, 0))
        segs.append(_capi.FdgsAdamSegment(b, e, 1e-3 * (i + 1), 5e-2 * (i + 1), period, head))
    tiling = (_capi.FdgsAdamSegment * len(segs))(*segs)
    # the same table with the last segment cut one element short of n: does not tile -> general kernel; element n - 1 gets lr = 0
    segs2 = list(segs)
    segs2[-1] = _capi.FdgsAdamSegment(bounds[-2], n - 1, segs[-1].lr, segs[-1].lr_head, segs[-1].period, segs[-1].head)
    general = (_capi.FdgsAdamSegment * len(segs2))(*segs2)

    def run(table):
        gen = torch.Generator(device="cpu").manual_seed(4)
        p = torch.randn(n, generator=gen).to(gpu_device)
        m = (0.01 * torch.randn(n, generator=gen)).to(gpu_device)
        v = (1e-4 * torch.rand(n, generator=gen)).to(gpu_device)
        for step in (1, 2, 3):
            rc = _capi.lib.fdgs_adam_step(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, table, len(segs), 0.9, 0.999, 1e-15, step,
                                          _capi.current_stream_handle(gpu_device))
            assert rc == 0
        torch.cuda.synchronize()
        return p, m, v
    a, b = run(tiling), run(general)
    for x, y in zip(a, b):
        assert torch.equal(x[:n - 1], y[:n - 1])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])          # the moments of the last element update either way
    assert float((a[0][n - 1] - b[0][n - 1]).abs()) > 0.0               # ... its parameter only where a segment covers it
    p0 = torch.randn(n, generator=torch.Generator(device="cpu").manual_seed(4)).to(gpu_device)
    assert float((a[0] - p0).abs().min()) > 0.0 or bool((grad == 0).any())


@pytest.mark.parametrize("shape", [(3, 64, 64), (3, 77, 131), (1, 40, 33), (3, 31, 97), (3, 1014, 1352)])
def test_one_kernel_value_and_grad_equals_the_two_kernel_path(shape, gpu_device):
    """fdgs_l1_ssim_value_and_grad (the training step's loss: a workgroup rebuilds the derivative maps around its tile instead of
    reading them back) against fdgs_l1_ssim_forward + fdgs_l1_ssim_backward: same arithmetic per pixel -- gradient, partial sums and
    loss value equal to fp32 rounding; sizes with ragged tiles on both axes, one smaller than a tile, and the bench's."""
    from fdgs import loss as fl
    g = torch.Generator().manual_seed(11)
    img = torch.rand(shape, generator=g).to(gpu_device)
    gt = torch.rand(shape, generator=g).to(gpu_device)
    up = torch.full((1,), 0.37, dtype=torch.float32, device=gpu_device)
    res = {}
    for fused in (False, True):
        fl.ssim_options["fused"] = fused
        try:
            gr, handle = fl.l1_ssim_grad(img, gt, 0.2, up)
            val = fl.l1_ssim_loss(handle)
            torch.cuda.synchronize()
            res[fused] = (gr.clone(), handle[0].clone(), float(val))
        finally:
            fl.ssim_options["fused"] = False
    assert torch.isfinite(res[True][0]).all() and float(res[True][0].abs().max()) > 0.0
    # (the two paths are two compilations of the same expressions: where the compiler contracts a multiply-add differs here and
    # there -- observed 6e-7 of the gradient's scale)
    scale = float(res[False][0].abs().max())
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-6 * scale
    assert float((res[True][1] - res[False][1]).abs().max()) <= 2e-6 * float(res[False][1].abs().max())
    assert abs(res[True][2] - res[False][2]) <= 1e-6


def test_loss_values_of_a_step_in_one_launch(gpu_device):
    """fdgs_l1_ssim_loss_batch: the loss-value reductions of all views of an optimizer step in one launch (a workgroup per view) --
    bit-identical to one fdgs_l1_ssim_loss per view (what fdgs.pipeline.StepPipeline used to enqueue: four one-workgroup launches)."""
    from fdgs.loss import l1_ssim_grad, l1_ssim_loss, l1_ssim_loss_batch, partials_buffer
    C, H, W, B = 3, 131, 203, 5
    g = torch.Generator(device="cpu").manual_seed(11)
    imgs = [torch.rand(C, H, W, generator=g).to(gpu_device) for _ in range(B)]
    gts = [torch.rand(C, H, W, generator=g).to(gpu_device) for _ in range(B)]
    up = torch.full((1,), 0.25, device=gpu_device)
    buf = partials_buffer(B, C, H, W, gpu_device)
    singles, handles, grads = [], [], []
    for v in range(B):
        g1, h1 = l1_ssim_grad(imgs[v], gts[v], 0.2, up)
        singles.append(l1_ssim_loss(h1))
        g2, h2 = l1_ssim_grad(imgs[v], gts[v], 0.2, up, parts=buf[v])
        handles.append(h2)
        grads.append((g1, g2))
    batch = l1_ssim_loss_batch(buf, handles)
    torch.cuda.synchronize()
    for v in range(B):
        assert torch.equal(batch[v], singles[v]) and torch.equal(*grads[v]), v
    assert len({float(x) for x in batch}) == B
    with pytest.raises(RuntimeError):
        l1_ssim_grad(imgs[0], gts[0], 0.2, up, parts=buf[0][:, :-1])
