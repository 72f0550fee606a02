"""Fused photometric loss (1 - lambda) L1 + lambda (1 - SSIM) on the GPU (csrc/ssim.hip).

Drop-in for the reference's ``(1.0 - opt.lambda_dssim) * l1_loss(image, gt) + opt.lambda_dssim * (1.0 - ssim(image, gt))``
(train.py:115-117, utils/loss_utils.py:17-64).  One forward kernel and one backward kernel instead of five
grouped 11x11 convolutions and their autograd graph.  ``gt`` is treated as a constant (no gradient), as in
training.  There is no CPU path: CPU tensors raise.  ``fdgs.train_host.photometric_loss`` is the PyTorch
statement of the same loss, used as the reference in the tests.
"""
import torch

from . import _capi


class _FusedL1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, lambda_dssim):
        if not img.is_cuda or not gt.is_cuda:
            raise RuntimeError("fdgs: fused_l1_ssim needs GPU tensors; there is no CPU path")
        img_c, gt_c = img.contiguous().float(), gt.contiguous().float()
        C, H, W = img_c.shape[-3], img_c.shape[-2], img_c.shape[-1]
        dev = img_c.device
        d1, d2, d3 = (torch.empty_like(img_c) for _ in range(3))
        nparts = _capi.lib.fdgs_l1_ssim_num_partials(C, H, W)
        parts = torch.empty((2, nparts), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_l1_ssim_forward(img_c.data_ptr(), gt_c.data_ptr(), C, H, W, d1.data_ptr(), d2.data_ptr(),
                                                d3.data_ptr(), parts[0].data_ptr(), parts[1].data_ptr(),
                                                _capi.current_stream_handle(dev))
            _capi._check(rc, "fdgs_l1_ssim_forward")
            out = torch.empty(3, dtype=torch.float32, device=dev)
            rc = _capi.lib.fdgs_l1_ssim_loss(parts[0].data_ptr(), parts[1].data_ptr(), nparts, C, H, W, float(lambda_dssim),
                                             out.data_ptr(), _capi.current_stream_handle(dev))
        _capi._check(rc, "fdgs_l1_ssim_loss")
        ctx.save_for_backward(img_c, gt_c, d1, d2, d3)
        ctx.lambda_dssim = float(lambda_dssim)
        ctx.shape = img.shape
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        img_c, gt_c, d1, d2, d3 = ctx.saved_tensors
        C, H, W = img_c.shape[-3], img_c.shape[-2], img_c.shape[-1]
        dev = img_c.device
        up = grad_out.reshape(1).contiguous().float()
        g = torch.empty_like(img_c)
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_l1_ssim_backward(img_c.data_ptr(), gt_c.data_ptr(), C, H, W, d1.data_ptr(), d2.data_ptr(),
                                                 d3.data_ptr(), up.data_ptr(), ctx.lambda_dssim, g.data_ptr(),
                                                 _capi.current_stream_handle(dev))
        _capi._check(rc, "fdgs_l1_ssim_backward")
        return g.view(ctx.shape), None, None


def fused_l1_ssim(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2) -> torch.Tensor:
    """(1 - lambda) * L1(image, gt) + lambda * (1 - SSIM(image, gt)); image, gt: [3, H, W] on the GPU."""
    return _FusedL1SSIM.apply(image, gt, lambda_dssim)


# l1_ssim_grad: the forward + backward pair with the three derivative maps in memory (default), or, "fused": True, ONE kernel that
# rebuilds the maps around every tile (fdgs_l1_ssim_value_and_grad: a third of the HBM traffic, 1.7 x the window arithmetic).  Measured
# on MI355X at 3x1014x1352, one stream: pair 64 us, one kernel 76 us; the two-stream C3 step: 2.53 against 2.56 ms -- the pair stays
# the default (FDGS_SSIM_FUSED=1 in the environment selects the one-kernel form at import).
import os as _os
ssim_options = {"fused": _os.environ.get("FDGS_SSIM_FUSED", "0") == "1"}


def l1_ssim_grad(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float, upstream: torch.Tensor, parts: torch.Tensor = None):
    """Value and gradient kernels only: returns ``(d(upstream * loss)/d image, handle)``; ``l1_ssim_loss(handle)`` reduces
    the per-tile partial sums to the loss value later (e.g. after the rasterizer backward has been enqueued, so that the
    small reduction is off the critical path).  ``parts``: a [2, num_partials] float32 slice to leave the partial sums in (a row of
    ``partials_buffer``: ``l1_ssim_loss_batch`` then reduces all views of a step in one launch)."""
    if not image.is_cuda or not gt.is_cuda:
        raise RuntimeError("fdgs: l1_ssim_grad needs GPU tensors; there is no CPU path")
    img_c, gt_c = image.contiguous().float(), gt.contiguous().float()
    C, H, W = img_c.shape[-3], img_c.shape[-2], img_c.shape[-1]
    dev = img_c.device
    nparts = _capi.lib.fdgs_l1_ssim_num_partials(C, H, W)
    if parts is None:
        parts = torch.empty((2, nparts), dtype=torch.float32, device=dev)
    elif tuple(parts.shape) != (2, nparts) or not parts.is_contiguous() or parts.dtype != torch.float32:
        raise RuntimeError("fdgs: parts must be a contiguous float32 [2, %d] tensor" % nparts)
    if ssim_options["fused"]:
        g = torch.empty_like(img_c)
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_l1_ssim_value_and_grad(img_c.data_ptr(), gt_c.data_ptr(), C, H, W, upstream.data_ptr(), float(lambda_dssim),
                                                       g.data_ptr(), parts[0].data_ptr(), parts[1].data_ptr(), _capi.current_stream_handle(dev))
        _capi._check(rc, "fdgs_l1_ssim_value_and_grad")
        return g, (parts, nparts, C, H, W, float(lambda_dssim))
    d1, d2, d3, g = (torch.empty_like(img_c) for _ in range(4))
    with torch.cuda.device(dev):
        st = _capi.current_stream_handle(dev)
        rc = _capi.lib.fdgs_l1_ssim_forward(img_c.data_ptr(), gt_c.data_ptr(), C, H, W, d1.data_ptr(), d2.data_ptr(),
                                            d3.data_ptr(), parts[0].data_ptr(), parts[1].data_ptr(), st)
        _capi._check(rc, "fdgs_l1_ssim_forward")
        rc = _capi.lib.fdgs_l1_ssim_backward(img_c.data_ptr(), gt_c.data_ptr(), C, H, W, d1.data_ptr(), d2.data_ptr(),
                                             d3.data_ptr(), upstream.data_ptr(), float(lambda_dssim), g.data_ptr(), st)
        _capi._check(rc, "fdgs_l1_ssim_backward")
    return g, (parts, nparts, C, H, W, float(lambda_dssim))


def l1_ssim_loss(handle) -> torch.Tensor:
    """Loss value of a ``l1_ssim_grad`` call (deterministic reduction kernel, current stream)."""
    parts, nparts, C, H, W, lam = handle
    dev = parts.device
    out = torch.empty(3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _capi.lib.fdgs_l1_ssim_loss(parts[0].data_ptr(), parts[1].data_ptr(), nparts, C, H, W, lam, out.data_ptr(),
                                         _capi.current_stream_handle(dev))
    _capi._check(rc, "fdgs_l1_ssim_loss")
    return out[0]


def partials_buffer(num_views: int, C: int, H: int, W: int, device) -> torch.Tensor:
    """[num_views, 2, num_partials]: one row per view for ``l1_ssim_grad(parts=buffer[v])``."""
    return torch.empty((num_views, 2, _capi.lib.fdgs_l1_ssim_num_partials(C, H, W)), dtype=torch.float32, device=device)


def l1_ssim_loss_batch(buffer: torch.Tensor, handles) -> list:
    """The loss values of the ``l1_ssim_grad`` calls that left their partial sums in rows 0 .. len(handles) - 1 of ``buffer``: ONE
    launch (a workgroup per view) instead of one per view; bit-identical to ``l1_ssim_loss`` per handle.  Returns a list of 0-d tensors."""
    n = len(handles)
    _parts, nparts, C, H, W, lam = handles[0]
    dev = buffer.device
    out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _capi.lib.fdgs_l1_ssim_loss_batch(buffer.data_ptr(), n, nparts, C, H, W, lam, out.data_ptr(), _capi.current_stream_handle(dev))
    _capi._check(rc, "fdgs_l1_ssim_loss_batch")
    return [out[v, 0] for v in range(n)]


def l1_ssim_value_and_grad(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float, upstream: torch.Tensor):
    """The same loss without autograd: returns ``(loss, d(upstream * loss)/d image)``.  ``upstream`` is a 1-element
    float32 GPU tensor (e.g. 1 / batch_size).  All kernels are enqueued on the current stream."""
    g, handle = l1_ssim_grad(image, gt, lambda_dssim, upstream)
    return l1_ssim_loss(handle), g
