"""``distCUDA2`` -- drop-in for ``simple_knn._C.distCUDA2`` (simple-knn/spatial.cu:15-27; used once, at initialisation,
scene/gaussian_model.py:274): mean squared distance of every point to its three nearest neighbours (csrc/knn.hip)."""
import torch

from . import _capi


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("fdgs: distCUDA2 needs a GPU tensor; there is no CPU path")
    pts = points.contiguous().float()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("fdgs: distCUDA2 expects a [P, 3] tensor")
    P, dev = int(pts.shape[0]), pts.device
    means = torch.zeros(P, dtype=torch.float32, device=dev)           # torch::full({P}, 0.0), spatial.cu:21
    if P == 0:
        return means
    scratch = torch.empty(_capi.lib.fdgs_knn_scratch_bytes(P), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _capi.lib.fdgs_dist2_knn3(P, pts.data_ptr(), means.data_ptr(), scratch.data_ptr(), _capi.current_stream_handle(dev))
    _capi._check(rc, "fdgs_dist2_knn3")
    return means
