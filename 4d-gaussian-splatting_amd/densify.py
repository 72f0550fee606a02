"""Densification / pruning of the flat-bucket model (csrc/densify.hip).

``densify_and_prune`` mirrors ``GaussianModel.densify_and_prune`` (scene/gaussian_model.py:584-610) including its
quirks: ``max_radii2D`` is reset by the densification before the screen-size test looks at it (so that test only acts
with ``prune_only``), new points start with zero Adam moments, all statistics restart from zero.  The surviving rows
come out in the reference's order: kept originals, clones, split children (copy 1 of every parent, then copy 2, ...).

The decisions depend only on the statistics (identical on every rank, fdgs.harness.DensificationStats) and on the
normal samples: with the same ``generator`` seed on every rank the replicas stay identical without a broadcast.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import _capi

CLONE, SPLIT, PRUNE, PRUNE_CHILD = 1, 2, 4, 8


def _p(t):
    return None if t is None else t.data_ptr()


@torch.no_grad()
def densify_and_prune(model, optimizer, stats, max_grad: float, min_opacity: float, extent: float,
                      max_screen_size: Optional[float], max_grad_t: Optional[float] = None, prune_only: bool = False,
                      percent_dense: float = 0.01, N: int = 2, generator: Optional[torch.Generator] = None,
                      samples: Optional[torch.Tensor] = None, samples_t: Optional[torch.Tensor] = None) -> Dict[str, int]:
    """In place on ``model`` (fdgs.train_host.GaussianParams), ``optimizer`` (FlatAdam) and ``stats``
    (DensificationStats).  ``max_grad_t`` is accepted and ignored, as in the reference (its uses are commented out).
    ``samples`` / ``samples_t``: the draws of ``torch.normal(mean=0, std=stds)`` for ALL selected parents in the
    reference's order (tests inject the reference's own draws); drawn here from ``generator`` when None."""
    dev = model.flat.device
    if not model.flat.is_cuda:
        raise RuntimeError("fdgs: densify_and_prune needs the model on the GPU; there is no CPU path")
    P = model.P
    st = _capi.current_stream_handle(dev)
    flags = torch.empty(P, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _capi.lib.fdgs_densify_classify(P, _p(stats.xyz_gradient_accum), _p(stats.denom), _p(model._scaling), _p(model._opacity),
                                             _p(stats.max_radii2D), float(max_grad), float(min_opacity), float(extent),
                                             float(max_screen_size) if max_screen_size else -1.0, float(percent_dense), int(N),
                                             int(bool(prune_only)), _p(flags), st)
    _capi._check(rc, "fdgs_densify_classify")

    # ---- index plan (host plumbing: three nonzero() calls) ----
    pruned = (flags & PRUNE) != 0
    idx_orig = torch.nonzero(((flags & SPLIT) == 0) & ~pruned).flatten()
    idx_clone = torch.nonzero(((flags & CLONE) != 0) & ~pruned).flatten()
    idx_sel = torch.nonzero((flags & SPLIT) != 0).flatten()                      # all parents that split, in order
    k = int(idx_sel.numel())
    keep_sel = ((flags[idx_sel] & PRUNE_CHILD) == 0) if k else torch.zeros(0, dtype=torch.bool, device=dev)
    four = model.gaussian_dim == 4
    if k and samples is None:
        # torch.normal(mean=0, std=stds) over the N-fold repeated parents, gaussian_model.py:500-522
        if model.rot_4d:
            stds = torch.exp(torch.cat([model._scaling[idx_sel], model._scaling_t[idx_sel]], 1)).repeat(N, 1)
        else:
            stds = torch.exp(model._scaling[idx_sel]).repeat(N, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        if four and not model.rot_4d:
            stds_t = torch.exp(model._scaling_t[idx_sel]).repeat(N, 1)
            samples_t = torch.normal(mean=torch.zeros_like(stds_t), std=stds_t, generator=generator)
    kept_parents = idx_sel[keep_sel]
    kk = int(kept_parents.numel())
    child_parent = kept_parents.repeat(N)                                        # copy 1 of all, then copy 2, ...
    if kk:
        rows = (torch.arange(N, device=dev).unsqueeze(1) * k + torch.nonzero(keep_sel).flatten().unsqueeze(0)).flatten()
        child_samples = samples.to(dev, torch.float32)[rows].contiguous()
        child_samples_t = samples_t.to(dev, torch.float32).reshape(-1)[rows].contiguous() if (four and not model.rot_4d) else None
    src = torch.cat([idx_orig, idx_clone, child_parent]).to(torch.int32).contiguous()
    kind = torch.cat([torch.zeros(idx_orig.numel(), dtype=torch.uint8, device=dev),
                      torch.ones(idx_clone.numel(), dtype=torch.uint8, device=dev),
                      torch.full((child_parent.numel(),), 2, dtype=torch.uint8, device=dev)]).contiguous()
    P_new = int(src.numel())
    n_orig, n_clone, n_child = int(idx_orig.numel()), int(idx_clone.numel()), int(child_parent.numel())

    # ---- new flat buffers in one gather ----
    per = model.floats_per_gaussian()
    f = dict(dtype=torch.float32, device=dev)
    new_flat, new_m, new_v = (torch.empty(P_new * per, **f) for _ in range(3))
    rows_arr = (C.c_int32 * len(model.row_floats()))(*model.row_floats())
    old_flat, old = model.flat, {n: model.params[n] for n in model.NAMES}
    with torch.cuda.device(dev):
        rc = _capi.lib.fdgs_densify_gather(len(model.row_floats()), rows_arr, P, P_new, _p(src), _p(kind), _p(old_flat),
                                           _p(optimizer.exp_avg), _p(optimizer.exp_avg_sq), _p(new_flat), _p(new_m), _p(new_v), st)
    _capi._check(rc, "fdgs_densify_gather")
    model._bind(new_flat, torch.zeros(P_new * per, **f), P_new)
    if n_child:
        first = n_orig + n_clone
        parent32 = child_parent.to(torch.int32).contiguous()
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_densify_split(
                n_child, int(N), int(model.rot_4d), int(model.gaussian_dim), _p(parent32), _p(child_samples), _p(child_samples_t),
                _p(old["_xyz"]), _p(old["_t"]), _p(old["_scaling"]), _p(old["_scaling_t"]), _p(old["_rotation"]), _p(old["_rotation_r"]),
                _p(model._xyz[first:]), _p(model._t[first:]), _p(model._scaling[first:]), _p(model._scaling_t[first:]), st)
        _capi._check(rc, "fdgs_densify_split")
    optimizer.rebind(new_m, new_v)

    # ---- statistics: restart from zero after a densification (densification_postfix), masked after prune_only ----
    if prune_only:
        sel = src.long()
        stats.xyz_gradient_accum = stats.xyz_gradient_accum[sel]
        stats.t_gradient_accum = stats.t_gradient_accum[sel]
        stats.denom = stats.denom[sel]
        stats.max_radii2D = stats.max_radii2D[sel]
    else:
        stats.xyz_gradient_accum = torch.zeros((P_new, 1), **f)
        stats.t_gradient_accum = torch.zeros((P_new, 1), **f)
        stats.denom = torch.zeros((P_new, 1), **f)
        stats.max_radii2D = torch.zeros((P_new,), **f)
    del old, old_flat
    return {"P_old": P, "P_new": P_new, "kept": n_orig, "cloned": n_clone, "split_parents": k, "children": n_child}


@torch.no_grad()
def reset_opacity(model, optimizer):
    """gaussian_model.py:371-374 + replace_tensor_to_optimizer: opacity := min(opacity, 0.01), its Adam moments := 0."""
    op = torch.sigmoid(model._opacity)
    new = torch.minimum(op, torch.full_like(op, 0.01))
    model._opacity.copy_(torch.log(new / (1 - new)))
    b, e = model.offsets["_opacity"]
    optimizer.exp_avg[b:e].zero_()
    optimizer.exp_avg_sq[b:e].zero_()
