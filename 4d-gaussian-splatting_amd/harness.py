"""Frame-parallel training harness: what sits around the step pipeline in
the reference's ``training()`` loop (train.py:80-260), arranged for one process per GPU.

* ``FrameShard``   -- the reference's ``DataLoader(shuffle=True, drop_last=True, batch_size=B)`` (train.py:80)
                      for N ranks: every epoch one shared-seed permutation of the views, cut into global batches
                      of N*B, rank r taking views r*B .. r*B+B-1 of each.  No communication: the permutation is a
                      pure function of (seed, epoch).
* ``expon_lr``     -- utils/general_utils.py:30-63, the xyz learning-rate schedule (scene/gaussian_model.py:354-365).
* ``DensificationStats`` -- the statistics the reference accumulates for densification (train.py:162-184, 229-236;
                      scene/gaussian_model.py:631-642), reduced over ranks with three small all-reduces per step
                      so that every rank holds IDENTICAL statistics: any densification decision derived from them
                      is then the same on all ranks without a broadcast ("deterministic densification").
* ``train``        -- the loop: learning rate, SH degree schedule, one ``StepPipeline.step`` per iteration (forward,
                      fused loss, backward, gradient all-reduce, Adam), statistics, logging.  Densification / pruning
                      itself (rank 4 of the same table) is a hook: ``on_densify(model, optimizer, stats, iteration)``.
"""
import math
from typing import Callable, Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .pipeline import StepPipeline


def expon_lr(step: int, lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
             max_steps: int = 1000000) -> float:
    """utils/general_utils.py:48-61: log-linear interpolation lr_init -> lr_final over max_steps, optional warm-up."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    log_lerp = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return float(delay_rate * log_lerp)


class FrameShard:
    """Per-rank view indices: epoch -> global batches of world*batch_size (drop_last) -> this rank's slice."""

    def __init__(self, n_views: int, batch_size: int, world_size: int = 1, rank: int = 0, seed: int = 0, shuffle: bool = True):
        if n_views < batch_size * world_size:
            raise ValueError("need at least world_size * batch_size = %d views, got %d" % (batch_size * world_size, n_views))
        self.n, self.B, self.world, self.rank, self.seed, self.shuffle = n_views, batch_size, world_size, rank, seed, shuffle

    def batches_per_epoch(self) -> int:
        return self.n // (self.B * self.world)

    def epoch(self, epoch: int) -> List[List[int]]:
        if self.shuffle:
            g = torch.Generator(device="cpu").manual_seed(self.seed + epoch)
            perm = torch.randperm(self.n, generator=g).tolist()
        else:
            perm = list(range(self.n))
        G = self.B * self.world
        return [perm[k * G + self.rank * self.B: k * G + (self.rank + 1) * self.B] for k in range(self.batches_per_epoch())]

    def __iter__(self) -> Iterator[List[int]]:
        e = 0
        while True:
            for b in self.epoch(e):
                yield b
            e += 1


class DensificationStats:
    """xyz_gradient_accum, t_gradient_accum, denom, max_radii2D of scene/gaussian_model.py, identical on every rank."""

    def __init__(self, P: int, device, world_size: int = 1):
        f = dict(dtype=torch.float32, device=device)
        self.xyz_gradient_accum = torch.zeros((P, 1), **f)
        self.t_gradient_accum = torch.zeros((P, 1), **f)
        self.denom = torch.zeros((P, 1), **f)
        self.max_radii2D = torch.zeros((P,), **f)
        self.world = int(world_size)

    @torch.no_grad()
    def update(self, results: Sequence[Dict[str, torch.Tensor]], t_grad: Optional[torch.Tensor], global_batch: int):
        """``results``: this rank's per-view outputs of StepPipeline.step (radii, viewspace_grad = dL/dmeans2D with the
        loss scaled by 1/global_batch, as train.py:162); ``t_grad``: the accumulated, all-reduced dL/dt [P,1].
        train.py:164-184 with batch_size = global_batch."""
        if self.xyz_gradient_accum.is_cuda:
            self._update_gpu(results, t_grad, global_batch)
            return
        vis = torch.stack([r["radii"] > 0 for r in results], 1)
        count = vis.sum(1).to(torch.float32)
        radii = torch.stack([r["radii"] for r in results], 1).max(1)[0].to(torch.float32)
        pgrad = torch.stack([torch.norm(r["viewspace_grad"][:, :2], dim=-1) for r in results], 1).sum(1)
        if self.world > 1:
            import torch.distributed as dist
            packed = torch.stack([count, pgrad], 0)
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
            dist.all_reduce(radii, op=dist.ReduceOp.MAX)
            count, pgrad = packed[0], packed[1]
        seen = count > 0
        cnt = count.clamp(min=1.0)
        pgrad = torch.where(seen, pgrad * float(global_batch) / cnt, pgrad).unsqueeze(1)   # same operation order as train.py:173
        self.max_radii2D[seen] = torch.max(self.max_radii2D[seen], radii[seen])       # train.py:231
        self.xyz_gradient_accum[seen] += pgrad[seen]                                   # gaussian_model.py:638-639
        self.denom[seen] += 1
        if t_grad is not None:
            tg = torch.where(seen, t_grad[:, 0] * float(global_batch) / cnt, t_grad[:, 0]).unsqueeze(1)
            self.t_gradient_accum[seen] += tg[seen]                                    # train.py:178-181, gaussian_model.py:641


    def _update_gpu(self, results, t_grad, global_batch):
        """The same update in two small kernels (csrc/densify.hip) instead of ~25 PyTorch ops over [P] tensors (1.4 ms per
        step at 300 k Gaussians -- a quarter of the step -- against 20 us)."""
        import ctypes as C
        from . import _capi
        dev = self.xyz_gradient_accum.device
        P = self.xyz_gradient_accum.shape[0]
        radii = [r["radii"].contiguous() for r in results]
        grads = [r["viewspace_grad"].contiguous() for r in results]
        tmp = torch.empty((3, P), dtype=torch.float32, device=dev)   # count, pgrad | radii_max
        st = _capi.current_stream_handle(dev)
        with torch.cuda.device(dev):
            # the kernel takes at most 16 views per call: larger per-rank batches go in groups, combined like the ranks are
            GROUP = 16
            for g0 in range(0, len(results), GROUP):
                rad, grd = radii[g0:g0 + GROUP], grads[g0:g0 + GROUP]
                n = len(rad)
                rp = (C.c_void_p * n)(*[t.data_ptr() for t in rad])
                gp = (C.c_void_p * n)(*[t.data_ptr() for t in grd])
                dst = tmp if g0 == 0 else torch.empty_like(tmp)
                rc = _capi.lib.fdgs_densify_stats_local(P, n, rp, gp, dst[0].data_ptr(), dst[1].data_ptr(), dst[2].data_ptr(), st)
                _capi._check(rc, "fdgs_densify_stats_local")
                if g0 > 0:
                    tmp[:2] += dst[:2]
                    torch.maximum(tmp[2], dst[2], out=tmp[2])
            if self.world > 1:
                import torch.distributed as dist
                dist.all_reduce(tmp[:2], op=dist.ReduceOp.SUM)
                dist.all_reduce(tmp[2], op=dist.ReduceOp.MAX)
            tg = None if t_grad is None else t_grad.contiguous()
            rc = _capi.lib.fdgs_densify_stats_apply(P, tmp[0].data_ptr(), tmp[1].data_ptr(), tmp[2].data_ptr(),
                                                    None if tg is None else tg.data_ptr(), float(global_batch),
                                                    self.xyz_gradient_accum.data_ptr(), self.t_gradient_accum.data_ptr(),
                                                    self.denom.data_ptr(), self.max_radii2D.data_ptr(), st)
            _capi._check(rc, "fdgs_densify_stats_apply")


def psnr(img: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """utils/image_utils.py: 20 log10(1 / sqrt(mse))."""
    mse = ((img - gt) ** 2).reshape(img.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def train(model, optimizer, cameras: Sequence, gts: Sequence[torch.Tensor], pipe, bg: torch.Tensor, iterations: int,
          batch_size: int = 4, world_size: int = 1, rank: int = 0, seed: int = 0, lambda_dssim: float = 0.2,
          position_lr_init: float = 1.6e-4, position_lr_final: float = 1.6e-6, position_lr_delay_mult: float = 0.01,
          position_lr_max_steps: int = 30000, sh_increase_interval: int = 1000, sh_degree_start: Optional[Sequence[int]] = None,
          white_background: bool = False,
          densify_until_iter: int = 15000, densify_from_iter: int = 500, densification_interval: int = 100,
          opacity_reset_interval: int = 3000, densify_grad_threshold: float = 2e-4, densify_grad_t_threshold: float = 2e-4 / 40,
          thresh_opa_prune: float = 0.005, percent_dense: float = 0.01, cameras_extent: Optional[float] = None,
          densify_until_num_points: int = -1, on_densify: Optional[Callable] = None, log_every: int = 0,
          log: Callable[[str], None] = print, spatial_order: bool = True,
          on_resort: Optional[Callable] = None) -> Dict[str, List[float]]:
    """The reference's training loop (train.py:82-254) over ``cameras`` / ``gts`` (all views, identical on every rank;
    each rank renders its FrameShard slice).  Returns the logged history {"iteration", "loss", "psnr"}.
    Densification (train.py:229-244) runs when ``cameras_extent`` is given: every rank takes the same decisions from the
    all-reduced statistics and draws the split samples from a generator seeded with (seed, iteration), so the replicas
    stay identical without a broadcast.  ``on_densify(model, optimizer, stats, iteration)`` replaces that default.
    SH schedule (train.py:93-94, gaussian_model.py:253-257): every ``sh_increase_interval`` iterations ``oneupSHdegree()``
    raises the spatial degree until it reaches the model's maximum, then the time degree.  ``sh_degree_start`` = (0, 0) is
    the reference's training from scratch (GaussianModel starts at degree 0 / 0); None keeps the model's current active
    degrees (resuming / fine-tuning a model whose coefficients are already populated).  ``white_background`` adds the
    reference's extra opacity reset at ``densify_from_iter`` (train.py:243).
    ``spatial_order``: the model is kept in Morton order of the Gaussians' positions (train_host.spatial_sort: at the start and
    after every densification, when the statistics have just been reset) -- a memory-layout choice with no effect on the
    arithmetic; every rank derives the same permutation from its (identical) parameters.  NOTE: the caller's ``model`` and the
    optimizer's moments are REORDERED IN PLACE (row j becomes the old row perm[j]); a caller that keeps per-Gaussian side data
    indexed like the model passes ``spatial_order=False`` or applies ``on_resort(perm)`` to it (called with every permutation)."""
    from .train_host import spatial_sort

    def resort(stats=None):
        perm = spatial_sort(model, optimizer)
        if on_resort is not None:
            on_resort(perm)
        if stats is not None:   # everything else that is indexed by Gaussian follows the model
            stats.xyz_gradient_accum, stats.t_gradient_accum = stats.xyz_gradient_accum[perm], stats.t_gradient_accum[perm]
            stats.denom, stats.max_radii2D = stats.denom[perm], stats.max_radii2D[perm]

    if spatial_order:
        resort()
    shard = iter(FrameShard(len(cameras), batch_size, world_size, rank, seed))
    steppipe = StepPipeline(model, optimizer, world_size=world_size, lambda_dssim=lambda_dssim)
    stats = DensificationStats(model.P, model.flat.device, world_size)
    if sh_degree_start is not None:
        model.active_sh_degree = min(int(sh_degree_start[0]), model.max_sh_degree)
        model.active_sh_degree_t = min(int(sh_degree_start[1]), model.max_sh_degree_t)
    hist: Dict[str, List[float]] = {"iteration": [], "loss": [], "psnr": []}
    for iteration in range(1, iterations + 1):
        optimizer.set_lr("_xyz", expon_lr(iteration, position_lr_init, position_lr_final, 0, position_lr_delay_mult,
                                          position_lr_max_steps))                       # gaussian_model.py:359-365
        if iteration % sh_increase_interval == 0:                                       # train.py:93-94
            model.oneupSHdegree()
        idx = next(shard)
        results, losses = steppipe.step([cameras[i] for i in idx], [gts[i] for i in idx], pipe, bg)
        if iteration < densify_until_iter:                                              # train.py:229-244
            t_grad = model.params["_t"].grad if model.gaussian_dim == 4 else None      # already all-reduced (mean over the batch)
            if densify_until_num_points < 0 or model.P < densify_until_num_points:
                stats.update(results, t_grad, batch_size * world_size)
                if on_densify is not None:
                    on_densify(model, optimizer, stats, iteration)
                    steppipe.sink = model.grad_sink()
                elif cameras_extent is not None:
                    if iteration > densify_from_iter and iteration % densification_interval == 0:        # train.py:238-240
                        from .densify import densify_and_prune
                        gen = torch.Generator(device=model.flat.device).manual_seed(seed * 1000003 + iteration)
                        size_threshold = 20 if iteration > opacity_reset_interval else None
                        rep = densify_and_prune(model, optimizer, stats, densify_grad_threshold, thresh_opa_prune, cameras_extent,
                                                size_threshold, densify_grad_t_threshold, percent_dense=percent_dense, generator=gen)
                        if spatial_order:
                            resort(stats)
                        steppipe.sink = model.grad_sink()
                        if rank == 0 and log_every:
                            log("[it %5d] densify: %d -> %d Gaussians (%d cloned, %d split)" % (iteration, rep["P_old"], rep["P_new"],
                                                                                              rep["cloned"], rep["split_parents"]))
                    if iteration % opacity_reset_interval == 0 or (white_background and iteration == densify_from_iter):   # train.py:243
                        from .densify import reset_opacity
                        reset_opacity(model, optimizer)
        if log_every and (iteration % log_every == 0 or iteration == 1 or iteration == iterations):
            with torch.no_grad():
                loss = float(torch.stack(losses).mean())
                p = float(psnr(results[-1]["render"], gts[idx[-1]]).mean())
            hist["iteration"].append(iteration); hist["loss"].append(loss); hist["psnr"].append(p)
            if rank == 0:
                log("[it %5d] loss %.5f  psnr %.2f dB  (%d Gaussians, SH degree %d / time %d)" % (iteration, loss, p, model.P,
                                                                                                 model.active_sh_degree, model.active_sh_degree_t))
    return hist
