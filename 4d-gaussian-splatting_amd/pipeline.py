"""One optimizer step = the B views of a batch on two HIP streams (MI355X: overlap instead of a longer queue).

Within one optimizer step the parameters are constant, so the forward of view b+1 does not depend on the backward
of view b (reference train.py:104-170 runs them strictly one after the other through autograd).  The forward front
end is a chain of short latency-bound kernels (preprocess, two radix sorts, scans: HBM / launch latency) while the
backward is dominated by the VALU-bound blend kernel: run on two streams they fill each other's gaps.

    stream F : fwd(0)         fwd(1)         fwd(2) ...
    stream B :        loss+bwd(0)    loss+bwd(1)    ...   all-reduce, Adam

No autograd: forward, fused L1+SSIM (value and gradient) and backward are called explicitly; parameter gradients
are written (view 0) / added (views 1..) straight into the flat bucket (``GaussianParams.grad_sink``), the loss is
pre-scaled by 1 / (B * world) so that the all-reduce SUM is already the mean.  Same arithmetic as
``render_raw`` + ``fused_l1_ssim`` + ``backward()`` per view (tests/test_gpu_api.py compares the two).
"""
import weakref
from typing import List, Sequence

import torch

from . import _capi
from .fused import raw_backward, raw_forward, raw_preprocess_batch, raw_settings
from .gaussian_renderer import diff_gaussian_rasterization as _dgr
from .loss import l1_ssim_grad, l1_ssim_loss, l1_ssim_loss_batch, partials_buffer
from .train_host import allreduce_and_step, allreduce_sh_begin, gather_view_stage_begin, timed_wait


_LAST_STEPPER = {}    # id(model) -> weak reference to the pipeline whose step() touched the model last (overlap_steps: see _model_token)


class StepPipeline:
    def __init__(self, model, optimizer, world_size: int = 1, lambda_dssim: float = 0.2, overlap: bool = True,
                 fuse_sh_adam: bool = True, gather_max_views: int = 32, split_colour: bool = False, batch_views: bool = False,
                 sh_group: int = 1, tile_cull: bool = True, lazy: bool = True, sparse_lists: bool = True, overlap_steps: bool = False):
        """``fuse_sh_adam``: on one rank the SH coefficients are updated straight from the views'
        staged SH gradients (FlatAdam.step_sh_staged) and ``_features.grad`` is NOT materialised for the step; False keeps
        the flush into the gradient bucket followed by the plain Adam step (always the case on several ranks, where the
        bucket is what the all-reduce sums)."""
        self.model, self.opt, self.world, self.lam = model, optimizer, int(world_size), float(lambda_dssim)
        self.fuse_sh_adam = bool(fuse_sh_adam)
        # several ranks, up to this many views per step over all ranks: the ranks exchange the views' SH stages (32 B per
        # Gaussian and view, all-gather) instead of all-reducing the dense SH gradient (12 M B per Gaussian), and every rank
        # runs the fused update on all of them.  A view's stage is final when ITS SH backward has run, so its all-gather is
        # started right there (train_host.gather_view_stage_begin) and travels while the following views are rendered -- the
        # dense gradient is a sum over the step's views and can only leave at the end.  8 GPUs x 4 views: 4 x 67 MB received
        # per GPU, three of the four hidden, against a 2 x 7/8 x 171 MB ring all-reduce after the last view; beyond 32 views
        # the stages outweigh the dense gradient
        self.gather_max_views = int(gather_max_views)
        # fdgs_forward_out.tile_cull: the tile lists hold a Gaussian only where it can reach alpha >= 1/255 (same pixels and
        # gradients as with the reference's lists, a quarter fewer instances at C3)
        self.tile_cull = bool(tile_cull)
        # fdgs_forward_out.lazy (one rank): no forward of the step waits for its num_rendered -- the host enqueues all B views without
        # touching the device (the reference stops in the middle of every forward, rasterizer_impl.cu:302) and reads the views' reports
        # ONCE, before the last view's backward, i.e. before anything of the optimizer step is enqueued; a view whose run-ahead buffers
        # turned out too small (its image is invalid) sends the whole step through the waiting path again -- nothing irreversible has
        # happened by then: the gradient bucket and the SH stages are simply overwritten.  ``lazy_redone`` counts those steps.
        # Several ranks: off (the decision to start over would have to be collective).
        import os
        self.lazy = bool(lazy) and os.environ.get("FDGS_PIPELINE_LAZY", "1") != "0"   # FDGS_PIPELINE_LAZY=0: debugging switch
        self.lazy_redone = 0
        # fdgs_forward_out.sparse_lists (with lazy): every tile's list at a fixed offset of the binning buffer -- the count and scan launches
        # leave the forward's critical chain (FDGS_PIPELINE_SPARSE=0: debugging / A-B switch)
        self.sparse_lists = bool(sparse_lists) and os.environ.get("FDGS_PIPELINE_SPARSE", "1") != "0"
        # ``overlap_steps`` (one rank, two streams, fused SH update; opt-in because it is a promise of the caller's): the head of step
        # k + 1 under the tail of step k.  89 % of the parameters are SH coefficients and their update (HBM-bound, 1.08 GB at C3: ~216 us)
        # is the last thing of a step -- but geometry, binning and sort of the next step's first view read no SH coefficient.  The SH
        # update goes onto a third stream A; the first view of the next step is a split_colour forward whose colour launch goes onto A
        # as well (fdgs_forward_out.colour_stream: in order behind the update, no event), while its geometry + binning + sort run on
        # stream F as soon as the GEOMETRY parameters' Adam step (23 us, stream B) is through.  Same kernels on the same numbers: losses and
        # parameters follow the plain pipeline's to the float-atomics noise two runs of ONE pipeline differ by (tests/test_gpu_api.py).
        # The promise: between two step() calls the caller enqueues nothing on ITS stream that writes the model / optimizer state or
        # that the next forwards depend on, and is done with the previous step's result tensors -- or it calls barrier() first (stream F
        # does not wait for the caller's stream at the start of such a step).  A model whose flat tensor was replaced or modified
        # through torch (densification, reset_opacity: the version counter moves) is noticed and treated like barrier().
        self.overlap_steps = (bool(overlap_steps) and bool(overlap) and int(world_size) == 1 and bool(fuse_sh_adam)
                              and os.environ.get("FDGS_PIPELINE_OVERLAP_STEPS", "1") != "0")
        self.finish_on_F = bool(overlap) and os.environ.get("FDGS_PIPELINE_LOSS_FINISH", "F") == "F"
        # ... and all views' reductions in ONE launch (fdgs_l1_ssim_loss_batch; FDGS_PIPELINE_LOSS_BATCH=0: one launch per view, A/B)
        self.loss_batch = self.finish_on_F and os.environ.get("FDGS_PIPELINE_LOSS_BATCH", "1") != "0"
        # one rank, fused SH update: the Adam step of the 17 geometry parameters per Gaussian is taken INSIDE the last view's geometry
        # backward (fdgs_backward_out.adam: the kernel has just completed their gradient) instead of by a launch of its own at the very end
        # of the step (FDGS_PIPELINE_GEO_ADAM=0: the separate launch, A/B); bit-identical parameters and moments
        self.fuse_geo_adam = os.environ.get("FDGS_PIPELINE_GEO_ADAM", "1") != "0"
        self._geo_adam_done = False
        self._carry = None    # what the model looked like when the last step left its SH update running on stream A
        self.steps_carried = 0
        # several ranks, measurement aid: with ``exchange_pairs`` a list, every wait of stream B for a collective at the end of the step
        # is bracketed by two timing events appended to it (train_host.timed_wait): the exchange time nothing overlapped
        self.exchange_pairs = None
        self.split_colour = bool(split_colour)   # fdgs_forward_out.split_colour for the views' forwards (A/B; off: see DESIGN)
        # View batching (opt-in, B > 1): the SH coefficients -- 12 M bytes per Gaussian, most of what preprocess and SH backward
        # read -- are the same for every view of the step.  ``batch_views``: the views' geometry still runs per view, but their SH
        # colours come from ONE pass over the coefficients before the first view's binning (fdgs_preprocess_batch).  ``sh_group``
        # (below): the SH backward of groups of views from ONE pass (fdgs_sh_backward_batch).
        # Measured at C3, 4 views per step: 120 us less kernel time per step, 3.66 -> 3.56 ms on one stream; with the two
        # streams it is a wash (3.04 -> 3.04-3.10 ms): the batched head and tail of the step have nothing to overlap with.  Off by
        # default.
        self.batch_views = bool(batch_views)
        # SH backward of ``sh_group`` consecutive views in one pass over the coefficients (fdgs_sh_backward_batch) on stream B, with
        # the forwards left per view: the batch kernel takes two views in 66 us against 2 x 50 us for the per-view kernel (its lanes
        # pair up on a Gaussian, one view each), four in 120-135 us.  1 (default): the per-view SH backward inside
        # fdgs_rasterize_backward -- in the two-stream step the grouping is a loss (pairs: 1330 -> 1308 images/s at C3, all four: 1278),
        # because a group's SH + geometry backward waits for the group's last blend backward.
        self.sh_group = int(sh_group)
        self._gacc_b = None   # [B, P, 16] persistent always-zero accumulators, one per view (the batched SH backward reads all of them)
        dev = model.flat.device
        self.dev = dev
        # (Tried and dropped, with measurements on MI355X: a high-priority F stream and a CU-masked B stream change
        # nothing or hurt; persistent blend kernels that leave wave slots free for the other stream lose more to
        # load imbalance / ~100 ns same-address atomics than the overlap returns.  See DESIGN.md.)
        self.sF = torch.cuda.Stream(dev)
        self.sB = torch.cuda.Stream(dev) if overlap else self.sF
        self.sA = torch.cuda.Stream(dev) if self.overlap_steps else None
        self.sink = model.grad_sink()
        self._up = {}
        self._gacc = None   # persistent, always-zero blend-backward accumulator (no memset per view)
        self._parts, self._parts_hw = None, None   # [B, 2, num_partials]: the views' partial loss sums (one reduction launch per step)
        self._sh_stage = None   # [B, P, 8]: deferred SH gradient (fdgs_backward_out.sh_stage), flushed once per step
        self._gathered = None   # [B, world, P, 8]: the stages of all ranks (several ranks, gather mode)

    def _upstream(self, B):
        if B not in self._up:
            # filled on the stream that reads it (B): created on the caller's stream AFTER sB.wait_stream(main) had been recorded, the
            # fill raced with the first step's loss backward
            with torch.cuda.stream(self.sB):
                self._up[B] = torch.full((1,), 1.0 / (B * self.world), dtype=torch.float32, device=self.dev)
        return self._up[B]

    def barrier(self):
        """overlap_steps: the caller has touched the model, the optimizer state or anything else the next step reads on its stream --
        the next step() waits for that stream before its first launch (as every step does without overlap_steps)."""
        self._carry = None

    def _model_token(self):
        m = self.model
        last = _LAST_STEPPER.get(id(m))
        # (another pipeline that stepped the same model in between counts as the caller having touched it)
        return (id(m.flat), m.flat._version, m.P, m.flat.data_ptr(), last is not None and last() is self)

    def step(self, cams: Sequence, gts: Sequence[torch.Tensor], pipe, bg: torch.Tensor, scaling_modifier: float = 1.0):
        """Runs forward + loss + backward of every view, the gradient all-reduce and the optimizer step.
        Returns (list of per-view results dict(render, radii, depth, alpha_T, flow, viewspace_grad, num_rendered), list
        of losses); the tensors may be used on the caller's stream until the next call of step()."""
        if self.lazy and self.world == 1 and not (self.sh_group > 1 and len(cams) > 1):
            out = self._step_views(cams, gts, pipe, bg, scaling_modifier, True)
            if out is not None:
                return out
            self.lazy_redone += 1
        return self._step_views(cams, gts, pipe, bg, scaling_modifier, False)

    def _step_views(self, cams, gts, pipe, bg, scaling_modifier, lazy):
        """step(); ``lazy``: see __init__ -- returns None when a view did not fit its run-ahead buffers (nothing of the optimizer step
        has been enqueued then)."""
        B = len(cams)
        main = torch.cuda.current_stream(self.dev)
        m = self.model
        # overlap_steps: stream F has waited for stream B (the geometry parameters' Adam step included) at the end of the previous step;
        # the caller's stream has nothing new for the forwards (the promise) but waits for the SH update on stream A -- so F must not
        # wait for it.  B does: its first launch of the step is the first view's loss, behind that view's colours anyway
        carried = self.overlap_steps and self._carry is not None and self._carry == self._model_token()
        self._carry = None
        _LAST_STEPPER[id(m)] = weakref.ref(self)
        # A carried step is only in order when its FIRST forward is a split_colour forward whose colour launch goes onto stream A, behind
        # the previous step's SH update.  The view-batched colour pass (batch_views, B > 1) and the batched step (sh_group > 1) evaluate SH
        # colours on stream F: they must not start before that update is through (nondeterministic colours otherwise, with no error)
        if carried and ((self.batch_views and B > 1) or (self.sh_group > 1 and B > 1)):
            self.sF.wait_stream(self.sA)
            carried = False
        if carried:
            self.steps_carried += 1
        else:
            self.sF.wait_stream(main)
        if self.sB is not self.sF:
            self.sB.wait_stream(main)
        up = self._upstream(B)
        if self._gacc is None or self._gacc.shape[0] != m.P:
            with torch.cuda.stream(self.sB):
                self._gacc = torch.zeros((m.P, 16), dtype=torch.float32, device=self.dev)
        # deferred SH gradient: with B > 1 views per step every view stages the 8 numbers it contributes to dL_dsh and ONE
        # flush per step writes the 3 M floats per Gaussian (instead of a read-modify-write of them per view)
        # (on one rank also for B = 1: the stage then feeds the fused SH flush + Adam kernel and dL_dsh is never written at all)
        fuse = self.fuse_sh_adam and self.world == 1
        gather = self.fuse_sh_adam and self.world > 1 and self.world * B <= self.gather_max_views
        defer_sh = B > 1 or fuse or gather
        if defer_sh and (self._sh_stage is None or self._sh_stage.shape[0] != B or self._sh_stage.shape[1] != m.P):
            with torch.cuda.stream(self.sB):
                self._sh_stage = torch.empty((B, m.P, 8), dtype=torch.float32, device=self.dev)
        if gather and (self._gathered is None or self._gathered.shape[0] != B or self._gathered.shape[2] != m.P):
            with torch.cuda.stream(self.sB):
                self._gathered = torch.empty((B, self.world, m.P, 8), dtype=torch.float32, device=self.dev)
        results, losses, keep, pend_loss = [], [], [], []
        R_last = -1
        sh_handle = []
        sh_gather = []     # gather: the work handles of the views' stage exchanges
        sh_stepped = []    # fuse: did the SH update run (on stream F) behind the last view's SH backward?
        if self.sh_group > 1 and B > 1 and defer_sh:
            return self._step_batched(cams, gts, pipe, bg, scaling_modifier, main, up, fuse, gather, sh_handle, sh_gather, sh_stepped)
        handles = [None] * B
        if self.batch_views and B > 1:
            # the SH colours of all views in one pass over the coefficients, ahead of the first view's binning
            with torch.cuda.stream(self.sF):
                sets = [raw_settings(c, m, pipe, bg, scaling_modifier) for c in cams]
                handles = raw_preprocess_batch([s_[0] for s_ in sets], *sets[0][1], tile_cull=self.tile_cull)
        for b in range(B):
            with torch.cuda.stream(self.sF):
                rs, (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var) = raw_settings(
                    cams[b], m, pipe, bg, scaling_modifier)
                (R, color, flow, depth, T, radii, geom, binb, img, _covs, out_means3D) = raw_forward(
                    rs, xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var, preprocessed=handles[b],
                    split_colour=(self.split_colour or (b == 0 and self.sA is not None and fuse)) and handles[b] is None,
                    colour_stream=self.sA if (b == 0 and self.sA is not None and fuse and handles[b] is None) else None,
                    tile_cull=self.tile_cull, lazy=lazy and handles[b] is None, sparse_lists=self.sparse_lists and lazy and handles[b] is None)
                ev = torch.cuda.Event()
                ev.record(self.sF)
            with torch.cuda.stream(self.sB):
                self.sB.wait_event(ev)
                if self.loss_batch and (self._parts is None or self._parts.shape[0] != B or self._parts_hw != tuple(color.shape)):
                    # the views' partial loss sums in one buffer: ONE reduction launch per step (below) instead of one per view
                    self._parts, self._parts_hw = partials_buffer(B, color.shape[0], color.shape[1], color.shape[2], self.dev), tuple(color.shape)
                g_color, loss_handle = l1_ssim_grad(color, gts[b], self.lam, up, parts=self._parts[b] if self.loss_batch else None)
                if self.finish_on_F and b == B - 1:
                    ev_parts = torch.cuda.Event()
                    ev_parts.record(self.sB)   # every view's partial sums are there
                if lazy and b == B - 1:
                    # the one look at the device per step: did every view's lists fit?  (the last forward's tile scan has usually run
                    # by now -- the host is about one view ahead of the device here, not inside every forward)
                    with torch.cuda.stream(self.sF):
                        _pend, failed, reported = _capi.forward_lazy_status(self.dev, wait=True)
                    if failed:
                        main.wait_stream(self.sB)
                        main.wait_stream(self.sF)
                        if self.sA is not None:
                            main.wait_stream(self.sA)
                        self.sF.wait_stream(self.sB)
                        return None
                    lazy_ix = [i for i, r_ in enumerate(results) if r_["num_rendered"] < 0] + ([b] if R < 0 else [])
                    for i, r_val in zip(lazy_ix, reported[-len(lazy_ix):] if lazy_ix else []):
                        if i < b:
                            results[i]["num_rendered"] = r_val
                        else:
                            R_last = r_val
                # last view of the step on several ranks: the SH gradients (88 % of the bucket) are final once this view's
                # SH backward has run -- their all-reduce starts there and travels while the geometry backward runs
                after_sh = None
                if b == B - 1 and fuse and self.sB is not self.sF:
                    # the SH stages are complete once this view's SH backward has run: the fused SH flush + Adam (HBM-bound)
                    # goes onto the idle F stream and runs next to the geometry backward (latency-bound) of stream B
                    # (overlap_steps: onto stream A, which the next step's first view puts its colour launch on)
                    def after_sh():
                        done = torch.cuda.Event()
                        done.record(self.sB)
                        s_up = self.sA if self.sA is not None else self.sF
                        with torch.cuda.stream(s_up):
                            s_up.wait_event(done)
                            self.opt.step_count += 1
                            sh_stepped.append(self.opt.step_sh_staged(self._sh_stage, rs, _dgr.analytic_sh_gradients()))
                elif gather:
                    def after_sh(b=b):   # this view's stage is final: its exchange travels while the following views are rendered
                        sh_gather.append(gather_view_stage_begin(self._sh_stage[b], self._gathered[b]))
                elif b == B - 1 and ((defer_sh and not fuse) or self.world > 1):
                    def after_sh():
                        if defer_sh:
                            _capi.sh_flush(self._sh_stage, self.sink["dL_dsh"], rs.sh_degree, rs.sh_degree_t, rs.gaussian_dim,
                                           rs.force_sh_3d, _dgr.analytic_sh_gradients())
                        if self.world > 1:
                            sh_handle.append(allreduce_sh_begin(m, self.world))
                geo_adam = None
                self._geo_adam_done = False
                if b == B - 1 and fuse and self.sB is not self.sF and self.fuse_geo_adam and m.rot_4d and m.gaussian_dim == 4:
                    def geo_adam():
                        # (called after after_sh: the step count is this step's, and it is known whether the fused SH update ran --
                        # the tail's fall-back, flush + one Adam over the whole bucket, must not meet parameters already stepped)
                        if not (sh_stepped and sh_stepped[0]):
                            return None
                        self._geo_adam_done = True
                        lr = {s_["name"]: s_["lr"] for s_ in self.opt.named_segments()}
                        return dict(flat=m.flat, exp_avg=self.opt.exp_avg, exp_avg_sq=self.opt.exp_avg_sq, betas=self.opt.betas, eps=self.opt.eps,
                                    step=self.opt.step_count,
                                    lr=dict(means3D=lr["_xyz"], opacities=lr["_opacity"], ts=lr["_t"], scales=lr["_scaling"], scales_t=lr["_scaling_t"],
                                            rotations=lr["_rotation"], rotations_r=lr["_rotation_r"]))
                grads = raw_backward(rs, xyz, out_means3D, radii, feats, opacity, ts, scaling, scaling_t, rotation,
                                     rotation_r, prefilter_var, geom, R, binb, img, g_color, None, None, None,
                                     self.sink, b > 0, grad_accum=self._gacc, after_sh=after_sh,
                                     sh_stage=self._sh_stage[b] if defer_sh else None, per_view_outputs=False, geometry_adam=geo_adam)
                # the small reduction of the loss VALUE: behind the backward (nothing of the step waits for it) -- or, with two streams, all
                # views' reductions on stream F behind its last forward (below): stream B's chain is the step's critical path, and 4 x ~6 us
                # of a one-workgroup kernel were part of it
                loss = None if self.finish_on_F else l1_ssim_loss(loss_handle)
                pend_loss.append(loss_handle)
            # buffers allocated on F are read on B: keep them alive until F has waited for B (end of the step)
            keep.append((geom, binb, img, out_means3D, g_color, T))
            results.append({"render": color, "radii": radii, "depth": depth, "alpha_T": T, "flow": flow,
                            "viewspace_grad": grads[0], "num_rendered": R_last if (lazy and b == B - 1 and R < 0) else R})
            losses.append(loss)
        if self.finish_on_F:
            with torch.cuda.stream(self.sF):
                self.sF.wait_event(ev_parts)
                losses = l1_ssim_loss_batch(self._parts, pend_loss) if self.loss_batch else [l1_ssim_loss(h) for h in pend_loss]
        self._optimizer_tail(rs, fuse, gather, sh_handle, sh_gather, sh_stepped)
        main.wait_stream(self.sB)
        main.wait_stream(self.sF)
        self.sF.wait_stream(self.sB)
        if self.sA is not None:
            main.wait_stream(self.sA)
            if fuse and sh_stepped and sh_stepped[0]:
                self._carry = self._model_token()    # the next step may start under this step's SH update (see __init__)
            else:
                self.sF.wait_stream(self.sA)
        # The returned tensors live in the F / B streams' allocator pools.  `main` has waited for both streams, and the
        # next step() makes both streams wait for `main` first, so they are safe to read on `main` until then
        # (no record_stream: it would defer every free by an event query and grow the pools).
        del keep, handles
        return results, losses

    def _optimizer_tail(self, rs, fuse, gather, sh_handle, sh_gather, sh_stepped):
        """Exchange (several ranks) + optimizer step on stream B, after the last view's backward."""
        m = self.model
        with torch.cuda.stream(self.sB):
            # the losses were scaled by 1 / (B * world): SUM = mean; Adam on chunk k overlaps the all-reduce of chunk k+1
            if fuse:
                if not sh_stepped:
                    self.opt.step_count += 1
                    sh_stepped.append(self.opt.step_sh_staged(self._sh_stage, rs, _dgr.analytic_sh_gradients()))
                if sh_stepped[0]:
                    if not self._geo_adam_done:     # (else: taken inside the last view's geometry backward)
                        self.opt.step_range(0, m.offsets["_features"][0])
                else:   # layout the fused kernel does not take: the two passes
                    _capi.sh_flush(self._sh_stage, self.sink["dL_dsh"], rs.sh_degree, rs.sh_degree_t, rs.gaussian_dim,
                                   rs.force_sh_3d, _dgr.analytic_sh_gradients())
                    self.opt.step_range(0, m.flat.numel())
            elif gather:
                import torch.distributed as dist
                feat = m.offsets["_features"][0]
                geo = dist.all_reduce(m.flat_grad[:feat], op=dist.ReduceOp.SUM, async_op=True)   # 17 floats per Gaussian
                self.opt.step_count += 1
                for work in sh_gather:
                    timed_wait(work, self.exchange_pairs)
                stages = self._gathered.view(-1, m.P, 8)   # [B x world] views: view-major, rank-minor, the same on every rank
                ok = self.opt.step_sh_staged(stages, rs, _dgr.analytic_sh_gradients())
                if not ok:   # layout the fused kernel does not take: every rank builds the same summed dL_dsh from all the stages
                    _capi.sh_flush(stages, self.sink["dL_dsh"], rs.sh_degree, rs.sh_degree_t, rs.gaussian_dim, rs.force_sh_3d,
                                   _dgr.analytic_sh_gradients())
                timed_wait(geo, self.exchange_pairs)
                self.opt.step_range(0, feat if ok else m.flat.numel())
            else:
                allreduce_and_step(m, self.opt, self.world, chunks=4, average=False, sh_handle=sh_handle[0] if sh_handle else None,
                                   wait_pairs=self.exchange_pairs)

    def _after_sh(self, rs, fuse, gather, defer_sh, sh_handle, sh_gather, sh_stepped):
        """What starts as soon as the SH stages of the step are complete (stream B is current): the fused SH update on the idle F
        stream (one rank), the exchange of the stages, or the flush + the all-reduce of the SH part of the bucket (several ranks)."""
        m = self.model
        if fuse and self.sB is not self.sF:
            done = torch.cuda.Event()
            done.record(self.sB)
            with torch.cuda.stream(self.sF):
                self.sF.wait_event(done)
                self.opt.step_count += 1
                sh_stepped.append(self.opt.step_sh_staged(self._sh_stage, rs, _dgr.analytic_sh_gradients()))
        elif gather:
            for b in range(self._sh_stage.shape[0]):
                sh_gather.append(gather_view_stage_begin(self._sh_stage[b], self._gathered[b]))
        elif (defer_sh and not fuse) or self.world > 1:
            if defer_sh:
                _capi.sh_flush(self._sh_stage, self.sink["dL_dsh"], rs.sh_degree, rs.sh_degree_t, rs.gaussian_dim,
                               rs.force_sh_3d, _dgr.analytic_sh_gradients())
            if self.world > 1:
                sh_handle.append(allreduce_sh_begin(m, self.world))

    def _step_batched(self, cams, gts, pipe, bg, scaling_modifier, main, up, fuse, gather, sh_handle, sh_gather, sh_stepped):
        """step() with the SH backward of ``sh_group`` consecutive views done in one pass over the coefficients (see __init__):
        stream B runs loss + blend backward per view and, after every ``sh_group`` views, ONE SH backward pass for the group
        followed by the group's geometry backward.  (``batch_views``: stream F starts with the geometry of every view and ONE
        colour pass, then per view binning + blend.)  Same arithmetic per view as
        the unbatched step (forward bit-identical; tests/test_gpu_api.py)."""
        B, m = len(cams), self.model
        G = max(1, min(self.sh_group, B))
        if self._gacc_b is None or self._gacc_b.shape[0] != B or self._gacc_b.shape[1] != m.P:
            with torch.cuda.stream(self.sB):
                self._gacc_b = torch.zeros((B, m.P, 16), dtype=torch.float32, device=self.dev)
        results, losses, keep, pend, grads_of = [], [], [], [], {}
        with torch.cuda.stream(self.sF):
            sets = [raw_settings(c, m, pipe, bg, scaling_modifier) for c in cams]
            (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var) = sets[0][1]
            handles = [None] * B
            if self.batch_views:
                handles = raw_preprocess_batch([s[0] for s in sets], xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var,
                                               tile_cull=self.tile_cull)
        for b in range(B):
            rs = sets[b][0]
            with torch.cuda.stream(self.sF):
                (R, color, flow, depth, T, radii, geom, binb, img, _covs, out_means3D) = raw_forward(
                    rs, xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var, preprocessed=handles[b],
                    split_colour=self.split_colour and handles[b] is None, tile_cull=self.tile_cull)
                ev = torch.cuda.Event()
                ev.record(self.sF)
            with torch.cuda.stream(self.sB):
                self.sB.wait_event(ev)
                g_color, loss_handle = l1_ssim_grad(color, gts[b], self.lam, up)
                pend.append(raw_backward(rs, xyz, out_means3D, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r,
                                         prefilter_var, geom, R, binb, img, g_color, None, None, None, self.sink, b > 0,
                                         grad_accum=self._gacc_b[b], sh_stage=self._sh_stage[b], begin_only=True, per_view_outputs=False))
                losses.append(l1_ssim_loss(loss_handle))
                if (b + 1) % G == 0 or b == B - 1:
                    first = b - (b % G)
                    _dgr._C.sh_backward_batch(pend[first:b + 1])
                    if b == B - 1:   # the stages of the step are complete
                        self._after_sh(rs, fuse, gather, True, sh_handle, sh_gather, sh_stepped)
                    for v in range(first, b + 1):
                        results_v = _dgr._C.backward_finish(pend[v])
                        keep.append(results_v)
                        grads_of[v] = results_v[0]
            keep.append((geom, binb, img, out_means3D, g_color, T))
            results.append({"render": color, "radii": radii, "depth": depth, "alpha_T": T, "flow": flow, "num_rendered": R})
        for v in range(B):
            results[v]["viewspace_grad"] = grads_of[v]
        self._optimizer_tail(rs, fuse, gather, sh_handle, sh_gather, sh_stepped)
        main.wait_stream(self.sB)
        main.wait_stream(self.sF)
        self.sF.wait_stream(self.sB)
        del keep, pend, handles
        return results, losses
