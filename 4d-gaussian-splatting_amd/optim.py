"""``fdgs.optim.Adam`` -- a ``torch.optim.Optimizer`` for the REFERENCE's model that brings the fused path to the reference's boundary.

The reference builds ``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` over nine param groups of one tensor each, named
"xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "t", "scaling_t", "rotation_r" (scene/gaussian_model.py:331-357), and
its densification edits ``optimizer.state`` / ``group["params"][0]`` directly (``replace_tensor_to_optimizer``,
``_prune_optimizer``, ``cat_tensors_to_optimizer``, :376-452).  Swapping that one constructor for this class keeps all of it
working and changes what happens underneath:

* **homing** -- the parameters are moved into ONE flat fp32 bucket (``train_host.GaussianParams`` layout: geometry first, the SH
  coefficients ``[P, M, 3]`` last).  The user's ``nn.Parameter`` objects stay the same objects; their ``.data`` become views of the
  bucket -- ``f_dc`` / ``f_rest`` are the strided views ``[:, :1, :]`` / ``[:, 1:, :]`` of the one coefficient tensor, so ``render()``
  hands the kernels the contiguous ``[P, M, 3]`` array instead of ``torch.cat``-ing 173 MB per view
  (scene/gaussian_model.py:211-215) -- and ``state[p]["exp_avg" / "exp_avg_sq"]`` are views of two flat moment buffers.
  Whenever the model replaced a parameter or a state tensor (densification, pruning, opacity reset, ``load_state_dict``) the
  next ``render()`` / ``step()`` notices (pointer checks) and homes again: three bucket-sized copies per densification.
* **gradients** -- ``render()`` (this package's) feeds the RAW parameters to the kernels (``fdgs_scene.raw_params``: the
  activations of :179-209 and their derivatives run inside preprocess / preprocess-backward) and the backward writes every
  parameter gradient straight into the flat gradient bucket behind ``p.grad`` (first backward after ``zero_grad``: overwrite;
  further views of the batch, train.py:104-166: add) -- no autograd accumulation pass, no memset.
* **SH gradient deferred** (``defer_sh=True``) -- a view's backward stages the 8 numbers per Gaussian it contributes to dL/dSH
  (``fdgs_backward_out.sh_stage``) and ``step()`` updates the coefficients straight from the staged views
  (``fdgs_adam_step_sh``): the dense ``[P, M, 3]`` gradient (89 % of all gradient bytes) is never written.
  ``_features_dc.grad`` / ``_features_rest.grad`` are then ``None`` after ``backward()``; ``defer_sh=False`` materialises them.
* **step** -- one fused HIP launch over the bucket (``fdgs_adam_step``; ``torch.optim.Adam`` arithmetic: no amsgrad, no weight
  decay) with the groups' CURRENT ``lr`` (``update_learning_rate`` keeps working), instead of 9 x ~12 small kernels.

**Parameters whose ``grad`` is ``None`` at ``step()``**: ``torch.optim.Adam`` skips them.  Here the nine homed tensors are stepped
together as soon as ANY of them received a gradient since the gradients were last cleared; one that received none is stepped with a
zero gradient (its moments decay, it moves on them) -- the state the reference's loop is in anyway, where every backward reaches
all nine.  If ALL of them are ``None`` (cleared by ``zero_grad()`` or by hand, no backward since) nothing is stepped, as in torch.
**One step count for the bucket**: ``torch.optim.Adam`` counts steps per parameter and does not count a skipped one, so after the
reference's ``reset_opacity`` (the opacity tensor is replaced between backward() and step(): no gradient, skipped, train.py:243-249)
its opacity lags one step behind the other eight tensors in the bias corrections ``1 - beta^t``; here all nine share the bucket's
count.  The reference resets at iteration 3000 at the earliest, where one step changes the corrections by 5e-5 relative.

Anything else in ``param_groups`` (a group without one of the nine names) is stepped tensor by tensor with the same kernel.
Gradients that reach a parameter through plain autograd (another loss term, the Python-SH branch of ``render()``) are absorbed:
``p.grad`` is the bucket view, so autograd adds into it in place.
"""
from typing import Dict, Optional

import torch

from . import _capi
from .train_host import FlatAdam, GaussianParams

# param-group name (scene/gaussian_model.py:331-350) -> GaussianParams segment
_GEOMETRY = {"xyz": "_xyz", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation", "t": "_t",
             "scaling_t": "_scaling_t", "rotation_r": "_rotation_r"}
_REQUIRED = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


class _StageSettings:
    """What fdgs_sh_flush / fdgs_adam_step_sh need to know about the staged views (the views' raster settings)."""

    def __init__(self, rs):
        self.sh_degree, self.sh_degree_t = int(rs.sh_degree), int(rs.sh_degree_t)
        self.gaussian_dim, self.force_sh_3d = int(rs.gaussian_dim), bool(rs.force_sh_3d)

    def key(self):
        return (self.sh_degree, self.sh_degree_t, self.gaussian_dim, self.force_sh_3d)


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, defer_sh: bool = True,
                 lazy_forward: bool = False):
        """Same arguments as ``torch.optim.Adam`` (``weight_decay`` / ``amsgrad`` must stay off: the reference does not use them).
        ``defer_sh``: see the module docstring.  ``lazy_forward``: ``render()`` does not wait for ``num_rendered``
        (fdgs_forward_out.lazy: the host never blocks inside the forward); ``step()`` then looks at the views' reports before it
        updates anything and SKIPS the update (returning False, with a warning) if a view of the batch had outgrown its run-ahead
        buffers -- that view's image was background only.  Off by default: a skipped step is invisible to the training loop."""
        if weight_decay or amsgrad:
            raise ValueError("fdgs.optim.Adam: weight_decay / amsgrad are not supported (the reference trains with neither)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.defer_sh, self.lazy_forward = bool(defer_sh), bool(lazy_forward)
        self._gp: Optional[GaussianParams] = None      # bucket layout (offsets, views) of the homed parameters
        self._fa: Optional[FlatAdam] = None            # the fused step + moment buckets
        self._homed: Dict[str, tuple] = {}             # group name -> (param object, data_ptr, exp_avg ptr, exp_avg_sq ptr)
        self._views: Dict[str, tuple] = {}             # group name -> (data view, grad view, exp_avg view, exp_avg_sq view)
        self._fresh = True                             # no backward has written the gradient bucket since zero_grad()
        self._stages = None                            # [cap, P, 8] staged SH gradients of the views since the last step
        self._n_staged = 0
        self._stage_rs: Optional[_StageSettings] = None
        self._gacc = None                              # [P, 16] persistent all-zero accumulator of the blend backward
        self._lazy_views = 0                           # lazy forwards since the last step()
        self._dense_sh = False                         # a dense SH gradient exists in the bucket this step (not only staged views)
        self.steps_skipped = 0
        self.homings = 0

    # ------------------------------------------------------------------ homing
    def _named(self):
        return {g.get("name"): g for g in self.param_groups if g.get("name") is not None and len(g["params"]) == 1}

    def bucketable(self) -> bool:
        g = self._named()
        if not all(n in g for n in _REQUIRED):
            return False
        xyz = g["xyz"]["params"][0]
        return xyz.is_cuda and xyz.dtype == torch.float32 and xyz.dim() == 2 and xyz.shape[1] == 3

    def is_homed(self) -> bool:
        if self._gp is None or not self._homed:
            return False
        g = self._named()
        for name, (p, dptr, aptr, sptr) in self._homed.items():
            grp = g.get(name)
            if grp is None or grp["params"][0] is not p or p.data_ptr() != dptr:
                return False
            st = self.state.get(p)
            if not st or st.get("exp_avg") is None or st.get("exp_avg_sq") is None:   # (a state that holds only "step": not homed)
                return False
            if st["exp_avg"].data_ptr() != aptr or st["exp_avg_sq"].data_ptr() != sptr:
                return False
        return len([n for n in g if n in _GEOMETRY or n in ("f_dc", "f_rest")]) == len(self._homed)

    def ensure_homed(self) -> bool:
        """Moves the reference's nine parameter tensors (and their Adam state) into the flat buckets if they are not there (any
        more).  False: the param groups are not the reference's (nothing is bucketed; step() handles every tensor on its own)."""
        if not self.bucketable():
            return False
        if not self.is_homed():
            self._home()
        return True

    @torch.no_grad()
    def _home(self):
        g = self._named()
        xyz = g["xyz"]["params"][0]
        dev, P = xyz.device, int(xyz.shape[0])
        dc, rest = g["f_dc"]["params"][0], g["f_rest"]["params"][0]
        if dc.dim() != 3 or rest.dim() != 3 or dc.shape[1] != 1 or dc.shape[2] != 3 or rest.shape[2] != 3 or dc.shape[0] != P or rest.shape[0] != P:
            raise ValueError("fdgs.optim.Adam: f_dc / f_rest must be [P, 1, 3] / [P, M - 1, 3] (scene/gaussian_model.py:297-298)")
        M = 1 + int(rest.shape[1])
        # gradients that exist already (plain autograd before the first homing / after a re-layout) move into the new bucket
        old_grad = {n: grp["params"][0].grad for n, grp in g.items()}
        old_state = {n: dict(self.state.get(grp["params"][0], {})) for n, grp in g.items()}
        step = 0
        for st in old_state.values():
            if "step" in st:
                step = max(step, int(float(st["step"])))
        if self._fa is not None:
            step = max(step, self._fa.step_count)
        gp = GaussianParams.__new__(GaussianParams)
        gp.M = M
        total = P * gp.floats_per_gaussian()
        gp._bind(torch.zeros(total, dtype=torch.float32, device=dev), torch.zeros(total, dtype=torch.float32, device=dev), P)
        betas, eps = g["xyz"]["betas"], g["xyz"]["eps"]
        fa = FlatAdam(gp, betas=tuple(betas), eps=float(eps))
        fa.step_count = step
        shapes = {"_xyz": (P, 3), "_opacity": (P, 1), "_scaling": (P, 3), "_rotation": (P, 4), "_t": (P, 1), "_scaling_t": (P, 1), "_rotation_r": (P, 4)}

        def seg(buf, name):
            b, e = gp.offsets[name]
            return buf[b:e]

        views = {}
        for gname, seg_name in _GEOMETRY.items():
            if gname in g:
                views[gname] = tuple(seg(buf, seg_name).view(shapes[seg_name]) for buf in (gp.flat, gp.flat_grad, fa.exp_avg, fa.exp_avg_sq))
        feat = tuple(seg(buf, "_features").view(P, M, 3) for buf in (gp.flat, gp.flat_grad, fa.exp_avg, fa.exp_avg_sq))
        views["f_dc"] = tuple(t[:, :1, :] for t in feat)
        views["f_rest"] = tuple(t[:, 1:, :] for t in feat)
        self._homed = {}
        for gname, (vd, vg, va, vs) in views.items():
            p = g[gname]["params"][0]
            if tuple(p.shape) != tuple(vd.shape):
                raise ValueError("fdgs.optim.Adam: group %r has shape %s, expected %s" % (gname, tuple(p.shape), tuple(vd.shape)))
            vd.copy_(p.data)
            st = old_state.get(gname, {})
            if "exp_avg" in st and tuple(st["exp_avg"].shape) == tuple(va.shape):
                va.copy_(st["exp_avg"])
                vs.copy_(st["exp_avg_sq"])
            if old_grad[gname] is not None and tuple(old_grad[gname].shape) == tuple(vg.shape):
                vg.copy_(old_grad[gname])
            p.data = vd
            p.grad = vg if old_grad[gname] is not None else None
            self.state[p] = {"step": torch.tensor(float(step)), "exp_avg": va, "exp_avg_sq": vs}
            self._homed[gname] = (p, vd.data_ptr(), va.data_ptr(), vs.data_ptr())
        self._views, self._gp, self._fa = views, gp, fa
        self._features = feat[0]
        self._gacc = torch.zeros((P, 16), dtype=torch.float32, device=dev)
        if self._stages is not None and (self._stages.shape[1] != P or self._stages.device != dev):
            # the Gaussians changed: staged views of the old layout cannot be applied (the reference densifies between backward() and step(), train.py:239-249: every parameter is replaced, none has a gradient, and torch.optim.Adam skips the step too)
            self._stages, self._n_staged = None, 0
        self._fresh = all(v is None for v in old_grad.values())
        self._dense_sh = old_grad.get("f_dc") is not None or old_grad.get("f_rest") is not None
        self.homings += 1

    # ------------------------------------------------------------------ what render() asks for
    def features(self) -> torch.Tensor:
        """The contiguous ``[P, M, 3]`` SH coefficient array the model's ``_features_dc`` / ``_features_rest`` are views of."""
        return self._features

    def model_tensors(self, gaussian_dim: int, rot_4d: bool):
        """(xyz, features, opacity, t, scaling, scaling_t, rotation, rotation_r): the homed raw parameter tensors for the kernels."""
        e = torch.Tensor([])
        v = self._views
        is_4d = gaussian_dim == 4
        return (v["xyz"][0], self._features, v["opacity"][0], v["t"][0] if is_4d and "t" in v else e, v["scaling"][0],
                v["scaling_t"][0] if is_4d and "scaling_t" in v else e, v["rotation"][0],
                v["rotation_r"][0] if is_4d and rot_4d and "rotation_r" in v else e)

    @torch.no_grad()
    def backward_begin(self, rs):
        """Called by render()'s backward: where this view's parameter gradients go.  Returns (sink, accumulate, grad_accum, sh_stage)."""
        v = self._views
        g = self._named()
        grads = {n: g[n]["params"][0].grad for n in v}
        self._note_cleared(grads)
        foreign = [n for n, pg in grads.items() if pg is not None and pg.data_ptr() != v[n][1].data_ptr()]
        if foreign:
            # gradients autograd produced on its own since zero_grad() (another loss term, a render() branch outside the fast path):
            # the bucket takes them over, and from here on autograd adds into the bucket in place
            for n in v:
                pg = grads[n]
                if pg is None:
                    if self._fresh:
                        v[n][1].zero_()
                elif n in foreign:
                    if self._fresh:
                        v[n][1].copy_(pg)
                    else:
                        v[n][1].add_(pg)
                g[n]["params"][0].grad = v[n][1]
            if "f_dc" in foreign or "f_rest" in foreign:
                self._dense_sh = True
            self._fresh = False
        accumulate = not self._fresh
        sink = {"dL_dmeans3D": v["xyz"][1], "dL_dopacity": v["opacity"][1], "dL_dscales": v["scaling"][1], "dL_drotations": v["rotation"][1]}
        for key, gname in (("dL_dts", "t"), ("dL_dscales_t", "scaling_t"), ("dL_drotations_r", "rotation_r")):
            if gname in v:
                sink[key] = v[gname][1]
        stage = None
        if self.defer_sh:
            ss = _StageSettings(rs)
            if self._n_staged and self._stage_rs is not None and self._stage_rs.key() != ss.key():
                self._flush_stages()   # the active SH degrees changed inside one step: the earlier views go into the dense gradient
            self._stage_rs = ss
            P = self._gp.P
            if self._stages is None or self._stages.shape[1] != P:
                self._stages = torch.empty((4, P, 8), dtype=torch.float32, device=self._gacc.device)
                self._n_staged = 0
            if self._n_staged == self._stages.shape[0]:
                bigger = torch.empty((2 * self._n_staged, P, 8), dtype=torch.float32, device=self._gacc.device)
                bigger[:self._n_staged].copy_(self._stages)
                self._stages = bigger
            stage = self._stages[self._n_staged]
            self._n_staged += 1
        else:
            fv = self._gp.params["_features"].grad
            if self._fresh:
                fv.zero_()   # (the dense SH gradient is read-modify-written by the kernel: it has to start from zero)
            sink["dL_dsh"] = fv
            for gname in ("f_dc", "f_rest"):
                g[gname]["params"][0].grad = v[gname][1]
        for gname in v:
            if gname not in ("f_dc", "f_rest"):
                g[gname]["params"][0].grad = v[gname][1]
        self._fresh = False
        return sink, accumulate, self._gacc, stage

    def _note_cleared(self, grads):
        """Gradients cleared WITHOUT ``zero_grad()`` (``p.grad = None`` per parameter, a model-level helper): after a backward of
        this step at least the geometry tensors' ``.grad`` are the bucket views, so "every homed parameter has ``grad is None``" can
        only mean that the loop threw the last step's gradients away -- the bucket's contents and the staged views are stale, the
        next backward has to OVERWRITE (as after ``zero_grad(set_to_none=True)``), not add."""
        if not self._fresh and all(pg is None for pg in grads.values()):
            self._fresh = True
            self._dense_sh = False
            self._n_staged = 0

    def note_forward(self, lazy: bool):
        if lazy:
            self._lazy_views += 1

    @torch.no_grad()
    def _flush_stages(self):
        """The staged views into the dense SH gradient (added), e.g. because something else wrote a dense SH gradient too."""
        if self._n_staged:
            ss = self._stage_rs
            from .gaussian_renderer.diff_gaussian_rasterization import analytic_sh_gradients
            fv = self._gp.params["_features"].grad
            g = self._named()
            gdc, grest = g["f_dc"]["params"][0].grad, g["f_rest"]["params"][0].grad
            have = gdc is not None or grest is not None
            if have:
                # only ONE of f_dc / f_rest carries a dense gradient: the other half of the [P, M, 3] bucket view still holds an
                # earlier step's values -- the flush below ADDS into the whole array, so that half has to start from zero
                # (torch.optim.Adam would see "no gradient yet" there, then the staged views' sum)
                if gdc is None:
                    self._views["f_dc"][1].zero_()
                if grest is None:
                    self._views["f_rest"][1].zero_()
            _capi.sh_flush(self._stages[:self._n_staged], fv, ss.sh_degree, ss.sh_degree_t, ss.gaussian_dim, ss.force_sh_3d,
                           analytic_sh_gradients(), accumulate=have)
            for gname in ("f_dc", "f_rest"):
                g[gname]["params"][0].grad = self._views[gname][1]
            self._n_staged = 0

    # ------------------------------------------------------------------ torch.optim.Optimizer interface
    def zero_grad(self, set_to_none: bool = True):
        """``set_to_none=True`` (what the reference calls, train.py:249): ``p.grad = None``; the bucket is kept and the next
        backward overwrites it -- no memset.  ``False``: the bucket is zeroed and stays behind ``p.grad``."""
        if self._gp is None:
            return super().zero_grad(set_to_none=set_to_none)
        homed = {id(t[0]) for t in self._homed.values()}
        for grp in self.param_groups:
            for p in grp["params"]:
                if id(p) in homed:
                    p.grad = None
                elif p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_()
                        p.grad.zero_()
        if not set_to_none:
            self._gp.flat_grad.zero_()
            for name, (p, *_r) in self._homed.items():
                p.grad = self._views[name][1]
            self._fresh = False
        else:
            self._fresh = True
        self._n_staged = 0
        self._dense_sh = False

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .gaussian_renderer.diff_gaussian_rasterization import analytic_sh_gradients
        bucketed = self.ensure_homed()
        homed_ids = {id(t[0]) for t in self._homed.values()} if bucketed else set()
        if bucketed:
            g = self._named()
            if self._lazy_views:
                dev = self._gacc.device
                _pend, failed, _r = _capi.forward_lazy_status(dev, wait=True)
                self._lazy_views = 0
                if failed:
                    import warnings
                    warnings.warn("fdgs.optim.Adam: %d view(s) of this step outgrew their run-ahead buffers (lazy_forward): their images were "
                                  "invalid; the step is skipped" % failed)
                    self.steps_skipped += 1
                    self._n_staged = 0
                    return False
            v = self._views
            self._note_cleared({n: g[n]["params"][0].grad for n in v})   # cleared by hand since the last backward: nothing to step
            # autograd-made gradients the fast path has not seen (no render() backward since they appeared)
            for gname in v:
                p = g[gname]["params"][0]
                if p.grad is not None and p.grad.data_ptr() != v[gname][1].data_ptr():
                    if self._fresh:
                        for other in v:
                            if g[other]["params"][0].grad is None:
                                v[other][1].zero_()
                    v[gname][1].copy_(p.grad)
                    p.grad = v[gname][1]
                    self._fresh = False
                    if gname in ("f_dc", "f_rest"):
                        self._dense_sh = True
            if not self._fresh:
                fa = self._fa
                fa.betas, fa.eps = tuple(g["xyz"]["betas"]), float(g["xyz"]["eps"])
                for gname, seg_name in _GEOMETRY.items():
                    if gname in g:
                        fa.set_lr(seg_name, float(g[gname]["lr"]))
                    else:
                        fa.set_lr(seg_name, 0.0)
                fa.set_lr("_features", float(g["f_rest"]["lr"]), float(g["f_dc"]["lr"]))
                dense_sh = self._dense_sh or any(g[n]["params"][0].grad is not None for n in ("f_dc", "f_rest"))
                fa.step_count += 1
                feat_begin = self._gp.offsets["_features"][0]
                if self._n_staged and not dense_sh and fa.step_sh_staged(self._stages[:self._n_staged], self._stage_rs, analytic_sh_gradients()):
                    fa.step_range(0, feat_begin)
                else:
                    if self._n_staged:
                        self._flush_stages()
                        dense_sh = True
                    if dense_sh or not self.defer_sh:
                        fa.step_range(0, self._gp.flat.numel())
                    else:
                        fa.step_range(0, feat_begin)   # no view touched the coefficients this step
                self._n_staged = 0
                for name, (p, *_r) in self._homed.items():
                    self.state[p]["step"] = torch.tensor(float(fa.step_count))
        # everything that is not one of the reference's nine tensors: the same kernel, tensor by tensor
        for grp in self.param_groups:
            for p in grp["params"]:
                if id(p) in homed_ids or p.grad is None:
                    continue
                self._step_single(p, grp)
        return loss

    @torch.no_grad()
    def _step_single(self, p, grp):
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("fdgs.optim.Adam: parameters must be contiguous float32 GPU tensors")
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
        st["step"] = st["step"] + 1
        n = p.numel()
        seg = (_capi.FdgsAdamSegment * 1)(_capi.FdgsAdamSegment(0, n, float(grp["lr"]), float(grp["lr"]), 0, 0))
        g = p.grad.contiguous()
        with torch.cuda.device(p.device):
            rc = _capi.lib.fdgs_adam_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n, seg, 1,
                                          float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]), int(float(st["step"])),
                                          _capi.current_stream_handle(p.device))
        _capi._check(rc, "fdgs_adam_step")
