"""Host-side (PyTorch) pieces of one frame-parallel train step around the native rasterizer.

These mirror what the reference trainer does on either side of the hot path
(train.py:104-166, 247-249) so that ``bench.py`` times the metric BASELINE.md defines
(render forward -> L1 + D-SSIM -> backward -> optimizer step) and so that the multi-GPU
exchange of SURVEY.md section 8e has one implementation that the CPU ``gloo`` tests
and the RCCL bench share:

* ``GaussianParams``  -- the reference's parameter set and activations
  (scene/gaussian_model.py:179-219: exp scales, sigmoid opacity, normalised
  quaternions), duck-typed for ``render()``.  All parameters are views into ONE flat
  fp32 buffer and all gradients into ONE flat gradient buffer, so the data-parallel
  exchange is a single collective and the optimizer step a single kernel.
* ``FlatAdam``        -- Adam over that bucket (fused HIP kernel on the GPU).
* ``allreduce_gradients`` -- one ``all_reduce(SUM)`` over the flat gradient bucket
  (161 floats per Gaussian at M = 48 -> 193 MB at 300 k Gaussians), then the 1/world
  scale of ``loss / batch_size`` (train.py:162).  Frames / timesteps are independent
  given the parameters, so there is no collective inside the rasterizer itself.
* ``l1_loss`` / ``ssim`` / ``photometric_loss`` -- utils/loss_utils.py:17-64 semantics
  (11x11 Gaussian window, sigma 1.5, C1 = 0.01^2, C2 = 0.03^2; lambda_dssim = 0.2).
"""
import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


def _inv_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianParams:
    """Flat-bucket parameter set with the reference model's getters (post-activation).

    Differences from scene/gaussian_model.py that do not change the mathematics: the SH coefficients are ONE
    ``[P, M, 3]`` parameter (the reference keeps ``_features_dc`` / ``_features_rest`` and concatenates them in
    ``get_features`` every call, :211-215); their two learning rates (feature_lr and feature_lr / 20) are
    applied per coefficient by the optimizer's segment table instead.
    """

    # bucket order: the small geometry tensors first (17 floats per Gaussian), the SH coefficients last (3 M = 144): the two
    # parts are final at different moments of a backward pass and are all-reduced separately (allreduce_and_step)
    NAMES = ("_xyz", "_opacity", "_scaling", "_rotation", "_t", "_scaling_t", "_rotation_r", "_features")

    def __init__(self, scene: Dict[str, object], device):
        """``scene``: post-activation tensors from fdgs.synth.make_scene; raw parameters are their inverses."""
        P, M = int(scene["means3D"].shape[0]), int(scene["M"])
        shapes = {"_xyz": (P, 3), "_features": (P, M, 3), "_opacity": (P, 1), "_scaling": (P, 3), "_rotation": (P, 4),
                  "_t": (P, 1), "_scaling_t": (P, 1), "_rotation_r": (P, 4)}
        init = {"_xyz": scene["means3D"], "_features": scene["shs"],
                "_opacity": _inv_sigmoid(scene["opacities"].clamp(1e-6, 1 - 1e-6)), "_scaling": torch.log(scene["scales"]),
                "_rotation": scene["rotations"], "_t": scene["ts"], "_scaling_t": torch.log(scene["scales_t"]),
                "_rotation_r": scene["rotations_r"]}
        self.M = M
        total = P * self.floats_per_gaussian()
        self._bind(torch.empty(total, dtype=torch.float32, device=device), torch.zeros(total, dtype=torch.float32, device=device), P)
        with torch.no_grad():
            for name in self.NAMES:
                self.params[name].copy_(init[name].to(device).reshape(shapes[name]))
        # the coefficient storage (M) is laid out for these maxima (scene["max_sh_degree*"]: synth.make_scene(alloc=...), the
        # reference's state from iteration 0: scene/gaussian_model.py:65,92); a model built from a scene starts at the scene's ACTIVE
        # degrees -- all of them for a trained model; training from scratch starts at (0, 0) and ramps with oneupSHdegree()
        self.max_sh_degree = int(scene.get("max_sh_degree", scene["sh_degree"]))
        self.max_sh_degree_t = int(scene.get("max_sh_degree_t", scene["sh_degree_t"]))
        self.active_sh_degree = int(scene["sh_degree"])
        self.active_sh_degree_t = int(scene["sh_degree_t"])
        self.time_duration = [0.0, float(scene["time_duration"])]
        self.rot_4d, self.gaussian_dim = bool(scene["rot_4d"]), int(scene["gaussian_dim"])
        self.force_sh_3d = bool(scene["force_sh_3d"])
        self.prefilter_var = -1.0
        self.env_map = None
        self.get_max_sh_channels = M

    def oneupSHdegree(self):
        """scene/gaussian_model.py:253-257: the spatial degree climbs to its maximum first, then the time degree."""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
        elif self.max_sh_degree_t and self.active_sh_degree_t < self.max_sh_degree_t:
            self.active_sh_degree_t += 1

    def row_floats(self) -> List[int]:
        """floats per Gaussian of every segment of the flat bucket, in NAMES order"""
        return [3, 1, 3, 4, 1, 1, 4, self.M * 3]

    def floats_per_gaussian(self) -> int:
        return sum(self.row_floats())

    def _bind(self, flat: torch.Tensor, flat_grad: torch.Tensor, P: int):
        """(Re)creates the parameter views over ``flat`` / ``flat_grad`` for P Gaussians (segments back to back)."""
        shapes = {"_xyz": (P, 3), "_features": (P, self.M, 3), "_opacity": (P, 1), "_scaling": (P, 3), "_rotation": (P, 4),
                  "_t": (P, 1), "_scaling_t": (P, 1), "_rotation_r": (P, 4)}
        assert self.NAMES[-1] == "_features"
        assert flat.numel() == flat_grad.numel() == P * self.floats_per_gaussian()
        self.flat, self.flat_grad = flat, flat_grad
        self.params = {}
        self.offsets = {}
        off = 0
        for name, rf in zip(self.NAMES, self.row_floats()):
            n = P * rf
            p = self.flat[off:off + n].view(shapes[name]).requires_grad_(True)
            p.grad = self.flat_grad[off:off + n].view(shapes[name])
            self.params[name] = p
            setattr(self, name, p)  # the reference's attribute names (scene/gaussian_model.py:70-80)
            self.offsets[name] = (off, off + n)
            off += n
        self.P = P

    # ---- scene/gaussian_model.py:179-219 ----
    get_xyz = property(lambda s: s.params["_xyz"])
    get_t = property(lambda s: s.params["_t"])
    get_scaling = property(lambda s: torch.exp(s.params["_scaling"]))
    get_scaling_t = property(lambda s: torch.exp(s.params["_scaling_t"]))
    get_rotation = property(lambda s: F.normalize(s.params["_rotation"]))
    get_rotation_r = property(lambda s: F.normalize(s.params["_rotation_r"]))
    get_opacity = property(lambda s: torch.sigmoid(s.params["_opacity"]))
    get_features = property(lambda s: s.params["_features"])

    def lr_segments(self) -> List[dict]:
        """Learning rates of arguments/__init__.py:84-92 (spatial_lr_scale = 1) as a segment table."""
        lr = {"_xyz": 1.6e-4, "_opacity": 5e-2, "_scaling": 5e-3, "_rotation": 1e-3, "_t": 1.6e-4, "_scaling_t": 5e-3,
              "_rotation_r": 1e-3}
        segs = []
        for n in self.NAMES:
            b, e = self.offsets[n]
            if n == "_features":  # DC coefficient: feature_lr; the rest: feature_lr / 20
                segs.append(dict(begin=b, end=e, lr=2.5e-3 / 20.0, lr_head=2.5e-3, period=self.M * 3, head=3))
            else:
                segs.append(dict(begin=b, end=e, lr=lr[n], lr_head=lr[n], period=0, head=0))
        return segs

    def zero_grad(self):
        self.flat_grad.zero_()

    def grad_sink(self) -> Dict[str, torch.Tensor]:
        """Gradient destinations for fdgs.fused.render_raw: the rasterizer backward writes each parameter's
        gradient straight into its slice of the flat bucket (no autograd accumulation pass, no zero_grad)."""
        g = {n: self.params[n].grad for n in self.NAMES}
        return {"dL_dmeans3D": g["_xyz"], "dL_dsh": g["_features"], "dL_dopacity": g["_opacity"],
                "dL_dscales": g["_scaling"], "dL_drotations": g["_rotation"], "dL_dts": g["_t"],
                "dL_dscales_t": g["_scaling_t"], "dL_drotations_r": g["_rotation_r"]}


class ReferenceStyleModel:
    """The model a user of the REFERENCE holds when they swap in this package's ``render`` and nothing else: separate
    ``nn.Parameter``s incl. ``_features_dc`` [P,1,3] / ``_features_rest`` [P,M-1,3] concatenated by ``get_features`` on every call,
    PyTorch activations in the getters (scene/gaussian_model.py:179-219), ``torch.optim.Adam(lr=0, eps=1e-15)`` over one param
    group per tensor (:331-357).  bench.py's drop-in leg times ``render()`` + autograd + this optimizer on it; nothing else uses it."""

    def __init__(self, scene: Dict[str, object], device, optimizer: str = "torch"):
        """``optimizer``: "torch" = ``torch.optim.Adam`` as the reference builds it (scene/gaussian_model.py:353); "fdgs" =
        ``fdgs.optim.Adam`` over the same param groups (the one-line swap INTEGRATION.md describes)."""
        M = int(scene["M"])
        mk = lambda t: torch.nn.Parameter(t.to(device).float().contiguous().requires_grad_(True))  # noqa: E731
        self._xyz = mk(scene["means3D"])
        self._features_dc = mk(scene["shs"][:, :1, :])
        self._features_rest = mk(scene["shs"][:, 1:, :])
        self._opacity = mk(_inv_sigmoid(scene["opacities"].clamp(1e-6, 1 - 1e-6)).reshape(-1, 1))
        self._scaling = mk(torch.log(scene["scales"]))
        self._rotation = mk(scene["rotations"])
        self._t = mk(scene["ts"].reshape(-1, 1))
        self._scaling_t = mk(torch.log(scene["scales_t"]).reshape(-1, 1))
        self._rotation_r = mk(scene["rotations_r"])
        self.active_sh_degree, self.active_sh_degree_t = int(scene["sh_degree"]), int(scene["sh_degree_t"])
        self.time_duration = [0.0, float(scene["time_duration"])]
        self.rot_4d, self.gaussian_dim = bool(scene["rot_4d"]), int(scene["gaussian_dim"])
        self.force_sh_3d = bool(scene["force_sh_3d"])
        self.prefilter_var = -1.0
        self.env_map = None
        self.get_max_sh_channels = M
        groups = [dict(params=[self._xyz], lr=1.6e-4, name="xyz"), dict(params=[self._features_dc], lr=2.5e-3, name="f_dc"),
                  dict(params=[self._features_rest], lr=2.5e-3 / 20.0, name="f_rest"), dict(params=[self._opacity], lr=5e-2, name="opacity"),
                  dict(params=[self._scaling], lr=5e-3, name="scaling"), dict(params=[self._rotation], lr=1e-3, name="rotation")]
        if self.gaussian_dim == 4:
            groups += [dict(params=[self._t], lr=1.6e-4, name="t"), dict(params=[self._scaling_t], lr=5e-3, name="scaling_t")]
            if self.rot_4d:
                groups.append(dict(params=[self._rotation_r], lr=1e-3, name="rotation_r"))
        if optimizer == "fdgs":
            from .optim import Adam as FdgsAdam
            self.optimizer = FdgsAdam(groups, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    # the attribute every parameter group's tensor lives under (scene/gaussian_model.py:411-425, 456-470)
    _ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
             "rotation": "_rotation", "t": "_t", "scaling_t": "_scaling_t", "rotation_r": "_rotation_r"}

    def _adopt(self, tensors: Dict[str, torch.Tensor]):
        for name, t in tensors.items():
            setattr(self, self._ATTR[name], t)

    # ---- how the reference's densification edits the optimizer (scene/gaussian_model.py:376-452): restated so that the tests can
    # put ANY optimizer through the same manipulations: state dicts moved from the old Parameter to the new one, moments gathered /
    # extended with zeros, ``group["params"][0]`` replaced by a fresh nn.Parameter ----
    def replace_tensor_to_optimizer(self, tensor: torch.Tensor, name: str):
        """:376-389 (reset_opacity): new values, zeroed moments."""
        for group in self.optimizer.param_groups:
            if group["name"] != name:
                continue
            old = group["params"][0]
            st = self.optimizer.state.get(old, None)
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
            del self.optimizer.state[old]
            group["params"][0] = torch.nn.Parameter(tensor.requires_grad_(True))
            self.optimizer.state[group["params"][0]] = st
            self._adopt({name: group["params"][0]})

    def prune_points(self, mask: torch.Tensor):
        """:391-429 (_prune_optimizer + prune_points): rows where ``mask`` is True go."""
        keep = ~mask
        out = {}
        for group in self.optimizer.param_groups:
            old = group["params"][0]
            st = self.optimizer.state.get(old, None)
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep], st["exp_avg_sq"][keep]
                del self.optimizer.state[old]
            group["params"][0] = torch.nn.Parameter(old[keep].requires_grad_(True))
            if st is not None:
                self.optimizer.state[group["params"][0]] = st
            out[group["name"]] = group["params"][0]
        self._adopt(out)

    def densification_postfix(self, new: Dict[str, torch.Tensor]):
        """:431-470 (cat_tensors_to_optimizer + densification_postfix): rows appended, their moments zero."""
        out = {}
        for group in self.optimizer.param_groups:
            ext = new[group["name"]]
            old = group["params"][0]
            st = self.optimizer.state.get(old, None)
            if st is not None:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del self.optimizer.state[old]
            group["params"][0] = torch.nn.Parameter(torch.cat((old, ext), dim=0).requires_grad_(True))
            if st is not None:
                self.optimizer.state[group["params"][0]] = st
            out[group["name"]] = group["params"][0]
        self._adopt(out)

    get_xyz = property(lambda s: s._xyz)
    get_t = property(lambda s: s._t)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_scaling_t = property(lambda s: torch.exp(s._scaling_t))
    get_scaling_xyzt = property(lambda s: torch.exp(torch.cat([s._scaling, s._scaling_t], dim=1)))
    get_rotation = property(lambda s: F.normalize(s._rotation))
    get_rotation_r = property(lambda s: F.normalize(s._rotation_r))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    # ---- the Python-side covariance of ``pipe.compute_cov3D_python`` (gaussian_renderer/__init__.py:73-81) ----
    # Restated from scene/gaussian_model.py:28-52, 230-251 and utils/general_utils.py:64-151 (device-agnostic; the reference
    # hard-codes "cuda"); pinned against the reference's own functions by tests/golden/pycov_*.npz (make_golden_pycov.py).
    def get_covariance(self, scaling_modifier=1):
        """3D: Sigma = (S R)^T (S R), upper triangle [xx, xy, xz, yy, yz, zz] (gaussian_model.py:28-32, 244-245)."""
        L = scaling_rotation_3d(scaling_modifier * self.get_scaling, self._rotation)
        return upper_triangle(L.transpose(1, 2) @ L)

    def get_current_covariance_and_mean_offset(self, scaling_modifier=1, timestamp=0.0):
        """rot_4d: the 3D covariance conditioned on t = timestamp and the shift of the mean (gaussian_model.py:34-47, 247-251)."""
        L = scaling_rotation_4d(scaling_modifier * self.get_scaling_xyzt, self._rotation, self._rotation_r)
        sigma = L @ L.transpose(1, 2)
        c12, ct = sigma[:, 0:3, 3:4], sigma[:, 3:4, 3:4]
        cond = sigma[:, :3, :3] - c12 @ c12.transpose(1, 2) / ct
        dt = timestamp - self.get_t
        return upper_triangle(cond), c12.squeeze(-1) / ct.squeeze(-1) * dt

    def get_cov_t(self, scaling_modifier=1):
        """Temporal variance (gaussian_model.py:230-236): Sigma_tt with rot_4d, else the activated scaling_t itself."""
        if self.rot_4d:
            L = scaling_rotation_4d(scaling_modifier * self.get_scaling_xyzt, self._rotation, self._rotation_r)
            return (L @ L.transpose(1, 2))[:, 3, 3].unsqueeze(1)
        return self.get_scaling_t * scaling_modifier

    def get_marginal_t(self, timestamp, scaling_modifier=1):
        """exp(-(t - timestamp)^2 / (2 (sigma_t [+ prefilter_var]))) (gaussian_model.py:238-242)."""
        sigma = self.get_cov_t(scaling_modifier)
        if self.prefilter_var > 0.0:
            sigma = sigma + self.prefilter_var
        return torch.exp(-0.5 * (self.get_t - timestamp) ** 2 / sigma)


def upper_triangle(sym: torch.Tensor) -> torch.Tensor:
    """[n,3,3] symmetric -> [n,6] = xx, xy, xz, yy, yz, zz (utils/general_utils.py:64-77)."""
    return torch.stack([sym[:, 0, 0], sym[:, 0, 1], sym[:, 0, 2], sym[:, 1, 1], sym[:, 1, 2], sym[:, 2, 2]], dim=1)


def scaling_rotation_3d(s: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """diag(s) @ R(q / |q|), q = (w, x, y, z) (utils/general_utils.py:79-111)."""
    q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
    return s.unsqueeze(2) * R


def scaling_rotation_4d(s: torch.Tensor, ql: torch.Tensor, qr: torch.Tensor) -> torch.Tensor:
    """R4(ql, qr) @ diag(s): the product of the left- and right-isoclinic rotations of the two unit quaternions, rows and
    columns reversed (utils/general_utils.py:113-151)."""
    ql = ql / torch.norm(ql, dim=-1, keepdim=True)
    qr = qr / torch.norm(qr, dim=-1, keepdim=True)
    a, b, c, d = ql.unbind(-1)
    p, q, r, t = qr.unbind(-1)
    Ml = torch.stack([a, -b, -c, -d, b, a, -d, c, c, d, a, -b, d, -c, b, a], dim=1).view(-1, 4, 4)
    Mr = torch.stack([p, q, r, t, -q, p, -t, r, -r, t, p, -q, -t, -r, q, p], dim=1).view(-1, 4, 4)
    A = (Ml @ Mr).flip(1, 2)
    return A * s.unsqueeze(1)


class FlatAdam:
    """torch.optim.Adam(lr=0, eps=1e-15) semantics (scene/gaussian_model.py:353) over the flat bucket.

    On the GPU the step is one fused HIP kernel (csrc/adam.hip, fdgs_adam_step); on the CPU (gloo tests) the same
    arithmetic in a few PyTorch ops with a per-element learning-rate vector -- also the reference the GPU test
    compares the kernel against."""

    def __init__(self, model: GaussianParams, betas=(0.9, 0.999), eps=1e-15):
        self.model, self.betas, self.eps = model, betas, eps
        self.exp_avg = torch.zeros_like(model.flat)
        self.exp_avg_sq = torch.zeros_like(model.flat)
        self.step_count = 0
        self.segments = model.lr_segments()
        self._native = None
        self._lr_vec = None

    def lr_vector(self) -> torch.Tensor:
        lr = torch.zeros_like(self.model.flat)
        for s in self.segments:
            lr[s["begin"]:s["end"]] = s["lr"]
            if s["period"] > 0:
                seg = lr[s["begin"]:s["end"]].view(-1, s["period"])
                seg[:, :s["head"]] = s["lr_head"]
        return lr

    def rebind(self, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor):
        """After the model was re-laid out (densification): new moment buffers, segment table for the new offsets;
        the learning rates set so far are kept."""
        old = self.segments
        self.exp_avg, self.exp_avg_sq = exp_avg, exp_avg_sq
        self.segments = self.model.lr_segments()
        for s_new, s_old in zip(self.segments, old):
            s_new["lr"], s_new["lr_head"] = s_old["lr"], s_old["lr_head"]
        self._native = None
        self._lr_vec = None

    def named_segments(self):
        """The segment table with the parameter tensor each segment belongs to (``name``)."""
        by_range = {tuple(v): k for k, v in self.model.offsets.items()}
        return [dict(s, name=by_range[(s["begin"], s["end"])]) for s in self.segments]

    def set_lr(self, name: str, lr: float, lr_head: float = None):
        """Learning rate of one parameter tensor (the reference's ``param_group['lr'] = lr``, gaussian_model.py:359-365)."""
        b, e = self.model.offsets[name]
        for i, s in enumerate(self.segments):
            if s["begin"] == b and s["end"] == e:
                s["lr"] = float(lr)
                s["lr_head"] = float(lr if lr_head is None else lr_head)
                if self._native is not None:
                    self._native[i].lr, self._native[i].lr_head = s["lr"], s["lr_head"]
                self._lr_vec = None
                return
        raise KeyError(name)

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        self.step_range(0, self.model.flat.numel())

    @torch.no_grad()
    def step_range(self, begin: int, end: int):
        """Adam update of flat[begin:end] for the CURRENT step_count (``step()`` = increment + whole range).  Used to
        update one chunk of the bucket while the all-reduce of the next chunk is still in flight."""
        m = self.model
        n = end - begin
        if n <= 0:
            return
        if begin % 4 != 0:
            raise ValueError("FlatAdam.step_range: begin must be a multiple of 4 elements (the kernel updates float4s; "
                             "allreduce_and_step cuts its chunks accordingly), got %d" % begin)
        if m.flat.is_cuda:
            from . import _capi
            # segment table relative to `begin` (a segment that starts before the chunk keeps its phase: negative begin)
            arr = (_capi.FdgsAdamSegment * len(self.segments))()
            for i, s in enumerate(self.segments):
                arr[i] = _capi.FdgsAdamSegment(s["begin"] - begin, s["end"] - begin, s["lr"], s["lr_head"], s["period"], s["head"])
            self._native = arr
            with torch.cuda.device(m.flat.device):
                rc = _capi.lib.fdgs_adam_step(m.flat.data_ptr() + 4 * begin, m.flat_grad.data_ptr() + 4 * begin,
                                              self.exp_avg.data_ptr() + 4 * begin, self.exp_avg_sq.data_ptr() + 4 * begin, n,
                                              arr, len(self.segments), self.betas[0], self.betas[1], self.eps,
                                              self.step_count, _capi.current_stream_handle(m.flat.device))
            _capi._check(rc, "fdgs_adam_step")
            return
        if self._lr_vec is None:
            self._lr_vec = self.lr_vector()
        b1, b2 = self.betas
        sl = slice(begin, end)
        g = m.flat_grad[sl]
        self.exp_avg[sl].mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq[sl].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
        denom = self.exp_avg_sq[sl].sqrt().div_(math.sqrt(bc2)).add_(self.eps)
        m.flat[sl].sub_(self._lr_vec[sl] / bc1 * (self.exp_avg[sl] / denom))


    @torch.no_grad()
    def step_sh_staged(self, stages: torch.Tensor, rs, analytic_sh_grad: bool = False, write_grad: bool = False) -> bool:
        """The update of the ``_features`` segment for the CURRENT step_count, with its gradient taken from the staged views
        of the deferred SH backward (``stages`` [B, P, 8], csrc/sh_bwd.hip sh_adam_kernel): fdgs_sh_flush + step_range on
        that segment in one pass, without dL_dsh travelling through memory.  ``rs``: the views' raster settings (SH degrees,
        gaussian_dim ...).  ``write_grad``: also leave the summed gradient in ``_features.grad``.  Returns False -- nothing
        done -- when the layout is not supported (rows not whole float4s / segment not 16-byte aligned)."""
        from . import _capi
        m = self.model
        b, e = m.offsets["_features"]
        seg = next(s for s in self.segments if s["begin"] == b and s["end"] == e)
        shape = m.params["_features"].shape
        return _capi.adam_step_sh(m.flat[b:e].view(shape), self.exp_avg[b:e].view(shape), self.exp_avg_sq[b:e].view(shape), stages,
                                  rs.sh_degree, rs.sh_degree_t, rs.gaussian_dim, rs.force_sh_3d, analytic_sh_grad,
                                  seg["lr"], seg["lr_head"], self.betas[0], self.betas[1], self.eps, self.step_count,
                                  dL_dsh=m.params["_features"].grad if write_grad else None)


def make_optimizer(model: GaussianParams) -> FlatAdam:
    return FlatAdam(model)


def morton_permutation(xyz: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation (on xyz's device) that puts points into Morton (Z-order) order of their positions quantised to ``bits`` bits
    per axis; ties keep their order (stable), so the result is a pure function of the positions."""
    p = xyz.detach().double()
    if p.shape[0] == 0:   # everything pruned: nothing to order (min / max over an empty dimension raise)
        return torch.arange(0, dtype=torch.long, device=p.device)
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).long().clamp_(0, (1 << bits) - 1)
    code = torch.zeros(p.shape[0], dtype=torch.long, device=p.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, stable=True)


@torch.no_grad()
def reorder_gaussians(model: GaussianParams, optimizer: "FlatAdam", perm: torch.Tensor) -> None:
    """Stores the model's Gaussians in the order ``perm`` (row j of every parameter tensor and of both Adam moments becomes the old
    row perm[j]).  Renders the same images and takes the same optimizer steps: only the memory order changes."""
    bufs = [model.flat] + ([optimizer.exp_avg, optimizer.exp_avg_sq] if optimizer is not None else [])
    for name, rf in zip(model.NAMES, model.row_floats()):
        b, e = model.offsets[name]
        for buf in bufs:
            seg = buf[b:e].view(model.P, rf)
            seg.copy_(seg[perm])


def spatial_sort(model: GaussianParams, optimizer: "FlatAdam" = None) -> torch.Tensor:
    """Keeps the model in Morton order of the Gaussians' positions (call it when the set of Gaussians changes: after loading,
    after every densification; harness.train does).  Consecutive Gaussians then project to neighbouring pixels from any
    camera, so a workgroup of the tile-binning passes touches a few dozen tile lists instead of all of them: its scattered
    8-byte stores land in runs, its counter flushes in a few cache lines (measured at C3: scatter 53 -> 34 us, count 14 -> 10 us,
    forward 0.281 -> 0.264 ms, step +2 %).  The reference appends new Gaussians at the end (gaussian_model.py:441-470);
    the order of the model is not part of its semantics.  Returns the permutation applied."""
    perm = morton_permutation(model.params["_xyz"].data)
    reorder_gaussians(model, optimizer, perm)
    return perm


def allreduce_gradients(model: GaussianParams, world_size: int, average: bool = True) -> None:
    """One collective over the flat gradient bucket; grads become the mean over ranks (loss / batch_size).
    ``average=False``: plain SUM, for callers that already scaled their loss by 1 / world_size (saves one pass
    over the bucket)."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(model.flat_grad, op=dist.ReduceOp.SUM)
        if average:
            model.flat_grad.mul_(1.0 / world_size)


def _bounds(begin: int, end: int, pieces: int):
    """[begin, end) cut into at most ``pieces`` runs whose starts are multiples of 4 elements (the Adam kernel's float4 path)."""
    n = end - begin
    if n <= 0:
        return []   # e.g. P == 0 after everything was pruned
    pieces = max(1, min(int(pieces), n))
    step = -(-n // pieces)
    step += (-step) % 4
    return [(b, min(b + step, end)) for b in range(begin, end, step)]


def _sh_split(model: GaussianParams) -> int:
    """First element of the part of the bucket that is all-reduced early: the start of the SH gradients rounded up to a
    multiple of 4 elements (Adam's float4 path); the up to 3 SH values before it travel with the geometry part."""
    b = model.offsets["_features"][0]
    return b + (-b) % 4


def allreduce_sh_begin(model: GaussianParams, world_size: int, chunks: int = 3):
    """Starts the all-reduce of the SH-gradient part of the bucket (asynchronously, in ``chunks`` pieces) and returns the
    handle for ``allreduce_and_step(..., sh_handle=...)``.  To be called as soon as the last view's SH backward has been
    enqueued -- the geometry backward of that view has not run yet, and does not touch this part of the bucket.
    The caller's CURRENT stream must be one on which the SH gradients are complete (torch.distributed orders the
    collective behind the current stream's work)."""
    if world_size <= 1:
        return None
    import torch.distributed as dist
    bounds = _bounds(_sh_split(model), model.flat.numel(), chunks)
    return [(dist.all_reduce(model.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True), lo, hi) for lo, hi in bounds]


def gather_sh_stages_begin(stages: torch.Tensor, world_size: int):
    """Starts the exchange of the ranks' staged SH gradients ([B, P, 8] each, fdgs_backward_out.sh_stage) and returns
    (work, gathered): after ``work.wait()`` ``gathered`` is [world * B, P, 8], rank-major, bit-identical on every rank.

    With few views per step this replaces the all-reduce of the dense SH gradient: a view contributes 32 bytes per Gaussian to
    the exchange instead of the 12 M bytes per Gaussian of dL_dsh (world * B = 8 views at M = 48: 77 MB gathered per rank
    against 2 * 7/8 * 173 MB moved by a ring all-reduce), and every rank then feeds the same stages, in the same order, to the
    fused SH flush + Adam kernel (FlatAdam.step_sh_staged) -- the replicas stay bit-identical.
    RCCL: one all-gather.  Other backends (the gloo debug / test runs): the same result as an all-reduce of a buffer that is
    zero outside the rank's own slice."""
    import torch.distributed as dist
    B = stages.shape[0]
    shape = (world_size * B,) + tuple(stages.shape[1:])
    if dist.get_backend() == "nccl" and hasattr(dist, "all_gather_into_tensor"):
        gathered = torch.empty(shape, dtype=stages.dtype, device=stages.device)
        return dist.all_gather_into_tensor(gathered, stages.contiguous(), async_op=True), gathered
    gathered = torch.zeros(shape, dtype=stages.dtype, device=stages.device)
    r = dist.get_rank()
    gathered[r * B:(r + 1) * B].copy_(stages)
    return dist.all_reduce(gathered, op=dist.ReduceOp.SUM, async_op=True), gathered


def gather_view_stage_begin(stage: torch.Tensor, out: torch.Tensor):
    """The exchange of ONE view's staged SH gradient, started as soon as that view's SH backward has been enqueued: ``stage``
    [P, 8] of this rank -> ``out`` [world, P, 8] (rank-major, the same on every rank).  Returns the work handle.

    Per view instead of once per step (gather_sh_stages_begin): a view's 32 bytes per Gaussian are final when ITS SH backward
    has run, so the all-gather of view b travels over xGMI while views b+1.. are rendered; only the last view's exchange is
    left at the end of the step.  (The dense gradient cannot do that: it is the SUM over the step's views.)"""
    import torch.distributed as dist
    if dist.get_backend() == "nccl" and hasattr(dist, "all_gather_into_tensor"):
        return dist.all_gather_into_tensor(out, stage.contiguous(), async_op=True)
    out.zero_()
    out[dist.get_rank()].copy_(stage)
    return dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True)


def timed_wait(work, pairs=None):
    """``work.wait()`` (the current stream waits for the collective); with ``pairs`` a list: bracketed by two timing events on the
    current stream, appended as (before, after) -- their elapsed time is what the stream stood still for this collective, i.e. the
    part of the exchange that nothing overlapped."""
    if pairs is None:
        work.wait()
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    work.wait()
    b.record()
    pairs.append((a, b))


def allreduce_and_step(model: GaussianParams, optimizer: FlatAdam, world_size: int, chunks: int = 4, average: bool = False,
                       sh_handle=None, wait_pairs=None) -> None:
    """Gradient all-reduce + Adam with the two overlapped: the bucket is cut into pieces, all all-reduces are issued
    at once (they run back to back on the collective stream) and a piece is updated as soon as ITS all-reduce has
    finished, while the next one is still on the wire.  ``sh_handle``: the SH part is already in flight
    (allreduce_sh_begin), only the geometry part (17 floats per Gaussian) is issued here.
    Same result as allreduce_gradients + step() (sums are element-wise: the cut does not change them)."""
    if world_size <= 1:
        optimizer.step()
        return
    import torch.distributed as dist
    n = model.flat.numel()
    geo_end = _sh_split(model)
    if sh_handle is None:
        pieces = _bounds(0, n, chunks)
    else:
        pieces = _bounds(0, geo_end, 1)
    works = [(dist.all_reduce(model.flat_grad[b:e], op=dist.ReduceOp.SUM, async_op=True), b, e) for b, e in pieces]
    if sh_handle is not None:
        works = list(sh_handle) + works     # the SH pieces were issued first and finish first
    optimizer.step_count += 1
    for w, b, e in works:
        timed_wait(w, wait_pairs if model.flat.is_cuda else None)
        if average:
            model.flat_grad[b:e].mul_(1.0 / world_size)
        optimizer.step_range(b, e)


# ------------------------- utils/loss_utils.py:17-64 -------------------------
def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


_WINDOWS: Dict[Tuple[int, int, str], torch.Tensor] = {}


def _window(window_size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    key = (window_size, channel, str(like.device))
    w = _WINDOWS.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
        w = w2.expand(channel, 1, window_size, window_size).contiguous().to(like.device).type_as(like)
        _WINDOWS[key] = w
    return w


def ssim(img1, img2, window_size=11):
    channel = img1.size(-3)
    w = _window(window_size, channel, img1)
    pad = window_size // 2
    x1, x2 = img1.unsqueeze(0) if img1.dim() == 3 else img1, img2.unsqueeze(0) if img2.dim() == 3 else img2
    mu1 = F.conv2d(x1, w, padding=pad, groups=channel)
    mu2 = F.conv2d(x2, w, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    sigma1_sq = F.conv2d(x1 * x1, w, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(x2 * x2, w, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(x1 * x2, w, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


def photometric_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda) L1 + lambda (1 - SSIM), train.py:115-117."""
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))


class SyntheticCamera:
    """Duck-typed scene.cameras.Camera built from a fdgs.synth scene dict."""

    def __init__(self, scene, device, timestamp=None):
        self.FoVx, self.FoVy = scene["FoVx"], scene["FoVy"]
        self.image_height, self.image_width = scene["H"], scene["W"]
        self.world_view_transform = scene["world_view_transform"].to(device)
        self.full_proj_transform = scene["full_proj_transform"].to(device)
        self.camera_center = scene["camera_center"].to(device)
        self.timestamp = scene["timestamp"] if timestamp is None else timestamp
        # pinhole intrinsics of the synthetic camera (scene/cameras.py:33-36: cx, cy, fl_x, fl_y), read by get_rays
        self.fl_x = self.image_width / (2.0 * math.tan(0.5 * self.FoVx))
        self.fl_y = self.image_height / (2.0 * math.tan(0.5 * self.FoVy))
        self.cx, self.cy = 0.5 * self.image_width, 0.5 * self.image_height

    def get_rays(self):
        """(origin [1,1,3], unit directions [H,W,3]) of the pixel centres in world space (scene/cameras.py:75-82; used by the
        environment-map branch of render(), gaussian_renderer/__init__.py:165-176)."""
        dev = self.world_view_transform.device
        ys, xs = torch.meshgrid(torch.arange(self.image_height, dtype=torch.float32, device=dev) + 0.5,
                                torch.arange(self.image_width, dtype=torch.float32, device=dev) + 0.5, indexing="ij")
        one = torch.ones_like(xs)
        pts_view = torch.stack([(xs - self.cx) / self.fl_x, (ys - self.cy) / self.fl_y, one, one], dim=-1)
        c2w = torch.linalg.inv(self.world_view_transform.transpose(0, 1))
        d = (pts_view @ c2w.T)[..., :3] - self.camera_center[None, None, :]
        return self.camera_center[None, None], d / torch.norm(d, dim=-1, keepdim=True)


class PipelineFlags:
    """arguments/__init__.py:70-79 defaults (in-kernel covariance and SH)."""
    compute_cov3D_python = False
    convert_SHs_python = False
    debug = False
    env_map_res = 0
