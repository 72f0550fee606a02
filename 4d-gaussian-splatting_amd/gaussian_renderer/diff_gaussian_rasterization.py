"""Drop-in for the reference module ``gaussian_renderer.diff_gaussian_rasterization``.

Same public surface (gaussian_renderer/diff_gaussian_rasterization.py in the
reference): ``GaussianRasterizationSettings`` (:227-245), ``GaussianRasterizer``
(:247-318, ``forward`` and ``markVisible``), ``rasterize_gaussians`` (:34-65) and
the autograd function ``_RasterizeGaussians`` (:67-225) -- same argument
orders, defaults, output order / shapes / dtypes and exceptions.  The reference
JIT-compiles and calls a CUDA pybind module ``_C``; here ``_C`` is a small object
with the same three entry points (ext.cpp:15-19) that forwards raw device
pointers to the hand-written HIP kernels in ``csrc/libfdgs.so`` through the C
ABI of ``include/fdgs.h``.  No rasterization arithmetic happens in Python and
there is no CPU fallback.
"""
import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from .. import _capi


_ANALYTIC_SH_GRAD = False


def set_analytic_sh_gradients(on: bool) -> None:
    """Opt-in (process-wide): the 4D-SH backward returns the analytic gradient of the forward pass instead of reproducing
    the reference's deviations Q1-Q3 (fdgs_scene.analytic_sh_grad; backward.cu:190, 303 / 384, 403).  Default off:
    gradients are bug-compatible with the reference."""
    global _ANALYTIC_SH_GRAD
    _ANALYTIC_SH_GRAD = bool(on)


def analytic_sh_gradients() -> bool:
    """The mode set_analytic_sh_gradients selected (callers of fdgs_sh_flush / fdgs_adam_step_sh pass it on)."""
    return _ANALYTIC_SH_GRAD


def _is_given(t) -> bool:
    return t is not None and t.numel() > 0


class _Scratch:
    """The three opaque byte buffers of one forward call (geometry / binning / image).

    Plays the role of the reference's resizeFunctional lambdas
    (rasterize_points.cu:28-34): the C library asks for ``bytes`` of device
    memory and torch owns the allocation.
    """

    def __init__(self, device):
        self.device = device
        self.buf = {}
        self.reuse = False   # a view preprocessed by fdgs_preprocess_batch: its forward call gets the same geometry / image buffers
        self.callback = _capi.ALLOC_FN(self._alloc)

    def _alloc(self, _user, which, nbytes):
        if self.reuse and int(which) in (_capi.FDGS_BUF_GEOMETRY, _capi.FDGS_BUF_IMAGE):
            t = self.buf.get(int(which))
            if t is not None and t.numel() >= int(nbytes):
                return t.data_ptr()
        try:
            t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
        except Exception:  # out of memory: report NULL to the C side, which returns FDGS_ERR_ALLOC
            return None
        self.buf[int(which)] = t
        return t.data_ptr()

    def get(self, which):
        t = self.buf.get(which)
        return t if t is not None else torch.empty(0, dtype=torch.uint8, device=self.device)

    def release(self):
        """The three buffers; breaks the self -> ctypes callback -> bound method -> self cycle, which would otherwise
        keep ~0.5 GB per call alive until Python's cyclic GC runs (and make the caching allocator grow)."""
        out = (self.get(_capi.FDGS_BUF_GEOMETRY), self.get(_capi.FDGS_BUF_BINNING), self.get(_capi.FDGS_BUF_IMAGE))
        self.callback = None
        self.buf = {}
        return out


class _NativeRasterizer:
    """Stand-in for the reference's pybind module ``_C`` (ext.cpp:15-19)."""

    @staticmethod
    def _scene(bg, means3D, colors, flows, opacity, ts, scales, scales_t, rotations, rotations_r, scale_modifier,
               cov3D_precomp, prefilter_var, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, degree_t,
               campos, timestamp, time_duration, rot_4d, gaussian_dim, force_sh_3d, prefiltered, debug, raw_params=False):
        if means3D.ndim != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:69-71
        f = _capi._dev_f32
        keep = {
            "bg": f(bg, "bg"), "means3D": f(means3D, "means3D"), "shs": f(sh, "sh"),
            "colors_precomp": f(colors, "colors_precomp"), "flows": f(flows, "flow_2d"),
            "opacities": f(opacity, "opacities"), "ts": f(ts, "ts"), "scales": f(scales, "scales"),
            "scales_t": f(scales_t, "scales_t"), "rotations": f(rotations, "rotations"),
            "rotations_r": f(rotations_r, "rotations_r"), "cov3D_precomp": f(cov3D_precomp, "cov3D_precomp"),
            "viewmatrix": f(viewmatrix, "viewmatrix"), "projmatrix": f(projmatrix, "projmatrix"),
            "campos": f(campos, "campos"),
        }
        s = _capi.FdgsScene()
        s.P = int(means3D.shape[0])
        s.D, s.D_t = int(degree), int(degree_t)
        s.M = int(sh.shape[1]) if _is_given(sh) else 0  # rasterize_points.cu:99-103
        s.W, s.H = int(W), int(H)
        for k, t in keep.items():
            setattr(s, k, _capi._ptr(t))
        s.scale_modifier, s.prefilter_var = float(scale_modifier), float(prefilter_var)
        s.tan_fovx, s.tan_fovy = float(tan_fovx), float(tan_fovy)
        s.timestamp, s.time_duration = float(timestamp), float(time_duration)
        s.rot_4d, s.gaussian_dim, s.force_sh_3d = int(bool(rot_4d)), int(gaussian_dim), int(bool(force_sh_3d))
        s.prefiltered, s.debug = int(bool(prefiltered)), int(bool(debug))
        s.raw_params = int(bool(raw_params))
        s.analytic_sh_grad = int(_ANALYTIC_SH_GRAD)
        return s, keep

    def rasterize_gaussians(self, bg, means3D, colors, flows, opacity, ts, scales, scales_t, rotations, rotations_r,
                            scale_modifier, cov3D_precomp, prefilter_var, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                            image_height, image_width, sh, degree, degree_t, campos, timestamp, time_duration, rot_4d,
                            gaussian_dim, force_sh_3d, prefiltered, debug, *, raw_params=False, split_colour=False, preprocessed=None,
                            tile_cull=False, lazy=False, sparse_lists=False, colour_stream=None):
        """30 positional arguments and the 11-tuple result of the reference binding (rasterize_points.h:18-49).
        Keyword-only extensions: ``raw_params``: the scale / opacity / rotation tensors are the model's raw
        parameters and the kernels apply the activations (fdgs_scene.raw_params); ``split_colour``: the SH colour evaluation
        runs on the library's second stream next to the tile binning (fdgs_forward_out.split_colour; forward-only rendering);
        ``tile_cull``: a Gaussian is only listed in the tiles it can reach with alpha >= 1/255 (fdgs_forward_out.tile_cull: same
        pixels and gradients, shorter tile lists); ``lazy``: fdgs_forward_out.lazy -- the call does not wait for num_rendered
        (returned as -1; the backward takes it) and ``_capi.forward_lazy_status`` later says whether the run-ahead buffers fitted;
        ``sparse_lists`` (with ``lazy``): fdgs_forward_out.sparse_lists -- every tile's list at a fixed offset of the binning buffer, no
        count / scan launches; ``colour_stream`` (with ``split_colour``): the torch.cuda.Stream the colour launch goes onto instead of
        the library's own (fdgs_forward_out.colour_stream)."""
        if not means3D.is_cuda:
            raise RuntimeError("fdgs: means3D must live on the GPU; there is no CPU path")
        dev = means3D.device
        if preprocessed is not None:
            # ``preprocessed``: this view's handle from preprocess_batch (same arguments): geometry and SH colours are already
            # enqueued on this stream; the call continues with the tile binning and the blend
            scene, keep = preprocessed["scene"], preprocessed["keep"]
        else:
            scene, keep = self._scene(bg, means3D, colors, flows, opacity, ts, scales, scales_t, rotations, rotations_r,
                                      scale_modifier, cov3D_precomp, prefilter_var, viewmatrix, projmatrix, tan_fovx,
                                      tan_fovy, image_height, image_width, sh, degree, degree_t, campos, timestamp,
                                      time_duration, rot_4d, gaussian_dim, force_sh_3d, prefiltered, debug, raw_params)
        P, H, W = scene.P, scene.H, scene.W
        fo = dict(dtype=torch.float32, device=dev)
        # every output is fully written by the kernels: torch.empty, not torch.full (rasterize_points.cu:80-85)
        out_color = torch.empty((3, H, W), **fo)
        out_flow = torch.empty((2, H, W), **fo)
        out_depth = torch.empty((1, H, W), **fo)
        out_T = torch.empty((1, H, W), **fo)
        if preprocessed is not None:
            radii, out_means3D, covs_com, scratch = (preprocessed[k] for k in ("radii", "out_means3D", "covs_com", "scratch"))
        else:
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            out_means3D = torch.empty((P, 3), **fo)
            covs_com = torch.empty((P, 6), **fo)  # owning, not a from_blob alias of the scratch (rasterize_points.cu:144-147)
            scratch = _Scratch(dev)
        out = _capi.FdgsForwardOut(out_color.data_ptr(), out_flow.data_ptr(), out_depth.data_ptr(), out_T.data_ptr(),
                                   _capi._ptr(radii), _capi._ptr(out_means3D), _capi._ptr(covs_com),
                                   int(preprocessed is not None), int(bool(split_colour)),
                                   int(preprocessed["tile_cull"] if preprocessed is not None else bool(tile_cull)), int(bool(lazy)),
                                   int(bool(sparse_lists)), colour_stream.cuda_stream if colour_stream is not None else None)
        R = C.c_int32(0)
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_rasterize_forward(C.byref(scene), C.byref(out), scratch.callback, None,
                                                  _capi.current_stream_handle(dev), C.byref(R))
        geom, binb, img = scratch.release()
        _capi._check(rc, "fdgs_rasterize_forward")
        del keep
        return (int(R.value), out_color, out_flow, out_depth, out_T, radii, geom, binb, img, covs_com, out_means3D)

    def preprocess_batch(self, views, *, raw_params=False, tile_cull=False):
        """View-batched preprocess (fdgs_preprocess_batch): ``views`` = the 30-tuples of positional arguments of
        ``rasterize_gaussians`` for the views of ONE optimizer step (same Gaussian tensors, own camera / timestamp).  The
        geometry runs per view, the SH colours of all views in one pass over the coefficients.  Returns one handle per view;
        ``rasterize_gaussians(*views[v], raw_params=..., preprocessed=handles[v])`` then completes view v on the same stream."""
        dev = views[0][1].device
        if not views[0][1].is_cuda:
            raise RuntimeError("fdgs: means3D must live on the GPU; there is no CPU path")
        handles = []
        for a in views:
            scene, keep = self._scene(*a, raw_params)
            P = scene.P
            fo = dict(dtype=torch.float32, device=dev)
            h = {"scene": scene, "keep": keep, "radii": torch.empty((P,), dtype=torch.int32, device=dev),
                 "out_means3D": torch.empty((P, 3), **fo), "covs_com": torch.empty((P, 6), **fo), "scratch": _Scratch(dev),
                 "tile_cull": bool(tile_cull)}
            h["scratch"].reuse = True
            h["out"] = _capi.FdgsForwardOut(None, None, None, None, _capi._ptr(h["radii"]), _capi._ptr(h["out_means3D"]),
                                            _capi._ptr(h["covs_com"]), 0, 0, int(bool(tile_cull)), 0, 0, None)
            handles.append(h)
        B = len(handles)
        scenes = (C.POINTER(_capi.FdgsScene) * B)(*[C.pointer(h["scene"]) for h in handles])
        outs = (C.POINTER(_capi.FdgsForwardOut) * B)(*[C.pointer(h["out"]) for h in handles])
        users = (C.c_void_p * B)(*[v + 1 for v in range(B)])
        # one callback for the batch: the user word says which view's buffer is asked for
        cb = _capi.ALLOC_FN(lambda user, which, nbytes: handles[int(user) - 1]["scratch"]._alloc(None, which, nbytes))
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_preprocess_batch(B, scenes, outs, cb, users, _capi.current_stream_handle(dev))
        _capi._check(rc, "fdgs_preprocess_batch")
        return handles

    def sh_backward_batch(self, pendings):
        """View-batched SH backward (fdgs_sh_backward_batch) of the views whose blend backward ``backward_begin`` has enqueued."""
        B = len(pendings)
        dev = pendings[0]["dev"]
        scenes = (C.POINTER(_capi.FdgsScene) * B)(*[C.pointer(p["scene"]) for p in pendings])
        ins = (C.POINTER(_capi.FdgsBackwardIn) * B)(*[C.pointer(p["bin"]) for p in pendings])
        outs = (C.POINTER(_capi.FdgsBackwardOut) * B)(*[C.pointer(p["bout"]) for p in pendings])
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_sh_backward_batch(B, scenes, ins, outs, _capi.current_stream_handle(dev))
        _capi._check(rc, "fdgs_sh_backward_batch")

    def backward_begin(self, *args, **kw):
        """The blend backward of one view only (fdgs_backward_out.stage_mask = 5); same arguments as
        ``rasterize_gaussians_backward`` (``sh_stage`` and a ``grad_accum`` of the view's own are required).  Returns the pending
        call for ``sh_backward_batch`` / ``backward_finish``."""
        return self.rasterize_gaussians_backward(*args, _phase="begin", **kw)

    def backward_finish(self, pending):
        """The geometry backward (stage_mask = 2) of a view begun with ``backward_begin``, after ``sh_backward_batch``; returns the
        binding's 12-tuple."""
        dev = pending["dev"]
        pending["bout"].stage_mask = 2
        with torch.cuda.device(dev):
            rc = _capi.lib.fdgs_rasterize_backward(C.byref(pending["scene"]), C.byref(pending["bin"]), C.byref(pending["bout"]),
                                                   _capi.current_stream_handle(dev))
        if rc != 0 and pending["clean"]:
            pending["grad_accum"].zero_()
        _capi._check(rc, "fdgs_rasterize_backward")
        g = pending["g"]
        return (g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dopacity"], g["dL_dmeans3D"], g["dL_dcov3D"], g["dL_dsh"],
                g["dL_dflows"], g["dL_dts"], g["dL_dscales"], g["dL_dscales_t"], g["dL_drotations"], g["dL_drotations_r"])

    def rasterize_gaussians_backward(self, bg, means3D, out_means3D, radii, colors, flows_2d, opacities, ts, scales,
                                     scales_t, rotations, rotations_r, scale_modifier, cov3D_precomp, prefilter_var,
                                     viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                     dL_dout_mask, dL_dout_flow, sh, degree, degree_t, campos, timestamp,
                                     time_duration, rot_4d, gaussian_dim, force_sh_3d, geomBuffer, R, binningBuffer,
                                     imageBuffer, debug, *, raw_params=False, grad_out=None, accumulate=False, grad_accum=None,
                                     after_sh=None, sh_stage=None, per_view_outputs=True, geometry_adam=None, _phase=None):
        """37 positional arguments and the 12-tuple result of the reference binding (rasterize_points.h:51-89).
        Keyword-only extensions: ``raw_params`` as in the forward; ``grad_out`` maps gradient names
        (dL_dmeans3D, dL_dsh, dL_dopacity, dL_dts, dL_dscales, dL_dscales_t, dL_drotations, dL_drotations_r) to
        preallocated contiguous tensors the kernels write into (e.g. views of a flat gradient bucket);
        ``accumulate``: the kernels ADD into those parameter gradients instead of overwriting them;
        ``grad_accum``: a persistent all-zero [P,16] float32 scratch tensor owned by the caller -- the call then skips
        its memset and leaves the tensor all zero again (fdgs_backward_out.grad_accum_clean);
        ``after_sh``: a callable invoked between the blend + SH backward and the geometry backward (two native calls,
        fdgs_backward_out.stage_mask): dL_dsh is final at that point, so a data-parallel caller can start its all-reduce;
        ``sh_stage``: a [P,8] float32 scratch tensor -> deferred SH gradient (fdgs_backward_out.sh_stage): dL_dsh is not
        touched by this call, ``_capi.sh_flush`` builds it from the stages of all views of the step;
        ``per_view_outputs=False``: dL_dcolors, dL_dcov3D and dL_dflows are not written (NULL at the C ABI) and come back as None;
        ``geometry_adam``: a callable evaluated right before the geometry backward is enqueued (after ``after_sh``) that returns None or
        dict(flat, exp_avg, exp_avg_sq, lr={means3D, opacities, ts, scales, scales_t, rotations, rotations_r}, betas, eps, step): the
        geometry backward then also takes the Adam step of the geometry parameters (fdgs_backward_out.adam; raw_params only)."""
        dev = means3D.device
        # The reference always receives four dense tensors (autograd materialises zeros).  Here an image gradient may
        # be None = "no upstream gradient": the kernels then skip that term (colour-only backward when only
        # dL_dout_color is given, the usual photometric-loss case).
        given = [t for t in (dL_dout_color, dL_dout_depth, dL_dout_mask, dL_dout_flow) if t is not None]
        if not given:
            raise RuntimeError("fdgs: backward needs at least one upstream image gradient")
        H, W = int(given[0].shape[1]), int(given[0].shape[2])  # rasterize_points.cu:192-193
        scene, keep = self._scene(bg, means3D, colors, flows_2d, opacities, ts, scales, scales_t, rotations,
                                  rotations_r, scale_modifier, cov3D_precomp, prefilter_var, viewmatrix, projmatrix,
                                  tan_fovx, tan_fovy, H, W, sh, degree, degree_t, campos, timestamp, time_duration,
                                  rot_4d, gaussian_dim, force_sh_3d, False, debug, raw_params)
        P, M = scene.P, scene.M
        fo = dict(dtype=torch.float32, device=dev)
        # fully written by the kernels (the reference zero-fills all 13 with torch::zeros, rasterize_points.cu:201-213)
        g = {
            "dL_dmeans2D": torch.empty((P, 3), **fo), "dL_dcolors": torch.empty((P, 3), **fo),
            "dL_dopacity": torch.empty((P, 1), **fo), "dL_dmeans3D": torch.empty((P, 3), **fo),
            "dL_dcov3D": torch.empty((P, 6), **fo), "dL_dsh": torch.empty((P, M, 3), **fo),
            "dL_dflows": torch.empty((P, 2), **fo), "dL_dts": torch.empty((P, 1), **fo),
            "dL_dscales": torch.empty((P, 3), **fo), "dL_dscales_t": torch.empty((P, 1), **fo),
            "dL_drotations": torch.empty((P, 4), **fo), "dL_drotations_r": torch.empty((P, 4), **fo),
            "grad_accum": None if grad_accum is not None else torch.empty((P, 16), **fo),  # packed blend-backward accumulators
        }
        if grad_out:
            for name, t in grad_out.items():
                if t is None:
                    continue
                if name not in g or t.numel() != g[name].numel() or not t.is_contiguous() or t.dtype != torch.float32:
                    raise RuntimeError("fdgs: grad_out[%r] must be a contiguous float32 tensor with %d elements" %
                                       (name, g[name].numel() if name in g else -1))
                g[name] = t
        gin = [_capi._dev_f32(t, n) for t, n in ((dL_dout_color, "dL_dout_color"), (dL_dout_depth, "dL_dout_depth"),
                                                 (dL_dout_mask, "dL_dout_mask"), (dL_dout_flow, "dL_dout_flow"))]
        radii_c, om_c = radii.contiguous(), out_means3D.contiguous()
        bin_ = _capi.FdgsBackwardIn(_capi._ptr(gin[0]), _capi._ptr(gin[1]), _capi._ptr(gin[2]), _capi._ptr(gin[3]),
                                    _capi._ptr(radii_c), _capi._ptr(om_c), _capi._ptr(geomBuffer),
                                    _capi._ptr(binningBuffer), _capi._ptr(imageBuffer), int(R))
        if accumulate and not grad_out:
            raise RuntimeError("fdgs: accumulate=True needs grad_out buffers that already hold gradients")
        if not per_view_outputs:
            g["dL_dcolors"] = g["dL_dcov3D"] = g["dL_dflows"] = None
        ptrs = [_capi._ptr(g[k]) for k in (
            "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dflows", "dL_dts",
            "dL_dscales", "dL_dscales_t", "dL_drotations", "dL_drotations_r")]
        clean = 0
        if grad_accum is not None:
            if grad_accum.numel() != P * 16 or grad_accum.dtype != torch.float32 or not grad_accum.is_contiguous():
                raise RuntimeError("fdgs: grad_accum must be a contiguous float32 tensor with %d elements" % (P * 16))
            g["grad_accum"], clean = grad_accum, 1
        if sh_stage is not None and (sh_stage.numel() != P * 8 or sh_stage.dtype != torch.float32 or not sh_stage.is_contiguous()):
            raise RuntimeError("fdgs: sh_stage must be a contiguous float32 tensor with %d elements" % (P * 8))
        bout = _capi.FdgsBackwardOut(*ptrs, int(bool(accumulate)), _capi._ptr(g["grad_accum"]), clean, _capi._ptr(sh_stage), 0)
        if _phase == "begin":
            if sh_stage is None or grad_accum is None:
                raise RuntimeError("fdgs: backward_begin needs sh_stage and a grad_accum of the view's own")
            bout.stage_mask = 5   # blend backward only; the SH backward is left to sh_backward_batch
            with torch.cuda.device(dev):
                rc = _capi.lib.fdgs_rasterize_backward(C.byref(scene), C.byref(bin_), C.byref(bout), _capi.current_stream_handle(dev))
            if rc != 0:
                grad_accum.zero_()
            _capi._check(rc, "fdgs_rasterize_backward")
            return {"scene": scene, "keep": keep, "bin": bin_, "bout": bout, "g": g, "dev": dev, "clean": clean, "grad_accum": grad_accum,
                    "alive": (gin, radii_c, om_c, geomBuffer, binningBuffer, imageBuffer, sh_stage)}
        adam_keep = None

        def attach_adam():
            ga = geometry_adam() if geometry_adam is not None else None
            if ga is None:
                return None
            lr = ga["lr"]
            st = _capi.FdgsGeometryAdam(_capi._ptr(ga["flat"]), _capi._ptr(ga["exp_avg"]), _capi._ptr(ga["exp_avg_sq"]), float(lr["means3D"]),
                                        float(lr["opacities"]), float(lr.get("ts", 0.0)), float(lr["scales"]), float(lr.get("scales_t", 0.0)),
                                        float(lr["rotations"]), float(lr.get("rotations_r", 0.0)), float(ga["betas"][0]), float(ga["betas"][1]),
                                        float(ga["eps"]), int(ga["step"]))
            bout.adam = C.pointer(st)
            return (st, ga)
        with torch.cuda.device(dev):
            if after_sh is None:
                adam_keep = attach_adam()
                rc = _capi.lib.fdgs_rasterize_backward(C.byref(scene), C.byref(bin_), C.byref(bout),
                                                       _capi.current_stream_handle(dev))
            else:
                bout.stage_mask = 1
                rc = _capi.lib.fdgs_rasterize_backward(C.byref(scene), C.byref(bin_), C.byref(bout),
                                                       _capi.current_stream_handle(dev))
                if rc == 0:
                    after_sh()
                    bout.stage_mask = 2
                    adam_keep = attach_adam()
                    rc = _capi.lib.fdgs_rasterize_backward(C.byref(scene), C.byref(bin_), C.byref(bout),
                                                           _capi.current_stream_handle(dev))
        del adam_keep
        if rc != 0 and clean:
            # a failed call may have left partial sums behind: restore the caller's "all zero on entry" invariant
            grad_accum.zero_()
        _capi._check(rc, "fdgs_rasterize_backward")
        del keep
        return (g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dopacity"], g["dL_dmeans3D"], g["dL_dcov3D"], g["dL_dsh"],
                g["dL_dflows"], g["dL_dts"], g["dL_dscales"], g["dL_dscales_t"], g["dL_drotations"],
                g["dL_drotations_r"])

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        """rasterize_points.cu:272-291: bool[P], True where view-space z > 0.2."""
        if not means3D.is_cuda:
            raise RuntimeError("fdgs: means3D must live on the GPU; there is no CPU path")
        dev = means3D.device
        P = int(means3D.shape[0])
        present = torch.zeros((P,), dtype=torch.bool, device=dev)
        m = _capi._dev_f32(means3D, "means3D")
        v = _capi._dev_f32(viewmatrix, "viewmatrix")
        p = _capi._dev_f32(projmatrix, "projmatrix")
        if P:
            with torch.cuda.device(dev):
                rc = _capi.lib.fdgs_mark_visible(P, m.data_ptr(), v.data_ptr(), _capi._ptr(p), present.data_ptr(),
                                                 _capi.current_stream_handle(dev))
            _capi._check(rc, "fdgs_mark_visible")
        return present


_C = _NativeRasterizer()


def cpu_deep_copy_tuple(input_tuple):
    return tuple(x.cpu().clone() if isinstance(x, torch.Tensor) else x for x in input_tuple)


def _call_native(fn, args, debug, dump_name, where):
    """debug mode keeps the reference behaviour: snapshot the arguments and dump them if the native call throws
    (gaussian_renderer/diff_gaussian_rasterization.py:122-129, 193-202)."""
    if not debug:
        return fn(*args)
    snapshot = cpu_deep_copy_tuple(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(snapshot, dump_name)
        print("\nAn error occured in %s. Please forward %s for debugging." % (where, dump_name))
        raise


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    sh_degree_t: int
    campos: torch.Tensor
    timestamp: float
    time_duration: float
    rot_4d: bool
    gaussian_dim: int
    force_sh_3d: bool
    prefiltered: bool
    debug: bool


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, flow_2d, opacities, ts, scales, scales_t, rotations,
                rotations_r, cov3Ds_precomp, prefilter_var, raster_settings):
        rs = raster_settings
        native_args = (
            rs.bg, means3D, colors_precomp, flow_2d, opacities, ts, scales, scales_t, rotations, rotations_r,
            rs.scale_modifier, cov3Ds_precomp, prefilter_var, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
            rs.image_height, rs.image_width, sh, rs.sh_degree, rs.sh_degree_t, rs.campos, rs.timestamp,
            rs.time_duration, rs.rot_4d, rs.gaussian_dim, rs.force_sh_3d, rs.prefiltered, rs.debug,
        )
        (num_rendered, color, flow, depth, T, radii, geomBuffer, binningBuffer, imgBuffer, covs_com,
         out_means3D) = _call_native(_C.rasterize_gaussians, native_args, rs.debug, "snapshot_fw.dump", "forward")

        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.prefilter_var = prefilter_var
        ctx.save_for_backward(colors_precomp, means3D, out_means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              flow_2d, opacities, ts, scales_t, rotations_r, geomBuffer, binningBuffer, imgBuffer)
        # outputs nobody differentiates arrive as None in backward instead of dense zeros (see the native binding)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, 1 - T, flow, covs_com

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_alpha, grad_flow, grad_covs_com):
        rs = ctx.raster_settings
        (colors_precomp, means3D, out_means3D, scales, rotations, cov3Ds_precomp, radii, sh, flow_2d, opacities, ts,
         scales_t, rotations_r, geomBuffer, binningBuffer, imgBuffer) = ctx.saved_tensors
        if grad_out_color is None and grad_depth is None and grad_alpha is None and grad_flow is None:
            # only covs_com was differentiated: the reference ignores that gradient, the images contribute zeros
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        native_args = (
            rs.bg, means3D, out_means3D, radii, colors_precomp, flow_2d, opacities, ts, scales, scales_t, rotations,
            rotations_r, rs.scale_modifier, cov3Ds_precomp, ctx.prefilter_var, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, grad_alpha, grad_flow, sh, rs.sh_degree,
            rs.sh_degree_t, rs.campos, rs.timestamp, rs.time_duration, rs.rot_4d, rs.gaussian_dim, rs.force_sh_3d,
            geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug,
        )
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_flows,
         grad_ts, grad_scales, grad_scales_t, grad_rotations, grad_rotations_r) = _call_native(
            _C.rasterize_gaussians_backward, native_args, rs.debug, "snapshot_bw.dump", "backward")

        def shaped(given, g):
            """gradient reshaped like its input, or None when that optional input was absent"""
            return g.reshape(given.shape) if _is_given(given) else None

        # order = forward's inputs (gaussian_renderer/diff_gaussian_rasterization.py:208-225)
        return (
            grad_means3D, grad_means2D, shaped(sh, grad_sh), shaped(colors_precomp, grad_colors_precomp),
            shaped(flow_2d, grad_flows), shaped(opacities, grad_opacities), shaped(ts, grad_ts),
            shaped(scales, grad_scales), shaped(scales_t, grad_scales_t), shaped(rotations, grad_rotations),
            shaped(rotations_r, grad_rotations_r), shaped(cov3Ds_precomp, grad_cov3Ds_precomp), None, None,
        )


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, flow_2d, opacities, ts, scales, scales_t, rotations,
                        rotations_r, cov3Ds_precomp, prefilter_var, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, flow_2d, opacities, ts, scales, scales_t,
                                     rotations, rotations_r, cov3Ds_precomp, prefilter_var, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the points in front of the near plane (frustum culling for the camera)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, flow_2d=None, ts=None, scales=None,
                scales_t=None, rotations=None, rotations_r=None, cov3D_precomp=None, prefilter_var=-1.0):
        rs = self.raster_settings

        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if rs.rot_4d and cov3D_precomp is None and (rotations_r is None or scales_t is None or ts is None):
            raise Exception('Please provide exactly rotations_r and scales_t and ts if rot_4d and cov3D_precomp is None!')

        # absent tensors travel as empty CPU tensors == NULL pointers
        # (gaussian_renderer/diff_gaussian_rasterization.py:282-300)
        def absent(t):
            return torch.Tensor([]) if t is None else t

        return rasterize_gaussians(means3D, means2D, absent(shs), absent(colors_precomp), absent(flow_2d), opacities,
                                   absent(ts), absent(scales), absent(scales_t), absent(rotations),
                                   absent(rotations_r), absent(cov3D_precomp), prefilter_var, rs)
