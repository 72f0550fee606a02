"""TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the package): the reference's OWN CUDA rasterizer compiled for gfx950
(oracle/_ref/liboracle_ref_hip.so, built in the build container by oracle/refbuild/build_ref_hip.py from /root/reference; the .so
travels to the GPU box, the sources do not).  Same role on the GPU as oracle/_ref/liboracle_ref.so has on the CPU:

  * a same-node baseline: the reference's kernels on the same MI355X, same inputs (bench.py's baseline leg, tests/test_gpu_ref_hip.py);
  * a second oracle whose float atomics run in a REAL GPU order (the CPU builds emulate one);
  * pixels / gradients of the product against the reference as a GPU actually executes it (FMA contraction on, as nvcc's default).

The reference launches on the legacy default stream (no stream arguments anywhere in cuda_rasterizer/*.cu): callers synchronise.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_ref", "liboracle_ref_hip.so")


class _Scene(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("P", "D", "D_t", "M", "W", "H")] + \
               [(n, C.c_void_p) for n in ("background", "means3D", "shs", "colors_precomp", "flows", "opacities", "ts", "scales", "scales_t",
                                          "rotations", "rotations_r", "cov3D_precomp", "viewmatrix", "projmatrix", "campos")] + \
               [(n, C.c_float) for n in ("scale_modifier", "prefilter_var", "tan_fovx", "tan_fovy", "timestamp", "time_duration")] + \
               [(n, C.c_int32) for n in ("rot_4d", "gaussian_dim", "force_sh_3d")]


_lib = None


def available():
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(SO)
        _lib.refhip_forward.restype = C.c_int
        _lib.refhip_forward.argtypes = [C.POINTER(_Scene)] + [C.c_void_p] * 6
        _lib.refhip_backward.restype = C.c_int
        _lib.refhip_backward.argtypes = [C.POINTER(_Scene), C.c_int] + [C.c_void_p] * 19
        _lib.refhip_kind.restype = C.c_char_p
    return _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class RefHip:
    """One scene (a dict as fdgs.synth.make_scene returns, tensors moved to ``device``) through the reference's kernels on the GPU."""

    def __init__(self, scene, device):
        self.dev = device
        f = lambda k: (scene[k].to(device=device, dtype=torch.float32).contiguous() if scene.get(k) is not None else None)   # noqa: E731
        self.t = {k: f(k) for k in ("bg", "means3D", "shs", "colors_precomp", "flow_2d", "opacities", "ts", "scales", "scales_t", "rotations",
                                    "rotations_r", "cov3D_precomp", "world_view_transform", "full_proj_transform", "camera_center")}
        if self.t["flow_2d"] is None:      # Q9: the reference dereferences `flows` unconditionally
            self.t["flow_2d"] = torch.zeros((self.t["means3D"].shape[0], 2), device=device)
        P = int(self.t["means3D"].shape[0])
        M = int(self.t["shs"].shape[1]) if self.t["shs"] is not None else 0
        s = _Scene()
        s.P, s.D, s.D_t, s.M, s.W, s.H = P, int(scene["sh_degree"]), int(scene["sh_degree_t"]), M, int(scene["W"]), int(scene["H"])
        for name, key in (("background", "bg"), ("means3D", "means3D"), ("shs", "shs"), ("colors_precomp", "colors_precomp"), ("flows", "flow_2d"),
                          ("opacities", "opacities"), ("ts", "ts"), ("scales", "scales"), ("scales_t", "scales_t"), ("rotations", "rotations"),
                          ("rotations_r", "rotations_r"), ("cov3D_precomp", "cov3D_precomp"), ("viewmatrix", "world_view_transform"),
                          ("projmatrix", "full_proj_transform"), ("campos", "camera_center")):
            setattr(s, name, self.t[key].data_ptr() if self.t[key] is not None else None)
        s.scale_modifier, s.prefilter_var = float(scene.get("scale_modifier", 1.0)), float(scene.get("prefilter_var", -1.0))
        s.tan_fovx, s.tan_fovy, s.timestamp, s.time_duration = float(scene["tanfovx"]), float(scene["tanfovy"]), float(scene["timestamp"]), float(scene["time_duration"])
        s.rot_4d, s.gaussian_dim, s.force_sh_3d = int(bool(scene["rot_4d"])), int(scene["gaussian_dim"]), int(bool(scene["force_sh_3d"]))
        self.s, self.P, self.M, self.W, self.H = s, P, M, s.W, s.H
        self.R = None

    def forward(self):
        """rasterize_points.cu:80-92's output initialisation, then Rasterizer::forward.  Returns a dict of device tensors."""
        dev, P, W, H = self.dev, self.P, self.W, self.H
        o = {"out_color": torch.zeros((3, H, W), device=dev), "out_flow": torch.zeros((2, H, W), device=dev), "out_depth": torch.zeros((1, H, W), device=dev),
             "out_T": torch.zeros((1, H, W), device=dev), "radii": torch.zeros((P,), dtype=torch.int32, device=dev), "out_means3D": self.t["means3D"].clone()}
        torch.cuda.synchronize(dev)
        with torch.cuda.device(dev):
            self.R = lib().refhip_forward(C.byref(self.s), _p(o["out_means3D"]), _p(o["out_color"]), _p(o["out_flow"]), _p(o["out_depth"]), _p(o["out_T"]), _p(o["radii"]))
        torch.cuda.synchronize(dev)
        if self.R < 0:
            raise RuntimeError("refhip_forward failed")
        self.out = o
        return o

    def alloc_grads(self):
        dev, P, M = self.dev, self.P, self.M
        z = lambda *s: torch.zeros(s, device=dev)   # noqa: E731
        return {"dL_dmean2D": z(P, 3), "dL_dconic": z(P, 2, 2), "dL_dopacity": z(P, 1), "dL_dcolor": z(P, 3), "dL_dmean3D": z(P, 3), "dL_dcov3D": z(P, 6),
                "dL_dsh": z(P, M, 3), "dL_dflows": z(P, 2), "dL_dts": z(P, 1), "dL_dscale": z(P, 3), "dL_dscale_t": z(P, 1), "dL_drot": z(P, 4), "dL_drot_r": z(P, 4)}

    def backward(self, grad_color, grad_depth, grad_alpha, grad_flow, grads=None, sync=True):
        """rasterize_points.cu:201-266 (zero-filled gradients), then Rasterizer::backward on the forward's buffers."""
        g = grads if grads is not None else self.alloc_grads()
        up = [t.to(self.dev).contiguous() for t in (grad_color, grad_depth, grad_alpha, grad_flow)]
        if sync:
            torch.cuda.synchronize(self.dev)
        with torch.cuda.device(self.dev):
            rc = lib().refhip_backward(C.byref(self.s), self.R, _p(self.out["out_means3D"]), _p(self.out["radii"]), _p(up[0]), _p(up[1]), _p(up[2]), _p(up[3]),
                                       _p(g["dL_dmean2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolor"]), _p(g["dL_dmean3D"]), _p(g["dL_dcov3D"]),
                                       _p(g["dL_dsh"]), _p(g["dL_dflows"]), _p(g["dL_dts"]), _p(g["dL_dscale"]), _p(g["dL_dscale_t"]), _p(g["dL_drot"]), _p(g["dL_drot_r"]))
        if sync:
            torch.cuda.synchronize(self.dev)
        if rc != 0:
            raise RuntimeError("refhip_backward failed")
        return g
