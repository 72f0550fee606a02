"""TEST INFRASTRUCTURE ONLY -- ctypes front-end for the CPU oracles.

Loads ``oracle/liboracle_port.so`` (our scalar restatement, oracle/fdgs_oracle.c)
or ``oracle/_ref/liboracle_ref.so`` (the reference's own kernel source compiled
verbatim for the CPU, oracle/refbuild/) and runs them on numpy arrays through
the common interface in oracle/oracle_api.h.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "liboracle_port.so")
REF_SO = os.path.join(HERE, "_ref", "liboracle_ref.so")
REF_FMA_SO = os.path.join(HERE, "_ref", "liboracle_ref_fma.so")   # the same source with FP contraction ON (build_ref.py --contract)

_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
_I32 = C.POINTER(C.c_int32)
_U8 = C.POINTER(C.c_uint8)
_U64 = C.POINTER(C.c_uint64)


class OracleIO(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("D_t", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("bg", _F), ("means3D", _F), ("shs", _F), ("colors_precomp", _F), ("flows", _F), ("opacities", _F),
        ("ts", _F), ("scales", _F), ("scales_t", _F), ("rotations", _F), ("rotations_r", _F),
        ("cov3D_precomp", _F), ("viewmatrix", _F), ("projmatrix", _F), ("campos", _F),
        ("scale_modifier", C.c_float), ("prefilter_var", C.c_float),
        ("timestamp", C.c_float), ("time_duration", C.c_float),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
        ("rot_4d", C.c_int), ("gaussian_dim", C.c_int), ("force_sh_3d", C.c_int), ("prefiltered", C.c_int),
        ("out_color", _F), ("out_flow", _F), ("out_depth", _F), ("out_T", _F), ("n_contrib", _U32),
        ("radii", _I32), ("out_means3D", _F), ("means2D", _F), ("depths", _F), ("cov3D", _F), ("rgb", _F),
        ("conic_opacity", _F), ("tiles_touched", _U32), ("point_offsets", _U32), ("clamped", _U8),
        ("ranges", _U32), ("border", _U8), ("border_g", _U8),
        ("R", C.c_int), ("keys_sorted", _U64), ("point_list", _U32),
        ("dL_dpix", _F), ("dL_ddepth", _F), ("dL_dmask", _F), ("dL_dflow", _F),
        ("dL_dmean2D", _F), ("dL_dconic", _F), ("dL_dopacity", _F), ("dL_dcolor", _F), ("dL_dmean3D", _F),
        ("dL_dcov3D", _F), ("dL_dsh", _F), ("dL_dflows", _F), ("dL_dts", _F), ("dL_dscale", _F),
        ("dL_dscale_t", _F), ("dL_drot", _F), ("dL_drot_r", _F),
        ("analytic_sh", C.c_int),
    ]


def build_port(force=False):
    """Compile oracle/fdgs_oracle.c -> liboracle_port.so (gcc, -ffp-contract=off, OpenMP)."""
    src = os.path.join(HERE, "fdgs_oracle.c")
    if (not force) and os.path.exists(PORT_SO) and os.path.getmtime(PORT_SO) >= max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "oracle_api.h"))):
        return PORT_SO
    cmd = ["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
           "-o", PORT_SO, src, "-lm"]
    subprocess.check_call(cmd)
    return PORT_SO


def build_ref(contract=False):
    """Compile the verbatim-reference oracle if /root/reference is present; returns path or None.
    ``contract``: also the build with FP contraction on (liboracle_ref_fma.so)."""
    script = os.path.join(HERE, "refbuild", "build_ref.py")
    rc = subprocess.call([sys.executable, script] + (["--contract"] if contract else []), stdout=subprocess.DEVNULL)
    return REF_SO if rc == 0 and os.path.exists(REF_SO) else None


def have_ref():
    return os.path.exists(REF_SO)


_LIBS = {}


def _load(kind):
    if kind in _LIBS:
        return _LIBS[kind]
    if kind == "port":
        path = build_port()
    elif kind == "reference":
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (run oracle/refbuild/build_ref.py where /root/reference exists)")
        path = REF_SO
    elif kind == "reference_fma":
        if not os.path.exists(REF_FMA_SO):
            raise FileNotFoundError(REF_FMA_SO + " (run oracle/refbuild/build_ref.py --contract where /root/reference exists)")
        path = REF_FMA_SO
    else:
        raise ValueError(kind)
    lib = C.CDLL(path)
    lib.oracle_forward.argtypes = [C.POINTER(OracleIO)]
    lib.oracle_forward.restype = C.c_int
    lib.oracle_backward.argtypes = [C.POINTER(OracleIO)]
    lib.oracle_backward.restype = C.c_int
    lib.oracle_free.argtypes = [C.POINTER(OracleIO)]
    lib.oracle_free.restype = None
    lib.oracle_mark_visible.argtypes = [C.c_int, _F, _F, _F, _U8]
    lib.oracle_mark_visible.restype = C.c_int
    lib.oracle_kind.restype = C.c_char_p
    lib.oracle_threads.restype = C.c_int
    assert lib.oracle_kind().decode() == kind
    _LIBS[kind] = lib
    return lib


def _np(x, dtype=np.float32):
    """numpy view of a torch tensor / array-like, C-contiguous, or None."""
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
    return a if a.size > 0 else None


def _ptr(a, ctype):
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))


class Oracle:
    """One oracle run (forward, optionally backward) on a scene dict from fdgs.synth.make_scene.

    ``scene`` keys used: means3D, opacities, shs | colors_precomp, flow_2d, ts,
    scales, scales_t, rotations, rotations_r | cov3D_precomp, bg,
    world_view_transform, full_proj_transform, camera_center, tanfovx, tanfovy,
    W, H, sh_degree, sh_degree_t, timestamp, time_duration, rot_4d, gaussian_dim,
    force_sh_3d, scale_modifier, prefilter_var.
    """

    def __init__(self, scene, kind="port"):
        self.kind = kind
        self.lib = _load(kind)
        s = scene
        P = int(s["means3D"].shape[0])
        W, H = int(s["W"]), int(s["H"])
        self.P, self.W, self.H = P, W, H
        self.T = ((W + 15) // 16) * ((H + 15) // 16)
        inp = self.inp = {}
        inp["bg"] = _np(s["bg"])
        inp["means3D"] = _np(s["means3D"])
        inp["shs"] = _np(s.get("shs"))
        inp["colors_precomp"] = _np(s.get("colors_precomp"))
        flows = s.get("flow_2d")
        inp["flows"] = _np(flows) if flows is not None else np.zeros((P, 2), np.float32)
        inp["opacities"] = _np(s["opacities"])
        for k in ("ts", "scales", "scales_t", "rotations", "rotations_r", "cov3D_precomp"):
            inp[k] = _np(s.get(k))
        inp["viewmatrix"] = _np(s["world_view_transform"])
        inp["projmatrix"] = _np(s["full_proj_transform"])
        inp["campos"] = _np(s["camera_center"])
        M = 0 if inp["shs"] is None else int(inp["shs"].shape[1])
        self.M = M
        io = self.io = OracleIO()
        io.P, io.D, io.D_t, io.M, io.W, io.H = P, int(s["sh_degree"]), int(s["sh_degree_t"]), M, W, H
        for k in ("bg", "means3D", "shs", "colors_precomp", "flows", "opacities", "ts", "scales", "scales_t",
                  "rotations", "rotations_r", "cov3D_precomp", "viewmatrix", "projmatrix", "campos"):
            setattr(io, k, _ptr(inp[k], C.c_float))
        io.scale_modifier = float(s.get("scale_modifier", 1.0))
        io.prefilter_var = float(s.get("prefilter_var", -1.0))
        io.timestamp = float(s["timestamp"])
        io.time_duration = float(s["time_duration"])
        io.tan_fovx, io.tan_fovy = float(s["tanfovx"]), float(s["tanfovy"])
        io.rot_4d, io.gaussian_dim = int(bool(s["rot_4d"])), int(s["gaussian_dim"])
        io.force_sh_3d, io.prefiltered = int(bool(s["force_sh_3d"])), 0
        io.analytic_sh = int(bool(s.get("analytic_sh_grad", False)))   # port oracle only
        N = W * H
        out = self.out = {
            "out_color": np.zeros((3, H, W), np.float32), "out_flow": np.zeros((2, H, W), np.float32),
            "out_depth": np.zeros((H, W), np.float32), "out_T": np.zeros((H, W), np.float32),
            "n_contrib": np.zeros((H, W), np.uint32), "radii": np.zeros((P,), np.int32),
            "out_means3D": np.zeros((P, 3), np.float32), "means2D": np.zeros((P, 2), np.float32),
            "depths": np.zeros((P,), np.float32), "cov3D": np.zeros((P, 6), np.float32),
            "rgb": np.zeros((P, 3), np.float32), "conic_opacity": np.zeros((P, 4), np.float32),
            "tiles_touched": np.zeros((P,), np.uint32), "point_offsets": np.zeros((P,), np.uint32),
            "clamped": np.zeros((P, 3), np.uint8), "ranges": np.zeros((self.T, 2), np.uint32),
            "border": np.zeros((H, W), np.uint8), "border_g": np.zeros((P,), np.uint8),
        }
        ctypes_of = {np.dtype(np.float32): C.c_float, np.dtype(np.uint32): C.c_uint32,
                     np.dtype(np.int32): C.c_int32, np.dtype(np.uint8): C.c_uint8}
        for k, a in out.items():
            setattr(io, k, _ptr(a, ctypes_of[a.dtype]))
        self.R = None
        self.grads = None
        self._freed = True

    def forward(self):
        R = self.lib.oracle_forward(C.byref(self.io))
        if R < 0:
            raise RuntimeError("oracle_forward failed: %d" % R)
        self._freed = False
        self.R = R
        if R > 0:
            self.out["keys_sorted"] = np.ctypeslib.as_array(self.io.keys_sorted, shape=(R,)).copy()
            self.out["point_list"] = np.ctypeslib.as_array(self.io.point_list, shape=(R,)).copy()
        else:
            self.out["keys_sorted"] = np.zeros((0,), np.uint64)
            self.out["point_list"] = np.zeros((0,), np.uint32)
        return self.out

    def backward(self, grad_color, grad_depth, grad_alpha, grad_flow):
        """Upstream grads w.r.t. (color[3,H,W], depth[1,H,W], alpha[1,H,W], flow[2,H,W]).

        alpha = 1 - T is formed in Python by the reference
        (gaussian_renderer/diff_gaussian_rasterization.py:140); the kernels receive
        grad_alpha unchanged as dL_dmask (``:176``), which is what is passed here.
        """
        assert self.R is not None, "call forward() first"
        P, M = self.P, self.M
        gi = self.gin = {
            "dL_dpix": _np(grad_color).reshape(3, self.H, self.W),
            "dL_ddepth": _np(grad_depth).reshape(self.H, self.W),
            "dL_dmask": _np(grad_alpha).reshape(self.H, self.W),
            "dL_dflow": _np(grad_flow).reshape(2, self.H, self.W),
        }
        for k, a in gi.items():
            setattr(self.io, k, _ptr(a, C.c_float))
        g = self.grads = {
            "dL_dmean2D": np.zeros((P, 3), np.float32), "dL_dconic": np.zeros((P, 4), np.float32),
            "dL_dopacity": np.zeros((P,), np.float32), "dL_dcolor": np.zeros((P, 3), np.float32),
            "dL_dmean3D": np.zeros((P, 3), np.float32), "dL_dcov3D": np.zeros((P, 6), np.float32),
            "dL_dsh": np.zeros((P, max(M, 1), 3), np.float32), "dL_dflows": np.zeros((P, 2), np.float32),
            "dL_dts": np.zeros((P,), np.float32), "dL_dscale": np.zeros((P, 3), np.float32),
            "dL_dscale_t": np.zeros((P,), np.float32), "dL_drot": np.zeros((P, 4), np.float32),
            "dL_drot_r": np.zeros((P, 4), np.float32),
        }
        for k, a in g.items():
            setattr(self.io, k, _ptr(a, C.c_float))
        rc = self.lib.oracle_backward(C.byref(self.io))
        if rc != 0:
            raise RuntimeError("oracle_backward failed: %d" % rc)
        if M == 0:
            g["dL_dsh"] = np.zeros((P, 0, 3), np.float32)
        return g

    def close(self):
        if not self._freed:
            self.lib.oracle_free(C.byref(self.io))
            self._freed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mark_visible(means3D, viewmatrix, projmatrix, kind="port"):
    lib = _load(kind)
    m, v, p = _np(means3D), _np(viewmatrix), _np(projmatrix)
    out = np.zeros((m.shape[0],), np.uint8)
    lib.oracle_mark_visible(m.shape[0], _ptr(m, C.c_float), _ptr(v, C.c_float), _ptr(p, C.c_float),
                            _ptr(out, C.c_uint8))
    return out.astype(bool)


def set_accumulation(mode, kind="port"):
    """oracle_set_accumulation (oracle_api.h): 0 = the reference's fp32 atomics in index order, 1 = the same atomics with tiles
    and threads in reverse order (another legal execution order of the same kernel), 2 = double accumulation (port only),
    3 = mode 2 with every term perturbed by a relative 1e-6 (port only: a conditioning probe, see oracle/fdgs_oracle.c)."""
    lib = _load(kind)
    lib.oracle_set_accumulation.argtypes = [C.c_int]
    lib.oracle_set_accumulation.restype = C.c_int
    if lib.oracle_set_accumulation(int(mode)) != 0:
        raise ValueError("oracle %r does not support accumulation mode %r" % (kind, mode))


def threads(kind="port"):
    return int(_load(kind).oracle_threads())


def ref_dist2_knn3(points):
    """simple-knn's distCUDA2 through the reference's OWN device code (simple_knn.cu:28-191 compiled verbatim into
    oracle/_ref/liboracle_ref.so, oracle/refbuild/ref_knn.cpp; contraction off): [P,3] float32 -> [P] float32."""
    lib = _load("reference")
    lib.oracle_ref_dist2_knn3.argtypes = [C.c_int, _F, _F]
    lib.oracle_ref_dist2_knn3.restype = C.c_int
    p = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
    assert p.ndim == 2 and p.shape[1] == 3
    out = np.zeros((p.shape[0],), np.float32)   # spatial.cu:22 torch::full({P}, 0.0)
    rc = lib.oracle_ref_dist2_knn3(p.shape[0], _ptr(p, C.c_float), _ptr(out, C.c_float))
    if rc != 0:
        raise RuntimeError("oracle_ref_dist2_knn3 failed: %d" % rc)
    return out


def block_any_pixel_passes(tuples):
    """Port oracle: per tuple [x, y, conic xx, xy, yy, opacity, rx0, rx1, ry0, ry1], does any pixel centre of the rectangle pass
    the forward blend's per-pixel test (forward.cu:585-590) in the oracle's own fp32 arithmetic."""
    lib = _load("port")
    lib.oracle_block_any_pixel_passes.argtypes = [C.c_int, _F, _U8]
    lib.oracle_block_any_pixel_passes.restype = C.c_int
    t = np.ascontiguousarray(np.asarray(tuples, dtype=np.float32))
    assert t.ndim == 2 and t.shape[1] == 10
    out = np.zeros((t.shape[0],), np.uint8)
    lib.oracle_block_any_pixel_passes(t.shape[0], _ptr(t, C.c_float), _ptr(out, C.c_uint8))
    return out.astype(bool)
