"""ctypes binding of libfdgs.so (C ABI: include/fdgs.h).

This is the thin host glue the north star asks for: PyTorch owns every tensor
and the current HIP stream; the library only borrows raw device pointers.  It
plays the role of the reference's pybind module ``_C``
(diff-gaussian-rasterization/ext.cpp:15-19) and of the tensor allocation code in
rasterize_points.cu:36-149, 151-270.

There is NO fallback: if the shared library has not been built
(``4d-gaussian-splatting_amd/csrc/build.sh`` or ``__graft_entry__.build()``) the
import of this module raises, and every entry point raises if handed CPU tensors.
"""
import ctypes as C
import os

import torch  # must be imported before libfdgs.so so both share one HIP runtime (same SONAME)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDGS_LIB: another build of the same library (A/B timing of kernel variants, tools/ab_build.sh); default: the in-tree build
LIB_PATH = os.environ.get("FDGS_LIB") or os.path.join(_HERE, "csrc", "libfdgs.so")

FDGS_BUF_GEOMETRY, FDGS_BUF_BINNING, FDGS_BUF_IMAGE = 0, 1, 2

_fp = C.c_void_p  # all device pointers travel as void*


FDGS_VERSION = 502  # include/fdgs.h; checked against fdgs_version() at import


class _Sized(C.Structure):
    """Every struct of the ABI starts with ``struct_size`` = sizeof(struct) (include/fdgs.h: the library rejects any other value);
    filled in here so that the call sites keep passing the remaining fields positionally."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class FdgsScene(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("P", C.c_int32), ("D", C.c_int32), ("D_t", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("bg", _fp), ("means3D", _fp), ("shs", _fp), ("colors_precomp", _fp), ("flows", _fp), ("opacities", _fp),
        ("ts", _fp), ("scales", _fp), ("scales_t", _fp), ("rotations", _fp), ("rotations_r", _fp),
        ("cov3D_precomp", _fp), ("viewmatrix", _fp), ("projmatrix", _fp), ("campos", _fp),
        ("scale_modifier", C.c_float), ("prefilter_var", C.c_float),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
        ("timestamp", C.c_float), ("time_duration", C.c_float),
        ("rot_4d", C.c_int32), ("gaussian_dim", C.c_int32), ("force_sh_3d", C.c_int32),
        ("prefiltered", C.c_int32), ("debug", C.c_int32), ("raw_params", C.c_int32), ("analytic_sh_grad", C.c_int32),
    ]


class FdgsForwardOut(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("out_color", _fp), ("out_flow", _fp), ("out_depth", _fp), ("out_T", _fp), ("radii", _fp),
                ("out_means3D", _fp), ("covs_com", _fp), ("preprocessed", C.c_int32), ("split_colour", C.c_int32), ("tile_cull", C.c_int32),
                ("lazy", C.c_int32), ("sparse_lists", C.c_int32), ("colour_stream", C.c_void_p)]


class FdgsBackwardIn(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("dL_dout_color", _fp), ("dL_dout_depth", _fp), ("dL_dout_alpha", _fp), ("dL_dout_flow", _fp),
                ("radii", _fp), ("out_means3D", _fp), ("geom_buffer", _fp), ("binning_buffer", _fp),
                ("image_buffer", _fp), ("num_rendered", C.c_int32)]


class FdgsGeometryAdam(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("flat", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp),
                ("lr_means3D", C.c_float), ("lr_opacities", C.c_float), ("lr_ts", C.c_float), ("lr_scales", C.c_float), ("lr_scales_t", C.c_float),
                ("lr_rotations", C.c_float), ("lr_rotations_r", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int32)]


class FdgsBackwardOut(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("dL_dmeans2D", _fp), ("dL_dcolors", _fp), ("dL_dopacity", _fp), ("dL_dmeans3D", _fp),
                ("dL_dcov3D", _fp), ("dL_dsh", _fp), ("dL_dflows", _fp), ("dL_dts", _fp), ("dL_dscales", _fp),
                ("dL_dscales_t", _fp), ("dL_drotations", _fp), ("dL_drotations_r", _fp), ("accumulate", C.c_int32),
                ("grad_accum", _fp), ("grad_accum_clean", C.c_int32), ("sh_stage", _fp), ("stage_mask", C.c_int32), ("adam", C.POINTER(FdgsGeometryAdam))]


class FdgsDebugView(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("depths", _fp), ("records", _fp), ("cov3D", _fp), ("tiles_touched", _fp), ("clamped", _fp),
                ("point_list", _fp), ("ranges", _fp), ("n_contrib", _fp),
                ("final_T", _fp), ("tile_order", _fp)]


class FdgsAdamSegment(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("lr", C.c_float), ("lr_head", C.c_float),
                ("period", C.c_int32), ("head", C.c_int32)]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)

# every symbol include/fdgs.h declares
EXPORTED = ("fdgs_rasterize_forward", "fdgs_forward_lazy_status", "fdgs_rasterize_backward", "fdgs_preprocess_batch", "fdgs_sh_backward_batch", "fdgs_mark_visible", "fdgs_geometry_bytes",
            "fdgs_image_bytes", "fdgs_binning_bytes", "fdgs_debug_views", "fdgs_debug_activations", "fdgs_debug_tile_sort_limits", "fdgs_debug_block_reaches", "fdgs_debug_run_ahead_stats", "fdgs_set_run_ahead", "fdgs_set_sparse_lists_budget", "fdgs_debug_sparse_lists_stats", "fdgs_debug_clock_sample", "fdgs_sh_flush", "fdgs_profile_enable", "fdgs_profile_sample_every", "fdgs_profile_read",
            "fdgs_profile_reset", "fdgs_stage_name", "fdgs_l1_ssim_forward", "fdgs_l1_ssim_backward", "fdgs_l1_ssim_loss", "fdgs_l1_ssim_loss_batch",
            "fdgs_l1_ssim_num_partials", "fdgs_l1_ssim_value_and_grad", "fdgs_adam_step", "fdgs_adam_step_sh", "fdgs_densify_classify", "fdgs_densify_gather", "fdgs_densify_split", "fdgs_densify_stats_local", "fdgs_densify_stats_apply", "fdgs_knn_scratch_bytes", "fdgs_dist2_knn3", "fdgs_last_error", "fdgs_version")
NUM_STAGES = 11


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfdgs.so not found at %s -- build it with 4d-gaussian-splatting_amd/csrc/build.sh "
            "(or __graft_entry__.build()); there is no CPU / PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.fdgs_rasterize_forward.argtypes = [C.POINTER(FdgsScene), C.POINTER(FdgsForwardOut), ALLOC_FN, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_int32)]
    lib.fdgs_rasterize_forward.restype = C.c_int
    lib.fdgs_forward_lazy_status.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.c_int32, C.POINTER(C.c_int32)]
    lib.fdgs_forward_lazy_status.restype = C.c_int
    lib.fdgs_rasterize_backward.argtypes = [C.POINTER(FdgsScene), C.POINTER(FdgsBackwardIn),
                                            C.POINTER(FdgsBackwardOut), C.c_void_p]
    lib.fdgs_rasterize_backward.restype = C.c_int
    lib.fdgs_preprocess_batch.argtypes = [C.c_int32, C.POINTER(C.POINTER(FdgsScene)), C.POINTER(C.POINTER(FdgsForwardOut)), ALLOC_FN,
                                          C.POINTER(C.c_void_p), C.c_void_p]
    lib.fdgs_preprocess_batch.restype = C.c_int
    lib.fdgs_sh_backward_batch.argtypes = [C.c_int32, C.POINTER(C.POINTER(FdgsScene)), C.POINTER(C.POINTER(FdgsBackwardIn)),
                                           C.POINTER(C.POINTER(FdgsBackwardOut)), C.c_void_p]
    lib.fdgs_sh_backward_batch.restype = C.c_int
    lib.fdgs_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fdgs_mark_visible.restype = C.c_int
    lib.fdgs_geometry_bytes.argtypes = [C.c_int32]
    lib.fdgs_geometry_bytes.restype = C.c_size_t
    lib.fdgs_image_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.fdgs_image_bytes.restype = C.c_size_t
    lib.fdgs_binning_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.fdgs_binning_bytes.restype = C.c_size_t
    lib.fdgs_debug_views.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(FdgsDebugView)]
    lib.fdgs_debug_views.restype = C.c_int
    lib.fdgs_debug_activations.argtypes = [C.c_int32] + [C.c_void_p] * 11
    lib.fdgs_debug_activations.restype = C.c_int
    lib.fdgs_debug_tile_sort_limits.argtypes = [C.c_int32, C.c_int32]
    lib.fdgs_debug_tile_sort_limits.restype = None
    lib.fdgs_debug_block_reaches.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fdgs_debug_block_reaches.restype = C.c_int
    lib.fdgs_debug_run_ahead_stats.argtypes = [C.POINTER(C.c_int64)]
    lib.fdgs_debug_run_ahead_stats.restype = None
    lib.fdgs_set_run_ahead.argtypes = [C.c_int32]
    lib.fdgs_set_run_ahead.restype = None
    lib.fdgs_set_sparse_lists_budget.argtypes = [C.c_int64, C.c_int32]
    lib.fdgs_set_sparse_lists_budget.restype = C.c_int
    lib.fdgs_debug_sparse_lists_stats.argtypes = [C.POINTER(C.c_int64)]
    lib.fdgs_debug_sparse_lists_stats.restype = None
    lib.fdgs_debug_clock_sample.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    lib.fdgs_debug_clock_sample.restype = C.c_int
    lib.fdgs_sh_flush.argtypes = [C.c_int32] * 8 + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.fdgs_sh_flush.restype = C.c_int
    lib.fdgs_profile_enable.argtypes = [C.c_int]
    lib.fdgs_profile_enable.restype = C.c_int
    lib.fdgs_profile_sample_every.argtypes = [C.c_int32]
    lib.fdgs_profile_sample_every.restype = C.c_int
    lib.fdgs_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.fdgs_profile_read.restype = C.c_int
    lib.fdgs_profile_reset.restype = C.c_int
    lib.fdgs_stage_name.argtypes = [C.c_int]
    lib.fdgs_stage_name.restype = C.c_char_p
    lib.fdgs_l1_ssim_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fdgs_l1_ssim_forward.restype = C.c_int
    lib.fdgs_l1_ssim_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    lib.fdgs_l1_ssim_backward.restype = C.c_int
    lib.fdgs_l1_ssim_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
    lib.fdgs_l1_ssim_loss.restype = C.c_int
    lib.fdgs_l1_ssim_loss_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
    lib.fdgs_l1_ssim_loss_batch.restype = C.c_int
    lib.fdgs_l1_ssim_value_and_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p]
    lib.fdgs_l1_ssim_value_and_grad.restype = C.c_int
    lib.fdgs_l1_ssim_num_partials.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.fdgs_l1_ssim_num_partials.restype = C.c_int
    lib.fdgs_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.POINTER(FdgsAdamSegment), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32,
                                   C.c_void_p]
    lib.fdgs_adam_step.restype = C.c_int
    lib.fdgs_adam_step_sh.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 8 + [C.c_void_p] + [C.c_float] * 5 + [C.c_int32, C.c_void_p]
    lib.fdgs_adam_step_sh.restype = C.c_int
    lib.fdgs_densify_classify.argtypes = [C.c_int32] + [C.c_void_p] * 5 + [C.c_float] * 5 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.fdgs_densify_classify.restype = C.c_int
    lib.fdgs_densify_gather.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_int64] + [C.c_void_p] * 9
    lib.fdgs_densify_gather.restype = C.c_int
    lib.fdgs_densify_split.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 14
    lib.fdgs_densify_split.restype = C.c_int
    lib.fdgs_densify_stats_local.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
    lib.fdgs_densify_stats_local.restype = C.c_int
    lib.fdgs_densify_stats_apply.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 5
    lib.fdgs_densify_stats_apply.restype = C.c_int
    lib.fdgs_knn_scratch_bytes.argtypes = [C.c_int32]
    lib.fdgs_knn_scratch_bytes.restype = C.c_size_t
    lib.fdgs_dist2_knn3.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fdgs_dist2_knn3.restype = C.c_int
    lib.fdgs_last_error.restype = C.c_char_p
    lib.fdgs_version.restype = C.c_int
    if lib.fdgs_version() != FDGS_VERSION:
        raise ImportError("%s is version %d, this binding was written for FDGS_VERSION %d (include/fdgs.h): rebuild the library "
                          "(4d-gaussian-splatting_amd/csrc/build.sh)" % (LIB_PATH, lib.fdgs_version(), FDGS_VERSION))
    return lib


lib = _load()


def last_error() -> str:
    return lib.fdgs_last_error().decode()


def _check(rc: int, what: str):
    if rc != 0:
        msg = last_error()
        if rc == 1:
            # argument errors keep the reference's Python-level wording / type
            # (gaussian_renderer/diff_gaussian_rasterization.py:271-280 raise plain Exception)
            raise Exception(msg)
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg))


def _ptr(t):
    """Device pointer of a tensor; None / empty tensor -> NULL (the reference's absent-tensor convention)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _dev_f32(t, name):
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError("fdgs: tensor '%s' must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def debug_activations(opacity_raw=None, scales_raw=None, scales_t_raw=None, rotations_raw=None, rotations_r_raw=None):
    """The activated tensors the kernels derive from raw parameters (fdgs_debug_activations): bit-identical to what
    preprocess computes in flight with fdgs_scene.raw_params = 1.  Returns a 5-tuple (None where the input was None)."""
    ins = [opacity_raw, scales_raw, scales_t_raw, rotations_raw, rotations_r_raw]
    ref = next(t for t in ins if t is not None)
    ins = [_dev_f32(t, "raw parameter") for t in ins]
    outs = [None if t is None else torch.empty_like(t) for t in ins]
    P = int(ref.shape[0])
    with torch.cuda.device(ref.device):
        rc = lib.fdgs_debug_activations(P, *[_ptr(t) for t in ins], *[_ptr(t) for t in outs], current_stream_handle(ref.device))
    _check(rc, "fdgs_debug_activations")
    return tuple(outs)


def sh_flush(stages, dL_dsh, sh_degree, sh_degree_t, gaussian_dim, force_sh_3d, analytic_sh_grad=False, accumulate=False):
    """dL_dsh from the staged views of the deferred SH backward (fdgs_sh_flush); ``stages``: [num_views, P, 8] float32
    tensor whose slices stages[v] were the ``sh_stage`` of the views' backward calls; ``analytic_sh_grad``: the mode those
    calls ran in (set_analytic_sh_gradients)."""
    P, M = int(dL_dsh.shape[0]), int(dL_dsh.shape[1])
    if stages.dim() != 3 or stages.shape[1] != P or stages.shape[2] != 8 or not stages.is_contiguous() or stages.dtype != torch.float32:
        raise RuntimeError("fdgs: stages must be a contiguous float32 tensor [num_views, %d, 8]" % P)
    with torch.cuda.device(dL_dsh.device):
        rc = lib.fdgs_sh_flush(P, int(sh_degree), int(sh_degree_t), M, int(gaussian_dim), int(bool(force_sh_3d)),
                               int(bool(analytic_sh_grad)), int(stages.shape[0]), stages.data_ptr(), dL_dsh.data_ptr(), int(bool(accumulate)),
                               current_stream_handle(dL_dsh.device))
    _check(rc, "fdgs_sh_flush")


def adam_step_sh(params, exp_avg, exp_avg_sq, stages, sh_degree, sh_degree_t, gaussian_dim, force_sh_3d, analytic_sh_grad,
                 lr, lr_dc, beta1, beta2, eps, step, dL_dsh=None) -> bool:
    """Adam over the SH coefficients ``params`` [P, M, 3] with the gradient built from the staged views (fdgs_adam_step_sh).
    Returns False -- nothing done -- if the shape / alignment is not supported (the caller then uses sh_flush + the plain step)."""
    P, M = int(params.shape[0]), int(params.shape[1])
    if stages.dim() != 3 or stages.shape[1] != P or stages.shape[2] != 8 or not stages.is_contiguous() or stages.dtype != torch.float32:
        raise RuntimeError("fdgs: stages must be a contiguous float32 tensor [num_views, %d, 8]" % P)
    ptrs = [params.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr()] + ([dL_dsh.data_ptr()] if dL_dsh is not None else [])
    if (3 * M) % 4 != 0 or any(q % 16 for q in ptrs):
        return False
    with torch.cuda.device(params.device):
        rc = lib.fdgs_adam_step_sh(params.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                   dL_dsh.data_ptr() if dL_dsh is not None else None, P, int(sh_degree), int(sh_degree_t), M,
                                   int(gaussian_dim), int(bool(force_sh_3d)), int(bool(analytic_sh_grad)), int(stages.shape[0]),
                                   stages.data_ptr(), float(lr), float(lr_dc), float(beta1), float(beta2), float(eps), int(step),
                                   current_stream_handle(params.device))
    _check(rc, "fdgs_adam_step_sh")
    return True


def forward_lazy_status(device, wait=True):
    """fdgs_forward_lazy_status for the calling thread on ``device``: (pending, failed, [num_rendered of the reported forwards])."""
    pend, failed, n = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rs = (C.c_int32 * 64)()
    with torch.cuda.device(device):
        rc = lib.fdgs_forward_lazy_status(int(bool(wait)), current_stream_handle(device), C.byref(pend), C.byref(failed), rs, 64, C.byref(n))
    _check(rc, "fdgs_forward_lazy_status")
    return int(pend.value), int(failed.value), [int(rs[i]) for i in range(n.value)]


def run_ahead_stats():
    """(kept, sorted again, exact) counts of this process's forward calls (fdgs_debug_run_ahead_stats)."""
    a = (C.c_int64 * 3)()
    lib.fdgs_debug_run_ahead_stats(a)
    return tuple(int(x) for x in a)


def sparse_lists_stats():
    """(forwards with sparse lists, forwards that asked for them and kept compact lists because of the byte budget, bytes of the last
    run-ahead binning buffer) -- fdgs_debug_sparse_lists_stats."""
    a = (C.c_int64 * 3)()
    lib.fdgs_debug_sparse_lists_stats(a)
    return tuple(int(x) for x in a)


class ClockSample:
    """Shader clock sustained over an interval (fdgs_debug_clock_sample): start() enqueues the one-wave sampler on its own stream,
    ghz() waits for it and returns the measured clock in GHz."""

    def __init__(self, device):
        self.dev = device
        self.stream = torch.cuda.Stream(device)
        self.out = torch.zeros(5, dtype=torch.int64, device=device)

    def start(self, span_ms: float):
        with torch.cuda.device(self.dev):
            rc = lib.fdgs_debug_clock_sample(self.out.data_ptr(), float(span_ms), C.c_void_p(self.stream.cuda_stream))
        _check(rc, "fdgs_debug_clock_sample")

    def ghz(self):
        self.stream.synchronize()
        w0, s0, w1, s1, khz = [int(x) for x in self.out.cpu().tolist()]
        if w1 <= w0 or khz <= 0:
            return None
        return (s1 - s0) / (w1 - w0) * khz * 1e-6


class ClockPair:
    """The same measurement without a stream of its own (a step that uses four streams leaves no hardware queue for a sampler that
    sits in one for the whole interval): two short samples on the CALLER's stream, mark() where the interval starts and mark() where it
    ends; s_memtime is one chip-wide counter, so the shader cycles between the first sample's start and the second one's end over the
    constant-rate ticks between them is the clock sustained in between (bench.py checks it against ClockSample: FDGS_BENCH_CLOCK=both)."""

    def __init__(self, device):
        self.dev = device
        self.out = torch.zeros((2, 5), dtype=torch.int64, device=device)
        self.n = 0

    def mark(self):
        with torch.cuda.device(self.dev):
            rc = lib.fdgs_debug_clock_sample(self.out[self.n].data_ptr(), 0.002, current_stream_handle(self.dev))
        _check(rc, "fdgs_debug_clock_sample")
        self.n += 1

    def ghz(self):
        if self.n != 2:
            return None
        torch.cuda.synchronize(self.dev)
        (w0, s0, _w, _s, khz), (_w2, _s2, w1, s1, _k) = [[int(x) for x in row] for row in self.out.cpu().tolist()]
        if w1 <= w0 or khz <= 0:
            return None
        return (s1 - s0) / (w1 - w0) * khz * 1e-6


def current_stream_handle(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def profile_enable(on: bool = True, stages=None, every: int = 1):
    """Bracket stages with HIP events on the caller's stream (fdgs_profile_enable).  ``stages``: iterable of stage
    names to restrict the events to; ``every``: only every n-th launch of a stage gets its event pair (fdgs_profile_sample_every: a
    pair costs the stream ~13 us of idle time around the launch)."""
    lib.fdgs_profile_sample_every(max(1, int(every)))
    if not on:
        mask = 0
    elif stages is None:
        mask = -1
    else:
        names = [lib.fdgs_stage_name(i).decode() for i in range(NUM_STAGES)]
        mask = 0
        for s in stages:
            mask |= 1 << names.index(s)
    lib.fdgs_profile_enable(mask)


def profile_reset():
    lib.fdgs_profile_reset()


def profile_read():
    """{stage name: (total_ms, samples)} accumulated since the last reset; synchronises pending events."""
    out = {}
    for st in range(NUM_STAGES):
        ms, n = C.c_double(0.0), C.c_int64(0)
        lib.fdgs_profile_read(st, C.byref(ms), C.byref(n))
        out[lib.fdgs_stage_name(st).decode()] = (ms.value, int(n.value))
    return out
