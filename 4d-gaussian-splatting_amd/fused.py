"""Fused-activation rendering: ``render_raw``.

``render()`` (gaussian_renderer/__init__.py, the drop-in) receives post-activation tensors from the model's
getters, exactly like the reference: exp / sigmoid / normalize run as separate PyTorch kernels forward and
backward, and autograd then accumulates every parameter gradient with one more pass.  ``render_raw`` feeds the
model's RAW parameters (the reference's attribute names ``_xyz, _opacity, _scaling, _rotation, _t, _scaling_t,
_rotation_r`` and ``get_features``) to the same kernels with ``fdgs_scene.raw_params = 1``: the activations of
scene/gaussian_model.py:179-219 are applied inside preprocess and their derivatives inside preprocess-backward.
With ``grad_sink`` (e.g. ``GaussianParams.grad_sink()``) the backward writes each gradient straight into the
caller's buffers -- the slices of the flat data-parallel bucket -- and returns no gradient to autograd for those
inputs, so there is no accumulation pass and no zero_grad.  A sink is OVERWRITTEN by a backward with
``accumulate=False`` (the first view of an optimizer step) and ADDED to with ``accumulate=True`` (the following
views of the same step: the reference sums ``loss / batch_size`` over ``batch_size`` views, train.py:104-166).

Covers the default pipeline (in-kernel covariance and SH, rot_4d or not, no env map); everything else goes
through ``render()``.  Same result dict as ``render()``.
"""
import math

import torch

from .gaussian_renderer.diff_gaussian_rasterization import GaussianRasterizationSettings, _C, _is_given


def _raw_forward_args(rs, means3D, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw, prefilter_var):
    e = torch.Tensor([])
    return (rs.bg, means3D, e, e, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw,
            rs.scale_modifier, e, prefilter_var, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
            rs.image_height, rs.image_width, sh, rs.sh_degree, rs.sh_degree_t, rs.campos, rs.timestamp,
            rs.time_duration, rs.rot_4d, rs.gaussian_dim, rs.force_sh_3d, rs.prefiltered, rs.debug)


def raw_forward(rs, means3D, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw, prefilter_var,
                split_colour=False, preprocessed=None, tile_cull=False, lazy=False, sparse_lists=False, colour_stream=None):
    """Native forward on RAW parameters (fdgs_scene.raw_params = 1); the reference binding's 11-tuple.
    ``preprocessed``: the view's handle from ``raw_preprocess_batch``; ``tile_cull``: fdgs_forward_out.tile_cull; ``lazy``:
    fdgs_forward_out.lazy (num_rendered comes back as -1, the host does not wait); ``sparse_lists``: fdgs_forward_out.sparse_lists; ``colour_stream``: fdgs_forward_out.colour_stream (a torch.cuda.Stream)."""
    args = _raw_forward_args(rs, means3D, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw, prefilter_var)
    return _C.rasterize_gaussians(*args, raw_params=True, split_colour=split_colour, preprocessed=preprocessed, tile_cull=tile_cull, lazy=lazy,
                                  sparse_lists=sparse_lists, colour_stream=colour_stream)


def raw_preprocess_batch(settings, means3D, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw, prefilter_var,
                         tile_cull=False):
    """View-batched preprocess on RAW parameters (fdgs_preprocess_batch): ``settings`` = the views' raster settings (the views of
    one optimizer step share every parameter tensor).  One handle per view for ``raw_forward(..., preprocessed=handle)``."""
    views = [_raw_forward_args(rs, means3D, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw, prefilter_var)
             for rs in settings]
    return _C.preprocess_batch(views, raw_params=True, tile_cull=tile_cull)


def raw_backward(rs, means3D, out_means3D, radii, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw,
                 prefilter_var, geom, R, binb, img, g_color, g_depth, g_alpha, g_flow, sink, accumulate, grad_accum=None, after_sh=None,
                 sh_stage=None, begin_only=False, per_view_outputs=True, geometry_adam=None):
    """Native backward on RAW parameters; gradients go into ``sink`` where given; the binding's 12-tuple.
    ``begin_only``: only the blend backward (``_C.backward_begin``): returns the pending call for ``_C.sh_backward_batch`` /
    ``_C.backward_finish``.  ``per_view_outputs=False``: dL_dcolors / dL_dcov3D / dL_dflows are not written (None in the tuple).
    ``geometry_adam``: see ``_C.rasterize_gaussians_backward`` (the geometry parameters' Adam step inside the geometry backward)."""
    e = torch.Tensor([])
    args = (rs.bg, means3D, out_means3D, radii, e, e, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw,
            rotation_r_raw, rs.scale_modifier, e, prefilter_var, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, g_color, g_depth, g_alpha, g_flow, sh, rs.sh_degree, rs.sh_degree_t, rs.campos,
            rs.timestamp, rs.time_duration, rs.rot_4d, rs.gaussian_dim, rs.force_sh_3d, geom, R, binb, img, rs.debug)
    if begin_only:
        return _C.backward_begin(*args, raw_params=True, grad_out=sink, accumulate=accumulate, grad_accum=grad_accum, sh_stage=sh_stage,
                                 per_view_outputs=per_view_outputs)
    return _C.rasterize_gaussians_backward(*args, raw_params=True, grad_out=sink, accumulate=accumulate, grad_accum=grad_accum,
                                           after_sh=after_sh, sh_stage=sh_stage, per_view_outputs=per_view_outputs, geometry_adam=geometry_adam)


def raw_settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0):
    """GaussianRasterizationSettings + the raw parameter tensors of ``pc`` for the default pipeline."""
    if pipe.compute_cov3D_python or pipe.convert_SHs_python or pipe.env_map_res:
        raise ValueError("render_raw covers the default pipeline only; use render() for the Python covariance / SH / env-map branches")
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        sh_degree_t=pc.active_sh_degree_t, campos=viewpoint_camera.camera_center, timestamp=viewpoint_camera.timestamp,
        time_duration=pc.time_duration[1] - pc.time_duration[0], rot_4d=pc.rot_4d, gaussian_dim=pc.gaussian_dim,
        force_sh_3d=pc.force_sh_3d, prefiltered=False, debug=pipe.debug)
    e = torch.Tensor([])
    is_4d = pc.gaussian_dim == 4
    ts = pc._t if is_4d else e
    scaling_t = pc._scaling_t if is_4d else e
    rotation_r = pc._rotation_r if (is_4d and pc.rot_4d) else e
    prefilter_var = pc.prefilter_var if (is_4d and pc.prefilter_var > 0.0) else -1.0
    return rs, (pc._xyz, pc.get_features, pc._opacity, ts, pc._scaling, scaling_t, pc._rotation, rotation_r, prefilter_var)


class _RasterizeRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw,
                prefilter_var, raster_settings, grad_sink, accumulate, tile_cull=False):
        rs = raster_settings
        (R, color, flow, depth, T, radii, geom, binb, img, covs_com, out_means3D) = raw_forward(
            rs, means3D, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw, rotation_r_raw, prefilter_var, tile_cull=tile_cull)
        ctx.rs, ctx.R, ctx.prefilter_var, ctx.sink, ctx.accumulate = rs, R, prefilter_var, grad_sink, bool(accumulate)
        ctx.save_for_backward(means3D, out_means3D, scaling_raw, rotation_raw, radii, sh, opacity_raw, ts, scaling_t_raw,
                              rotation_r_raw, geom, binb, img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # unused outputs -> None gradients -> colour-only backward
        return color, radii, depth, 1 - T, flow

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha, g_flow):
        rs = ctx.rs
        (means3D, out_means3D, scaling_raw, rotation_raw, radii, sh, opacity_raw, ts, scaling_t_raw, rotation_r_raw,
         geom, binb, img) = ctx.saved_tensors
        sink = ctx.sink
        (d_means2D, _d_colors, d_opacity, d_means3D, _d_cov3D, d_sh, _d_flows, d_ts, d_scales, d_scales_t, d_rot,
         d_rot_r) = raw_backward(rs, means3D, out_means3D, radii, sh, opacity_raw, ts, scaling_raw, scaling_t_raw, rotation_raw,
                                 rotation_r_raw, ctx.prefilter_var, geom, ctx.R, binb, img, g_color, g_depth, g_alpha, g_flow,
                                 sink, ctx.accumulate)

        def ret(name, given, g):
            if not _is_given(given):
                return None
            if sink and sink.get(name) is not None:
                return None  # already written into the caller's buffer
            return g.reshape(given.shape)

        return (ret("dL_dmeans3D", means3D, d_means3D), d_means2D, ret("dL_dsh", sh, d_sh),
                ret("dL_dopacity", opacity_raw, d_opacity), ret("dL_dts", ts, d_ts),
                ret("dL_dscales", scaling_raw, d_scales), ret("dL_dscales_t", scaling_t_raw, d_scales_t),
                ret("dL_drotations", rotation_raw, d_rot), ret("dL_drotations_r", rotation_r_raw, d_rot_r),
                None, None, None, None, None)


def render_raw(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, grad_sink=None, accumulate=False, tile_cull=False):
    """``render()`` with the activations fused into the kernels; see the module docstring.  ``tile_cull``:
    fdgs_forward_out.tile_cull (shorter tile lists, same pixels and gradients)."""
    rs, (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var) = raw_settings(
        viewpoint_camera, pc, pipe, bg_color, scaling_modifier)
    screenspace_points = torch.zeros_like(xyz, requires_grad=True)
    color, radii, depth, alpha, flow = _RasterizeRaw.apply(
        xyz, screenspace_points, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r,
        prefilter_var, rs, grad_sink, accumulate, tile_cull)
    return {"render": color, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
            "depth": depth, "alpha": alpha, "flow": flow}
