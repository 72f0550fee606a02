"""Drop-in for the reference's ``gaussian_renderer`` package: ``render()``.

Same signature, branches and result dict as gaussian_renderer/__init__.py:19-194
of the reference; the rasterizer underneath is the MI355X-native one
(``.diff_gaussian_rasterization`` -> csrc/libfdgs.so).  ``pc`` / ``viewpoint_camera`` /
``pipe`` are duck-typed exactly as in the reference (SURVEY.md section 8b lists the
members that are read).  Unlike the reference, the device is taken from the
model (``pc.get_xyz.device``) instead of the literal "cuda".
"""
import math

import torch
from torch.nn import functional as F

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from ..sh_utils import eval_sh, eval_shfs_4d


def _select(mask, *tensors):
    return tuple(None if t is None else t[mask] for t in tensors)


# Options of the fast path below (process-wide).  tile_cull: fdgs_forward_out.tile_cull -- shorter tile lists, same pixels and
# gradients (off: point_list / ranges / n_contrib are the reference's, bit for bit).
# fast_path: False sends every model through the reference's own sequence (getters -> activations in PyTorch -> rasterizer).
render_options = {"tile_cull": False, "fast_path": True}

_RAW_ATTRS = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


class _RasterizeModel(torch.autograd.Function):
    """The rasterizer on a reference-style model's RAW parameters (fdgs_scene.raw_params: activations inside the kernels).
    With an ``fdgs.optim.Adam`` that owns the parameters (``opt``), the backward writes the parameter gradients straight into the
    optimizer's flat gradient bucket behind ``p.grad`` and returns nothing for them; otherwise it returns them to autograd."""

    @staticmethod
    def forward(ctx, means2D, xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var, rs, opt, tile_cull, lazy):
        from ..fused import raw_forward
        (R, color, flow, depth, T, radii, geom, binb, img, _covs, out_means3D) = raw_forward(
            rs, xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var, tile_cull=tile_cull, lazy=lazy,
            sparse_lists=lazy)   # (a lazy forward never hands its lists out: they may as well sit at fixed offsets -- no count / scan launch)
        ctx.rs, ctx.R, ctx.prefilter_var, ctx.opt = rs, R, prefilter_var, opt
        ctx.save_for_backward(xyz, out_means3D, scaling, rotation, radii, feats, opacity, ts, scaling_t, rotation_r, geom, binb, img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # outputs nobody differentiates arrive as None: colour-only backward
        if opt is not None:
            opt.note_forward(R < 0)
        return color, radii, depth, 1 - T, flow

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha, g_flow):
        from ..fused import raw_backward
        from .diff_gaussian_rasterization import _is_given
        rs, opt = ctx.rs, ctx.opt
        (xyz, out_means3D, scaling, rotation, radii, feats, opacity, ts, scaling_t, rotation_r, geom, binb, img) = ctx.saved_tensors
        if g_color is None and g_depth is None and g_alpha is None and g_flow is None:
            g_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=xyz.device)
        sunk = opt is not None and opt.is_homed() and opt.features().data_ptr() == feats.data_ptr()
        if opt is not None and not sunk:
            # the forward was handed the optimizer's bucket views (plain tensors, no requires_grad): returning gradients for them
            # would be dropped by autograd without a word and no parameter would receive anything
            raise RuntimeError("fdgs render(): the optimizer's parameters or Adam state were replaced between this view's forward and its "
                               "backward (densification / prune / load_state_dict); the gradients of this view have nowhere to go -- "
                               "run backward() before editing the optimizer, as the reference's loop does (train.py:160-249)")
        if sunk:
            sink, accumulate, gacc, stage = opt.backward_begin(rs)
            grads = raw_backward(rs, xyz, out_means3D, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, ctx.prefilter_var,
                                 geom, ctx.R, binb, img, g_color, g_depth, g_alpha, g_flow, sink, accumulate, grad_accum=gacc, sh_stage=stage)
            return (grads[0],) + (None,) * 13
        (d_means2D, _dc, d_opacity, d_means3D, _dcov, d_sh, _df, d_ts, d_scales, d_scales_t, d_rot, d_rot_r) = raw_backward(
            rs, xyz, out_means3D, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, ctx.prefilter_var, geom, ctx.R, binb, img,
            g_color, g_depth, g_alpha, g_flow, None, False)

        def shaped(given, g):
            return g.reshape(given.shape) if _is_given(given) else None

        return (d_means2D, d_means3D, shaped(feats, d_sh), shaped(opacity, d_opacity), shaped(ts, d_ts), shaped(scaling, d_scales),
                shaped(scaling_t, d_scales_t), shaped(rotation, d_rot), shaped(rotation_r, d_rot_r), None, None, None, None, None)


def _fast_path(viewpoint_camera, pc, pipe, raster_settings, means2D):
    """A reference-style model (raw ``_scaling`` / ``_rotation`` / ``_opacity`` / ``_features_dc`` / ``_features_rest`` ... attributes,
    scene/gaussian_model.py:70-80) on the default pipeline: the kernels take the RAW parameters and apply the activations of
    :179-209 themselves (no exp / sigmoid / normalize kernels forward and backward); with ``fdgs.optim.Adam`` as ``pc.optimizer``
    also no ``torch.cat`` of the SH coefficients and no gradient accumulation pass (see fdgs/optim.py).  None: not applicable."""
    if pipe.compute_cov3D_python or pipe.convert_SHs_python or pipe.debug or not render_options["fast_path"]:
        return None
    if not all(isinstance(getattr(pc, a, None), torch.Tensor) for a in _RAW_ATTRS):
        return None
    xyz = pc._xyz
    if not xyz.is_cuda or xyz.dtype != torch.float32 or pc._features_dc.dim() != 3:
        return None
    is_4d = pc.gaussian_dim == 4
    if is_4d and not (isinstance(getattr(pc, "_t", None), torch.Tensor) and isinstance(getattr(pc, "_scaling_t", None), torch.Tensor)):
        return None
    if is_4d and pc.rot_4d and not isinstance(getattr(pc, "_rotation_r", None), torch.Tensor):
        return None
    from ..optim import Adam as _FdgsAdam
    e = torch.Tensor([])
    opt = getattr(pc, "optimizer", None)
    prefilter_var = pc.prefilter_var if (is_4d and pc.prefilter_var > 0.0) else -1.0
    if isinstance(opt, _FdgsAdam) and opt.ensure_homed() and opt._homed["xyz"][0] is xyz:
        tensors = opt.model_tensors(pc.gaussian_dim, pc.rot_4d)
        lazy = opt.lazy_forward
    else:
        opt, lazy = None, False
        tensors = (xyz, pc.get_features, pc._opacity, pc._t if is_4d else e, pc._scaling, pc._scaling_t if is_4d else e, pc._rotation,
                   pc._rotation_r if (is_4d and pc.rot_4d) else e)
    (x, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r) = tensors
    return _RasterizeModel.apply(means2D, x, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, prefilter_var,
                                 raster_settings, opt, bool(render_options["tile_cull"]), lazy)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render the scene seen by ``viewpoint_camera``.  ``bg_color`` must live on the model's device."""
    xyz = pc.get_xyz
    device = xyz.device

    # zero tensor whose .grad receives the screen-space mean gradients (densification statistics)
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color if not pipe.env_map_res else torch.zeros(3, device=device),
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        sh_degree_t=pc.active_sh_degree_t,
        campos=viewpoint_camera.camera_center,
        timestamp=viewpoint_camera.timestamp,
        time_duration=pc.time_duration[1] - pc.time_duration[0],
        rot_4d=pc.rot_4d,
        gaussian_dim=pc.gaussian_dim,
        force_sh_3d=pc.force_sh_3d,
        prefiltered=False,
        debug=pipe.debug,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    fast = _fast_path(viewpoint_camera, pc, pipe, raster_settings, screenspace_points) if override_color is None else None
    if fast is not None:
        rendered_image, radii, depth, alpha, flow = fast
        return _finish(viewpoint_camera, pc, pipe, screenspace_points, rendered_image, radii, depth, alpha, flow, None)

    means3D, means2D, opacity = xyz, screenspace_points, pc.get_opacity
    scales = scales_t = rotations = rotations_r = ts = cov3D_precomp = None
    prefilter_var = -1.0
    marginal_t = None
    is_4d = pc.gaussian_dim == 4

    # covariance: Python-side (pipe.compute_cov3D_python) or inside the preprocess kernel
    if pipe.compute_cov3D_python:
        if pc.rot_4d:
            cov3D_precomp, delta_mean = pc.get_current_covariance_and_mean_offset(scaling_modifier, viewpoint_camera.timestamp)
            means3D = means3D + delta_mean
        else:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        if is_4d:
            marginal_t = pc.get_marginal_t(viewpoint_camera.timestamp)
            opacity = opacity * marginal_t
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
        if is_4d:
            scales_t, ts = pc.get_scaling_t, pc.get_t
            if pc.rot_4d:
                rotations_r = pc.get_rotation_r
            if pc.prefilter_var > 0.0:
                prefilter_var = pc.prefilter_var

    # colour: override > Python SH (pipe.convert_SHs_python) > SH inside the preprocess kernel
    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif pipe.convert_SHs_python:
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).view(-1, 3, pc.get_max_sh_channels)
        cam = viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
        if pipe.compute_cov3D_python:
            dir_pp = (means3D - cam).detach()
        else:
            _, delta_mean = pc.get_current_covariance_and_mean_offset(scaling_modifier, viewpoint_camera.timestamp)
            dir_pp = ((means3D + delta_mean) - cam).detach()
        dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        if pc.gaussian_dim == 3 or pc.force_sh_3d:
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp)
        elif is_4d:
            dir_t = (pc.get_t - viewpoint_camera.timestamp).detach()
            sh2rgb = eval_shfs_4d(pc.active_sh_degree, pc.active_sh_degree_t, shs_view, dir_pp, dir_t,
                                  pc.time_duration[1] - pc.time_duration[0])
        colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
    else:
        shs = pc.get_features
        if is_4d and ts is None:
            ts = pc.get_t

    flow_2d = torch.zeros_like(xyz[:, :2])

    # temporal pre-filter when the marginal was folded into opacity in Python
    mask = None
    if pipe.compute_cov3D_python and is_4d:
        mask = marginal_t[:, 0] > 0.05
        (means2D, means3D, ts, shs, colors_precomp, opacity, scales, scales_t, rotations, rotations_r, cov3D_precomp,
         flow_2d) = _select(mask, means2D, means3D, ts, shs, colors_precomp, opacity, scales, scales_t, rotations,
                            rotations_r, cov3D_precomp, flow_2d)

    rendered_image, radii, depth, alpha, flow, covs_com = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, flow_2d=flow_2d,
        opacities=opacity, ts=ts, scales=scales, scales_t=scales_t, rotations=rotations, rotations_r=rotations_r,
        cov3D_precomp=cov3D_precomp, prefilter_var=prefilter_var)

    return _finish(viewpoint_camera, pc, pipe, screenspace_points, rendered_image, radii, depth, alpha, flow, mask)


def _finish(viewpoint_camera, pc, pipe, screenspace_points, rendered_image, radii, depth, alpha, flow, mask):
    """Environment map behind the Gaussians, radii scattered back through the marginal_t mask, the result dict
    (gaussian_renderer/__init__.py:165-194)."""
    if pipe.env_map_res:
        # composite an environment map behind the Gaussians (sphere of radius 60)
        assert pc.env_map is not None
        sphere_r = 60
        rays_o, rays_d = viewpoint_camera.get_rays()
        od = (rays_o * rays_d).sum(-1)
        dd = (rays_d ** 2).sum(-1)
        delta = od ** 2 - dd * ((rays_o ** 2).sum(-1) - sphere_r ** 2)
        assert (delta > 0).all()
        t_inter = -od + torch.sqrt(delta) / dd
        xyz_inter = rays_o + rays_d * t_inter.unsqueeze(-1)
        tu = torch.atan2(xyz_inter[..., 1:2], xyz_inter[..., 0:1]) / (2 * torch.pi) + 0.5
        tv = torch.acos(xyz_inter[..., 2:3] / sphere_r) / torch.pi
        texcoord = torch.cat([tu, tv], dim=-1) * 2 - 1
        bg_from_envmap = F.grid_sample(pc.env_map[None], texcoord[None])[0]
        rendered_image = rendered_image + (1 - alpha) * bg_from_envmap

    if mask is not None:
        radii_all = radii.new_zeros(mask.shape)
        radii_all[mask] = radii
    else:
        radii_all = radii

    # Gaussians that were culled or had radius 0 were not visible (excluded from densification statistics)
    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii_all > 0,
            "radii": radii_all,
            "depth": depth,
            "alpha": alpha,
            "flow": flow}
