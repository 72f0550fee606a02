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
