"""Seeded synthetic 4D Gaussian scenes + camera.

The reference ships no data; this generator is grounded in its own
initialisation (xyz ~ U[-1.3,1.3]^3 as scene/dataset_readers.py:329, unit
quaternions and scales_t as scene/gaussian_model.py:268-286) and its camera
conventions (utils/graphics_utils.py:38-72, scene/cameras.py:65-71: matrices
are stored transposed, "row-vector" convention, and read column-major by the
kernels).  Everything is generated on the CPU from a seeded generator so the
tensors are identical on every device; callers move them with ``.to(device)``.
"""
import math
from typing import Dict, NamedTuple

import numpy as np
import torch


class SceneConfig(NamedTuple):
    name: str
    P: int
    W: int
    H: int
    sh_degree: int
    sh_degree_t: int
    s0: float          # base spatial scale
    duration: float    # time_duration
    rot_4d: bool
    gaussian_dim: int
    force_sh_3d: bool
    cluster: float = 0.0   # fraction of the Gaussians drawn inside a box that covers 15 % of the image instead of the whole volume


# BASELINE.json configs[0..4]
CONFIGS: Dict[str, SceneConfig] = {
    "C1": SceneConfig("C1", 10_000, 400, 400, 0, 0, 0.03, 1.0, True, 4, True),
    "C2": SceneConfig("C2", 100_000, 800, 800, 3, 0, 0.02, 1.0, True, 4, True),
    "C3": SceneConfig("C3", 300_000, 1352, 1014, 3, 2, 0.015, 10.0, True, 4, False),
    "C5": SceneConfig("C5", 2_000_000, 2704, 2028, 3, 0, 0.006, 1.0, True, 4, True),
    # not a BASELINE config -- a skewed variant of C3: trained scenes (configs/dynerf/*.yaml) concentrate their Gaussians on the
    # subject; 70 % of them in a box that projects onto 15 % of the image: tile lists five times the average there (thousands of
    # entries, some beyond 4096: the 1024-thread instance of the per-tile sort), empty tiles elsewhere
    "C3-clustered": SceneConfig("C3-clustered", 300_000, 1352, 1014, 3, 2, 0.015, 10.0, True, 4, False, 0.7),
    # not a BASELINE config -- a scaling probe: C3 with 4 x the Gaussians on 4 x the image area at the same footprint per
    # Gaussian (focal doubles with W, so s0 halves): every per-tile quantity equals C3's, every launch is 4 x larger --
    # what one launch per stage for 4 views of C3 would look like (tools/probe/README.md)
    "C3x4": SceneConfig("C3x4", 1_200_000, 2704, 2028, 3, 2, 0.0075, 10.0, True, 4, False),
}


def num_sh_coeffs(sh_degree: int, sh_degree_t: int, force_sh_3d: bool, gaussian_dim: int) -> int:
    """Coefficients per Gaussian, scene/gaussian_model.py:222-228 (4D-SH: (D+1)^2 * (D_t+1))."""
    m = (sh_degree + 1) ** 2
    if gaussian_dim == 4 and not force_sh_3d:
        m *= (sh_degree_t + 1)
    return m


def make_camera(W: int, H: int, focal: float = None, cam_z: float = -4.0,
                znear: float = 0.01, zfar: float = 100.0, yaw: float = 0.0, pitch: float = 0.0, roll: float = 0.0,
                shift=(0.0, 0.0, 0.0), principal=None) -> Dict[str, object]:
    """Pinhole camera.  Default: at (0,0,cam_z) looking down +z, no rotation.

    ``yaw`` / ``pitch`` / ``roll`` (radians; camera-to-world R = Rz(roll) Rx(pitch) Ry(yaw), the ``R`` scene/cameras.py:65 receives)
    and ``shift`` (added to the camera centre, world units) give a general pose: the camera is first placed at (0,0,cam_z)+shift and
    then turned about its own centre, so every entry of the view matrix and 12 of the 16 of the full projection are non-zero.
    ``principal`` = (cx, cy) in pixels selects the reference's centre-shift projection (utils/graphics_utils.py:74-91, taken when
    the dataset carries cx / cy: scene/cameras.py:66-67), which fills the third column of the projection.

    Returns the tensors exactly as the reference Camera holds them (scene/cameras.py:65-71): world_view_transform =
    getWorld2View2(R, T).T (utils/graphics_utils.py:39-51), full_proj_transform = world_view_transform @ projection.T and
    camera_center = inverse(world_view_transform)[3, :3] -- the transposes of the mathematical (column-vector) matrices.
    """
    if focal is None:
        focal = 0.9 * W
    tanfovx = W / (2.0 * focal)
    tanfovy = H / (2.0 * focal)
    cy_, sy_ = math.cos(yaw), math.sin(yaw)
    cp_, sp_ = math.cos(pitch), math.sin(pitch)
    cr_, sr_ = math.cos(roll), math.sin(roll)
    Ry = np.array([[cy_, 0.0, sy_], [0.0, 1.0, 0.0], [-sy_, 0.0, cy_]])
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, cp_, -sp_], [0.0, sp_, cp_]])
    Rz = np.array([[cr_, -sr_, 0.0], [sr_, cr_, 0.0], [0.0, 0.0, 1.0]])
    R = Rz @ Rx @ Ry                                      # camera-to-world rotation
    centre = np.array([0.0, 0.0, cam_z]) + np.asarray(shift, dtype=np.float64)
    # world -> view: x_view = R^T (x - centre)  (utils/graphics_utils.py:39-51 with T = -R^T centre)
    Rt = np.eye(4, dtype=np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = -R.T @ centre
    V = torch.tensor(np.float32(Rt))
    Pm = torch.zeros(4, 4)
    if principal is None:
        # perspective (utils/graphics_utils.py:52-72)
        top = tanfovy * znear
        right = tanfovx * znear
        Pm[0, 0] = 2.0 * znear / (2.0 * right)
        Pm[1, 1] = 2.0 * znear / (2.0 * top)
    else:
        # centre-shift perspective (utils/graphics_utils.py:74-91)
        cx, cy = float(principal[0]), float(principal[1])
        top = cy / focal * znear
        bottom = -(H - cy) / focal * znear
        left = -(W - cx) / focal * znear
        right = cx / focal * znear
        Pm[0, 0] = 2.0 * znear / (right - left)
        Pm[1, 1] = 2.0 * znear / (top - bottom)
        Pm[0, 2] = (right + left) / (right - left)
        Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    world_view = V.transpose(0, 1).contiguous()
    proj = Pm.transpose(0, 1).contiguous()
    full = (world_view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    campos = world_view.inverse()[3, :3].contiguous()
    return {
        "image_width": W, "image_height": H,
        "tanfovx": tanfovx, "tanfovy": tanfovy,
        "FoVx": 2.0 * math.atan(tanfovx), "FoVy": 2.0 * math.atan(tanfovy),
        "world_view_transform": world_view, "full_proj_transform": full,
        "camera_center": campos,
    }


# General poses (yaw, pitch, roll in radians, shift in world units, principal-point offset in pixels from the image centre or None)
# used by the parity tests, the golden fixtures and the bench: a DyNeRF rig (configs/dynerf/*.yaml) is ~20 cameras on an arc around
# the subject, each turned towards it; "slant" is steep enough that Gaussians cross the 1.3 tanfov clamp (forward.cu:206-211) and the
# z <= 0.2 cull plane (auxiliary.h:153) at an angle.
POSES: Dict[str, dict] = {
    "axis": dict(),
    "rig0": dict(yaw=0.22, pitch=-0.09, roll=0.05, shift=(-0.9, 0.35, 0.2)),
    "rig1": dict(yaw=-0.17, pitch=0.12, roll=-0.08, shift=(0.7, -0.45, -0.3)),
    "rig2": dict(yaw=0.10, pitch=0.07, roll=0.03, shift=(-0.4, -0.3, 0.1), principal_off=(13.5, -9.25)),
    "rig3": dict(yaw=-0.26, pitch=-0.05, roll=0.11, shift=(1.0, 0.2, 0.4)),
    "slant": dict(yaw=0.55, pitch=0.40, roll=0.3, shift=(-1.1, 0.9, 2.7)),   # centre (-1.1, 0.9, -1.3): at the corner of the volume
}


def camera_for(pose, W: int, H: int) -> Dict[str, object]:
    """make_camera for a POSES name or a pose dict."""
    kw = dict(POSES[pose] if isinstance(pose, str) else pose)
    off = kw.pop("principal_off", None)
    if off is not None:
        kw["principal"] = (0.5 * W + off[0], 0.5 * H + off[1])
    return make_camera(W, H, **kw)


def make_scene(cfg: SceneConfig, seed: int = 0, P: int = None, W: int = None, H: int = None,
               random_flow: bool = False, bg=(0.0, 0.0, 0.0), timestamp_frac: float = 0.5, pose="axis",
               alloc=None, rot_sigma=0.05, st_scale: float = 1.0) -> Dict[str, object]:
    """Post-activation rasterizer inputs for ``cfg`` (CPU float32 tensors).

    ``pose``: a POSES name or a make_camera keyword dict (default: the unrotated on-axis camera).
    ``alloc`` = (D, D_t): allocate the SH coefficients of THAT degree pair (the reference always allocates its maximum,
    scene/gaussian_model.py:65,92,222-228: M = 48 for (3, 2)) while the ACTIVE degrees stay cfg.sh_degree / cfg.sh_degree_t -- the
    state of the first 5000 iterations of every training run (one degree up every 1000 iterations, gaussian_model.py:253-257,
    train.py:93-94).  The coefficients beyond the active ones are non-zero on purpose: the kernels must not read them.

    ``rot_sigma``: how far ``rotations`` / ``rotations_r`` are from the identity quaternion: normalize((1,0,0,0) + rot_sigma * N(0,1)^4)
    (0.05 = SURVEY 8d's generator: within ~6 degrees of identity), or "uniform" = normalize(N(0,1)^4): uniformly distributed unit
    quaternions, i.e. arbitrary orientations of the 3D ellipsoid and arbitrary 4D rotations M_l * M_r -- what a trained model holds
    (scene/gaussian_model.py:191-197 only normalises).  The same number of draws either way: every other tensor of the scene is the same.
    ``st_scale`` multiplies ``scales_t``: a general 4D rotation turns part of the temporal axis into space (cov_t = Sigma[3][3] shrinks to
    ~ scales_t^2 * R[3][3]^2, forward.cu:332), so fewer Gaussians pass the 0.05 temporal cull at the same scales_t.

    Keys mirror GaussianRasterizer.forward's arguments
    (gaussian_renderer/diff_gaussian_rasterization.py:263-267) plus the settings.
    """
    P = cfg.P if P is None else P
    W = cfg.W if W is None else W
    H = cfg.H if H is None else H
    g = torch.Generator(device="cpu").manual_seed(seed)
    dur = cfg.duration

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float32)

    def rand(*s):
        return torch.rand(*s, generator=g, dtype=torch.float32)

    xyz = (rand(P, 3) * 2.0 - 1.0) * 1.3
    if cfg.cluster > 0.0:
        # the first cluster * P Gaussians (then shuffled) sit in a box of sqrt(0.15) of the extent in x and y, off-centre
        nc = int(cfg.cluster * P)
        side = math.sqrt(0.15)
        centre = torch.tensor([0.35, -0.2, 0.0]) * 1.3
        xyz[:nc, 0:2] = centre[0:2] + xyz[:nc, 0:2] * side
        xyz = xyz[torch.randperm(P, generator=g)]
    ts = (rand(P, 1) * 1.2 - 0.1) * dur
    scales = cfg.s0 * torch.exp(0.3 * randn(P, 3))
    scales_t = math.sqrt(0.2) * torch.exp(0.3 * randn(P, 1)) * dur
    if st_scale != 1.0:
        scales_t = scales_t * float(st_scale)
    ident = torch.tensor([1.0, 0.0, 0.0, 0.0])
    if rot_sigma == "uniform":
        rot = torch.nn.functional.normalize(randn(P, 4), dim=1)
        rot_r = torch.nn.functional.normalize(randn(P, 4), dim=1)
    else:
        rot = torch.nn.functional.normalize(ident + float(rot_sigma) * randn(P, 4), dim=1)
        rot_r = torch.nn.functional.normalize(ident + float(rot_sigma) * randn(P, 4), dim=1)
    opacity = torch.sigmoid(randn(P, 1))
    if alloc is None:
        M = num_sh_coeffs(cfg.sh_degree, cfg.sh_degree_t, cfg.force_sh_3d, cfg.gaussian_dim)
    else:
        assert alloc[0] >= cfg.sh_degree and alloc[1] >= cfg.sh_degree_t, "allocated degrees below the active ones"
        M = num_sh_coeffs(alloc[0], alloc[1], cfg.force_sh_3d, cfg.gaussian_dim)
    shs = 0.2 * randn(P, M, 3)
    shs[:, 0, :] = rand(P, 3) * 2.0 - 1.0
    flow = 0.5 * randn(P, 2) if random_flow else torch.zeros(P, 2)
    cam = camera_for(pose, W, H)
    scene = {
        "cfg": cfg, "P": P, "W": W, "H": H, "M": M,
        "means3D": xyz.contiguous(), "ts": ts.contiguous(),
        "scales": scales.contiguous(), "scales_t": scales_t.contiguous(),
        "rotations": rot.contiguous(), "rotations_r": rot_r.contiguous(),
        "opacities": opacity.contiguous(), "shs": shs.contiguous(),
        "flow_2d": flow.contiguous(),
        "bg": torch.tensor(bg, dtype=torch.float32),
        "sh_degree": cfg.sh_degree, "sh_degree_t": cfg.sh_degree_t,
        "max_sh_degree": cfg.sh_degree if alloc is None else int(alloc[0]),
        "max_sh_degree_t": cfg.sh_degree_t if alloc is None else int(alloc[1]),
        "timestamp": float(timestamp_frac * dur), "time_duration": float(dur),
        "rot_4d": cfg.rot_4d, "gaussian_dim": cfg.gaussian_dim, "force_sh_3d": cfg.force_sh_3d,
        "scale_modifier": 1.0, "prefilter_var": -1.0,
    }
    scene.update(cam)
    return scene


def active_sh_coeffs(sh_degree: int, sh_degree_t: int, force_sh_3d: bool, gaussian_dim: int) -> int:
    """How many leading coefficients the kernels READ (and write gradients for) at the ACTIVE degrees: (D+1)^2 for 3D SH; for 4D SH
    the time blocks only exist on top of a complete degree-3 spatial block (forward.cu:142: the time terms sit inside ``deg > 2``),
    16 coefficients each."""
    if gaussian_dim == 4 and not force_sh_3d and sh_degree > 2 and sh_degree_t > 0:
        return 16 * (sh_degree_t + 1)
    return (sh_degree + 1) ** 2


def make_upstream_grads(W: int, H: int, seed: int = 1, scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded upstream gradients for (color, depth, alpha, flow)."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float32) * scale

    return {
        "grad_color": randn(3, H, W), "grad_depth": randn(1, H, W),
        "grad_alpha": randn(1, H, W), "grad_flow": randn(2, H, W),
    }


PER_GAUSSIAN_KEYS = ("means3D", "ts", "scales", "scales_t", "rotations", "rotations_r", "opacities", "shs", "flow_2d")


def morton_order(xyz: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation that sorts points by the Morton (Z-order) code of their position quantised to ``bits`` bits per axis."""
    p = xyz.detach().cpu().double()
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).long().clamp_(0, (1 << bits) - 1)
    code = torch.zeros(p.shape[0], dtype=torch.long)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, stable=True)


def permute_scene(scene: Dict[str, object], perm: torch.Tensor) -> Dict[str, object]:
    """The same scene with its Gaussians stored in another order (every per-Gaussian tensor permuted alike): renders the same
    images; only the memory order -- hence the coherence of everything the kernels do per consecutive Gaussians -- changes."""
    out = dict(scene)
    for k in PER_GAUSSIAN_KEYS:
        if isinstance(out.get(k), torch.Tensor):
            out[k] = out[k][perm].contiguous()
    return out
