"""Spherical-harmonic / 4D "spherindrical" colour evaluation in PyTorch.

Host-side helpers used only by ``render()``'s ``pipe.convert_SHs_python``
branch (reference gaussian_renderer/__init__.py:98-111, utils/sh_utils.py:58-223).
The default path evaluates SH inside the HIP preprocess kernel; these functions
exist so the drop-in ``render()`` honours the same pipeline flag.  Real SH basis
up to degree 3, optionally multiplied by the first two cosine time harmonics
cos(2 pi k dt / duration), k = 1, 2 (coefficient blocks 16..31 and 32..47).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def sh_basis(deg: int, dirs: torch.Tensor):
    """List of the (deg+1)^2 real SH basis values, each shaped like dirs[..., :1] (or a float for l=0)."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be in 0..3")
    basis = [SH_C0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        basis += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        basis += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        basis += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                  SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
                  SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return basis


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh: [..., C, >=(deg+1)^2], dirs: unit vectors [..., 3] -> [..., C]."""
    out = 0.0
    for k, b in enumerate(sh_basis(deg, dirs)):
        out = out + b * sh[..., k]
    return out


def eval_shfs_4d(deg: int, deg_t: int, sh: torch.Tensor, dirs: torch.Tensor, dirs_t: torch.Tensor, l: float) -> torch.Tensor:
    """4D SH: spatial basis (degree ``deg``) x {1, cos(2 pi dt / l), cos(4 pi dt / l)} truncated at ``deg_t``.

    sh: [..., C, 16 * (deg_t + 1)] when deg == 3; dirs_t: [..., 1] time offsets; l: time duration.
    """
    basis = sh_basis(deg, dirs)
    out = 0.0
    for k, b in enumerate(basis):
        out = out + b * sh[..., k]
    for level in range(1, min(deg_t, 2) + 1):
        if deg < 3:
            raise ValueError("time harmonics are defined on the full degree-3 spatial basis")
        tk = torch.cos(2 * math.pi * level * dirs_t / l)
        for k, b in enumerate(basis):
            out = out + tk * b * sh[..., 16 * level + k]
    return out


def RGB2SH(rgb):
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh):
    return sh * SH_C0 + 0.5
