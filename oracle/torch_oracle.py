"""TEST INFRASTRUCTURE ONLY -- a differentiable PyTorch (CPU) statement of the rasterizer's FORWARD.

Why it exists (BASELINE.json configs[0]): the reference has no PyTorch / CPU path for this operator -- its
``diff_gaussian_rasterization.py`` only wraps the CUDA extension -- so the "PyTorch CPU forward" the baseline file names is stated
here, under ``oracle/`` (never imported by the product: a fallback inside the package would void every parity claim).  It gives
  * the plumbing check of configs[0]: C1 (10 k Gaussians, 400 x 400, SH degree 0) rendered on the CPU without any native code of
    ours on the float path, compared with the C oracle (tests/test_oracle_torch.py);
  * autograd through the forward = the ANALYTIC gradient of what the reference's forward computes: a third opinion, next to the
    C port and the verbatim reference build, on which of the reference's backward formulas are the forward's derivative (all of
    the blend backward, the 3D covariance / SH chain, the 4D-SH chain in ``analytic_sh`` mode) and which are not (Q1-Q3, Q4 / Q5).

What is restated (float64 by default): preprocessCUDA's float path -- 3D covariance (forward.cu:242-276), 4D conditional
covariance, mean shift and temporal marginal (:279-352), EWA projection (:198-237), conic / opacity, SH and 4D-SH colours
(:20-195, Q4: view direction from the un-shifted mean) -- and renderCUDA's blend (:560-626).  What is TAKEN from the C oracle
(discrete, not differentiable): which Gaussians survive the culls and their radii, the per-tile depth-sorted lists
(``point_list`` / ``ranges``) and every pixel's last contributor (``n_contrib``) -- the same lists the reference's blend walks.
"""
import math

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)
REF_PI = 3.14159265   # auxiliary.h:20


def _basis(deg, d):
    """[P,16] SH basis values of the unit directions d [P,3] (forward.cu:87-131); columns beyond the degree are zero."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    cols = [torch.full_like(x, SH_C0)]
    if deg > 0:
        cols += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        cols += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                 SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                 SH_C3[6] * x * (xx - 3 * yy)]
    while len(cols) < 16:
        cols.append(torch.zeros_like(x))
    return torch.stack(cols, dim=1)


def _rot3(q):
    """GLM's (column-major) R of forward.cu:251-262 as a row-major torch matrix: entry [row, col]."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    # glm::mat3(a, b, c, d, ...) fills COLUMN 0 with (a, b, c): R[row][col]
    c0 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], dim=1)
    c1 = torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], dim=1)
    c2 = torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return torch.stack([c0, c1, c2], dim=2)   # [P, row, col]


def _cols4(rows):
    """A GLM mat4 given as its four COLUMNS (each a list of 4 [P] tensors) -> [P, row, col]."""
    return torch.stack([torch.stack(c, dim=1) for c in rows], dim=2)


def preprocess(sc, p):
    """The float path of preprocessCUDA for every Gaussian.  ``sc``: scene dict (settings), ``p``: dict of (possibly
    requires_grad) tensors means3D, opacities, shs | colors_precomp, ts, scales, scales_t, rotations, rotations_r.
    Returns dict(pix [P,2], conic [P,3], opacity [P], rgb [P,3], depth [P], marginal_ok [P] bool)."""
    dt_ = p["means3D"].dtype
    P = p["means3D"].shape[0]
    mod = float(sc.get("scale_modifier", 1.0))
    pv = float(sc.get("prefilter_var", -1.0))
    mean = p["means3D"]
    opacity = p["opacities"].reshape(P)
    ok = torch.ones(P, dtype=torch.bool)
    if sc["rot_4d"]:
        a, b, c, d = p["rotations"].unbind(1)
        pq, q, r, s = p["rotations_r"].unbind(1)
        Ml = _cols4([[a, b, -c, d], [-b, a, d, c], [c, -d, a, b], [-d, -c, -b, a]])       # forward.cu:315-320
        Mr = _cols4([[pq, q, -r, -s], [-q, pq, s, -r], [r, -s, pq, -q], [s, r, q, pq]])   # :321-326
        S = torch.diag_embed(mod * torch.cat([p["scales"], p["scales_t"].reshape(P, 1)], dim=1))
        M = S @ (Mr @ Ml)
        Sigma = M.transpose(1, 2) @ M
        cov_t = Sigma[:, 3, 3]
        dt = float(sc["timestamp"]) - p["ts"].reshape(P)
        marg = torch.exp(-0.5 * dt * dt / (cov_t + pv if pv > 0 else cov_t))
        ok = marg.detach() > 0.05
        opacity = opacity * marg
        c12 = Sigma[:, 0:3, 3]
        cov = Sigma[:, 0:3, 0:3] - c12.unsqueeze(2) * c12.unsqueeze(1) / cov_t.reshape(P, 1, 1)
        mean = mean + c12 / cov_t.reshape(P, 1) * dt.reshape(P, 1)
    else:
        S = torch.diag_embed(mod * p["scales"])
        M = S @ _rot3(p["rotations"])
        cov = M.transpose(1, 2) @ M
        if sc["gaussian_dim"] == 4:
            dt = p["ts"].reshape(P) - float(sc["timestamp"])
            sigma = p["scales_t"].reshape(P) * mod
            marg = torch.exp(-0.5 * dt * dt / (sigma + pv if pv > 0 else sigma))
            ok = marg.detach() > 0.05
            opacity = opacity * marg
    vm = sc["world_view_transform"].to(dt_)    # stored transposed: row-vector convention
    pm = sc["full_proj_transform"].to(dt_)
    hom = torch.cat([mean, torch.ones(P, 1, dtype=dt_)], dim=1)
    t = hom @ vm[:, :3]                         # auxiliary.h:59-67
    ph = hom @ pm
    pw = 1.0 / (ph[:, 3] + 0.0000001)
    W, H = int(sc["W"]), int(sc["H"])
    pix = torch.stack([((ph[:, 0] * pw + 1.0) * W - 1.0) * 0.5, ((ph[:, 1] * pw + 1.0) * H - 1.0) * 0.5], dim=1)
    # EWA (forward.cu:198-237)
    tanx, tany = float(sc["tanfovx"]), float(sc["tanfovy"])
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    tz = t[:, 2]
    tx = torch.clamp(t[:, 0] / tz, -1.3 * tanx, 1.3 * tanx) * tz
    ty = torch.clamp(t[:, 1] / tz, -1.3 * tany, 1.3 * tany) * tz
    z0 = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, z0, -(fx * tx) / (tz * tz)], dim=1),
                     torch.stack([z0, fy / tz, -(fy * ty) / (tz * tz)], dim=1)], dim=1)      # [P, 2, 3]
    Wm = vm[:3, :3].T                                                                         # world -> view rotation
    A = J @ Wm
    c2 = A @ cov @ A.transpose(1, 2)
    cx, cy, cz = c2[:, 0, 0] + 0.3, c2[:, 0, 1], c2[:, 1, 1] + 0.3
    det = cx * cz - cy * cy
    conic = torch.stack([cz / det, -cy / det, cx / det], dim=1)
    # colours
    if p.get("colors_precomp") is not None:
        rgb = p["colors_precomp"]
    else:
        campos = sc["camera_center"].to(dt_)
        dirs = p["means3D"] - campos               # Q4: the UN-shifted mean
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        D, D_t = int(sc["sh_degree"]), int(sc["sh_degree_t"])
        shs = p["shs"]
        sh3d = sc["gaussian_dim"] == 3 or sc["force_sh_3d"]
        bas = _basis(D, dirs)
        n0 = min(16, (D + 1) ** 2)
        rgb = (bas[:, :n0, None] * shs[:, :n0, :]).sum(1)
        if not sh3d and D > 2 and D_t > 0:
            dir_t = p["ts"].reshape(P) - float(sc["timestamp"])
            for k in range(1, min(D_t, 2) + 1):
                tk = torch.cos(2 * REF_PI * dir_t * k / float(sc["time_duration"]))
                rgb = rgb + tk.reshape(P, 1) * (bas[:, :, None] * shs[:, 16 * k:16 * k + 16, :]).sum(1)
        rgb = torch.clamp_min(rgb + 0.5, 0.0)
    return {"pix": pix, "conic": conic, "opacity": opacity, "rgb": rgb, "depth": tz, "marginal_ok": ok}


def blend(sc, pre, lists, flows=None):
    """renderCUDA (forward.cu:560-626) over the tile lists of the C oracle.  ``lists``: dict(point_list, ranges, n_contrib) numpy
    arrays from oracle_forward.  Returns dict(out_color [3,H,W], out_depth [H,W], out_T [H,W], out_flow [2,H,W])."""
    W, H = int(sc["W"]), int(sc["H"])
    dt_ = pre["pix"].dtype
    gx = (W + 15) // 16
    bg = sc["bg"].to(dt_)
    color = torch.zeros(3, H, W, dtype=dt_)
    depth = torch.zeros(H, W, dtype=dt_)
    Tout = torch.ones(H, W, dtype=dt_)
    flow = torch.zeros(2, H, W, dtype=dt_)
    pl = torch.from_numpy(lists["point_list"].astype(np.int64))
    ranges = lists["ranges"].astype(np.int64)
    ncon = torch.from_numpy(lists["n_contrib"].astype(np.int64))
    for t in range(ranges.shape[0]):
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        x0, y0 = (t % gx) * 16, (t // gx) * 16
        x1, y1 = min(x0 + 16, W), min(y0 + 16, H)
        if r1 <= r0:
            continue   # T stays 1, colour = background below
        ids = pl[r0:r1]
        ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt_), torch.arange(x0, x1, dtype=dt_), indexing="ij")
        px, py = xs.reshape(-1, 1), ys.reshape(-1, 1)                              # [n_pix, 1]
        dx = pre["pix"][ids, 0].reshape(1, -1) - px
        dy = pre["pix"][ids, 1].reshape(1, -1) - py
        co = pre["conic"][ids]
        power = -0.5 * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
        alpha = torch.clamp_max(pre["opacity"][ids].reshape(1, -1) * torch.exp(power), 0.99)
        nc = ncon[y0:y1, x0:x1].reshape(-1, 1)
        k = torch.arange(ids.shape[0]).reshape(1, -1)
        use = (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0) & (k < nc)    # forward.cu:585-600 + the T < 1e-4 stop
        a = torch.where(use, alpha, torch.zeros_like(alpha))
        Tbefore = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a[:, :-1]], dim=1), dim=1)
        w = a * Tbefore
        Tfin = Tbefore[:, -1] * (1 - a[:, -1])
        col = w @ pre["rgb"][ids]                                                    # [n_pix, 3]
        hh, ww = y1 - y0, x1 - x0
        color[:, y0:y1, x0:x1] = (col + Tfin.reshape(-1, 1) * bg.reshape(1, 3)).T.reshape(3, hh, ww)
        depth[y0:y1, x0:x1] = (w @ pre["depth"][ids]).reshape(hh, ww)
        Tout[y0:y1, x0:x1] = Tfin.reshape(hh, ww)
        if flows is not None:
            flow[:, y0:y1, x0:x1] = (w @ flows[ids]).T.reshape(2, hh, ww)
    empty = torch.from_numpy((ranges[:, 1] <= ranges[:, 0]))
    if bool(empty.any()):
        for t in torch.nonzero(empty).reshape(-1).tolist():
            x0, y0 = (t % gx) * 16, (t // gx) * 16
            color[:, y0:min(y0 + 16, H), x0:min(x0 + 16, W)] = bg.reshape(3, 1, 1)
    return {"out_color": color, "out_depth": depth, "out_T": Tout, "out_flow": flow}


PARAM_KEYS = ("means3D", "opacities", "shs", "colors_precomp", "ts", "scales", "scales_t", "rotations", "rotations_r", "flow_2d")


def render(scene, lists, dtype=torch.float64, requires_grad=()):
    """The whole forward on the CPU.  ``lists``: the C oracle's forward outputs for the same scene (radii, point_list, ranges,
    n_contrib).  Returns (outputs, params): params[name] are the (leaf) tensors that were fed, for autograd."""
    p = {}
    for k in PARAM_KEYS:
        v = scene.get(k)
        if v is None:
            p[k] = None
            continue
        t = v.detach().to(dtype).clone()
        if k in requires_grad:
            t.requires_grad_(True)
        p[k] = t
    # only the Gaussians the reference keeps (radius > 0) go through the float path: a culled one (behind the camera, det == 0)
    # would put inf / NaN into rows nobody reads -- and 0 x inf = NaN into their gradients
    vis = torch.from_numpy(np.nonzero(lists["radii"] > 0)[0].astype(np.int64))
    remap = torch.full((p["means3D"].shape[0],), -1, dtype=torch.int64)
    remap[vis] = torch.arange(vis.shape[0])
    sub = {k: (None if v is None else v[vis]) for k, v in p.items()}
    pre = preprocess(scene, sub)
    compact = dict(lists)
    compact["point_list"] = remap[torch.from_numpy(lists["point_list"].astype(np.int64))].numpy()
    assert (compact["point_list"] >= 0).all()
    out = blend(scene, pre, compact, flows=sub.get("flow_2d"))
    out["pre"], out["visible"] = pre, vis
    return out, p
