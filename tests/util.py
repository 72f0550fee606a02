"""Shared helpers for the parity tests: run the HIP product through its C ABI, run an oracle, compare."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fdgs import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

PIX_TOL = 1e-4     # north_star: pixels within 1e-4 abs
GRAD_TOL = 1e-4    # gradients: 1e-4 abs for O(1) tensors, else 1e-4 of the tensor's max magnitude
GRAD_SCALE = 1e-2  # upstream-gradient scale used by the tests (keeps most gradient tensors O(1..10))


def scene_to_device(scene, device):
    out = {}
    for k, v in scene.items():
        out[k] = v.to(device) if isinstance(v, torch.Tensor) else v
    return out


def native_args_fwd(sc):
    """The 30 positional arguments of _C.rasterize_gaussians from a scene dict (device tensors)."""
    e = torch.Tensor([])
    g = lambda k: sc[k] if sc.get(k) is not None else e  # noqa: E731
    return (sc["bg"], sc["means3D"], g("colors_precomp"), g("flow_2d"), sc["opacities"], g("ts"), g("scales"),
            g("scales_t"), g("rotations"), g("rotations_r"), sc.get("scale_modifier", 1.0), g("cov3D_precomp"),
            sc.get("prefilter_var", -1.0), sc["world_view_transform"], sc["full_proj_transform"], sc["tanfovx"],
            sc["tanfovy"], sc["H"], sc["W"], g("shs"), sc["sh_degree"], sc["sh_degree_t"], sc["camera_center"],
            sc["timestamp"], sc["time_duration"], sc["rot_4d"], sc["gaussian_dim"], sc["force_sh_3d"], False,
            sc.get("debug", False))


def _view(buf, ptr, count, dtype):
    """Slice of an opaque scratch tensor starting at device pointer ``ptr``."""
    if ptr is None or count == 0:
        return torch.empty(0, dtype=dtype)
    off = ptr - buf.data_ptr()
    nbytes = count * torch.empty(0, dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype)


def collect_forward(res, P, W, H):
    """numpy dict of every forward output + the introspection views of the opaque buffers (fdgs_debug_views)
    from the 11-tuple the native forward returns."""
    from fdgs import _capi
    (R, color, flow, depth, T, radii, geom, binb, img, covs_com, out_means3D) = res
    v = _capi.FdgsDebugView()
    rc = _capi.lib.fdgs_debug_views(P, W, H, R, _capi._ptr(geom), _capi._ptr(binb), _capi._ptr(img), C.byref(v))
    assert rc == 0, _capi.last_error()
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    torch.cuda.synchronize()
    rec = _view(geom, v.records, P * 12, torch.float32).reshape(P, 12).cpu().numpy()
    out = {
        "R": R,
        "out_color": color.cpu().numpy(), "out_flow": flow.cpu().numpy(), "out_depth": depth.cpu().numpy()[0],
        "out_T": T.cpu().numpy()[0], "radii": radii.cpu().numpy(), "out_means3D": out_means3D.cpu().numpy(),
        "covs_com": covs_com.cpu().numpy(),
        "depths": _view(geom, v.depths, P, torch.float32).cpu().numpy(),
        "cov3D": _view(geom, v.cov3D, P * 6, torch.float32).reshape(P, 6).cpu().numpy(),
        "tiles_touched": _view(geom, v.tiles_touched, P, torch.int32).cpu().numpy().astype(np.uint32),
        "clamped_bits": _view(geom, v.clamped, P, torch.uint8).cpu().numpy(),
        "point_list": _view(binb, v.point_list, R, torch.int32).cpu().numpy().astype(np.uint32),
        "ranges": _view(img, v.ranges, ntiles * 2, torch.int32).reshape(ntiles, 2).cpu().numpy().astype(np.uint32),
        "n_contrib": _view(img, v.n_contrib, W * H, torch.int32).reshape(H, W).cpu().numpy().astype(np.uint32),
        "final_T": _view(img, v.final_T, W * H, torch.float32).reshape(H, W).cpu().numpy(),
        "tile_order": _view(img, v.tile_order, ntiles, torch.int32).cpu().numpy().astype(np.int64),
        "means2D": rec[:, 0:2].copy(),
        "conic_opacity": np.concatenate([rec[:, 2:5], rec[:, 5:6]], axis=1),
        "rgb": rec[:, 6:9].copy(), "rec_depth": rec[:, 9].copy(), "rec_flow": rec[:, 10:12].copy(),
    }
    out["clamped"] = np.stack([(out["clamped_bits"] >> i) & 1 for i in range(3)], axis=1).astype(np.uint8)
    # the tile of sorted instance s is the tile whose range contains s (check_forward compares the ranges themselves too)
    rg = out["ranges"].astype(np.int64)
    out["tile_keys"] = np.repeat(np.arange(ntiles, dtype=np.uint32), (rg[:, 1] - rg[:, 0]))
    return out


def run_hip(scene, device, grads=None, tile_cull=False):
    """Forward (+ optional backward) of the HIP product through the C ABI; returns numpy dicts.
    ``tile_cull``: fdgs_forward_out.tile_cull (shorter tile lists, same pixels and gradients)."""
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import _C
    sc = scene_to_device(scene, device)
    res = _C.rasterize_gaussians(*native_args_fwd(sc), tile_cull=tile_cull)
    (R, color, flow, depth, T, radii, geom, binb, img, covs_com, out_means3D) = res
    P, W, H = int(sc["means3D"].shape[0]), int(sc["W"]), int(sc["H"])
    out = collect_forward(res, P, W, H)
    gout = None
    if grads is not None:
        e = torch.Tensor([])
        g = lambda k: sc[k] if sc.get(k) is not None else e  # noqa: E731
        gd = {k: (t.to(device) if t is not None else None) for k, t in grads.items()}  # None = no upstream gradient
        bargs = (sc["bg"], sc["means3D"], out_means3D, radii, g("colors_precomp"), g("flow_2d"), sc["opacities"],
                 g("ts"), g("scales"), g("scales_t"), g("rotations"), g("rotations_r"), sc.get("scale_modifier", 1.0),
                 g("cov3D_precomp"), sc.get("prefilter_var", -1.0), sc["world_view_transform"],
                 sc["full_proj_transform"], sc["tanfovx"], sc["tanfovy"], gd["grad_color"], gd["grad_depth"],
                 gd["grad_alpha"], gd["grad_flow"], g("shs"), sc["sh_degree"], sc["sh_degree_t"], sc["camera_center"],
                 sc["timestamp"], sc["time_duration"], sc["rot_4d"], sc["gaussian_dim"], sc["force_sh_3d"], geom, R,
                 binb, img, sc.get("debug", False))
        names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dflows", "dL_dts",
                 "dL_dscale", "dL_dscale_t", "dL_drot", "dL_drot_r")
        res = _C.rasterize_gaussians_backward(*bargs)
        torch.cuda.synchronize()
        gout = {n: t.cpu().numpy() for n, t in zip(names, res)}
        for n in ("dL_dopacity", "dL_dts", "dL_dscale_t"):
            gout[n] = gout[n].reshape(-1)
    return out, gout


def run_oracle(scene, grads=None, kind="port"):
    o = pyoracle.Oracle(scene, kind=kind)
    out = dict(o.forward())
    out["R"] = o.R
    gout = None
    if grads is not None:
        # the kernels receive d loss / d alpha unchanged as dL_dmask (reference diff_gaussian_rasterization.py:176)
        gout = dict(o.backward(grads["grad_color"], grads["grad_depth"], grads["grad_alpha"], grads["grad_flow"]))
    o.close()
    return out, gout


def check_culled_lists(hip, ref, W, H, label=""):
    """fdgs_forward_out.tile_cull: the tile lists against the reference's (``ref`` = the oracle's forward).
    * every tile's list is the reference's list of that tile with instances taken out, order kept;
    * every (Gaussian, tile) instance taken out fails the forward blend's per-pixel test (alpha >= 1/255, forward.cu:585-590)
      on EVERY pixel of the tile, in the oracle's own fp32 arithmetic (oracle_block_any_pixel_passes);
    * n_contrib, a list position, points at the same instance as the reference's (checked off the cliff pixels).
    Returns (instances kept, instances of the reference)."""
    gx = (W + 15) // 16
    ref_tile = (ref["keys_sorted"] >> np.uint64(32)).astype(np.int64)
    ref_key = (ref_tile << 32) | ref["point_list"].astype(np.int64)          # (tile, id): unique
    hip_key = (hip["tile_keys"].astype(np.int64) << 32) | hip["point_list"].astype(np.int64)
    order = np.argsort(ref_key, kind="stable")
    srt = ref_key[order]
    at = np.searchsorted(srt, hip_key)
    assert (at < srt.size).all() and np.array_equal(srt[np.minimum(at, srt.size - 1)], hip_key), label + ": a listed instance is not in the reference's list"
    pos = order[at]                                                          # position of every kept instance in the reference's array
    assert (np.diff(pos) > 0).all(), label + ": the kept instances are not in the reference's order"
    kept = np.zeros(ref_key.size, bool)
    kept[pos] = True
    drop = np.nonzero(~kept)[0]
    if drop.size:
        g = ref["point_list"][drop].astype(np.int64)
        t = ref_tile[drop]
        tx, ty = t % gx, t // gx
        tup = np.empty((drop.size, 10), np.float32)
        tup[:, 0:2] = ref["means2D"][g]
        tup[:, 2:6] = ref["conic_opacity"][g]
        tup[:, 6] = tx * 16; tup[:, 7] = np.minimum(tx * 16 + 15, W - 1)
        tup[:, 8] = ty * 16; tup[:, 9] = np.minimum(ty * 16 + 15, H - 1)
        passes = pyoracle.block_any_pixel_passes(tup)
        assert not passes.any(), "%s: %d of the %d instances left out reach alpha >= 1/255 on some pixel of their tile (first: Gaussian %d, tile %d)" % (
            label, int(passes.sum()), drop.size, int(g[np.argmax(passes)]), int(t[np.argmax(passes)]))
    # n_contrib: position k in the culled list <-> position of that instance in the reference's tile list + 1
    ok = ~ref["border"].astype(bool)
    yy, xx = np.nonzero(ok)
    tile = (yy // 16) * gx + xx // 16
    k = hip["n_contrib"][yy, xx].astype(np.int64)
    want = ref["n_contrib"][yy, xx].astype(np.int64)
    has = k > 0
    assert np.array_equal(want[~has], np.zeros((~has).sum(), np.int64)), label + ": n_contrib 0 where the reference has contributors"
    inst = hip["ranges"][tile[has], 0].astype(np.int64) + k[has] - 1
    assert (inst < hip["ranges"][tile[has], 1]).all(), label + ": n_contrib beyond the tile's list"
    got = pos[inst] - ref["ranges"][tile[has], 0].astype(np.int64) + 1
    bad = int((got != want[has]).sum())
    assert bad == 0, "%s: n_contrib points at another instance than the reference's on %d non-cliff pixels" % (label, bad)
    return int(hip_key.size), int(ref_key.size)


def check_forward(hip, ref, label="", precomp_cov=False, precomp_colors=False, max_border=1e-3, tile_cull=False, WH=None, pix_rel=False):
    """Bit-exact integer / key indexing, 1e-4 pixels (away from flagged threshold cliffs). Returns a report dict.
    ``tile_cull``: the forward ran with fdgs_forward_out.tile_cull -- the lists are checked by check_culled_lists (WH = (W, H))
    instead of bit for bit.  ``pix_rel``: the pixel bar is 1e-4 * max(1, max|ref|) per output (scenes whose depth image is far from O(1))."""
    rep = {}
    # Gaussians whose temporal marginal sits within 1e-5 (relative) of the 0.05 cull threshold: the two expf
    # implementations may decide differently.  Everything below assumes they did not -- say so if they did.
    bg = ref["border_g"].astype(bool)
    rep["cull_cliff_gaussians"] = int(bg.sum())
    if bg.any() and not np.array_equal(hip["radii"][bg] > 0, ref["radii"][bg] > 0):
        raise AssertionError("%s: %d Gaussians sit on the temporal-cull cliff and the kernel decided differently from "
                             "the oracle; pick another seed / timestamp" % (label, int(bg.sum())))
    vis = ref["radii"] > 0
    np.testing.assert_array_equal(hip["radii"], ref["radii"], err_msg=label + " radii")
    np.testing.assert_array_equal(hip["tiles_touched"], ref["tiles_touched"], err_msg=label + " tiles_touched")
    np.testing.assert_array_equal(hip["depths"][vis].view(np.uint32), ref["depths"][vis].view(np.uint32),
                                  err_msg=label + " depth bits")
    np.testing.assert_array_equal(hip["means2D"][vis].view(np.uint32), ref["means2D"][vis].view(np.uint32),
                                  err_msg=label + " means2D bits")
    if tile_cull:
        rep["instances"] = "%d of %d" % check_culled_lists(hip, ref, WH[0], WH[1], label)
    else:
        assert hip["R"] == ref["R"], "%s: num_rendered %d vs %d" % (label, hip["R"], ref["R"])
        np.testing.assert_array_equal(hip["point_list"], ref["point_list"], err_msg=label + " point_list")
        np.testing.assert_array_equal(hip["tile_keys"], (ref["keys_sorted"] >> np.uint64(32)).astype(np.uint32),
                                      err_msg=label + " sorted tile ids")
        np.testing.assert_array_equal(hip["ranges"], ref["ranges"], err_msg=label + " ranges")
    float_checks = [("out_means3D", 0.0), ("conic_opacity", 1e-6)]
    if precomp_colors is False:
        float_checks.append(("rgb", 2e-6))  # with colors_precomp the reference never writes its rgb scratch
    if precomp_cov is False:
        float_checks.append(("cov3D", 0.0))  # with cov3D_precomp the reference never writes its cov3D scratch
    for k, tol in float_checks:
        a, b = hip[k][vis], ref[k][vis]
        err = float(np.abs(a - b).max()) if a.size else 0.0
        rep[k] = err
        assert err <= tol * max(1.0, float(np.abs(b).max()) if b.size else 1.0), "%s: %s max err %g" % (label, k, err)
    if precomp_colors is False:
        np.testing.assert_array_equal(hip["clamped"][vis], ref["clamped"][vis], err_msg=label + " clamped")
    border = ref["border"].astype(bool)
    rep["border_frac"] = float(border.mean())
    bound = max(max_border, 3.0 / border.size)   # tiny images: allow a handful of pixels
    assert rep["border_frac"] < bound, "%s: too many cliff pixels %g (bound %g)" % (label, rep["border_frac"], bound)
    ok = ~border
    if not tile_cull:   # (with tile_cull: a position in the shorter list, checked by check_culled_lists)
        nc_diff = int((hip["n_contrib"][ok] != ref["n_contrib"][ok]).sum())
        rep["n_contrib_diff_nonborder"] = nc_diff
        assert nc_diff == 0, "%s: n_contrib differs on %d non-cliff pixels" % (label, nc_diff)
    for k in ("out_color", "out_flow", "out_depth", "out_T"):
        a, b = hip[k], ref[k]
        d = np.abs(a - b)
        d = d[:, ok] if d.ndim == 3 else d[ok]
        rep[k] = float(d.max())
        tol = PIX_TOL * (max(1.0, float(np.abs(b).max())) if pix_rel else 1.0)
        assert rep[k] <= tol, "%s: %s max abs err %g > %g" % (label, k, rep[k], tol)
    np.testing.assert_array_equal(hip["final_T"], hip["out_T"], err_msg=label + " final_T copy")
    return rep


def check_backward(hipg, refg, label="", tol=GRAD_TOL):
    rep = {}
    for k, b in refg.items():
        if k == "dL_dconic":
            continue
        a = hipg[k].reshape(b.shape)
        scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
        err = float(np.abs(a - b).max()) if b.size else 0.0
        rep[k] = (err, scale)
        assert np.isfinite(a).all(), "%s: %s has non-finite values" % (label, k)
        assert err <= tol * scale, "%s: %s max abs err %g > %g (max|ref| %g)" % (label, k, err, tol * scale, scale)
    return rep


# ----------------------------------------------------------------------------------------------------------------------------
# Gradients behind the per-Gaussian chain, beyond the plain bar: a bar that knows each Gaussian's conditioning (round 6)
# ----------------------------------------------------------------------------------------------------------------------------
# dL/d(scale, scale_t, rotation, rotation_r) -- and, for splats hundreds of pixels wide, dL/d(cov3D, mean3D, t) -- are cancelling sums
# of the blend backward's per-Gaussian sums (thousands of fp32 atomics per Gaussian, in whatever order the hardware issues them), which
# the covariance chain amplifies by two to three orders of magnitude in ANY implementation, the reference included.  What that
# amplification is for a given Gaussian can be read off the reference itself, deterministically: the oracle evaluates the same backward
# with its atomics in index order (mode 0), in the opposite order (mode 1), with the per-Gaussian sums accumulated in double (mode 2,
# "f64": the reference's arithmetic without accumulation error) and with every term of those sums perturbed by what two correct fp32
# evaluations of it differ by (mode 3, the conditioning probe: 2 ulp of the largest term of the exponent -- 1e-4 of G for a needle
# hundreds of pixels long -- and 1e-6 of everything else; oracle/fdgs_oracle.c).  nu_g = the largest deviation from f64, relative to
# its tensor's scale, that any of the three shows on ANY chain element of Gaussian g.
# The bar for an element of Gaussian g:   |hip - ref| <= 1e-4 * scale   (the plain bar: every well-conditioned Gaussian)
#                              else   |hip - f64| <= 1e-4 * scale + NOISE_K * scale * nu_g.
# The oracle side is deterministic; the HIP side is one draw of ITS order noise, NOISE_K is the margin for that (the largest ratio
# observed over repeated backward runs is printed by the tests; BASELINE.md section 5 records it).  A wrong term in a kernel shows on
# the thousands of well-conditioned Gaussians (nu_g < 1e-5), where the bar is the plain one.
CHAIN_ACTIVATED = ("dL_dscale", "dL_dscale_t", "dL_drot", "dL_drot_r")
CHAIN_ACTIVATED_WIDE = CHAIN_ACTIVATED + ("dL_dcov3D", "dL_dmean3D", "dL_dts")
NOISE_K = 8.0


def gaussian_noise_scale(orders, f64, names):
    """nu_g (see above): [P] float64.  ``orders``: the reference's gradients in its two accumulation orders (dicts), ``f64``: with
    double-accumulated sums, ``names``: the chain tensors (per-Gaussian leading dimension)."""
    nu = None
    for n in names:
        x64 = np.asarray(f64[n], np.float64)
        P = x64.shape[0]
        scale = max(1.0, float(np.abs(orders[0][n]).max()) if x64.size else 1.0)
        for o in orders:
            d = np.abs(np.asarray(o[n], np.float64).reshape(P, -1) - x64.reshape(P, -1)).max(1) / scale
            nu = d if nu is None else np.maximum(nu, d)
    return nu


def check_backward_noise_aware(hipg, refg, refg_rev, refg_f64, refg_probe, label="", chain=CHAIN_ACTIVATED, tol=GRAD_TOL, K=NOISE_K, scale_like=None):
    """check_backward with the conditioning-aware bar for the tensors in ``chain`` (every other tensor: the plain bar, no exception).
    ``scale_like``: {tensor: other tensor} -- a tensor that is a difference of terms of another tensor's magnitude is held to that
    tensor's scale if larger (dL_dscale_t <- dL_drot, as tests/test_oracle_pin.py and the golden test do).
    Returns {tensor: (how it passed, max |hip - ref|, scale, elements beyond the plain bar, worst |hip - f64| / bound)}."""
    chain = tuple(n for n in chain if n in refg)
    nu = gaussian_noise_scale([refg, refg_rev, refg_probe], refg_f64, chain) if chain else None
    rep = {}
    for k, b in refg.items():
        if k == "dL_dconic":
            continue
        a = np.asarray(hipg[k]).reshape(b.shape)
        assert np.isfinite(a).all(), "%s: %s has non-finite values" % (label, k)
        scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
        if scale_like and k in scale_like:
            scale = max(scale, float(np.abs(refg[scale_like[k]]).max()))
        d = np.abs(a - b)
        err = float(d.max()) if b.size else 0.0
        if err <= tol * scale:
            rep[k] = ("1e-4", err, scale, 0, 0.0)
            continue
        assert k in chain, "%s: %s max abs err %g > %g (max|ref| %g) -- not a tensor behind the covariance chain: the plain bar is the bar" % (
            label, k, err, tol * scale, scale)
        P = b.shape[0]
        e = np.abs(a.astype(np.float64) - np.asarray(refg_f64[k], np.float64).reshape(b.shape)).reshape(P, -1).max(1)
        bound = tol * scale + K * scale * nu
        ratio = e / bound
        g = int(np.argmax(ratio))
        beyond = int((d.reshape(P, -1).max(1) > tol * scale).sum())
        assert ratio[g] <= 1.0, ("%s: %s of Gaussian %d: |hip - f64| %g > %g = 1e-4 * scale + %g * (what the reference's own accumulation orders and a few-ulp perturbation of its terms do to that Gaussian, %g of scale); "
                                 "scale %g, %d Gaussians beyond the plain bar, max |hip - ref| %g") % (label, k, g, e[g], bound[g], K, nu[g], scale, beyond, err)
        rep[k] = ("noise-aware", err, scale, beyond, float(ratio[g]))
    return rep


def fmt_noise_rep(rep):
    return {k: ("%.2e/%.1e" % (v[1], v[2])) + ("" if v[0] == "1e-4" else " [%d Gaussians beyond 1e-4; worst |hip-f64| at %.2f of its conditioning-aware bound]" % (v[3], v[4]))
            for k, v in rep.items()}


def oracle_four_modes(o, grads):
    """The oracle's backward in its two accumulation orders, with double-accumulated sums and as the conditioning probe
    (oracle_set_accumulation 0 / 1 / 2 / 3): four dicts."""
    args = (grads["grad_color"], grads["grad_depth"], grads["grad_alpha"], grads["grad_flow"])
    out = []
    try:
        for mode in (0, 1, 2, 3):
            pyoracle.set_accumulation(mode)
            out.append({k: v.copy() for k, v in o.backward(*args).items()})
    finally:
        pyoracle.set_accumulation(0)
    return out


# ----------------------------------------------------------------------------------------------------------------------------
# Uniformly distributed orientations with a moderate footprint (tests/test_gpu_orientations.py, tests/golden/make_golden.py)
# ----------------------------------------------------------------------------------------------------------------------------

def splat_anisotropy(ref):
    """sqrt(lambda_max / lambda_min) of every Gaussian's 2D conic (= of its screen-space covariance incl. the 0.3 low-pass)."""
    a, b, c = (ref["conic_opacity"][:, i].astype(np.float64) for i in range(3))
    mid, det = 0.5 * (a + c), a * c - b * b
    root = np.sqrt(np.maximum(mid * mid - det, 0.0))
    return np.sqrt((mid + root) / np.maximum(mid - root, 1e-300))


def bounded_footprint(scene, limit=40, max_anisotropy=5.0, rounds=10, seed=77, views=None):
    """Redraws (uniformly) the quaternion pairs of the Gaussians whose splat is wider than ``limit`` pixels or more elongated than
    1 : ``max_anisotropy`` on screen, until none is left (the last few get the identity): orientations uniform on the sphere
    CONDITIONED on a moderate footprint.  ``views``: scene-dict overrides (camera tensors, timestamp) of every view the scene is
    rendered from -- a pair is redrawn if it is out of bounds in any of them.  Returns the fraction redrawn in the first round."""
    g = torch.Generator().manual_seed(seed)
    scene["rotations"], scene["rotations_r"] = scene["rotations"].clone(), scene["rotations_r"].clone()
    first = None
    for it in range(rounds + 1):
        bad = None
        for v in (views or [{}]):
            ref, _ = run_oracle(dict(scene, **v), None, kind="port")
            b = (ref["radii"] > limit) | ((ref["radii"] > 0) & (splat_anisotropy(ref) > max_anisotropy))
            bad = b if bad is None else (bad | b)
        bad = torch.from_numpy(bad)
        n = int(bad.sum())
        first = n if first is None else first
        if n == 0:
            break
        for k in ("rotations", "rotations_r"):
            if it < rounds:
                q = torch.randn(n, 4, generator=g)
                scene[k][bad] = q / q.norm(dim=1, keepdim=True)
            else:
                scene[k][bad] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    return first / float(bad.numel())
