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
