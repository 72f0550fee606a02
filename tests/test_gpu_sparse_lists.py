"""GPU: fdgs_forward_out.sparse_lists -- a lazy forward whose tile lists sit at fixed offsets of the binning buffer (tile t at
[t * cap, t * cap + n_t)), so that the count and scan launches (rasterizer_impl.cu:99-139's counting + InclusiveSum, here tile_count +
tile_scan) leave the forward: the scatter pass counts as it goes, the per-tile sort reads each tile's count and one extra workgroup of
its launch reports num_rendered and writes the blend kernels' tile order.

What must hold: the LISTS -- which instances, in which order -- are exactly the compact forward's (hence the reference's, or its
sub-lists with tile_cull: tests/test_gpu_parity.py, test_gpu_tile_cull.py), only their addresses differ; every image, radii, n_contrib bit
for bit; the reported num_rendered equal; the backward (which only sees `ranges`) gives the same gradients; a list that outgrows its
slots is reported as a failed lazy forward."""
import ctypes as C

import numpy as np
import pytest
import torch

from util import GRAD_SCALE, _view, native_args_fwd, run_oracle, scene_to_device, synth

pytestmark = pytest.mark.gpu
SC = synth.SceneConfig


def _lists(res, P, W, H):
    """per-tile lists (concatenated in tile order), ranges and n_contrib of a forward, whatever the layout of the binning buffer"""
    from fdgs import _capi
    (R, color, flow, depth, T, radii, geom, binb, img, covs_com, om) = res
    v = _capi.FdgsDebugView()
    rc = _capi.lib.fdgs_debug_views(P, W, H, max(R, 0), _capi._ptr(geom), _capi._ptr(binb), _capi._ptr(img), C.byref(v))
    assert rc == 0, _capi.last_error()
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    torch.cuda.synchronize()
    rg = _view(img, v.ranges, ntiles * 2, torch.int32).reshape(ntiles, 2).cpu().numpy().astype(np.int64)
    extent = int(rg[:, 1].max()) if ntiles else 0
    pl = _view(binb, v.point_list, extent, torch.int32).cpu().numpy() if extent else np.zeros(0, np.int32)
    lists = [pl[a:b] for a, b in rg]
    nc = _view(img, v.n_contrib, W * H, torch.int32).reshape(H, W).cpu().numpy()
    order = _view(img, v.tile_order, ntiles, torch.int32).cpu().numpy().astype(np.int64)
    return lists, rg, nc, order


def _fwd(sc, **kw):
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import _C
    return _C.rasterize_gaussians(*native_args_fwd(sc), **kw)


@pytest.mark.parametrize("tile_cull", [False, True], ids=["reference-lists", "tile-cull"])
@pytest.mark.parametrize("name,cfg,pose", [
    ("rot4d_sh3t2", SC("sp", 30017, 400, 304, 3, 2, 0.015, 10.0, True, 4, False), "rig1"),
    ("dim3_sh2", SC("sp", 8009, 256, 256, 2, 0, 0.03, 1.0, False, 3, False), "rig0"),
    ("ragged", SC("sp", 1501, 49, 33, 1, 0, 0.05, 1.0, True, 4, True), "axis"),
    ("long_lists", SC("sp", 40013, 96, 64, 0, 0, 0.03, 1.0, True, 4, True), "axis"),        # lists of thousands of entries: the long-list sort instances
    # at full size, where the compact forward is held to the oracle (tests/test_gpu_parity.py): the configuration the metric is quoted on
    # through one of the bench's cameras, and its clustered variant (lists of up to 5485 entries: 45 M slots of address space)
    ("C3", synth.CONFIGS["C3"], "rig0"),
    ("C3-clustered", synth.CONFIGS["C3-clustered"], "axis"),
])
def test_sparse_lists_are_the_compact_lists_elsewhere(name, cfg, pose, tile_cull, gpu_device):
    from fdgs import _capi
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import _C
    scene = synth.make_scene(cfg, seed=3, pose=pose, random_flow=True)
    sc = scene_to_device(scene, gpu_device)
    P, W, H = cfg.P, cfg.W, cfg.H
    _capi.forward_lazy_status(gpu_device, wait=True)
    first = _fwd(sc, tile_cull=tile_cull)      # the waiting forward with compact lists (it also leaves the run-ahead guess behind)
    assert first[0] > 0
    want_lists, want_rg, want_nc, _ = _lists(first, P, W, H)
    assert int((want_rg[:, 1] - want_rg[:, 0]).sum()) == first[0] and (want_rg[1:, 0][want_rg[1:, 1] > want_rg[1:, 0]] >= 0).all()
    before = _capi.run_ahead_stats()
    res = _fwd(sc, tile_cull=tile_cull, lazy=True, sparse_lists=True)
    assert res[0] == -1, "the second forward of a configuration must run ahead"
    pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
    assert (pend, failed, reported) == (0, 0, [first[0]]), (pend, failed, reported, first[0])
    assert tuple(b - a for a, b in zip(before, _capi.run_ahead_stats())) == (1, 0, 0)
    got_lists, got_rg, got_nc, order = _lists(res, P, W, H)
    n_want, n_got = want_rg[:, 1] - want_rg[:, 0], got_rg[:, 1] - got_rg[:, 0]
    np.testing.assert_array_equal(n_got, n_want, err_msg=name + ": list lengths")
    # the layout: fixed, 64-aligned slots, tile t at t * cap; empty tiles keep (0, 0) as identifyTileRanges leaves them
    nonempty = n_got > 0
    t_idx = np.nonzero(nonempty)[0]
    assert t_idx.size > 1
    cap = int((got_rg[t_idx[-1], 0] - got_rg[t_idx[0], 0]) // (t_idx[-1] - t_idx[0]))
    assert cap % 64 == 0 and cap >= n_got.max()
    np.testing.assert_array_equal(got_rg[t_idx, 0], t_idx * cap)
    assert (got_rg[~nonempty] == 0).all()
    for t in range(len(want_lists)):
        assert np.array_equal(got_lists[t], want_lists[t]), "%s: tile %d lists differ" % (name, t)
    np.testing.assert_array_equal(got_nc, want_nc)
    for i, key in ((1, "colour"), (2, "flow"), (3, "depth"), (4, "T"), (5, "radii")):
        assert torch.equal(res[i], first[i]), "%s: %s differs between the compact and the sparse forward" % (name, key)
    assert np.array_equal(np.sort(order), np.arange(len(order)))      # the tile order written by the sort launch: a permutation
    # the backward on the sparse lists (num_rendered = -1) against the backward on the compact ones
    grads = synth.make_upstream_grads(W, H, seed=1, scale=GRAD_SCALE)
    e = torch.Tensor([])
    g = lambda k: sc[k] if sc.get(k) is not None else e  # noqa: E731
    outs = []
    for r in (first, res):
        (R, color, flow, depth, T, radii, geom, binb, img, covs_com, om) = r
        gd = {k: v.to(gpu_device) for k, v in grads.items()}
        bargs = (sc["bg"], sc["means3D"], om, radii, g("colors_precomp"), g("flow_2d"), sc["opacities"], g("ts"), g("scales"), g("scales_t"), g("rotations"),
                 g("rotations_r"), 1.0, g("cov3D_precomp"), -1.0, sc["world_view_transform"], sc["full_proj_transform"], sc["tanfovx"], sc["tanfovy"],
                 gd["grad_color"], gd["grad_depth"], gd["grad_alpha"], gd["grad_flow"], g("shs"), sc["sh_degree"], sc["sh_degree_t"], sc["camera_center"],
                 sc["timestamp"], sc["time_duration"], sc["rot_4d"], sc["gaussian_dim"], sc["force_sh_3d"], geom, R, binb, img, False)
        outs.append([t.cpu().numpy() for t in _C.rasterize_gaussians_backward(*bargs)])
    for a, b in zip(*outs):
        scale = max(1.0, float(np.abs(a).max()) if a.size else 1.0)
        assert float(np.abs(a - b).max()) <= 3e-4 * scale if a.size else True    # (float atomics: order differs run to run)
    print(name, "tile_cull" if tile_cull else "reference lists", "R", first[0], "cap", cap, "longest", int(n_got.max()),
          "address space x%.1f" % (cap * len(n_got) / max(first[0], 1)))


def test_sparse_lists_against_the_oracle(gpu_device):
    """... and directly against the port oracle: every tile's sparse list is the reference's list of that tile, bit for bit."""
    from fdgs import _capi
    cfg = SC("spo", 12011, 320, 240, 2, 1, 0.02, 2.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=5, pose="rig3")
    sc = scene_to_device(scene, gpu_device)
    _capi.forward_lazy_status(gpu_device, wait=True)
    _fwd(sc)
    res = _fwd(sc, lazy=True, sparse_lists=True)
    assert res[0] == -1
    _pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
    ref, _ = run_oracle(scene, None, kind="port")
    assert failed == 0 and reported == [ref["R"]]
    lists, rg, nc, _ = _lists(res, cfg.P, cfg.W, cfg.H)
    rr = ref["ranges"].astype(np.int64)
    for t in range(rr.shape[0]):
        assert np.array_equal(lists[t].astype(np.uint32), ref["point_list"][rr[t, 0]:rr[t, 1]]), t
    ok = ~ref["border"].astype(bool)
    assert int((nc.astype(np.uint32)[ok] != ref["n_contrib"][ok]).sum()) == 0
    assert float(np.abs(res[1].cpu().numpy() - ref["out_color"])[:, ok].max()) <= 1e-4


def test_sparse_list_overflow_is_reported(gpu_device):
    """A view whose lists outgrow the slots provided (1.5 x the longest list of the last four reports + 64) is cut ON THE DEVICE and
    reported as failed; the next forwards have learnt the size."""
    from fdgs import _capi
    cfg = SC("spf", 20029, 320, 240, 0, 0, 0.03, 1.0, True, 4, True)
    scene = synth.make_scene(cfg, seed=8)
    sc = scene_to_device(scene, gpu_device)
    big = dict(scene)
    big["scales"] = (scene["scales"] * 2.0).contiguous()
    bsc = scene_to_device(big, gpu_device)
    _capi.forward_lazy_status(gpu_device, wait=True)
    first = _fwd(sc)
    ok = _fwd(sc, lazy=True, sparse_lists=True)
    assert ok[0] == -1 and _capi.forward_lazy_status(gpu_device, wait=True)[1:] == (0, [first[0]])
    res = _fwd(bsc, lazy=True, sparse_lists=True)      # lists 2.2 x as long as provided for
    assert res[0] == -1
    pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
    want = _fwd(bsc)                                     # the waiting forward: right, and it teaches the guess the new size
    assert pend == 0 and failed == 1 and reported == [want[0]], (pend, failed, reported, want[0])
    again = _fwd(bsc, lazy=True, sparse_lists=True)
    _pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
    assert again[0] == -1 and failed == 0 and reported == [want[0]]
    assert torch.equal(again[1], want[1]) and torch.equal(again[5], want[5])


def test_sparse_lists_budget_falls_back_to_compact_lists(gpu_device):
    """fdgs_set_sparse_lists_budget: T * cap entries of address space are only spent while they stay within max(min_bytes, factor x the
    compact buffer).  With the budget taken away a lazy forward that asks for sparse lists keeps COMPACT lists (prefix-sum ranges, count
    + scan launches) and everything it returns is what the sparse forward returns; the counters say which layout ran."""
    from fdgs import _capi
    cfg = SC("spb", 16033, 320, 240, 1, 0, 0.03, 1.0, True, 4, True)
    scene = synth.make_scene(cfg, seed=9, pose="rig2")
    sc = scene_to_device(scene, gpu_device)
    _capi.forward_lazy_status(gpu_device, wait=True)
    first = _fwd(sc)
    want_lists, want_rg, want_nc, _ = _lists(first, cfg.P, cfg.W, cfg.H)
    assert _capi.lib.fdgs_set_sparse_lists_budget(-1, 4) != 0 and _capi.lib.fdgs_set_sparse_lists_budget(0, 0) != 0
    try:
        s0 = _capi.sparse_lists_stats()
        sparse = _fwd(sc, lazy=True, sparse_lists=True)
        s1 = _capi.sparse_lists_stats()
        assert (s1[0] - s0[0], s1[1] - s0[1]) == (1, 0)
        assert _capi.lib.fdgs_set_sparse_lists_budget(0, 1) == 0          # never more than the compact buffer
        compact = _fwd(sc, lazy=True, sparse_lists=True)
        s2 = _capi.sparse_lists_stats()
        assert (s2[0] - s1[0], s2[1] - s1[1]) == (0, 1) and s2[2] < s1[2]
    finally:
        assert _capi.lib.fdgs_set_sparse_lists_budget(1 << 30, 4) == 0
    pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
    assert (pend, failed, reported) == (0, 0, [first[0], first[0]])
    got_lists, got_rg, got_nc, _ = _lists(compact, cfg.P, cfg.W, cfg.H)
    np.testing.assert_array_equal(got_rg, want_rg)                        # compact: the reference's prefix sums
    for t in range(len(want_lists)):
        assert np.array_equal(got_lists[t], want_lists[t])
    sp_rg = _lists(sparse, cfg.P, cfg.W, cfg.H)[1]
    assert not np.array_equal(sp_rg, want_rg)                             # (the sparse forward really had another layout)
    for i in (1, 2, 3, 4, 5):
        assert torch.equal(compact[i], first[i]) and torch.equal(sparse[i], first[i])


def test_one_hot_tile_does_not_buy_gigabytes(gpu_device):
    """A real capture has hot tiles.  2704 x 2028 (T = 21 717 tiles) with 17 000 Gaussians projecting into ONE tile: sparse lists would
    take T x cap(~25 k) = 550 M slots = 6.9 GB per forward in flight for 0.2 M instances; the default budget (max(1 GiB, 4 x compact))
    sends that forward through the compact layout: same image bit for bit, a binning buffer of a few MB."""
    from fdgs import _capi
    cfg = SC("hot", 60037, 2704, 2028, 0, 0, 0.004, 1.0, True, 4, True)
    scene = synth.make_scene(cfg, seed=10)
    n_hot = 17000
    g = torch.Generator().manual_seed(3)
    hot = scene["means3D"][:n_hot]
    # a spot a few pixels wide at depth 4 around a tile centre: pixel (1352 +- 256, 1014 +- 206) with focal 0.9 W
    focal = 0.9 * cfg.W
    hot[:, 0:2] = torch.tensor([256.0, -206.0]) * 4.0 / focal + 0.0012 * torch.randn(n_hot, 2, generator=g)
    hot[:, 2] = 0.0
    scene["scales"][:n_hot] = 0.0008
    scene["ts"][:n_hot] = scene["timestamp"]
    sc = scene_to_device(scene, gpu_device)
    _capi.forward_lazy_status(gpu_device, wait=True)
    first = _fwd(sc)
    lists, rg, _nc, _ = _lists(first, cfg.P, cfg.W, cfg.H)
    longest = int((rg[:, 1] - rg[:, 0]).max())
    assert longest > 12000 and first[0] < 600000, (longest, first[0])
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(gpu_device)
    base = torch.cuda.memory_allocated(gpu_device)
    s0 = _capi.sparse_lists_stats()
    res = _fwd(sc, lazy=True, sparse_lists=True)
    pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
    s1 = _capi.sparse_lists_stats()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(gpu_device) - base
    assert res[0] == -1 and (pend, failed, reported) == (0, 0, [first[0]])
    assert (s1[0] - s0[0], s1[1] - s0[1]) == (0, 1), "the hot tile went through the sparse layout"
    T = ((cfg.W + 15) // 16) * ((cfg.H + 15) // 16)
    would_be = T * (longest + longest // 2) * 12.5
    print("hot tile: longest list %d of R %d; binning buffer %.1f MB (sparse lists would take %.1f GB); peak allocation of the forward %.1f MB" % (
        longest, first[0], s1[2] / 2**20, would_be / 2**30, peak / 2**20))
    assert s1[2] < 64 << 20 and res[7].numel() == s1[2] and peak < 512 << 20 and would_be > 4 * 2**30
    for i in (1, 3, 4, 5):
        assert torch.equal(res[i], first[i])
