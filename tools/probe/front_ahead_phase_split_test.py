"""fdgs_forward_out.phase: a forward in two calls -- the front half (geometry + tile binning: reads no SH coefficient) and, later and
possibly on another stream, the back half (SH colours + blend) -- against the same forward in one call.  Same kernels on the same
numbers: every output, and every gradient of the backward that follows, must be the same BIT FOR BIT (the blend backward's float
atomics aside: the backward is compared on the forward's buffers, which are compared exactly)."""
import pytest
import torch

from fdgs import synth

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int32) if t.dtype == torch.float32 else t


CASES = [
    # name, P, W, H, D, D_t, dim, rot_4d, force_sh_3d, pose
    ("4d-t2-rig1", 9000, 208, 160, 3, 2, 4, True, False, "rig1"),
    ("4d-t0-axis", 6000, 176, 144, 3, 0, 4, True, False, "axis"),
    ("3d-deg2-rig2", 7000, 208, 160, 2, 0, 3, False, True, "rig2"),
    ("4d-norot-rig3", 6000, 176, 144, 3, 2, 4, False, False, "rig3"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("mode", ["waiting", "lazy", "lazy-sparse"])
@pytest.mark.parametrize("tile_cull", [False, True], ids=["ref-lists", "tile-cull"])
def test_forward_in_two_calls_equals_the_forward_in_one(gpu_device, case, mode, tile_cull):
    from fdgs import _capi, train_host
    from fdgs.fused import raw_forward, raw_settings
    name, P, W, H, D, D_t, dim, rot4, f3d, pose = case
    cfg = synth.SceneConfig(name, P, W, H, D, D_t, 0.03, 10.0 if dim == 4 else 1.0, rot4, dim, f3d)
    scene = synth.make_scene(cfg, seed=31, rot_sigma="uniform", pose=pose)
    model = train_host.GaussianParams(scene, gpu_device)
    pipe = train_host.PipelineFlags()
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    cam = train_host.SyntheticCamera(scene, gpu_device, timestamp=0.41 * scene["time_duration"])
    rs, tens = raw_settings(cam, model, pipe, bg)
    kw = dict(tile_cull=tile_cull, lazy=mode != "waiting", sparse_lists=mode == "lazy-sparse")
    whole0 = raw_forward(rs, *tens, tile_cull=tile_cull)          # (the first call of a configuration: exact sizes; gives the run-ahead its guess)
    whole = raw_forward(rs, *tens, **kw)
    other = torch.cuda.Stream(gpu_device)
    for rep in range(2):
        h = raw_forward(rs, *tens, phase=1, **kw)
        assert isinstance(h, dict) and h["split_forward"]
        # ... something else on the stream in between, the back half on ANOTHER stream behind an event
        torch.zeros(1 << 20, device=gpu_device).add_(1.0)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(other):
            other.wait_event(ev)
            split = raw_forward(rs, *tens, preprocessed=h, phase=2)
        torch.cuda.current_stream(gpu_device).wait_stream(other)
        torch.cuda.synchronize()
        if mode != "waiting":
            _pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
            assert not failed
        for ref in (whole0, whole):
            assert ref[0] < 0 or split[0] < 0 or ref[0] == split[0]
            for i in (1, 2, 3, 4, 5, 9, 10):
                assert torch.equal(_bits(ref[i]), _bits(split[i])), (rep, i)
    assert float(whole0[1].abs().max()) > 0


def test_split_forward_then_backward(gpu_device):
    """The buffers a split forward leaves are the whole forward's: the backward runs on them."""
    from fdgs import train_host
    from fdgs.fused import raw_backward, raw_forward, raw_settings
    cfg = synth.SceneConfig("sfb", 8000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=9, rot_sigma=0.3, pose="rig1")
    model = train_host.GaussianParams(scene, gpu_device)
    pipe = train_host.PipelineFlags()
    bg = torch.zeros(3, device=gpu_device)
    cam = train_host.SyntheticCamera(scene, gpu_device, timestamp=0.5 * scene["time_duration"])
    rs, tens = raw_settings(cam, model, pipe, bg)
    (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = tens
    up = (torch.randn(3, scene["H"], scene["W"], generator=torch.Generator(device="cpu").manual_seed(2)) * 1e-2).to(gpu_device)

    def run(split):
        if split:
            h = raw_forward(rs, *tens, phase=1, tile_cull=True)
            fwd = raw_forward(rs, *tens, preprocessed=h, phase=2)
        else:
            fwd = raw_forward(rs, *tens, tile_cull=True)
        (R, color, flow, depth, T, radii, geom, binb, img, _c, om) = fwd
        sink = {k: torch.zeros_like(v) for k, v in model.grad_sink().items()}
        out = raw_backward(rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img, up, None, None, None,
                           sink, False)
        torch.cuda.synchronize()
        return fwd, sink, out
    fa, sa, oa = run(False)
    fb, sb, ob = run(True)
    assert fa[0] == fb[0] > 0
    assert torch.equal(_bits(fa[1]), _bits(fb[1]))
    for k in sa:
        sc = max(1e-6, float(sa[k].abs().max()))
        # (two runs of ONE backward differ by the order of its float atomics; the covariance chain amplifies it)
        assert float((sa[k] - sb[k]).abs().max()) <= (5e-3 if k in ("dL_dscales", "dL_dscales_t", "dL_drotations", "dL_drotations_r") else 1e-4) * sc, k
