"""fdgs_backward_out.sh_live: the deferred SH backward as two launches -- one lane per Gaussian compacts the LIVE ones (a colour
gradient in this view) into a list, the evaluation takes 64 list entries per wave -- against the one-launch version in which every
wave searches its share of the model (sh_bwd.hip).  A Gaussian's outputs are its own (stage record, accumulator words 12..15), so
the order of the list (atomics: it differs from run to run) changes nothing: everything must be the same BIT FOR BIT.

The blend backward's float atomics differ from run to run, so the two variants are run on ONE copy of its result: blend backward
alone (stage_mask 5), then `fdgs_rasterize_backward` with num_rendered = 0 and a clean accumulator (= no blend launch, SH +
geometry backward on the accumulators as given)."""
import numpy as np
import pytest
import torch

from fdgs import synth

pytestmark = pytest.mark.gpu

CASES = [
    # name, P, W, H, D, D_t, dim, rot_4d, force_sh_3d, pose, rot_sigma, analytic, degree (active D, D_t or None)
    ("4d-t2-uniform-rig1", 9000, 208, 160, 3, 2, 4, True, False, "rig1", "uniform", False, None),
    ("4d-t2-axis", 9000, 208, 160, 3, 2, 4, True, False, "axis", 0.05, False, None),
    ("4d-t1-rig2-analytic", 6000, 176, 144, 3, 1, 4, True, False, "rig2", 0.3, True, None),
    ("4d-t0-rig0", 6000, 176, 144, 3, 0, 4, True, False, "rig0", "uniform", False, None),
    ("4d-norot-rig3", 6000, 176, 144, 3, 2, 4, False, False, "rig3", "uniform", False, None),
    ("4d-force3d", 6000, 176, 144, 3, 2, 4, True, True, "rig1", 0.3, False, None),
    ("3d-deg2", 7000, 208, 160, 2, 0, 3, False, True, "rig1", "uniform", False, None),
    ("3d-deg0", 5000, 160, 128, 0, 0, 3, False, True, "axis", 0.05, False, None),
    ("4d-below-allocated-(2,0)", 6000, 176, 144, 3, 2, 4, True, False, "rig1", "uniform", False, (2, 0)),
    ("4d-below-allocated-(3,1)", 6000, 176, 144, 3, 2, 4, True, False, "rig2", "uniform", False, (3, 1)),
    ("4d-partial-chunk", 4097, 160, 128, 3, 2, 4, True, False, "rig1", "uniform", False, None),
]


def _bits(t):
    return t.contiguous().view(torch.int32)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sh_backward_from_the_live_list_is_the_scan_version_bit_for_bit(gpu_device, case):
    from fdgs import train_host
    from fdgs.fused import raw_backward, raw_forward, raw_settings
    from fdgs.gaussian_renderer import diff_gaussian_rasterization as dgr
    name, P, W, H, D, D_t, dim, rot4, f3d, pose, rot_sigma, analytic, active = case
    cfg = synth.SceneConfig(name, P, W, H, D, D_t, 0.03, 10.0 if dim == 4 else 1.0, rot4, dim, f3d)
    scene = synth.make_scene(cfg, seed=77, rot_sigma=rot_sigma, pose=pose)
    model = train_host.GaussianParams(scene, gpu_device)
    if active is not None:
        model.active_sh_degree, model.active_sh_degree_t = active
    pipe = train_host.PipelineFlags()
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    cam = train_host.SyntheticCamera(scene, gpu_device, timestamp=0.37 * scene["time_duration"])
    gen = torch.Generator(device="cpu").manual_seed(11)
    up = (torch.randn(3, H, W, generator=gen) * 1e-2).to(gpu_device)
    was = dgr.analytic_sh_gradients()
    dgr.set_analytic_sh_gradients(analytic)
    try:
        rs, tens = raw_settings(cam, model, pipe, bg)
        (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = tens
        (R, color, flow, depth, T, radii, geom, binb, img, _covs, om) = raw_forward(rs, *tens)
        assert R > 0

        # blend backward once
        sink0 = {k: torch.zeros_like(v) for k, v in model.grad_sink().items()}
        gacc = torch.zeros((P, 16), device=gpu_device)
        stage0 = torch.zeros((P, 8), device=gpu_device)
        args = (rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img, up, None, None, None)
        pend = raw_backward(*args, sink0, False, grad_accum=gacc, sh_stage=stage0, begin_only=True)
        torch.cuda.synchronize()
        G = gacc.clone()
        assert float(G[:, :3].abs().max()) > 0 and float(G[:, 12:].abs().max()) == 0.0
        del pend

        def sh_and_geometry(use_list):
            sink = {k: torch.full_like(v, float("nan")) for k, v in model.grad_sink().items()}
            acc = G.clone()
            stage = torch.full((P, 8), float("nan"), device=gpu_device)
            seen = {}

            def after_sh():
                torch.cuda.synchronize()
                seen["acc"], seen["stage"] = acc.clone(), stage.clone()
            args0 = args[:13] + (0,) + args[14:]    # num_rendered = 0: no blend launch; the accumulators are taken as given
            scratch = torch.full((P + 64,), 0x7fffffff, dtype=torch.int32, device=gpu_device) if use_list else None   # garbage on entry
            out = raw_backward(*args0, sink, False, grad_accum=acc, sh_stage=stage, after_sh=after_sh, sh_live=scratch)
            if use_list:
                n = int(scratch[P])
                assert n == int((seen["stage"][:, :3] != 0).any(1).sum())          # the count ...
                assert sorted(scratch[:n].tolist()) == torch.nonzero((seen["stage"][:, :3] != 0).any(1)).flatten().tolist()   # ... and the entries
            torch.cuda.synchronize()
            assert float(acc.abs().max()) == 0.0
            return seen, sink, out
        a_seen, a_sink, a_out = sh_and_geometry(False)
        b_seen, b_sink, b_out = sh_and_geometry(True)
        live = (a_seen["stage"][:, :3] != 0).any(1)
        assert live.any()
        assert torch.equal(_bits(a_seen["stage"]), _bits(b_seen["stage"]))     # NaN where nobody writes (second half of a dead record): both
        assert torch.equal(_bits(a_seen["acc"]), _bits(b_seen["acc"]))
        if not (dim == 3 or f3d):
            assert float(a_seen["acc"][:, 15].abs().max()) > 0 or D_t == 0 or (active is not None and active[1] == 0)
        assert float(a_seen["acc"][:, 12:15].abs().max()) > 0 or D == 0 or (active is not None and active[0] == 0)
        for k in a_sink:
            if k == "dL_dsh":
                continue
            assert torch.equal(_bits(a_sink[k]), _bits(b_sink[k])), k
        assert torch.equal(_bits(a_out[0]), _bits(b_out[0]))    # dL_dmeans2D
    finally:
        dgr.set_analytic_sh_gradients(was)


def test_step_pipeline_with_and_without_the_live_list(gpu_device, monkeypatch):
    """StepPipeline(sh_live on / off): the same optimizer steps to the noise two runs of one pipeline differ by."""
    from fdgs import train_host
    from fdgs.pipeline import StepPipeline
    cfg = synth.SceneConfig("pl", 12000, 256, 192, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=5, rot_sigma="uniform")
    pipe = train_host.PipelineFlags()
    bg = torch.zeros(3, device=gpu_device)
    cams = []
    for i, pose in enumerate(("rig0", "rig1", "rig2", "rig3")):
        sc = dict(scene, **synth.camera_for(pose, scene["W"], scene["H"]))
        cams.append(train_host.SyntheticCamera(sc, gpu_device, timestamp=(i + 0.5) / 4 * scene["time_duration"]))
    gen = torch.Generator(device="cpu").manual_seed(3)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in cams]

    def run(flag):
        monkeypatch.setenv("FDGS_PIPELINE_SH_LIVE", flag)
        m = train_host.GaussianParams(scene, gpu_device)
        sp = StepPipeline(m, train_host.make_optimizer(m), world_size=1, lambda_dssim=0.2)
        assert sp.sh_live == (flag == "1")
        losses = []
        for _ in range(3):
            _res, ls = sp.step(cams, gts, pipe, bg)
            losses.append([float(x) for x in ls])
        torch.cuda.synchronize()
        assert (sp._sh_live is not None) == (flag == "1")
        return m.flat.detach().clone(), np.array(losses)
    pa, la = run("0")
    pb, lb = run("1")
    assert np.allclose(la, lb, rtol=1e-5, atol=1e-6), (la, lb)
    scale = max(1.0, float(pa.abs().max()))
    assert float((pa - pb).abs().max()) <= 2e-4 * scale
