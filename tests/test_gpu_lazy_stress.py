"""GPU stress: the forward that never waits for ``num_rendered`` (fdgs_forward_out.lazy, csrc/capi.hip: mailbox ring, run-ahead
buffers) -- what replaces the blocking read of the reference (rasterizer_impl.cu:302).  A lazy path that returned a stale count would
be a silent wrong-gradient bug, so: hundreds of iterations in one process, a second host thread hammering the caching allocator and
the device with its own stream, run-ahead switched on and off, and EXACT equality of every count (and of the images) with the waiting
forward on identical parameters in every iteration.

(History: one builder run of round 4 showed ``num_rendered`` 37335 vs 37332 between a lazy and a waiting StepPipeline.  That was the
11th view of the comparison -- the fourth optimizer step, after three Adam updates whose float-atomics noise differs from run to run,
so the two models no longer held identical parameters; the assertion of that day demanded equality there, commit 7e9634b relaxed it for
the later steps.  Here the parameters ARE identical in every iteration (restored before each step), so equality is exact.)"""
import threading

import numpy as np
import pytest
import torch

from util import native_args_fwd, scene_to_device, synth

pytestmark = pytest.mark.gpu
SC = synth.SceneConfig


class _Hammer:
    """A host thread that allocates / frees tensors of random sizes and launches small kernels on its own stream until stopped."""

    def __init__(self, dev):
        self.dev, self.stop, self.count, self.error = dev, False, 0, None
        self.t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        try:
            torch.cuda.set_device(self.dev)
            s = torch.cuda.Stream(self.dev)
            rng = np.random.default_rng(0)
            keep = []
            with torch.cuda.stream(s):
                while not self.stop:
                    n = int(rng.integers(1, 1 << 22))
                    t = torch.empty(n, dtype=torch.float32, device=self.dev)
                    t.fill_(1.0)
                    keep.append(t)
                    if len(keep) > 8:
                        del keep[int(rng.integers(0, len(keep)))]
                    if self.count % 64 == 63:
                        s.synchronize()
                        torch.cuda.empty_cache() if self.count % 512 == 511 else None
                    self.count += 1
        except Exception as e:   # surfaces in the main thread
            self.error = repr(e)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join()
        assert self.error is None, self.error
        assert self.count > 0


def test_lazy_forward_counts_are_exact_under_allocator_pressure(gpu_device):
    """200 iterations x 3 scenes of one size (different instance counts, so every lazy forward is sized by ANOTHER scene's report):
    the count a lazy forward reports and its image equal the waiting forward's, with run-ahead on and off, next to the hammer."""
    from fdgs import _capi
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import _C
    cfg = SC("lzs", 20041, 320, 240, 1, 0, 0.03, 1.0, True, 4, True)    # a P no other test uses
    scenes = []
    for k, pose in ((1.0, "axis"), (0.9, "rig1"), (1.1, "rig2")):
        sc = synth.make_scene(cfg, seed=8, pose=pose)
        sc["scales"] = (sc["scales"] * k).contiguous()
        scenes.append(scene_to_device(sc, gpu_device))
    _capi.forward_lazy_status(gpu_device, wait=True)
    want = []
    for sc in scenes:
        res = _C.rasterize_gaussians(*native_args_fwd(sc))
        want.append((res[0], res[1].clone(), res[5].clone()))
    assert len({w[0] for w in want}) == 3 and min(w[0] for w in want) > 0
    bad = []
    with _Hammer(gpu_device) as hammer:
        for it in range(200):
            run_ahead = (it // 10) % 2 == 0
            _capi.lib.fdgs_set_run_ahead(1 if run_ahead else 0)
            order = [(it + j) % 3 for j in range(3)]
            outs = [_C.rasterize_gaussians(*native_args_fwd(scenes[i]), lazy=True) for i in order]
            pend, failed, reported = _capi.forward_lazy_status(gpu_device, wait=True)
            lazy_ix = [j for j, o in enumerate(outs) if o[0] < 0]
            counts = [o[0] for o in outs]
            for j, r in zip(lazy_ix, reported[-len(lazy_ix):] if lazy_ix else []):
                counts[j] = r
            if pend != 0 or failed != 0 or len(reported) != len(lazy_ix):
                bad.append((it, "status", pend, failed, reported))
            for j, i in enumerate(order):
                if counts[j] != want[i][0] or not torch.equal(outs[j][1], want[i][1]) or not torch.equal(outs[j][5], want[i][2]):
                    bad.append((it, run_ahead, i, counts[j], want[i][0]))
            # every other iteration a WAITING forward in between: it must see its own count, not a pending lazy one's
            if it % 2:
                w = _C.rasterize_gaussians(*native_args_fwd(scenes[order[0]]))
                if w[0] != want[order[0]][0]:
                    bad.append((it, "waiting", w[0], want[order[0]][0]))
    _capi.lib.fdgs_set_run_ahead(1)
    assert not bad, bad[:10]
    assert hammer.count > 50


def test_step_pipeline_lazy_counts_are_exact_on_identical_parameters(gpu_device):
    """StepPipeline(lazy=True) against lazy=False, 100 steps each in one process next to the hammer, with the model AND the optimizer
    state restored before every step so that both see bit-identical parameters: every view's num_rendered, every step, must be equal
    -- and equal to the first waiting step's; no step may have been redone."""
    from fdgs import train_host
    from fdgs.pipeline import StepPipeline
    cfg = SC("lzq", 6011, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=4)
    bg = torch.tensor([0.1, 0.2, 0.3], device=gpu_device)
    pipe = train_host.PipelineFlags()
    B = 3
    cams = [train_host.SyntheticCamera(dict(scene, **synth.camera_for(p, scene["W"], scene["H"])), gpu_device, timestamp=(b + 0.5) / B * 10.0)
            for b, p in enumerate(["axis", "rig0", "rig3"])]
    gen = torch.Generator(device="cpu").manual_seed(7)
    gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(gpu_device) for _ in range(B)]
    runs = {}
    with _Hammer(gpu_device):
        for lazy in (False, True):
            m = train_host.GaussianParams(scene, gpu_device)
            opt = train_host.make_optimizer(m)
            sp = StepPipeline(m, opt, world_size=1, lambda_dssim=0.2, lazy=lazy)
            p0 = m.flat.detach().clone()
            rs, losses = [], []
            for it in range(100):
                with torch.no_grad():
                    m.flat.copy_(p0)
                    opt.exp_avg.zero_()
                    opt.exp_avg_sq.zero_()
                opt.step_count = 0
                results, ls = sp.step(cams, gts, pipe, bg)
                rs.append([r["num_rendered"] for r in results])
                losses.append([float(l) for l in ls])
            torch.cuda.synchronize()
            runs[lazy] = (rs, losses, sp.lazy_redone)
    assert runs[True][2] == 0 and runs[False][2] == 0
    first = runs[False][0][0]
    assert min(first) > 0 and len(set(first)) == B
    for lazy in (False, True):
        for it, r in enumerate(runs[lazy][0]):
            assert r == first, "lazy=%s step %d: num_rendered %r, the waiting forward on the same parameters: %r" % (lazy, it, r, first)
    np.testing.assert_allclose(np.array(runs[True][1]), np.array(runs[False][1]), rtol=1e-5, atol=1e-6)
