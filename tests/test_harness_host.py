"""CPU tests of the frame-parallel harness pieces (fdgs/harness.py): sharding, learning-rate schedule, statistics."""
import json
import os

import numpy as np
import torch

import util  # noqa: F401  (sys.path for the package)
from fdgs import harness


def test_expon_lr_matches_reference_golden():
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "expon_lr.json")))
    assert len(cases) >= 40
    for c in cases:
        got = harness.expon_lr(c["step"], c["lr_init"], c["lr_final"], c["lr_delay_steps"], c["lr_delay_mult"], c["max_steps"])
        assert got == c["lr"] or abs(got - c["lr"]) <= 1e-15 * max(1.0, abs(c["lr"])), (c, got)


def test_frame_shard_partitions_every_epoch():
    n, B, world = 53, 3, 4
    shards = [harness.FrameShard(n, B, world, r, seed=11) for r in range(world)]
    nb = shards[0].batches_per_epoch()
    for epoch in range(3):
        per_rank = [s.epoch(epoch) for s in shards]
        assert nb == n // (B * world) and all(len(p) == nb for p in per_rank)
        seen = []
        for k in range(nb):
            batch = [i for p in per_rank for i in p[k]]
            assert len(batch) == B * world and len(set(batch)) == B * world   # a global batch has no repeated view
            seen += batch
        assert len(set(seen)) == len(seen) == nb * B * world                   # drop_last: each view at most once per epoch
        assert set(seen) <= set(range(n))
    assert shards[0].epoch(0) != shards[0].epoch(1)                            # reshuffled
    it = iter(shards[1])
    first = [next(it) for _ in range(nb + 1)]
    assert first[:nb] == shards[1].epoch(0) and first[nb] == shards[1].epoch(1)[0]


def _reference_stats_update(xyz_acc, t_acc, denom, max_r, radii_list, grad_list, t_grad, batch_size):
    """train.py:164-184 + 229-236 + gaussian_model.py:637-642, restated literally."""
    vis_list = [r > 0 for r in radii_list]
    visibility_count = torch.stack(vis_list, 1).sum(1)
    visibility_filter = visibility_count > 0
    radii = torch.stack(radii_list, 1).max(1)[0]
    g = torch.stack([torch.norm(x[:, :2], dim=-1) for x in grad_list], 1).sum(1)
    g[visibility_filter] = g[visibility_filter] * batch_size / visibility_count[visibility_filter]
    g = g.unsqueeze(1)
    bt = t_grad.clone()[:, 0]
    bt[visibility_filter] = bt[visibility_filter] * batch_size / visibility_count[visibility_filter]
    bt = bt.unsqueeze(1)
    max_r[visibility_filter] = torch.max(max_r[visibility_filter], radii[visibility_filter].float())
    xyz_acc[visibility_filter] += g[visibility_filter]
    denom[visibility_filter] += 1
    t_acc[visibility_filter] += bt[visibility_filter]


def test_densification_stats_match_reference_statement():
    P, B = 400, 4
    g = torch.Generator().manual_seed(3)
    st = harness.DensificationStats(P, "cpu", 1)
    ref = [torch.zeros(P, 1), torch.zeros(P, 1), torch.zeros(P, 1), torch.zeros(P)]
    for _ in range(3):
        radii = [(torch.randint(-2, 9, (P,), generator=g).clamp(min=0)).to(torch.int32) for _ in range(B)]
        grads = [torch.randn(P, 3, generator=g) for _ in range(B)]
        t_grad = torch.randn(P, 1, generator=g)
        st.update([{"radii": r, "viewspace_grad": x} for r, x in zip(radii, grads)], t_grad, B)
        _reference_stats_update(ref[0], ref[1], ref[2], ref[3], radii, grads, t_grad, B)
    np.testing.assert_allclose(st.xyz_gradient_accum.numpy(), ref[0].numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(st.t_gradient_accum.numpy(), ref[1].numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(st.denom.numpy(), ref[2].numpy())
    np.testing.assert_array_equal(st.max_radii2D.numpy(), ref[3].numpy())


def test_sh_schedule_follows_oneupSHdegree():
    """scene/gaussian_model.py:253-257 + train.py:93-94: from (0, 0) the spatial degree climbs first, then the time degree;
    a model built from a scene starts with every degree active and the schedule leaves it alone."""
    import torch
    from fdgs import synth, train_host
    cfg = synth.SceneConfig("sh", 50, 32, 32, 3, 2, 0.05, 10.0, True, 4, False)
    m = train_host.GaussianParams(synth.make_scene(cfg, seed=0), torch.device("cpu"))
    assert (m.max_sh_degree, m.max_sh_degree_t, m.active_sh_degree, m.active_sh_degree_t) == (3, 2, 3, 2)
    m.oneupSHdegree()
    assert (m.active_sh_degree, m.active_sh_degree_t) == (3, 2)
    m.active_sh_degree, m.active_sh_degree_t = 0, 0
    seq = []
    for _ in range(7):
        m.oneupSHdegree()
        seq.append((m.active_sh_degree, m.active_sh_degree_t))
    assert seq == [(1, 0), (2, 0), (3, 0), (3, 1), (3, 2), (3, 2), (3, 2)]
    opt = train_host.FlatAdam(m)
    import pytest
    with pytest.raises(ValueError, match="multiple of 4"):
        opt.step_range(2, 10)


def test_spatial_sort_permutes_parameters_and_moments_alike():
    """train_host.spatial_sort: rows of every parameter tensor and of both Adam moments move together; the Morton permutation is a
    stable, deterministic function of the positions; neighbours in the new order are neighbours in space."""
    from fdgs import synth, train_host
    scene = synth.make_scene(synth.SceneConfig("h", 3000, 64, 48, 1, 0, 0.03, 1.0, True, 4, True), seed=2)
    m = train_host.GaussianParams(scene, torch.device("cpu"))
    o = train_host.make_optimizer(m)
    o.exp_avg.copy_(torch.arange(m.flat.numel(), dtype=torch.float32))
    o.exp_avg_sq.copy_(torch.arange(m.flat.numel(), dtype=torch.float32) * 2)
    before = {n: m.params[n].detach().clone() for n in m.NAMES}
    ea = {n: o.exp_avg[m.offsets[n][0]:m.offsets[n][1]].clone().view(m.P, -1) for n in m.NAMES}
    perm = train_host.spatial_sort(m, o)
    assert torch.equal(perm, train_host.morton_permutation(before["_xyz"])) and torch.equal(torch.sort(perm).values, torch.arange(m.P))
    for n in m.NAMES:
        assert torch.equal(m.params[n].detach(), before[n][perm]), n
        b, e = m.offsets[n]
        assert torch.equal(o.exp_avg[b:e].view(m.P, -1), ea[n][perm]), n
        assert torch.equal(o.exp_avg_sq[b:e].view(m.P, -1), 2 * ea[n][perm]), n
    # already sorted: the permutation is the identity (stable sort)
    assert torch.equal(train_host.morton_permutation(m.params["_xyz"].detach()), torch.arange(m.P))
    xyz = m.params["_xyz"].detach()
    step_sorted = (xyz[1:] - xyz[:-1]).norm(dim=1).mean().item()
    step_random = (before["_xyz"][1:] - before["_xyz"][:-1]).norm(dim=1).mean().item()
    assert step_sorted < 0.25 * step_random
