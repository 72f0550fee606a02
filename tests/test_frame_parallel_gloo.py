"""CPU, world_size 2 over gloo: the frame-parallel exchange of SURVEY.md section 8e -- ONE all-reduce over the flat
gradient bucket, replicas stay bit-identical after the optimizer step.  The same GaussianParams /
allreduce_gradients code runs over RCCL in bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fdgs import train_host
    cfg = synth.SceneConfig("dp", 257, 64, 48, 3, 2, 0.03, 10.0, True, 4, False)
    scene = synth.make_scene(cfg, seed=0)
    model = train_host.GaussianParams(scene, torch.device("cpu"))
    opt = train_host.make_optimizer(model)
    assert model.flat.numel() == 257 * 161  # 161 floats per Gaussian at M = 48
    # every parameter and gradient is a view into the two flat buffers
    for p in model.params.values():
        assert p.untyped_storage().data_ptr() == model.flat.untyped_storage().data_ptr()
        assert p.grad.untyped_storage().data_ptr() == model.flat_grad.untyped_storage().data_ptr()
    for step in range(3):
        model.zero_grad()
        # a rank-dependent "loss" over the activated parameters, through autograd like the real step
        g = torch.Generator().manual_seed(100 * step + rank)
        loss = 0.0
        for name in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features", "get_t", "get_scaling_t",
                     "get_rotation_r"):
            v = getattr(model, name)
            loss = loss + (v * torch.randn(v.shape, generator=g)).sum()
        loss.backward()
        local = model.flat_grad.clone()
        train_host.allreduce_gradients(model, world)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect = sum(gathered) / world
        assert torch.allclose(model.flat_grad, expect, rtol=0, atol=1e-6)
        if step == 1:
            # chunked path: all-reduce + Adam per chunk (allreduce_and_step) == all-reduce, then one Adam step
            twin = train_host.GaussianParams(scene, torch.device("cpu"))
            twin_opt = train_host.make_optimizer(twin)
            with torch.no_grad():
                twin.flat.copy_(model.flat); twin.flat_grad.copy_(local)
            twin_opt.exp_avg.copy_(opt.exp_avg); twin_opt.exp_avg_sq.copy_(opt.exp_avg_sq); twin_opt.step_count = opt.step_count
            train_host.allreduce_and_step(twin, twin_opt, world, chunks=5, average=True)
            opt.step()
            assert torch.equal(twin.flat, model.flat) and torch.equal(twin_opt.exp_avg_sq, opt.exp_avg_sq)
        elif step == 2:
            # split path of the step pipeline: the SH part of the bucket is all-reduced EARLY (while, on the GPU, the last
            # view's geometry backward still runs -- here the geometry gradients are simply written afterwards), the
            # geometry part later; must equal one all-reduce over the whole bucket followed by one Adam step, bit for bit
            twin = train_host.GaussianParams(scene, torch.device("cpu"))
            twin_opt = train_host.make_optimizer(twin)
            assert twin.offsets["_features"][0] == 257 * 17 and twin.NAMES[-1] == "_features"
            geo_end = train_host._sh_split(twin)     # 4369 -> 4372: cut on a float4 boundary
            assert geo_end == 4372
            with torch.no_grad():
                twin.flat.copy_(model.flat)
                twin.flat_grad[geo_end:].copy_(local[geo_end:])            # SH gradients final ...
                twin.flat_grad[:geo_end].fill_(float("nan"))               # ... geometry gradients not written yet
            twin_opt.exp_avg.copy_(opt.exp_avg); twin_opt.exp_avg_sq.copy_(opt.exp_avg_sq); twin_opt.step_count = opt.step_count
            handle = train_host.allreduce_sh_begin(twin, world, chunks=3)
            with torch.no_grad():
                twin.flat_grad[:geo_end].copy_(local[:geo_end])            # the geometry backward finishes
            train_host.allreduce_and_step(twin, twin_opt, world, average=True, sh_handle=handle)
            opt.step()
            assert torch.equal(twin.flat_grad, model.flat_grad)
            assert torch.equal(twin.flat, model.flat) and torch.equal(twin_opt.exp_avg_sq, opt.exp_avg_sq)
        else:
            opt.step()
        flats = [torch.zeros_like(model.flat) for _ in range(world)]
        dist.all_gather(flats, model.flat.detach())
        assert torch.equal(flats[0], flats[1]), "replicas diverged"
    # the exchange of staged SH gradients (few views per step: all-gather of 32 B per Gaussian and view instead of the dense
    # all-reduce): every rank ends up with everybody's stages, rank-major, bit-identical
    B, P = 3, 257
    mine = torch.randn(B, P, 8, generator=torch.Generator().manual_seed(500 + rank))
    work, stages = train_host.gather_sh_stages_begin(mine, world)
    work.wait()
    want = torch.cat([torch.randn(B, P, 8, generator=torch.Generator().manual_seed(500 + r)) for r in range(world)])
    assert stages.shape == (world * B, P, 8) and torch.equal(stages, want)
    # ... and view by view (the step pipeline starts a view's exchange right behind its SH backward): [B, world, P, 8]
    gathered = torch.empty(B, world, P, 8)
    works = [train_host.gather_view_stage_begin(mine[b], gathered[b]) for b in range(B)]
    for w in works:
        w.wait()
    for r in range(world):
        assert torch.equal(gathered[:, r], want[r * B:(r + 1) * B])
    q.put((rank, float(model.flat.detach().abs().sum())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_bucket_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res[0] == res[1]


def _stats_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fdgs import harness
    P, B = 300, 2
    # every rank can generate everybody's views (seeded): the union is what a single process with batch B*world would see
    def views(step, r):
        g = torch.Generator().manual_seed(1000 * step + r)
        return [{"radii": torch.randint(-3, 7, (P,), generator=g).clamp(min=0).to(torch.int32),
                 "viewspace_grad": torch.randn(P, 3, generator=g)} for _ in range(B)]
    st = harness.DensificationStats(P, "cpu", world)
    single = harness.DensificationStats(P, "cpu", 1)
    for step in range(3):
        t_grad = torch.randn(P, 1, generator=torch.Generator().manual_seed(77 + step))  # the (already all-reduced) dL/dt
        st.update(views(step, rank), t_grad, B * world)
        single.update([v for r in range(world) for v in views(step, r)], t_grad, B * world)
    for a, b in ((st.xyz_gradient_accum, single.xyz_gradient_accum), (st.t_gradient_accum, single.t_gradient_accum),
                 (st.denom, single.denom), (st.max_radii2D, single.max_radii2D)):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
    # identical on every rank -> densification decisions derived from them need no broadcast
    packed = torch.cat([st.xyz_gradient_accum.flatten(), st.t_gradient_accum.flatten(), st.denom.flatten(), st.max_radii2D])
    both = [torch.zeros_like(packed) for _ in range(world)]
    dist.all_gather(both, packed)
    assert torch.equal(both[0], both[1])
    q.put((rank, float(packed.abs().sum())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_densification_stats_identical_on_all_ranks_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stats_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res[0] == res[1]
