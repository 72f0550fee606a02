#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X: train-step images/s (+ forward Mpix/s) on the
C3 workload (300 k 4D Gaussians, 1352x1014, SH degree 3 + time degree 2 (M = 48), rot_4d + cov_t).

One *step* = one optimizer step = one pass of the hot path over a batch of B views per rank (default B = 4, the
reference's DyNeRF batch size), structured like the reference's training iteration (train.py:104-166, 247-249);
per view:
    activations (exp / sigmoid / normalize; fused into the preprocess kernels, --reference-host: PyTorch) ->
    render forward (HIP) -> (1-l) L1 + l (1 - SSIM) (fused HIP kernel, --torch-loss: PyTorch conv2d) ->
    backward (HIP; gradients land directly in the flat bucket) ->
then once per step the optimizer step (Adam, torch.optim.Adam arithmetic):
    N = 1: the SH coefficients (89 % of the parameters) are updated straight from the views' staged SH gradients by one fused
           kernel (fdgs_adam_step_sh), the 17 geometry floats per Gaussian by fdgs_adam_step;
    N > 1: the exchange over RCCL -- up to 32 views per step over all ranks: every view's SH stage (32 B per Gaussian) is
           all-gathered as soon as its SH backward has run, i.e. while the following views are rendered, + all-reduce of the
           geometry gradients at the end, then the same fused update on every rank; more views: all-reduce of the flat
           161*P-float gradient bucket, then Adam over it (--dense-sh-exchange forces this).
Frames / timesteps shard embarrassingly: rank r renders timestamp (r + 0.5) / N of the sequence with
replicated parameters (scaling = "weak": B views per GPU per step).  Inputs are synthetic
(fdgs.synth, seed 0) and resident in HBM before the timed region.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]
        N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                    --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line (see README / DESIGN.md for the field definitions).
"""
import argparse
import gc
import glob
import json
import os
import sys
import time
import weakref

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# Several ranks: the step's own streams (caller's, forward, backward, the clock sampler) already fill the runtime's default of 4
# hardware queues, and RCCL brings streams of its own; streams that share a hardware queue give steps that take several times as
# long now and then (measured on one GPU with a third stream in the step: DESIGN.md section 5a).  One rank stays on the default (8
# queues cost the two-stream step 1 %).  Must be set before the HIP runtime starts; inherited by the ranks a self-launch starts.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or any(a == "--gpus" and i + 1 < len(sys.argv) and sys.argv[i + 1] not in ("1", "0")
                                                      for i, a in enumerate(sys.argv)) or any(a.startswith("--gpus=") and a[7:] not in ("1", "0") for a in sys.argv):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

PROFILE_EVERY = int(os.environ.get("FDGS_BENCH_PROFILE_EVERY", "5"))   # the dominant stage's live event pairs: every n-th launch
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)

# Algorithmic HBM bytes per launch of each stage (BASELINE.md section 3 / SURVEY.md section 8d):
# P Gaussians, Pv visible, M SH coefficients, R instances, N pixels, T tiles.
ALGO_BYTES = {
    "preprocess_fwd": lambda P, Pv, M, R, N, T: 68 * P + (12 * M + 87) * Pv,
    "blend_fwd": lambda P, Pv, M, R, N, T: 52 * R + 32 * N,
    "blend_bwd": lambda P, Pv, M, R, N, T: 96 * R + 36 * N,
    # BASELINE.md's preprocess-bwd figure (68 + 24 M + 150) P_v + cov2D-bwd 104 P_v, split over our two kernels:
    # sh_bwd per mode; Pv here = the LIVE Gaussians (those that carry a colour gradient): the kernel scans
    # the colour words of all P records and evaluates the live ones only.  Direct: dL_dsh rows read-modify-written;
    # deferred (sh_stage): the row is only read, 32 B of stage written instead
    "sh_bwd": lambda P, Pv, M, R, N, T: 12 * P + (24 * M + 44) * Pv,
    "sh_bwd_deferred": lambda P, Pv, M, R, N, T: 12 * P + (12 * M + 76) * Pv,
    "preprocess_bwd": lambda P, Pv, M, R, N, T: (68 + 150 + 104 + 64 - 44) * Pv,  # incl. the fused cov2D backward + record read
    # tile binning (csrc/tilebin.hip): count reads the rectangles, scatter writes one (depth bits, id) pair per instance,
    # the per-tile local sort reads the pairs and writes point_list + ranges
    "tile_count": lambda P, Pv, M, R, N, T: 8 * P + 4 * T,
    "tile_scan": lambda P, Pv, M, R, N, T: 8 * T,
    "tile_scatter": lambda P, Pv, M, R, N, T: 12 * P + 8 * R + 8 * T,
    "tile_sort": lambda P, Pv, M, R, N, T: 12 * R + 12 * T,
    "grad_zero": lambda P, Pv, M, R, N, T: 64 * P,
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps before anything is measured (the shader clock takes ~100 ms of load to settle)")
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C5", "C3x4", "C3-clustered"])
    ap.add_argument("--views-per-step", type=int, default=4,
                    help="views rendered per rank and optimizer step.  4 (default): the reference's DyNeRF batch per GPU "
                         "(configs/dynerf/*.yaml:7; gradients accumulated, train.py:104-166), weak scaling: global batch "
                         "4 N.  1: BASELINE configs[3] as specified (N timesteps frame-parallel across N GPUs, one view "
                         "per rank and step, one gradient all-reduce per view)")
    ap.add_argument("--min-warmup-ms", type=float, default=150.0,
                    help="the warm-up goes on (whole steps) until this much wall time has passed, whatever --warmup says: the shader clock "
                         "needs ~100 ms of load to settle")
    ap.add_argument("--min-timed-ms", type=float, default=1000.0,
                    help="the timed region is --steps steps repeated (whole multiples) until it lasts at least this long; `steps_timed` in the "
                         "JSON line says how many steps that was (0: exactly --steps)")
    ap.add_argument("--reflists-steps", type=int, default=20,
                    help="steps of the extra leg with the reference's bit-identical tile lists (tile_cull = 0): value_reference_lists (0 = skip)")
    ap.add_argument("--clustered-steps", type=int, default=10,
                    help="workload C3 only: steps of the extra leg on C3-clustered (70 %% of the Gaussians on 15 %% of the image: tile lists of "
                         "thousands of entries, some beyond the 4096 the LDS sort takes) -- what the skew of a trained scene costs (0 = skip)")
    ap.add_argument("--cameras", choices=("rig", "axis"), default="rig",
                    help="rig (default): the views of a step come from DIFFERENT rotated, off-axis cameras (fdgs.synth.POSES rig0..rig3 by global "
                         "view index, one of them with the centre-shift projection) -- every real step does (a DyNeRF rig has ~20, "
                         "scene/cameras.py:65-71); axis: every view through the one unrotated on-axis camera of rounds 1-4 (timed as the "
                         "secondary leg value_axis_camera either way)")
    ap.add_argument("--axis-steps", type=int, default=20,
                    help="steps of the extra leg that re-times the step with the OTHER --cameras setting (0 = skip)")
    ap.add_argument("--c5-steps", type=int, default=6,
                    help="workload C3, N = 1 only: steps of the extra leg on C5 (BASELINE configs[4]: 2 M Gaussians, 2704x2028, the HBM stress "
                         "configuration): images/s, per-stage GB/s, end-to-end algorithmic GB/s, roofline of its dominant kernel (0 = skip)")
    ap.add_argument("--no-sparse-lists", action="store_true",
                    help="A/B: lazy forwards with the tile lists packed back to back (count + scan launches in the forward) instead of "
                         "fdgs_forward_out.sparse_lists (every tile's list at a fixed offset of the binning buffer: the same lists, no count / scan)")
    ap.add_argument("--no-overlap-steps", action="store_true",
                    help="A/B: StepPipeline(overlap_steps=False) -- every step starts behind the previous step's SH update (default: geometry, "
                         "binning and sort of a step's first view run next to it; one rank, two streams)")
    ap.add_argument("--no-lazy", action="store_true",
                    help="A/B: every forward waits for its num_rendered (as the reference does) instead of fdgs_forward_out.lazy")
    ap.add_argument("--cpu-samples", type=int, default=2, help="oracle forward+backward passes timed for cpu_baseline (0 = skip)")
    ap.add_argument("--host-cost-steps", type=int, default=30,
                    help="steps of the tiny-scene leg that measures the host cost per view (0 = skip, e.g. under rocprofv3)")
    ap.add_argument("--dropin-steps", type=int, default=3,
                    help="N = 1 only: steps of the DROP-IN leg -- a reference-style model (separate parameters, torch.cat features, "
                         "PyTorch activations, torch.optim.Adam over 9 groups) through this package's render() + autograd + the "
                         "reference's PyTorch loss: what a user of the reference gets by swapping one import (0 = skip)")
    ap.add_argument("--storage-order", choices=("morton", "random"), default="morton",
                    help="how the model stores its Gaussians.  morton (default): in Morton order of their positions, what fdgs.harness.train "
                         "keeps at the start and after every densification (train_host.spatial_sort; a memory layout, no effect on the arithmetic: "
                         "the same images, DESIGN.md section 4.6b); random: the generator's order.  The other order is timed as a secondary leg "
                         "(random_order_images_s / spatial_order_images_s)")
    ap.add_argument("--spatial-order", action="store_true", help="(kept for old command lines) = --storage-order morton")
    ap.add_argument("--spatial-order-steps", type=int, default=20,
                    help="steps of the extra leg that re-times the step with the model stored in Morton order of the Gaussians' "
                         "positions (train_host.spatial_sort -- what fdgs.harness.train keeps after every densification) instead of the "
                         "generator's random order: spatial_order_images_s / spatial_order_forward_ms (0 = skip)")
    ap.add_argument("--no-loss", action="store_true", help="debug: sum() loss instead of L1 + SSIM")
    ap.add_argument("--reference-host", action="store_true",
                    help="host side exactly as the reference: render() on PyTorch activations, autograd gradient accumulation")
    ap.add_argument("--torch-loss", action="store_true", help="L1 + SSIM through PyTorch conv2d (MIOpen) instead of the fused HIP kernel")
    ap.add_argument("--autograd", action="store_true",
                    help="fused kernels driven through autograd (render_raw + fused_l1_ssim + backward()) on one stream, "
                         "instead of the explicit two-stream step pipeline (fdgs/pipeline.py)")
    ap.add_argument("--no-overlap", action="store_true", help="step pipeline on a single stream (A/B for the overlap)")
    ap.add_argument("--sh-group", type=int, default=1,
                    help="SH backward of this many consecutive views in one pass over the coefficients (fdgs_sh_backward_batch); 1: per view")
    ap.add_argument("--batch-views", action="store_true",
                    help="A/B: SH colours and SH backward of the step's views in one pass over the coefficients each "
                         "(fdgs_preprocess_batch / fdgs_sh_backward_batch): fewer bytes, but a serial head and tail of the step")
    ap.add_argument("--split-colour", choices=("forward", "all", "off"), default="forward",
                    help="fdgs_forward_out.split_colour (SH colours on the library's second stream next to the binning): in the "
                         "forward-only loop (default), also in the training step, or nowhere")
    ap.add_argument("--no-tile-cull", action="store_true",
                    help="A/B: the reference's tile lists (every tile of the 3-sigma square) instead of fdgs_forward_out.tile_cull "
                         "(a Gaussian listed only where it can reach alpha >= 1/255: same pixels and gradients)")
    ap.add_argument("--dense-sh-exchange", action="store_true",
                    help="N > 1: always all-reduce the dense SH gradient (default: up to 32 views per step over all ranks exchange "
                         "the views' 32-byte SH stages by all-gather instead, train_host.gather_view_stage_begin)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec through torch.distributed.run (one process per GPU, RCCL),
    rendezvous on 127.0.0.1 with a free port.  Returns the children's exit code."""
    import socket
    import subprocess
    share = bool(os.environ.get("FDGS_BENCH_DEBUG_SHARE_GPU"))
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%d" % (args.gpus, world))
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ.get("FDGS_BENCH_DEBUG_SHARE_GPU"):
            # debug only: exercise the multi-rank control flow on a 1-GPU box (all ranks on cuda:0, gloo collectives)
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            if torch.cuda.device_count() <= local:
                raise SystemExit("bench.py: rank %d needs cuda:%d but only %d GPU(s) are visible" % (rank, local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        backend = dist.get_backend()
        assert dist.get_world_size() == world
    return world, rank, local, backend


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reference_kernels_on_this_gpu(scene, dev, passes=20):
    """Part of the baseline leg: the REFERENCE'S OWN CUDA kernels compiled for gfx950 (oracle/_ref/liboracle_ref_hip.so -- torch.utils.hipify
    + hipcc on a temporary copy of /root/reference's sources in the build container, oracle/refbuild/build_ref_hip.py; the .so travels, the
    sources do not) on the same scene, the same four upstream gradients, this GPU: rasterizer forward + backward per image incl. the
    zero-fills rasterize_points.cu does around them (:80-92, :201-213), synchronised per pass as `loss.backward()` + `optimizer.step()`
    leave it in the reference's loop.  The like-for-like partner of `raster_images_s`.  None when the library was not built."""
    from fdgs import synth
    from oracle import ref_hip
    if not ref_hip.available():
        return None
    g = {k: v.to(dev) for k, v in synth.make_upstream_grads(scene["W"], scene["H"], seed=1, scale=1e-2).items()}
    ref = ref_hip.RefHip(scene, dev)
    for _ in range(3):
        ref.forward()
        ref.backward(g["grad_color"], g["grad_depth"], g["grad_alpha"], g["grad_flow"])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(passes):
        ref.forward()
    t_f = (time.perf_counter() - t0) / passes
    t0 = time.perf_counter()
    for _ in range(passes):
        ref.forward()
        ref.backward(g["grad_color"], g["grad_depth"], g["grad_alpha"], g["grad_flow"])
    t_fb = (time.perf_counter() - t0) / passes
    ref_hip.lib().refhip_free()
    return {"images_s": round(1.0 / t_fb, 2), "ms_per_image": round(t_fb * 1e3, 3), "forward_ms": round(t_f * 1e3, 3), "num_rendered": int(ref.R),
            "kind": "reference (its own forward.cu / backward.cu / rasterizer_impl.cu, hipified and compiled for gfx950; hipcub radix sort)",
            "what": "rasterizer forward + backward of the same %s scene, all four upstream gradients, %d passes on this GPU" % (scene["cfg"].name, passes)}


def cpu_baseline(scene, samples):
    """The oracle ("port": scalar C restatement of the reference kernels, OpenMP over Gaussians / tiles)
    timed on this box's host cores on a bounded sample of the same workload: `samples` full rasterizer
    forward + backward passes of the same scene (no loss / optimizer: those are PyTorch on both sides)."""
    from fdgs import synth
    from oracle import pyoracle
    o = pyoracle.Oracle(scene, kind="port")
    g = synth.make_upstream_grads(scene["W"], scene["H"], seed=1, scale=1e-2)
    o.forward()  # touch pages / build the library
    t0 = time.perf_counter()
    for _ in range(samples):
        o.close()
        o.forward()
        o.backward(g["grad_color"], g["grad_depth"], g["grad_alpha"], g["grad_flow"])
    dt = time.perf_counter() - t0
    o.close()
    return {"value": samples / dt, "unit": "images/s", "cores": pyoracle.threads("port"), "kind": "port",
            "sample": "%d rasterizer forward+backward passes of the same %s scene (%.1f s of CPU work); "
                      "oracle/fdgs_oracle.c, gcc -O2 -fopenmp" % (samples, scene["cfg"].name, dt)}


def dropin_leg(args, scene, cams, gts, pipe, bg, dev, B):
    """What a user of the reference gets by swapping ONE import (gaussian_renderer.render / GaussianRasterizer -> this package) and
    keeping everything else of train.py:104-170, 247-249: a reference-style model (separate parameters, torch.cat'ed features,
    PyTorch activations), render() through autograd, the reference's PyTorch L1 + SSIM (utils/loss_utils.py), torch.optim.Adam over
    the reference's param groups, zero_grad(set_to_none=True).  Also the pieces, timed one by one, so that the gap to the step
    pipeline (`value`) can be attributed."""
    from fdgs import train_host
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    rm = train_host.ReferenceStyleModel(scene, dev)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / n * 1e3

    def step(loss_fn):
        for b in range(B):
            pkg = render(cams[b], rm, pipe, bg)
            loss = loss_fn(pkg["render"], gts[b])
            (loss / B).backward()                       # train.py:162
        rm.optimizer.step()                             # train.py:247-249
        rm.optimizer.zero_grad(set_to_none=True)

    n = max(1, args.dropin_steps)
    ms_step = timed(lambda: step(train_host.photometric_loss), n)
    ms_step_fused = timed(lambda: step(lambda a, g: fused_l1_ssim(a, g, 0.2)), n)

    def fwd():
        with torch.no_grad():
            render(cams[0], rm, pipe, bg)

    def fwd_bwd():
        pkg = render(cams[0], rm, pipe, bg)
        (pkg["render"].sum() * 1e-6).backward()
        rm.optimizer.zero_grad(set_to_none=True)

    img = torch.rand(3, scene["H"], scene["W"], device=dev)

    def torch_loss():
        x = img.clone().requires_grad_(True)
        train_host.photometric_loss(x, gts[0]).backward()

    def optim():
        for g in rm.optimizer.param_groups:
            for q in g["params"]:
                if q.grad is None:
                    q.grad = torch.zeros_like(q)
        rm.optimizer.step()

    ms_fwd, ms_fb, ms_loss, ms_opt = timed(fwd, 8), timed(fwd_bwd, 4), timed(torch_loss, 4), timed(optim, 3)
    rm.optimizer.zero_grad(set_to_none=True)
    del rm

    # The reference's loop with the three swaps INTEGRATION.md describes: this package's render(), fdgs.loss.fused_l1_ssim for the
    # loss expression, fdgs.optim.Adam for torch.optim.Adam in training_setup -- model class, getters, densification code, loop
    # structure (autograd, one view after the other, optimizer.step / zero_grad) untouched.
    import fdgs.gaussian_renderer as gr
    fast = {}
    for tag, lazy, cull in (("", False, False), ("_lazy", True, False), ("_lazy_tile_cull", True, True)):
        fm = train_host.ReferenceStyleModel(scene, dev, optimizer="fdgs")
        fm.optimizer.lazy_forward = lazy
        gr.render_options["tile_cull"] = cull
        try:
            def fstep():
                for b in range(B):
                    pkg = render(cams[b], fm, pipe, bg)
                    (fused_l1_ssim(pkg["render"], gts[b], 0.2) / B).backward()
                fm.optimizer.step()
                fm.optimizer.zero_grad(set_to_none=True)

            for _ in range(3):
                fstep()
            fast["images_s_fdgs_optim" + tag] = round(B * 1e3 / timed(fstep, max(n, 10)), 2)

            def ffwd():
                with torch.no_grad():
                    render(cams[0], fm, pipe, bg)

            if not tag:
                fast["forward_ms_fdgs_optim"] = round(timed(ffwd, 16), 4)
            fast["steps_skipped" + tag] = fm.optimizer.steps_skipped
        finally:
            gr.render_options["tile_cull"] = False
        del fm
    return {"images_s": round(B * 1e3 / ms_step, 2), "ms_per_image": round(ms_step / B, 4), "forward_ms": round(ms_fwd, 4),
            "images_s_with_fused_loss": round(B * 1e3 / ms_step_fused, 2),
            "fdgs_optim": dict(fast, what="the same loop with fdgs.optim.Adam (flat bucket behind the model's own Parameters, gradients "
                                          "written straight into it, SH update from the staged views) + fused loss; _lazy: optimizer.lazy_forward "
                                          "(no forward waits for num_rendered); _tile_cull: render_options['tile_cull']"),
            "pieces_ms": {"render_forward": round(ms_fwd, 4), "render_forward_backward_autograd": round(ms_fb, 4),
                          "pytorch_l1_ssim_forward_backward": round(ms_loss, 4), "torch_adam_step_per_step": round(ms_opt, 4)},
            "steps": n,
            "what": "reference-style model + render() + autograd + PyTorch L1/SSIM (utils/loss_utils.py) + torch.optim.Adam (9 groups): "
                    "the reference's train.py:104-170 with one import swapped; images_s_with_fused_loss: the same with fdgs.loss.fused_l1_ssim"}


_PIPES = weakref.WeakSet()


def new_pipeline(*a, **kw):
    """StepPipeline, registered so that models_touched() reaches it"""
    from fdgs.pipeline import StepPipeline
    p = StepPipeline(*a, **kw)
    _PIPES.add(p)
    return p


def warm_up(step_fn, dev, world=1, at_least=3, at_most=10):
    """Untimed steps in front of a timed leg, until the caller's allocator is quiet: the host runs up to a mailbox ring of forwards ahead
    of the device and every forward in flight holds its binning buffer (C3-clustered: 570 MB, C5: 656 MB with sparse lists) -- that
    depth, hence the last device allocations (tens of ms each on a busy GPU: a leg of 6-10 steps that contained four of them read
    half its rate), is only reached after three or four steps.  Several ranks: a fixed number (a step holds collectives)."""
    # Round 6: a leg REHEARSES its own trajectory -- `at_least` = the number of steps it will time, from the model state it will start
    # from (the leg restores that state again afterwards): Adam on noise targets shrinks the scene step by step, the run-ahead buffer
    # sizes follow it, and ONE device allocation inside a leg of 20 steps (25-33 ms next to a busy device: `reference_lists` read 1100
    # instead of 1720 images/s in every third run of rounds 5 and 6, `ms_per_step_max` 26 ms) is a third of its reading.
    gc.collect()   # (the collector is off during the legs: bench.py main)
    at_most = max(at_most, at_least + 6)
    for k in range(at_most):
        n0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
        step_fn()
        if world > 1:
            if k + 1 >= at_least + 2:
                break
        elif k + 1 >= at_least and torch.cuda.memory_stats(dev)["num_device_alloc"] == n0:
            break


def models_touched():
    """The bench has written parameters / optimizer state through torch on its own stream (the restores between legs and repetitions):
    a StepPipeline(overlap_steps=True) starts its next step only behind that (StepPipeline.barrier)."""
    for p in list(_PIPES):
        p.barrier()


def step_end_stages(dev, P, M, B, W, H, D, D_t, gaussian_dim, force_sh_3d, iters=20, workload=None):
    """The stages of a training step that are NOT inside the rasterizer's profile table: the fused SH flush + Adam of the SH coefficients
    (fdgs_adam_step_sh: the step's one pass over 89 % of the parameters), the Adam step of the 17 geometry floats per Gaussian
    (fdgs_adam_step) and the fused L1 + SSIM loss (value partials + gradient, two launches) -- each timed ALONE with HIP events on
    buffers of the workload's size (not the model's: nothing of the run is touched), with its algorithmic bytes
    and the fraction of the HBM peak they make.  `working_set_bytes` says whether the launch can live in the 256 MB Infinity Cache
    from one iteration to the next (the geometry Adam at C3 does: its figure is not an HBM figure)."""
    from fdgs import _capi
    from fdgs.loss import l1_ssim_grad
    out = {}

    def timed(fn, n):
        # the fastest of five back-to-back batches of n launches (mean of the batch): a stand-alone kernel time, not a sample of what else
        # the device was settling from
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        best = float("inf")
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize(dev)
            best = min(best, a.elapsed_time(b) / n)
        return best

    def entry(ms, algo, working):
        return {"ms": round(ms, 4), "algo_bytes": int(algo), "gbps": round(algo / (ms * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "working_set_bytes": int(working)}
    g = torch.Generator(device="cpu").manual_seed(0)
    if M > 0 and (3 * M) % 4 == 0:
        st = torch.randn(B, P, 8, generator=g)
        st[:, :, 0:3][torch.rand(B, P, generator=g) < 0.6] = 0     # 40 % of the Gaussians live per view, as on C3
        st = st.to(dev)
        prm = torch.randn(P, M, 3, device=dev)
        m1, m2 = torch.zeros_like(prm), torch.zeros_like(prm)
        ms = timed(lambda: _capi.adam_step_sh(prm, m1, m2, st, D, D_t, gaussian_dim, force_sh_3d, False, 1e-4, 2.5e-3, 0.9, 0.999, 1e-15, 1), iters)
        algo = 24 * 3 * M * P + 32 * B * P      # parameter + two moments read and written, the views' 32-byte stage records read
        out["sh_adam"] = entry(ms, algo, algo)
        del st, prm, m1, m2
    n_geo = 17 * P
    flat = torch.randn(n_geo, device=dev)
    grad = torch.randn(n_geo, device=dev)
    m1, m2 = torch.zeros_like(flat), torch.zeros_like(flat)
    seg = (_capi.FdgsAdamSegment * 1)(_capi.FdgsAdamSegment(0, n_geo, 1e-4, 1e-4, 0, 0))

    def geo():
        with torch.cuda.device(dev):
            _capi._check(_capi.lib.fdgs_adam_step(flat.data_ptr(), grad.data_ptr(), m1.data_ptr(), m2.data_ptr(), n_geo, seg, 1,
                                                  0.9, 0.999, 1e-15, 1, _capi.current_stream_handle(dev)), "fdgs_adam_step")
    ms = timed(geo, iters)
    out["geometry_adam"] = entry(ms, 28 * n_geo, 28 * n_geo)
    del flat, grad, m1, m2
    img = torch.rand(3, H, W, generator=g).to(dev)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    up = torch.ones((), device=dev)
    ms = timed(lambda: l1_ssim_grad(img, gt, 0.2, up), iters)
    N = W * H
    # forward: both images in, three derivative maps out; backward: the maps and both images in, the gradient out (3 channels, 4 bytes)
    out["l1_ssim"] = entry(ms, (24 + 36 + 60 + 12) * N, (24 + 36 + 12) * N)
    if workload in ("C3", "C5"):
        # HBM bytes per launch from the committed counter passes of the same step (not this run): read + write, gfx950 corrections applied
        for k, stage in (("sh_adam", "sh_flush"), ("geometry_adam", "adam")):
            if k in out:
                out[k]["traffic"] = pmc_traffic(stage, workload)
        if "l1_ssim" in out:
            f, b = pmc_traffic("ssim_fwd", workload), pmc_traffic("ssim_bwd", workload)
            out["l1_ssim"]["traffic"] = None if f is None or b is None else f + b
    best = max((k for k in out if out[k]["working_set_bytes"] > 256 << 20), key=lambda k: out[k]["frac_of_hbm_peak"], default=None)
    if best is not None:
        out["hbm_bound_best"] = {"stage": best, "frac_of_hbm_peak": out[best]["frac_of_hbm_peak"], "gbps": out[best]["gbps"], "ms": out[best]["ms"],
                                 "algo_bytes": out[best]["algo_bytes"], "traffic": out[best].get("traffic")}
    out["what"] = ("stand-alone HIP-event times (the fastest of five batches of launches, mean of the batch) of the step's stages outside the rasterizer (buffers of the workload's size): sh_adam = the fused SH flush + "
                   "Adam over the SH coefficients (24 B per coefficient + 32 B per Gaussian and view), geometry_adam = 28 B per geometry float, "
                   "l1_ssim = value partials + gradient (132 B per pixel); hbm_bound_best = the best of those whose working set exceeds the "
                   "256 MB Infinity Cache")
    return out


def c5_leg(args, dev, make_cams, pipe, B):
    """BASELINE configs[4] (2 M Gaussians, 2704x2028, SH degree 3: "HBM-bound stress; rocprof GB/s vs roofline") through the same
    step as `value`: images/s on two streams, the per-stage table of a single-stream pass with every stage's algorithmic bytes and
    GB/s, the end-to-end algorithmic GB/s (all stages' bytes / wall time per view) and the roofline entry of its dominant kernel
    with the HBM traffic of the committed C5 counter passes (profiles/pmc_traffic_r??_C5.json)."""
    from fdgs import _capi, synth, train_host
    cfg = synth.CONFIGS["C5"]
    scene = synth.make_scene(cfg, seed=0)
    model = train_host.GaussianParams(scene, dev)
    opt = train_host.make_optimizer(model)
    if args.storage_order == "morton":
        train_host.spatial_sort(model, opt)
    snap = (model.flat.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())

    def restore():
        model.flat.data.copy_(snap[0]); opt.exp_avg.copy_(snap[1]); opt.exp_avg_sq.copy_(snap[2]); opt.step_count = 0
        models_touched()

    cams = make_cams(scene, args.cameras)
    bg = scene["bg"].to(dev)
    W, H = scene["W"], scene["H"]
    gts = [torch.rand(3, H, W, generator=torch.Generator(device="cpu").manual_seed(4321 + b)).to(dev) for b in range(B)]
    kw = dict(world_size=1, lambda_dssim=0.2, tile_cull=not args.no_tile_cull, lazy=not args.no_lazy, sparse_lists=not args.no_sparse_lists)
    # single-stream stage pass (kernel time per stage)
    sp1 = new_pipeline(model, opt, overlap=False, **kw)
    sp1.step(cams, gts, pipe, bg)
    restore()
    torch.cuda.synchronize(dev)
    _capi.profile_reset()
    _capi.profile_enable(True)
    res = sp1.step(cams, gts, pipe, bg)[0]
    torch.cuda.synchronize(dev)
    _capi.profile_enable(False)
    prof = _capi.profile_read()
    del sp1
    R = sum(r["num_rendered"] for r in res) / len(res)
    Pv = sum(int((r["radii"] > 0).sum().item()) for r in res) / len(res)
    P_live = sum(int((r["viewspace_grad"] != 0).any(dim=1).sum().item()) for r in res) / len(res)
    P, M, N, T = model.P, model.M, W * H, ((W + 15) // 16) * ((H + 15) // 16)
    stages, total_bytes = {}, 0
    for name, (ms, n) in prof.items():
        if n == 0:
            continue
        e = {"ms": round(ms / n, 4)}
        key = "sh_bwd_deferred" if name == "sh_bwd" else name
        if key in ALGO_BYTES:
            b = ALGO_BYTES[key](P, P_live if name == "sh_bwd" else Pv, M, R, N, T)
            e["algo_bytes"] = int(b)
            e["gbps"] = round(b / (ms / n * 1e-3) / 1e9, 1) if ms > 0 else None
            e["frac_of_hbm_peak"] = round(b / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None
            # the bytes the kernel really moved (committed C5 counter passes, not this run) over THIS run's time: what the memory system was
            # doing, next to what the algorithm needed (traffic / algo_bytes > 1: partial lines, read-modify-write of the gradient
            # accumulators, the re-zeroing of the records)
            tr = pmc_traffic(name, "C5")
            if tr and ms > 0:
                e["traffic"] = int(tr)
                e["frac_of_hbm_peak_by_traffic"] = round(tr / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            total_bytes += b
        stages[name] = e
    # the timed two-stream steps, the dominant kernel re-measured live
    dom = max((k for k in prof if k != "readback"), key=lambda k: prof[k][0])
    sp2 = new_pipeline(model, opt, overlap=not args.no_overlap, overlap_steps=not args.no_overlap_steps, **kw)
    restore()
    warm_up(lambda: sp2.step(cams, gts, pipe, bg), dev, at_least=max(3, args.c5_steps))
    restore()
    _capi.profile_reset()
    _capi.profile_enable(True, stages=[dom], every=PROFILE_EVERY)
    torch.cuda.synchronize(dev)
    dbg = int(os.environ.get("FDGS_BENCH_DEBUG", "0"))     # 1: allocator / run-ahead statistics of the timed region; 2: + a synchronise per step
    if dbg:
        ms0, ra0 = torch.cuda.memory_stats(dev), _capi.run_ahead_stats()
        seg0 = {(g["address"], g["total_size"]) for g in torch.cuda.memory_snapshot()}
    t0 = time.perf_counter()
    for _ in range(args.c5_steps):
        ts = time.perf_counter()
        sp2.step(cams, gts, pipe, bg)
        if dbg > 1:
            th = time.perf_counter()
            torch.cuda.synchronize(dev)
            print("c5 step host %.2f ms total %.2f ms" % ((th - ts) * 1e3, (time.perf_counter() - ts) * 1e3), file=sys.stderr)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dbg:
        ms1, ra1 = torch.cuda.memory_stats(dev), _capi.run_ahead_stats()
        print("c5 dbg: device allocs %d frees %d reserved %.1f GB run-ahead %s" % (
            ms1["num_device_alloc"] - ms0["num_device_alloc"], ms1["num_device_free"] - ms0["num_device_free"],
            ms1["reserved_bytes.all.current"] / 1e9, tuple(b - a for a, b in zip(ra0, ra1))), file=sys.stderr)
        print("c5 dbg: new segments (MB, stream)", sorted((g["total_size"] >> 20, g["stream"]) for g in torch.cuda.memory_snapshot()
                                                          if (g["address"], g["total_size"]) not in seg0), file=sys.stderr)
    _capi.profile_enable(False)
    pd = _capi.profile_read()[dom]
    dom_ms = pd[0] / max(pd[1], 1)
    dom_bytes = ALGO_BYTES[dom](P, Pv, M, R, N, T)
    ms_view = dt / (args.c5_steps * B) * 1e3
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    out = {"images_s": round(B * args.c5_steps / dt, 2), "ms_per_step": round(dt / args.c5_steps * 1e3, 3), "ms_per_image": round(ms_view, 4),
           "steps": args.c5_steps, "num_rendered": int(round(R)), "visible": int(round(Pv)), "live_gaussians": int(round(P_live)),
           "lazy_steps_redone": sp2.lazy_redone, "steps_started_under_the_previous_sh_update": sp2.steps_carried, "cameras": args.cameras,
           "raster_ms_single_stream": round(sum(v["ms"] for v in stages.values()), 4),
           "algo_bytes_per_view": int(total_bytes),
           # all stages' algorithmic bytes over the WALL time per view of the two-stream step (loss + optimizer included in the time)
           "end_to_end_algorithmic_gbps": round(total_bytes / (ms_view * 1e-3) / 1e9, 1),
           "end_to_end_frac_of_hbm_peak": round(total_bytes / (ms_view * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "stages": stages,
           "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, "C5"),
                        "traffic_source": "committed rocprofv3 --pmc passes of the same C5 step (profiles/pmc_traffic_r??_C5.json), not this run",
                        "avg_kernel_ms": round(dom_ms, 4), "avg_kernel_ms_single_stream": stages[dom]["ms"], "launches_timed": int(pd[1]), "launches_bracketed": "every %d-th" % PROFILE_EVERY,
                        "algo_bytes_per_launch": int(dom_bytes)},
           "what": "BASELINE configs[4] (C5: %d Gaussians, %dx%d, SH degree %d, M = %d) through the same two-stream step as `value`, %d views per step; "
                   "stages: one single-stream step, HIP events per stage" % (P, W, H, cfg.sh_degree, M, B)}
    del sp2, model, opt, snap
    torch.cuda.empty_cache()
    ses = step_end_stages(dev, P, M, B, W, H, cfg.sh_degree, cfg.sh_degree_t, cfg.gaussian_dim, cfg.force_sh_3d, iters=8, workload="C5")
    out["step_end_stages"] = ses
    # the wall time per view covers the loss and the optimizer step as well: their algorithmic bytes belong into the same sum (per view: the
    # loss once, the two Adam launches of the step divided by its B views)
    extra = ses.get("l1_ssim", {}).get("algo_bytes", 0) + (ses.get("sh_adam", {}).get("algo_bytes", 0) + ses.get("geometry_adam", {}).get("algo_bytes", 0)) / B
    out["algo_bytes_per_view_with_loss_and_optimizer"] = int(total_bytes + extra)
    out["end_to_end_frac_of_hbm_peak_with_loss_and_optimizer"] = round((total_bytes + extra) / (ms_view * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    torch.cuda.empty_cache()
    return out


def pmc_traffic(stage, workload="C3"):
    """HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes (tools/pmc_traffic.py) of ``workload``, or None."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_traffic_r??.json" if workload == "C3" else "pmc_traffic_r??_%s.json" % workload)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return d.get(stage, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def pmc_valu(stage):
    """VALU issue statistics of the dominant kernel from the committed SQ-counter pass (tools/pmc_sq.sh), or None."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_sq_r??.txt")))
    if not files:
        return None
    try:
        cur, vals = None, {}
        for line in open(files[-1]):
            t = line.split()
            if not line.startswith(" ") and t:
                cur = line.strip()
            elif cur and stage in cur and len(t) >= 2 and (t[0].startswith("SQ_") or t[0] == "_AVG_DURATION_NS"):
                vals[t[0]] = float(t[1])
        if "SQ_ACTIVE_INST_VALU" not in vals:
            return None
        simds = 256 * 4
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's SIMDs (MI355X_MICROARCH.md, s_memtime row)
        return {"counters_from": "committed pass (not this run): " + os.path.relpath(files[-1], ROOT),
                "insts_valu_per_launch": int(vals.get("SQ_INSTS_VALU", 0)),
                "valu_issue_cycles_per_simd": int(vals["SQ_ACTIVE_INST_VALU"] * 4 / simds),
                # the kernel's duration in the counter pass itself (its views' lists differ a little from the timed region's mean)
                "kernel_ns_in_counter_pass": vals.get("_AVG_DURATION_NS"),
                "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called as the driver calls the N = 1 bench: launch the N ranks ourselves
        raise SystemExit(self_launch(args))
    world, rank, local, backend = init_dist(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from fdgs import _capi, synth, train_host
    from fdgs.gaussian_renderer import render
    from fdgs.loss import fused_l1_ssim
    from fdgs.fused import render_raw

    cfg = synth.CONFIGS[args.workload]
    scene = synth.make_scene(cfg, seed=0)
    model = train_host.GaussianParams(scene, dev)
    opt = train_host.make_optimizer(model)
    if args.spatial_order:
        args.storage_order = "morton"
    snap_random = (model.flat.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count)
    if args.storage_order == "morton":
        train_host.spatial_sort(model, opt)
    pipe = train_host.PipelineFlags()
    B = max(1, args.views_per_step)
    RIG = ("rig0", "rig1", "rig2", "rig3")

    def make_cams(sc, which):
        """frame-parallel: the N*B views of one optimizer step are N*B different timestamps (and, --cameras rig, cameras: pose
        rig{g % 4} for global view index g); rank r takes views r*B .. r*B+B-1"""
        out = []
        for b in range(B):
            g = rank * B + b
            cs = sc if which == "axis" else dict(sc, **synth.camera_for(RIG[g % len(RIG)], sc["W"], sc["H"]))
            out.append(train_host.SyntheticCamera(cs, dev, timestamp=(g + 0.5) / (world * B) * sc["time_duration"]))
        return out

    cams = make_cams(scene, args.cameras)
    cam = cams[0]
    bg = scene["bg"].to(dev)
    # one target per GLOBAL view index: N ranks x B views see the same data as one rank x N B views
    gts = [torch.rand(3, scene["H"], scene["W"], generator=torch.Generator(device="cpu").manual_seed(1234 + rank * B + b)).to(dev)
           for b in range(B)]
    sink = None if args.reference_host else model.grad_sink()
    use_pipeline = not (args.reference_host or args.autograd or args.torch_loss or args.no_loss)
    if use_pipeline:
        StepPipeline = new_pipeline
        steppipe = StepPipeline(model, opt, world_size=world, lambda_dssim=0.2, overlap=not args.no_overlap,
                                gather_max_views=0 if args.dense_sh_exchange else 32, split_colour=args.split_colour == "all", tile_cull=not args.no_tile_cull,
                                batch_views=args.batch_views, sh_group=args.sh_group, lazy=not args.no_lazy, sparse_lists=not args.no_sparse_lists,
                                overlap_steps=not args.no_overlap_steps)

    def step():
        if use_pipeline:
            results, _losses = steppipe.step(cams, gts, pipe, bg)
            return results
        if args.reference_host:
            model.zero_grad()
        pkg = None
        for b in range(B):
            if args.reference_host:
                pkg = render(cams[b], model, pipe, bg)
            else:
                # fused activations; gradients written (first view) / added (further views) straight into the flat bucket
                pkg = render_raw(cams[b], model, pipe, bg, grad_sink=sink, accumulate=b > 0)
            if args.no_loss:
                loss = pkg["render"].sum() * 1e-6
            elif args.torch_loss:
                loss = train_host.photometric_loss(pkg["render"], gts[b])
            else:
                loss = fused_l1_ssim(pkg["render"], gts[b], 0.2)
            (loss / (B * world)).backward()  # train.py:162; 1 / world: the all-reduce SUM is then the mean
        train_host.allreduce_gradients(model, world, average=False)
        opt.step()
        return [pkg]

    # The targets are noise, so hundreds of Adam steps at the reference's learning rates drive the model away from the workload the
    # metric is quoted on (opacities and scales shrink: a third of the instances are left after 400 steps).  Every timed leg therefore
    # starts from the SAME state: parameters, Adam moments and step count are put back (three device copies, INSIDE the timed region
    # where one is timed: extra work, nothing skipped) after the warm-up and before every repetition of the --steps steps.
    snap = (model.flat.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count)

    def restore():
        model.flat.data.copy_(snap[0])
        opt.exp_avg.copy_(snap[1])
        opt.exp_avg_sq.copy_(snap[2])
        opt.step_count = snap[3]
        models_touched()

    # warm-up: --warmup steps, and on until --min-warmup-ms of wall time have passed (the driver's --warmup 5 is 15 ms of C3 steps;
    # the shader clock takes ~100 ms of load to settle)
    torch.cuda.synchronize(dev)
    tw = time.perf_counter()
    warm_steps = 0
    for _ in range(args.warmup):
        step()
        warm_steps += 1
    while args.min_warmup_ms > 0:
        # every rank takes the same number of steps (a step holds collectives): the decision to go on is taken together, on time
        # that has actually been worked (the host runs ahead of the device)
        torch.cuda.synchronize(dev)
        more = 1 if (time.perf_counter() - tw) * 1e3 < args.min_warmup_ms else 0
        if world > 1:
            import torch.distributed as dist
            t_more = torch.tensor([more], dtype=torch.int64, device=dev)
            dist.all_reduce(t_more, op=dist.ReduceOp.MAX)
            more = int(t_more.item())
        if not more:
            break
        for _ in range(4):
            step()
        warm_steps += 4
    torch.cuda.synchronize(dev)
    # Untimed stage pass on ONE stream (kernel time, not queueing time behind the other stream's launches): every
    # rasterizer stage bracketed with HIP events -> the per-stage table and the dominant stage.
    if use_pipeline:
        stage_pipe = StepPipeline(model, opt, world_size=world, lambda_dssim=0.2, overlap=False,
                                  gather_max_views=0 if args.dense_sh_exchange else 32, batch_views=args.batch_views, sh_group=args.sh_group,
                                  tile_cull=not args.no_tile_cull, lazy=not args.no_lazy, sparse_lists=not args.no_sparse_lists)
        stage_step = lambda: stage_pipe.step(cams, gts, pipe, bg)[0]  # noqa: E731
    else:
        stage_step = step
    stage_step()
    torch.cuda.synchronize(dev)
    _capi.profile_reset()
    _capi.profile_enable(True)
    n_stage_steps = max(1, min(3, args.steps))
    r_seen = []
    for _ in range(n_stage_steps):
        r_seen += [r["num_rendered"] for r in stage_step() if "num_rendered" in r]
    torch.cuda.synchronize(dev)
    _capi.profile_enable(False)
    prof_all = _capi.profile_read()
    R_stage = (sum(r_seen) / len(r_seen)) if r_seen else (sum(_R_LOG) / max(len(_R_LOG), 1))
    dom = max((k for k in prof_all if k != "readback"), key=lambda k: prof_all[k][0])
    # Timed region (the two-stream pipeline): only the dominant kernel keeps its event pair -- the roofline figure is
    # measured live here -- and one event per step boundary on the main stream gives the per-step distribution.
    _capi.profile_reset()
    # (every PROFILE_EVERY-th launch of it: an event pair costs the stream ~13 us of idle time around the launch -- 2 % of the step if
    # every blend backward carried one; 5 is coprime to the 4 views of a step, so the samples rotate through the views)
    _capi.profile_enable(True, stages=[dom], every=PROFILE_EVERY)
    _R_LOG.clear()
    # the timed region: --steps steps, repeated (whole multiples) until it lasts >= --min-timed-ms
    torch.cuda.synchronize(dev)
    te0 = time.perf_counter()
    step(); step()
    torch.cuda.synchronize(dev)
    est_step_ms = (time.perf_counter() - te0) / 2 * 1e3
    reps = 1
    if args.min_timed_ms > 0:
        reps = max(1, int(-(-args.min_timed_ms // max(est_step_ms * args.steps, 1e-3))))
    if world > 1:   # every rank must time the same number of steps
        import torch.distributed as dist
        t_reps = torch.tensor([reps], dtype=torch.int64, device=dev)
        dist.all_reduce(t_reps, op=dist.ReduceOp.MAX)
        reps = int(t_reps.item())
    steps_timed = reps * args.steps
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps_timed)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps_timed)]
    # shader clock actually sustained during the timed steps: a one-wave sampler on its own stream spans ~80 % of the region
    # (its length estimated from two more untimed steps)
    est_ms = est_step_ms * steps_timed
    clock = None
    # (the step uses up to four streams -- the caller's, F, B, A -- and the runtime has four hardware queues by default: the sampler that sits
    # in a queue of its own for the whole region is the A/B aid now, FDGS_BENCH_CLOCK=span|both; the default takes two short samples on the
    # caller's stream, in front of the first and behind the last timed step)
    clock_mode = os.environ.get("FDGS_BENCH_CLOCK", "pair")
    clock_pair = None
    try:
        clock = _capi.ClockSample(dev) if clock_mode in ("span", "both") else None
        clock_pair = _capi.ClockPair(dev) if clock_mode in ("pair", "both") else None
    except Exception:
        clock = clock_pair = None
    # Rehearsal of the timed region's own pattern (round 6): two blocks of [restore, --steps steps], untimed.  The warm-up above walks ONE
    # long trajectory away from the restored state -- the scene shrinks, the run-ahead buffers with it, the caching allocator splits
    # its large blocks to serve the smaller requests -- and the first restore inside the timed region then asks for the large sizes
    # again: one or two device allocations of ~10 ms each next to a busy device (`ms_per_step_max` 12 ms, `value` 3 % under
    # `value_median` in one run of three).  The same cure as for the legs (warm_up).
    for _rb in range(2):
        restore()
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize(dev)
    # no collector pauses inside the timed region (the host runs ~1.8 ms ahead of the GPU per step; a generation-2 collection of the
    # step's many small Python objects takes longer than that)
    gc.collect()
    gc.disable()
    if use_pipeline and world > 1:
        steppipe.exchange_pairs = []   # stream B's waits for the collectives, bracketed by timing events
    barrier(world)
    t0 = time.perf_counter()
    if clock is not None:
        clock.start(min(max(0.8 * est_ms, 1.0), 1500.0))
    if clock_pair is not None:
        clock_pair.mark()
    _DBG, _host_ms = int(os.environ.get("FDGS_BENCH_DEBUG", "0")), []
    for i in range(steps_timed):
        if i % args.steps == 0:
            restore()   # inside the timed region: ~0.25 ms of copies per --steps steps (not part of any step's own event pair)
        starts[i].record()
        if _DBG:
            _th = time.perf_counter()
        pkg = step()
        if _DBG:
            _host_ms.append((time.perf_counter() - _th) * 1e3)
        ends[i].record()
    if clock_pair is not None:
        clock_pair.mark()
    torch.cuda.synchronize(dev)
    barrier(world)
    dt = time.perf_counter() - t0
    if _DBG:
        print("step dbg: event pairs (ms)", [round(starts[i].elapsed_time(ends[i]), 2) for i in range(min(steps_timed, 45))], file=sys.stderr)
        print("step dbg: gaps between pairs (ms)", [round(ends[i].elapsed_time(starts[i + 1]), 2) for i in range(min(steps_timed - 1, 45))], file=sys.stderr)
        print("step dbg: host (ms)", [round(h, 2) for h in _host_ms[:45]], file=sys.stderr)
        _ev = [starts[i].elapsed_time(ends[i]) for i in range(steps_timed)]
        _med = sorted(_ev)[len(_ev) // 2]
        print("step dbg: steps over twice the median (index, event ms, host ms)", [(i, round(_ev[i], 2), round(_host_ms[i], 2)) for i in range(steps_timed) if _ev[i] > 2 * _med],
              "host-only outliers", [(i, round(_host_ms[i], 2)) for i in range(steps_timed) if _host_ms[i] > 3 * _med and _ev[i] <= 2 * _med], file=sys.stderr)
    # (the collector stays off for the rest of the run -- every leg below is a few dozen steps, i.e. tens of milliseconds, and one
    # generation-2 pause of the interpreter in it is 20-40 % of its reading: legs that read 1000-1400 where the next run read 1700;
    # warm_up() collects by hand in front of every leg)
    gc.collect()
    _capi.profile_enable(False)
    shader_ghz = pair_ghz = None
    try:
        shader_ghz = clock.ghz() if clock is not None else None
        if clock_pair is not None:
            pair_ghz = clock_pair.ghz()
            if clock_mode == "both":
                print("clock: spanning sampler %s GHz, two samples on the caller's stream %s GHz" % (shader_ghz, pair_ghz), file=sys.stderr)
    except Exception:
        shader_ghz = None
    if clock is None and clock_mode == "pair" and use_pipeline:
        # the spanning sampler (the clock every round has quoted: it counts through the moments the shader array idles, the two samples
        # read 1.5 % less) on the same steps right behind the timed region, through a pipeline without overlap_steps: three streams + the
        # sampler's = the four hardware queues
        try:
            cpipe = new_pipeline(model, opt, world_size=world, lambda_dssim=0.2, overlap=not args.no_overlap,
                                 gather_max_views=0 if args.dense_sh_exchange else 32, split_colour=args.split_colour == "all",
                                 tile_cull=not args.no_tile_cull, batch_views=args.batch_views, sh_group=args.sh_group, lazy=not args.no_lazy,
                                 sparse_lists=not args.no_sparse_lists, overlap_steps=False)
            restore()
            for _ in range(3):
                cpipe.step(cams, gts, pipe, bg)
            torch.cuda.synchronize(dev)
            n_clock = max(args.steps, min(steps_timed, int(300.0 / max(est_step_ms, 1e-3))))
            clock = _capi.ClockSample(dev)
            clock.start(min(max(0.8 * est_step_ms * n_clock, 1.0), 1500.0))
            for _ in range(n_clock):
                cpipe.step(cams, gts, pipe, bg)
            torch.cuda.synchronize(dev)
            shader_ghz = clock.ghz()
            del cpipe
        except Exception:
            shader_ghz = pair_ghz
        finally:
            restore()
    prof_dom = _capi.profile_read()[dom]
    # N > 1: every rank's own wall time per step and the exchange time nothing overlapped (the sum of stream B's waits for a
    # collective per step, train_host.timed_wait), so that a scaling run explains itself
    per_rank = None
    if world > 1:
        import torch.distributed as dist
        exposed = None
        if use_pipeline and steppipe.exchange_pairs:
            exposed = sum(a.elapsed_time(b) for a, b in steppipe.exchange_pairs) / steps_timed
            steppipe.exchange_pairs = None
        allr = torch.zeros((world, 2), dtype=torch.float64, device=dev)
        allr[rank, 0], allr[rank, 1] = dt / steps_timed * 1e3, (-1.0 if exposed is None else exposed)
        dist.all_reduce(allr, op=dist.ReduceOp.SUM)   # (a gather spelt as a sum: works on every backend the debug runs use)
        per_rank = {"ms_per_step": [round(float(t[0]), 4) for t in allr],
                    "exchange_exposed_ms": None if exposed is None else [round(float(t[1]), 4) for t in allr],
                    "what": "per rank: wall time per step of the timed region; exchange_exposed_ms = per step, the time the backward stream stood still "
                            "waiting for a collective (stage all-gathers / gradient all-reduces): the part of the exchange nothing overlapped"}
    dt = max_over_ranks(dt, world, dev)
    step_ms = sorted(starts[i].elapsed_time(ends[i]) for i in range(steps_timed))
    r_last = [r["num_rendered"] for r in pkg if isinstance(r, dict) and r.get("num_rendered", -1) >= 0]
    R_timed = (sum(r_last) / len(r_last)) if r_last else (sum(_R_LOG) / max(len(_R_LOG), 1))
    lazy_redone = steppipe.lazy_redone if use_pipeline else None
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]  # noqa: E731

    # digest of the parameters after all steps (tests compare N ranks x B views with one rank x N B views: the same update)
    param_digest = [float(model.flat.double().sum()), float(model.flat.double().abs().sum())]
    # frame-parallel replicas must hold bit-identical parameters after the timed steps (every rank applied the same update)
    replicas_identical = None
    if world > 1:
        import torch.distributed as dist
        digest = torch.stack([model.flat.double().sum(), model.flat.double().abs().sum()])
        lo, hi = digest.clone(), digest.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))

    # forward-only rate (the metric's second half), outside the train-step timing
    n_fwd = min(steps_timed, 4 * args.steps) * B

    def forward_only(c):
        if use_pipeline:  # the same explicit call the step pipeline makes (no autograd bookkeeping)
            from fdgs.fused import raw_forward, raw_settings
            rs, (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = raw_settings(c, model, pipe, bg)
            return raw_forward(rs, xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv,
                               split_colour=args.split_colour != "off", tile_cull=not args.no_tile_cull, lazy=not args.no_lazy,
                               sparse_lists=not (args.no_lazy or args.no_sparse_lists))
        return render(c, model, pipe, bg) if args.reference_host else render_raw(c, model, pipe, bg)

    fwd_lazy_failed = 0

    restore()
    with torch.no_grad():
        for _ in range(20):   # right behind the timed steps (the shader clock is up); a few more calls settle the allocator on this stream
            forward_only(cam)
        torch.cuda.synchronize(dev)
        # n_fwd forwards in 5 slices, each bracketed by a synchronisation; the leg's time is the median slice x 5 (one host hiccup
        # -- 8 ms in 52 -- used to move this secondary figure by 15 %; the headline `value` above stays the plain mean of K steps)
        per = max(1, n_fwd // 5)
        slices = []
        for s_ in range(5):
            t1 = time.perf_counter()
            for i in range(per):
                forward_only(cams[(s_ * per + i) % B])
                if i % 32 == 31 and use_pipeline and not args.no_lazy:
                    fwd_lazy_failed += _capi.forward_lazy_status(dev, wait=False)[1]   # keeps the ring of unreported forwards short
            torch.cuda.synchronize(dev)
            slices.append(time.perf_counter() - t1)
            if use_pipeline and not args.no_lazy:
                fwd_lazy_failed += _capi.forward_lazy_status(dev, wait=True)[1]
        n_fwd = 5 * per
        dt_fwd = max_over_ranks(sorted(slices)[2] * 5, world, dev)

    # rasterizer-only rate: forward + backward of one view after the other, no loss, no optimizer -- the like-for-like partner of
    # cpu_baseline (same scene, the same four upstream gradients, all of them given: the general blend-backward variant)
    raster = None
    if use_pipeline and world == 1:
        from fdgs.fused import raw_backward, raw_forward, raw_settings
        up4 = synth.make_upstream_grads(scene["W"], scene["H"], seed=1, scale=1e-2)
        up4 = {k: v.to(dev) for k, v in up4.items()}
        gacc = torch.zeros((model.P, 16), dtype=torch.float32, device=dev)
        sink4 = model.grad_sink()

        def raster_only(c):
            rs, (xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = raw_settings(c, model, pipe, bg)
            (R, color, flow, depth, T, radii, geom, binb, img, _c, om) = raw_forward(rs, xyz, feats, opacity, ts, scaling, scaling_t,
                                                                                    rotation, rotation_r, pv,
                                                                                    tile_cull=not args.no_tile_cull)
            raw_backward(rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img,
                         up4["grad_color"], up4["grad_depth"], up4["grad_alpha"], up4["grad_flow"], sink4, False, grad_accum=gacc)

        with torch.no_grad():
            for b in range(B):
                raster_only(cams[b])
            torch.cuda.synchronize(dev)
            slices = []
            for s_ in range(5):   # median slice x 5, as the forward-only leg
                t2 = time.perf_counter()
                for i in range(per):
                    raster_only(cams[(s_ * per + i) % B])
                torch.cuda.synchronize(dev)
                slices.append(time.perf_counter() - t2)
            dt_r = sorted(slices)[2] * 5
        raster = {"images_s": round(n_fwd / dt_r, 2), "ms_per_image": round(dt_r / n_fwd * 1e3, 4),
                  "what": "rasterizer forward + backward only (all four upstream gradients given), one stream, %d views: pairs with cpu_baseline" % n_fwd}
        del gacc, up4

    # the same step with the reference's tile lists (fdgs_forward_out.tile_cull = 0: point_list / ranges / n_contrib bit-identical to the
    # reference's) -- `value` runs with tile_cull = 1 (same pixels and gradients, a quarter fewer list entries)
    reflists = None
    if use_pipeline and args.reflists_steps > 0 and not args.no_tile_cull:
        rp = StepPipeline(model, opt, world_size=world, lambda_dssim=0.2, overlap=not args.no_overlap,
                          gather_max_views=0 if args.dense_sh_exchange else 32, tile_cull=False, lazy=not args.no_lazy, sparse_lists=False,
                          overlap_steps=not args.no_overlap_steps)
        restore()
        warm_up(lambda: rp.step(cams, gts, pipe, bg), dev, world, at_least=max(3, args.reflists_steps))
        restore()
        torch.cuda.synchronize(dev)
        barrier(world)
        r_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reflists_steps)]
        tr0 = time.perf_counter()
        for i in range(args.reflists_steps):
            r_ev[i][0].record()
            rres, _l = rp.step(cams, gts, pipe, bg)
            r_ev[i][1].record()
        torch.cuda.synchronize(dev)
        barrier(world)
        dtr = max_over_ranks(time.perf_counter() - tr0, world, dev)
        r_ms = sorted(a.elapsed_time(b) for a, b in r_ev)
        reflists = {"images_s": round(world * B * args.reflists_steps / dtr, 2), "ms_per_step": round(dtr / args.reflists_steps * 1e3, 4),
                    # (rounds 5-6: this leg read 1100-1200 instead of 1700 in about every third run -- ONE step of 25-33 ms: a device allocation
                    # inside the leg, found in round 6: warm_up now rehearses the leg's whole trajectory)
                    "images_s_median_step": round(world * B / (r_ms[len(r_ms) // 2] * 1e-3), 2), "ms_per_step_max": round(r_ms[-1], 4),
                    "steps": args.reflists_steps, "num_rendered": int(round(sum(r["num_rendered"] for r in rres) / len(rres))),
                    "lazy_steps_redone": rp.lazy_redone,
                    "step_ms_sorted_tail": [round(x, 3) for x in r_ms[-3:]],
                    "what": "the same step with tile_cull = 0 and sparse_lists = 0: point_list / ranges / n_contrib are the reference's, bit for bit (tests/test_gpu_parity.py)"}
        del rp

    # the same step with the model stored in Morton order (a memory-layout choice of the trainer, no effect on the arithmetic)
    spatial = None
    if use_pipeline and args.spatial_order_steps > 0:
        if args.storage_order == "morton":   # the other order: the generator's
            model.flat.data.copy_(snap_random[0])
            opt.exp_avg.copy_(snap_random[1])
            opt.exp_avg_sq.copy_(snap_random[2])
            opt.step_count = snap_random[3]
            models_touched()
        else:
            restore()
            train_host.spatial_sort(model, opt)
            models_touched()
        warm_up(step, dev, world, at_least=max(3, args.spatial_order_steps))
        torch.cuda.synchronize(dev)
        barrier(world)
        ts0 = time.perf_counter()
        for _ in range(args.spatial_order_steps):
            step()
        torch.cuda.synchronize(dev)
        barrier(world)
        dts = max_over_ranks(time.perf_counter() - ts0, world, dev)
        with torch.no_grad():
            for _ in range(10):
                forward_only(cam)
            torch.cuda.synchronize(dev)
            tf0 = time.perf_counter()
            for i in range(n_fwd):
                forward_only(cams[i % B])
            torch.cuda.synchronize(dev)
            dtf = max_over_ranks(time.perf_counter() - tf0, world, dev)
        spatial = {"images_s": round(world * B * args.spatial_order_steps / dts, 2), "ms_per_step": round(dts / args.spatial_order_steps * 1e3, 4),
                   "forward_ms": round(dtf / n_fwd * 1e3, 4), "steps": args.spatial_order_steps,
                   "order": "random" if args.storage_order == "morton" else "morton",
                   "what": "the same step and forward with the model stored in the OTHER order (`order`); `value` is measured with "
                           "--storage-order " + args.storage_order + ". morton = Morton order of the positions (train_host.spatial_sort, "
                           "the order fdgs.harness.train keeps the model in); random = the generator's order"}

    # the same step through the OTHER camera set (rounds 1-4 quoted `value` on the single unrotated on-axis camera)
    other_cams = None
    if use_pipeline and args.axis_steps > 0:
        which = "axis" if args.cameras == "rig" else "rig"
        ocams = make_cams(scene, which)
        restore()
        warm_up(lambda: steppipe.step(ocams, gts, pipe, bg), dev, world, at_least=max(3, args.axis_steps))
        restore()
        torch.cuda.synchronize(dev)
        barrier(world)
        to0 = time.perf_counter()
        for _ in range(args.axis_steps):
            ores, _l = steppipe.step(ocams, gts, pipe, bg)
        torch.cuda.synchronize(dev)
        barrier(world)
        dto = max_over_ranks(time.perf_counter() - to0, world, dev)
        other_cams = {"cameras": which, "images_s": round(world * B * args.axis_steps / dto, 2), "ms_per_step": round(dto / args.axis_steps * 1e3, 4),
                      "steps": args.axis_steps, "num_rendered": int(round(sum(r["num_rendered"] for r in ores) / len(ores))),
                      "visible": int(round(sum(int((r["radii"] > 0).sum().item()) for r in ores) / len(ores))),
                      "what": "the same step with --cameras " + which + " (axis = the one unrotated camera on the z axis every view of rounds 1-4 "
                              "went through; rig = four different rotated off-axis cameras per step)"}
        restore()

    # BASELINE configs[4]: the HBM stress configuration, every round, in the driver's line
    c5 = None
    if use_pipeline and world == 1 and args.workload == "C3" and args.c5_steps > 0:
        c5 = c5_leg(args, dev, make_cams, pipe, B)

    # a skewed scene: the same step on C3-clustered (own model; same storage order as the main run)
    clustered = None
    if use_pipeline and world == 1 and args.workload == "C3" and args.clustered_steps > 0:
        cs = synth.make_scene(synth.CONFIGS["C3-clustered"], seed=0)
        cm = train_host.GaussianParams(cs, dev)
        co = train_host.make_optimizer(cm)
        if args.storage_order == "morton":
            train_host.spatial_sort(cm, co)
        csnap = (cm.flat.detach().clone(), co.exp_avg.clone(), co.exp_avg_sq.clone())
        cp = StepPipeline(cm, co, world_size=1, lambda_dssim=0.2, overlap=not args.no_overlap, tile_cull=not args.no_tile_cull, lazy=not args.no_lazy,
                          sparse_lists=not args.no_sparse_lists, overlap_steps=not args.no_overlap_steps)
        # (on the on-axis camera whatever --cameras says: the box is placed to project onto 15 % of THAT image, and the leg's full-size parity test uses it)
        ccams = [train_host.SyntheticCamera(cs, dev, timestamp=(b + 0.5) / B * cs["time_duration"]) for b in range(B)]
        warm_up(lambda: cp.step(ccams, gts, pipe, bg), dev, at_least=max(3, args.clustered_steps))
        cm.flat.data.copy_(csnap[0]); co.exp_avg.copy_(csnap[1]); co.exp_avg_sq.copy_(csnap[2]); co.step_count = 0
        models_touched()
        torch.cuda.synchronize(dev)
        tc0 = time.perf_counter()
        for _ in range(args.clustered_steps):
            cres, _l = cp.step(ccams, gts, pipe, bg)
        torch.cuda.synchronize(dev)
        dtc = time.perf_counter() - tc0
        from fdgs.fused import raw_forward, raw_settings
        with torch.no_grad():
            tf0 = time.perf_counter()
            for i in range(4 * args.clustered_steps):
                rs_, (xyz_, f_, o_, t_, s_c, st_, r_, rr_, pv_) = raw_settings(ccams[i % B], cm, pipe, bg)
                raw_forward(rs_, xyz_, f_, o_, t_, s_c, st_, r_, rr_, pv_, tile_cull=not args.no_tile_cull, lazy=not args.no_lazy,
                            sparse_lists=not (args.no_lazy or args.no_sparse_lists))
            torch.cuda.synchronize(dev)
            dtcf = time.perf_counter() - tf0
            _capi.forward_lazy_status(dev, wait=True)
            rs_, (xyz_, f_, o_, t_, s_c, st_, r_, rr_, pv_) = raw_settings(ccams[0], cm, pipe, bg)
            one = raw_forward(rs_, xyz_, f_, o_, t_, s_c, st_, r_, rr_, pv_, tile_cull=not args.no_tile_cull)
            torch.cuda.synchronize(dev)
        clustered = {"images_s": round(B * args.clustered_steps / dtc, 2), "ms_per_step": round(dtc / args.clustered_steps * 1e3, 4),
                     "forward_ms": round(dtcf / (4 * args.clustered_steps) * 1e3, 4), "num_rendered": int(one[0]), "steps": args.clustered_steps,
                     "lazy_steps_redone": cp.lazy_redone,
                     "what": "the same step on C3-clustered (fdgs.synth: 70 % of the 300 k Gaussians inside a box that projects onto 15 % of the image; "
                             "188 tile lists beyond 4096 entries, the longest 5485 with the reference's lists): parity at full size in "
                             "tests/test_gpu_parity.py::test_c3_clustered_full_size_vs_oracle"}
        del cm, co, cp, csnap

    # host cost per view: the same step on a scene so small that the GPU work is negligible (wall time ~ host time)
    host_ms_per_view = None
    if use_pipeline and rank == 0 and args.host_cost_steps > 0:
        tiny = synth.make_scene(synth.SceneConfig("tiny", 2000, 64, 48, cfg.sh_degree, cfg.sh_degree_t, 0.05, cfg.duration,
                                                  cfg.rot_4d, cfg.gaussian_dim, cfg.force_sh_3d), seed=0)
        tm = train_host.GaussianParams(tiny, dev)
        tp = StepPipeline(tm, train_host.make_optimizer(tm), world_size=1, lambda_dssim=0.2, overlap=not args.no_overlap)
        tcams = [train_host.SyntheticCamera(tiny, dev, timestamp=(b + 0.5) / B * tiny["time_duration"]) for b in range(B)]
        tgts = [torch.rand(3, tiny["H"], tiny["W"], device=dev) for _ in range(B)]
        tbg = tiny["bg"].to(dev)
        for _ in range(5):
            tp.step(tcams, tgts, pipe, tbg)
        torch.cuda.synchronize(dev)
        th = time.perf_counter()
        for _ in range(args.host_cost_steps):
            tp.step(tcams, tgts, pipe, tbg)
        torch.cuda.synchronize(dev)
        host_ms_per_view = (time.perf_counter() - th) / (args.host_cost_steps * B) * 1e3

    dropin = None
    if world == 1 and rank == 0 and args.dropin_steps > 0:
        dropin = dropin_leg(args, scene, cams, gts, pipe, bg, dev, B)

    if rank != 0:
        return
    P, M, W, H = model.P, model.M, scene["W"], scene["H"]
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    # visible Gaussians: mean over the views of the last step (every view has its own timestamp, hence its own cull)
    Pv = sum(int((r["radii"] > 0).sum().item()) for r in pkg) / len(pkg)
    # live Gaussians: those that contributed to some pixel (a non-zero screen-space gradient row), mean over the last step's views
    if use_pipeline:
        P_live = sum(int((r["viewspace_grad"] != 0).any(dim=1).sum().item()) for r in pkg) / len(pkg)
    else:
        P_live = Pv
    stages = {}
    for name, (ms, n) in prof_all.items():
        if n == 0:
            continue
        entry = {"ms": round(ms / n, 4)}
        if name in ALGO_BYTES:
            if name == "sh_bwd":   # the step pipeline stages the SH gradient (deferred) whenever it is used; bytes scale with the LIVE Gaussians
                b = ALGO_BYTES["sh_bwd_deferred" if use_pipeline else "sh_bwd"](P, P_live, M, R_stage, N, T)
            else:
                b = ALGO_BYTES[name](P, Pv, M, R_stage, N, T)
            entry["algo_bytes"] = int(b)
            entry["gbps"] = round(b / (ms / n * 1e-3) / 1e9, 1) if ms > 0 else None
        stages[name] = entry
    # roofline of the dominant kernel: live duration from the timed (two-stream) region, bytes from the mean R of the timed views
    dom_ms = prof_dom[0] / max(prof_dom[1], 1)
    dom_bytes = ALGO_BYTES[dom](P, Pv, M, R_timed, N, T)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # the committed counter passes are C3 passes: no figure for another workload
                "traffic": pmc_traffic(dom, cfg.name) if cfg.name in ("C3", "C5") else None,
                "traffic_source": ("committed rocprofv3 --pmc passes of the same step (profiles/pmc_traffic_r*.json: tools/collect_profiles.sh, "
                                   "tools/pmc_traffic.py), not this run" if cfg.name in ("C3", "C5") else None),
                "avg_kernel_ms": round(dom_ms, 4), "avg_kernel_ms_single_stream": stages[dom]["ms"],
                "launches_timed": int(prof_dom[1]), "launches_bracketed": "every %d-th" % PROFILE_EVERY, "algo_bytes_per_launch": int(dom_bytes),
                "note": "the blend kernels are VALU-issue-bound, not HBM-bound (DESIGN.md section 4): the fraction of the HBM "
                        "peak is reported as the contract asks; valu_issue_frac = VALU issue cycles per SIMD (SQ counters, "
                        "profiles/) / kernel cycles at the shader clock measured under the step's load is the bound this kernel actually runs against"}
    valu = pmc_valu(dom) if cfg.name == "C3" else None   # the committed SQ-counter pass is a C3 pass
    roofline["shader_clock_ghz_measured"] = None if shader_ghz is None else round(shader_ghz, 3)
    roofline["shader_clock_ghz_two_samples"] = None if pair_ghz is None else round(pair_ghz, 3)
    roofline["shader_clock_note"] = ("shader_clock_ghz_measured: one wave on a stream of its own spanning ~80 % of a loop of the same steps "
                                     "(inside the timed region with FDGS_BENCH_CLOCK=span; by default right behind it through a pipeline without "
                                     "overlap_steps, because the timed steps use all four default hardware queues); shader_clock_ghz_two_samples: "
                                     "two short samples on the caller's stream around the timed region itself (reads ~1.5 % less: the counter "
                                     "rests while the shader array idles)")
    if valu:
        # instruction counts are a property of the kernel + workload (committed SQ pass, same C3 scene); the time and the clock are live
        ghz = shader_ghz if shader_ghz else 2.4
        pass_ns = valu.get("kernel_ns_in_counter_pass")
        # cycles the kernel takes: its single-stream duration in THIS run (the stage pass above: since round 5 the counter pass --
        # tools/step_loop.py -- renders the same four views) x the clock measured in this run.  The duration inside the counter pass
        # belongs to a process of two optimizer steps whose shader clock has not settled (it is ~7 % longer than the live one for the
        # same instruction count); the fraction with it is kept next to the other one
        single_ms = stages[dom]["ms"] if dom in stages else dom_ms
        valu["kernel_cycles"] = int(single_ms * 1e-3 * ghz * 1e9)
        valu["kernel_cycles_from"] = ("the kernel's single-stream duration in this run (stage pass, same views as the counter pass) x the clock measured in this run; "
                                      "a fraction a few percent above 1 is within the method's accuracy: SQ_ACTIVE_INST_VALU sums the cycles of the main and "
                                      "the transcendental pipe (3.8 % of this kernel's VALU instructions), and the sampled clock averages over a loop of whole steps")
        valu["clock"] = ("measured under the step's load (s_memtime / s_memrealtime sampler, fdgs_debug_clock_sample; shader_clock_note)" if shader_ghz
                         else "2.4 GHz maximum clock assumed: valu_issue_frac is a LOWER bound")
        roofline["valu"] = valu
        # ONE pass on both sides of the fraction (round-5 review): the committed counter pass's VALU issue cycles over the SAME pass's kernel
        # duration (x the clock measured here) -- a physical fraction, <= 1.  The counter-pass cycles over THIS run's (shorter) single-stream
        # duration is kept beside it: it mixes two runs and reads a few percent above 1 when the clocks differ
        live = round(valu["valu_issue_cycles_per_simd"] / max(valu["kernel_cycles"], 1), 3)
        if pass_ns:
            roofline["valu_issue_frac"] = round(valu["valu_issue_cycles_per_simd"] / max(pass_ns * 1e-9 * ghz * 1e9, 1.0), 3)
            roofline["valu_issue_frac_with_this_runs_duration"] = live
        else:
            roofline["valu_issue_frac"] = live
    mode = ("weak scaling, %d views per GPU and step (the reference's DyNeRF batch per GPU)" % B if B > 1 else
            "BASELINE configs[3] as specified: one view per GPU and step, N timesteps frame-parallel")
    out = {
        "metric": "train-step images/sec + forward Mpix/s, 300k 4D Gaussians @1352x1014",
        "value": round(world * B * steps_timed / dt, 3),
        "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        # the timed region is --steps steps repeated `steps_timed / steps` times (>= --min-timed-ms of wall time), one barrier +
        # synchronize pair around all of it; the warm-up ran `warmup_steps_run` steps (>= --warmup, >= --min-warmup-ms)
        "steps_timed": steps_timed, "warmup_steps_run": warm_steps, "timed_region_ms": round(dt * 1e3, 2),
        "value_median": round(world * B * 1e3 / pct(0.5), 3),
        "ms_per_step": round(dt / steps_timed * 1e3, 4),
        "ms_per_step_median": round(pct(0.5), 4), "ms_per_step_p10": round(pct(0.1), 4), "ms_per_step_p90": round(pct(0.9), 4),
        "ms_per_step_max": round(step_ms[-1], 4), "steps_over_twice_the_median": int(sum(1 for t_ in step_ms if t_ > 2.0 * pct(0.5))),
        "ms_per_image": round(dt / (steps_timed * B) * 1e3, 4),
        "lazy_forward": bool(use_pipeline and not args.no_lazy and world == 1), "lazy_steps_redone": lazy_redone,
        # fdgs_forward_out.sparse_lists: lazy forwards keep every tile's list at a fixed offset of the binning buffer (the same lists; no count /
        # scan launch: their rows are then missing from `stages`)
        "sparse_lists": bool(use_pipeline and not args.no_lazy and not args.no_sparse_lists and world == 1),
        "overlap_steps": bool(use_pipeline and steppipe.overlap_steps),
        "steps_started_under_the_previous_sh_update": int(steppipe.steps_carried) if use_pipeline else 0,
        "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
        "host_ms_per_view": None if host_ms_per_view is None else round(host_ms_per_view, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "replicas_identical": replicas_identical, "param_digest": param_digest,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("[model stored in Morton order, as fdgs.harness.train keeps it] " if args.storage_order == "morton" else "[model stored in the generator's random order] ") + "%s: %d 4D Gaussians, %dx%d, SH degree %d + time degree %d (M=%d), rot_4d=%s, "
                               "%d views/GPU/step, L1+SSIM loss (%s), Adam" % (cfg.name, P, W, H, cfg.sh_degree, cfg.sh_degree_t,
                                                                        M, cfg.rot_4d, B, "PyTorch" if args.torch_loss else "fused HIP") + (", reference host path" if args.reference_host else (", fused activations, explicit fwd/loss/bwd on %s" % ("one stream" if args.no_overlap else "two HIP streams") if use_pipeline else ", fused activations, autograd")),
                   "num_rendered": int(round(R_timed)), "cameras": ("4 different rotated off-axis cameras per step (fdgs.synth.POSES rig0..rig3)" if args.cameras == "rig" else "one unrotated on-axis camera for every view"), "tile_cull": not args.no_tile_cull, "visible": int(round(Pv)), "views_per_step_per_gpu": B, "global_batch": B * world,
                   "parallelism": "frame-parallel dp%d" % world, "mode": mode},
        "forward_mpix_s": round(world * n_fwd * N / dt_fwd / 1e6, 1),
        "forward_split_colour": bool(use_pipeline and args.split_colour != "off"),
        "forward_ms": round(dt_fwd / n_fwd * 1e3, 4),
        "forward_lazy_failed": fwd_lazy_failed,   # forwards of the forward-only leg whose run-ahead buffers were too small (their images are invalid): must be 0
        "raster_ms": round(sum(v["ms"] for v in stages.values()), 4),
        "live_gaussians": int(round(P_live)),
        "stages": stages,
        "stages_note": "per view, HIP events on the launch stream, from an untimed SINGLE-stream pass of %d steps (kernel "
                       "time, not queueing time behind the other stream); mean num_rendered of that pass %d; the dominant "
                       "stage ('%s') is re-measured live inside the timed two-stream region: roofline.avg_kernel_ms" % (
                           n_stage_steps, int(round(R_stage)), dom),
        "roofline": roofline,
    }
    out["rccl_ranks"] = world
    if world > 1:
        # what the ranks exchanged per step (fdgs/pipeline.py): the views' 32-byte SH stages by all-gather + the 17 geometry floats by
        # all-reduce (up to 32 views per step over all ranks), or the dense 161 P-float gradient bucket by all-reduce
        out["sh_exchange"] = ("stage all-gather + geometry all-reduce" if (use_pipeline and not args.dense_sh_exchange and world * B <= 32)
                              else "dense gradient all-reduce")
    if per_rank:
        out["per_rank"] = per_rank
    out["backend"] = backend if backend else "none (single process)"
    if other_cams:
        out["value_%s_camera" % other_cams["cameras"]] = other_cams["images_s"]
        out["other_cameras"] = other_cams
    if c5:
        out["c5_images_s"] = c5["images_s"]
        out["c5"] = c5
    if clustered:
        out["clustered_images_s"] = clustered["images_s"]
        out["clustered"] = clustered
    if reflists:
        out["value_reference_lists"] = reflists["images_s"]
        out["reference_lists"] = reflists
    out["storage_order"] = args.storage_order
    if spatial:
        if args.storage_order == "morton":
            out["random_order_images_s"] = spatial["images_s"]
            out["spatial_order_images_s"] = out["value"]
        else:
            out["spatial_order_images_s"] = spatial["images_s"]
            out["random_order_images_s"] = out["value"]
        out["other_storage_order"] = spatial
    if raster:
        out["raster_images_s"] = raster["images_s"]
        out["raster"] = raster
    if dropin:
        out["dropin_images_s"] = dropin["images_s"]
        out["dropin_forward_ms"] = dropin["forward_ms"]
        out["dropin"] = dropin
    if world == 1 and args.cpu_samples > 0:
        cb_scene = scene if args.cameras == "axis" else dict(scene, **synth.camera_for("rig0", scene["W"], scene["H"]))
        out["cpu_baseline"] = cpu_baseline(cb_scene, args.cpu_samples)
        # (the baseline leg's second half: the reference's own kernels on THIS GPU, same scene -- the partner of raster_images_s)
        ref_gpu = reference_kernels_on_this_gpu(cb_scene, dev)
        if ref_gpu is not None:
            out["cpu_baseline"]["reference_kernels_on_this_gpu"] = ref_gpu
            out["reference_hipified_images_s"] = ref_gpu["images_s"]
    if world == 1:
        out["step_end_stages"] = step_end_stages(dev, P, M, B, W, H, cfg.sh_degree, cfg.sh_degree_t, cfg.gaussian_dim, cfg.force_sh_3d, workload=cfg.name)
        if "hbm_bound_best" in out["step_end_stages"]:
            out["hbm_bound_best"] = out["step_end_stages"]["hbm_bound_best"]
    print(json.dumps(front_loaded(out)))


def front_loaded(out):
    """The same line with everything a reader of its first 2 KB needs up front (the driver's record truncates the rest): the contract's
    keys, then every leg's headline number, a compact `roofline` / `cpu_baseline`; the verbose parts (notes, per-stage tables, per-leg
    detail) follow under the same keys as before, the notes of `roofline` under `roofline_notes`."""
    first = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "value_median", "value_reference_lists", "value_axis_camera", "value_rig_camera", "forward_ms", "forward_mpix_s", "c5_images_s",
             "clustered_images_s", "raster_images_s", "reference_hipified_images_s", "dropin_images_s", "random_order_images_s", "hbm_bound_best")
    head = {k: out[k] for k in first if k in out}
    if "c5" in out and isinstance(out["c5"], dict):
        c5r = out["c5"].get("roofline") or {}
        head["c5_summary"] = {"forward_ms": out["c5"].get("forward_ms"), "end_to_end_frac_of_hbm_peak": out["c5"].get("end_to_end_frac_of_hbm_peak"),
                              "end_to_end_frac_of_hbm_peak_with_loss_and_optimizer": out["c5"].get("end_to_end_frac_of_hbm_peak_with_loss_and_optimizer"),
                              "hbm_bound_best": (out["c5"].get("step_end_stages") or {}).get("hbm_bound_best"),
                              "roofline": {k: c5r.get(k) for k in ("kernel", "achieved", "peak", "frac", "traffic") if k in c5r}}
    if "roofline" in out:
        r = out["roofline"]
        keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "avg_kernel_ms_single_stream", "valu_issue_frac",
                "algo_bytes_per_launch", "launches_timed")
        head["roofline"] = {k: r[k] for k in keep if k in r}
        out = dict(out, roofline_notes={k: v for k, v in r.items() if k not in keep})
    if "cpu_baseline" in out:
        head["cpu_baseline"] = out["cpu_baseline"]
    head["config"] = out.get("config")
    for k, v in out.items():
        if k not in head:
            head[k] = v
    return head


_NUM_RENDERED = {}
_R_LOG = []


def _install_r_probe():
    from fdgs.gaussian_renderer import diff_gaussian_rasterization as m
    orig = m._C.rasterize_gaussians

    def wrapped(*a, **k):
        res = orig(*a, **k)
        _NUM_RENDERED["R"] = res[0]
        _R_LOG.append(res[0])
        return res
    m._C.rasterize_gaussians = wrapped


if __name__ == "__main__":
    _install_r_probe()
    main()
