"""Frame-parallel training on a synthetic 4D scene (no dataset needed): ground truth is rendered from a target model,
a perturbed copy is trained back.  One process per GPU:

    python examples/train_synthetic.py --iterations 300
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py

Uses fdgs.harness.train (FrameShard + StepPipeline + one gradient all-reduce per step over RCCL).
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # the step's own streams fill the runtime's default of 4 hardware queues; RCCL adds its own
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=4)
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    from fdgs import harness, synth, train_host
    from fdgs.fused import render_raw
    scene = synth.make_scene(synth.CONFIGS[args.workload], seed=0)
    pipe, bg = train_host.PipelineFlags(), scene["bg"].to(dev)
    target = train_host.GaussianParams(scene, dev)
    cams = [train_host.SyntheticCamera(scene, dev, timestamp=(v + 0.5) / args.views * scene["time_duration"]) for v in range(args.views)]
    with torch.no_grad():
        gts = [render_raw(c, target, pipe, bg)["render"].clone() for c in cams]
    student = train_host.GaussianParams(scene, dev)
    g = torch.Generator(device="cpu").manual_seed(1)   # same perturbation on every rank: replicas start identical
    with torch.no_grad():
        student.params["_features"].add_(0.3 * torch.randn(student.params["_features"].shape, generator=g).to(dev))
        student.params["_opacity"].add_(0.5 * torch.randn(student.params["_opacity"].shape, generator=g).to(dev))
    del target
    opt = train_host.make_optimizer(student)
    torch.cuda.synchronize(); t0 = time.time()
    harness.train(student, opt, cams, gts, pipe, bg, iterations=args.iterations, batch_size=args.batch_size,
                  world_size=world, rank=rank, log_every=max(1, args.iterations // 10))
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - t0
        print("%d iterations x %d views x %d ranks in %.2f s: %.0f images/s" % (args.iterations, args.batch_size, world, dt,
                                                                              args.iterations * args.batch_size * world / dt))


if __name__ == "__main__":
    main()
