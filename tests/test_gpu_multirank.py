"""GPU: the multi-rank control flow of bench.py on ONE GPU (FDGS_BENCH_DEBUG_SHARE_GPU: both ranks on cuda:0, gloo
collectives) -- what the driver launches on 2/4/8 GPUs over RCCL, minus the fabric: torch.distributed.run, the barrier /
max-over-ranks timing, the exchange of the SH gradient (all-gather of the views' stages + fused update; or, forced with
--dense-sh-exchange, the early all-reduce of the dense gradient with the split backward and chunked all-reduce + Adam), one JSON
line from rank 0, replicas bit-identical.  Both modes: 4 views per rank and step (weak scaling) and 1 view per rank and step (BASELINE configs[3])."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("views,dense", [(4, False), (1, False), (4, True)])
def test_bench_two_ranks_share_one_gpu(views, dense, gpu_device):
    env = dict(os.environ, FDGS_BENCH_DEBUG_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--min-warmup-ms", "0", "--min-timed-ms", "0",
           "--workload", "C2", "--views-per-step", str(views), "--cpu-samples", "0"] + (["--dense-sh-exchange"] if dense else [])
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["views_per_step_per_gpu"] == views and d["config"]["global_batch"] == 2 * views
    assert "roofline" in d and d["roofline"]["frac"] > 0
    assert d["replicas_identical"] is True   # both ranks hold bit-identical parameters after the steps


def test_bench_launches_its_own_ranks(gpu_device):
    """`python bench.py --gpus 2` as the driver calls it (no torch.distributed.run in front): bench.py re-executes itself
    through torch.distributed.run on 127.0.0.1, one process per rank, and rank 0 prints the one JSON line."""
    env = dict(os.environ, FDGS_BENCH_DEBUG_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # time-based warm-up and a minimum timed duration, as the driver's flags leave them on: how many steps that makes is decided by
    # the ranks TOGETHER (a step holds collectives: a rank that took one more would hang the others)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--min-warmup-ms", "40", "--min-timed-ms", "60", "--workload", "C2",
           "--cpu-samples", "0", "--host-cost-steps", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["replicas_identical"] is True
    assert d["config"]["global_batch"] == 8 and d["value"] > 0
    assert d["steps_timed"] >= 2 and d["steps_timed"] % 2 == 0 and d["warmup_steps_run"] >= 1


def test_bench_refuses_more_ranks_than_gpus(gpu_device):
    """Without the debug switch, asking for more GPUs than the node has must fail loudly (not hang, not share a device)."""
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDGS_BENCH_DEBUG_SHARE_GPU"):
        env.pop(k, None)
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--min-warmup-ms", "0", "--min-timed-ms", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU(s) visible" in (out.stderr + out.stdout)


def _run_bench(nproc, views, extra=()):
    env = dict(os.environ, FDGS_BENCH_DEBUG_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "2", "--min-warmup-ms", "0", "--min-timed-ms", "0", "--workload", "C2",
              "--views-per-step", str(views), "--cpu-samples", "0", "--host-cost-steps", "0", "--dropin-steps", "0", "--spatial-order-steps", "0"] + list(extra)
    if nproc == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + common
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("dense", [False, True])
def test_two_ranks_make_the_same_update_as_one(dense, gpu_device):
    """2 ranks x 2 views per step must train like 1 rank x 4 views: the same four timestamps per step, the same loss scale,
    the exchanged gradient = the accumulated one (float atomics and the order of a few sums aside).  Both exchanges: the SH
    stages gathered and fed to the fused update on every rank, and the dense bucket all-reduced."""
    one = _run_bench(1, 4)
    two = _run_bench(2, 2, ["--dense-sh-exchange"] if dense else [])
    assert two["replicas_identical"] is True and two["config"]["global_batch"] == one["config"]["global_batch"] == 4
    (s1, a1), (s2, a2) = one["param_digest"], two["param_digest"]
    assert abs(a1 - a2) <= 1e-5 * a1 and abs(s1 - s2) <= 1e-5 * a1, (one["param_digest"], two["param_digest"])


@pytest.mark.parametrize("views", [4, 1])
def test_eight_ranks_make_the_same_update_as_one(views, gpu_device):
    """The N = 8 control flow the driver's scaling run takes, before an 8-GPU box shows up: 8 ranks (all on cuda:0, gloo) x ``views``
    views per step against 1 rank x 8 ``views`` views.  views = 4: global batch 32 = gather_max_views, the LAST batch size that
    still exchanges the views' SH stages ([B, 8, P, 8] gather buffer); views = 1: BASELINE configs[3] as specified (8 timesteps, one
    per rank).  rank -> view mapping (global view g = rank * B + b: the same timestamps, cameras and targets as the single rank's),
    replicas bit-identical, the same update, per-rank exchange time reported."""
    one = _run_bench(1, 8 * views, ["--reflists-steps", "0", "--axis-steps", "0"])
    eight = _run_bench(8, views, ["--reflists-steps", "0", "--axis-steps", "0"])
    assert eight["n_gpus"] == 8 and eight["rccl_ranks"] == 8 and eight["replicas_identical"] is True
    assert eight["config"]["global_batch"] == one["config"]["global_batch"] == 8 * views
    assert eight["sh_exchange"].startswith("stage all-gather"), eight["sh_exchange"]
    assert eight["gpu_max_hw_queues"] == "8"
    (s1, a1), (s2, a2) = one["param_digest"], eight["param_digest"]
    assert abs(a1 - a2) <= 1e-5 * a1 and abs(s1 - s2) <= 1e-5 * a1, (one["param_digest"], eight["param_digest"])
    pr = eight["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and pr["exchange_exposed_ms"] is not None and len(pr["exchange_exposed_ms"]) == 8
    print("8 ranks x %d views on one GPU: per-rank ms/step %s, exchange exposed ms %s" % (views, pr["ms_per_step"], pr["exchange_exposed_ms"]))


def _run_bench_rccl(nproc, views, extra=()):
    """bench.py over RCCL: one rank per GPU (no debug sharing), launched as the driver launches the N > 1 bench."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDGS_BENCH_DEBUG_SHARE_GPU"):
        env.pop(k, None)
    common = [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "2", "--min-warmup-ms", "0", "--min-timed-ms", "0",
              "--workload", "C2", "--views-per-step", str(views), "--cpu-samples", "0", "--host-cost-steps", "0", "--dropin-steps", "0",
              "--spatial-order-steps", "0", "--reflists-steps", "0"] + list(extra)
    if nproc == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + common
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def _gpu_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: the exchange over RCCL / xGMI (the 1-GPU boxes run the gloo variant above)")
@pytest.mark.parametrize("dense", [False, True])
def test_rccl_two_gpus_make_the_same_update_as_one(dense, gpu_device):
    """The moment two GPUs are visible: 2 ranks x 2 views over **nccl (RCCL)** -- stage all-gather and dense all-reduce -- against
    1 rank x 4 views: replicas bit-identical, the same update (float atomics aside), per-rank timings reported."""
    one = _run_bench_rccl(1, 4)
    two = _run_bench_rccl(2, 2, ["--dense-sh-exchange"] if dense else [])
    assert two["backend"] == "nccl" and two["rccl_ranks"] == 2 and two["n_gpus"] == 2
    assert two["replicas_identical"] is True and two["config"]["global_batch"] == one["config"]["global_batch"] == 4
    (s1, a1), (s2, a2) = one["param_digest"], two["param_digest"]
    assert abs(a1 - a2) <= 1e-5 * a1 and abs(s1 - s2) <= 1e-5 * a1, (one["param_digest"], two["param_digest"])
    assert len(two["per_rank"]["ms_per_step"]) == 2 and two["per_rank"]["exchange_exposed_ms"] is not None


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs")
def test_rccl_bench_launches_itself_on_two_gpus(gpu_device):
    """`python bench.py --gpus 2` as the driver's N = 1 command line would be extended: self-launch, one rank per GPU, RCCL."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDGS_BENCH_DEBUG_SHARE_GPU"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "C2", "--cpu-samples", "0",
                          "--host-cost-steps", "0", "--dropin-steps", "0", "--spatial-order-steps", "0", "--reflists-steps", "0", "--min-timed-ms", "100"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["backend"] == "nccl" and d["replicas_identical"] is True and d["scaling"] == "weak"
