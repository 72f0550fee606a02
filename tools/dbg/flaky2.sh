#!/usr/bin/env bash
T=tests/test_gpu_api.py
K="test_gradient_accumulation or deferred or (test_step_pipeline_matches_autograd_step and batched-colours)"
echo "== values"; python -m pytest $T -q -k "$K" 2>&1 | grep -E "ACTUAL|DESIRED|FAILED"
echo "== HIP_LAUNCH_BLOCKING=1"; HIP_LAUNCH_BLOCKING=1 python -m pytest $T -q -k "$K" 2>&1 | grep -E "passed|failed|ACTUAL|DESIRED|^FAILED" | tail -4
echo "== AMD_SERIALIZE_KERNEL=3"; AMD_SERIALIZE_KERNEL=3 python -m pytest $T -q -k "$K" 2>&1 | grep -E "passed|failed|ACTUAL|DESIRED|^FAILED" | tail -4
echo "== gc disabled"; python - <<'PY'
import gc, sys, pytest
gc.disable()
sys.exit(pytest.main(["tests/test_gpu_api.py", "-q", "-k", "test_gradient_accumulation or deferred or (test_step_pipeline_matches_autograd_step and batched-colours)"]))
PY
echo "== reference loop repeated in one process (no pipeline at all)"
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from fdgs import synth, train_host
from fdgs.fused import render_raw
from fdgs.loss import fused_l1_ssim
dev = torch.device("cuda:0")
cfg = synth.SceneConfig("pipe", 6000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
scene = synth.make_scene(cfg, seed=4)
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
pipe = train_host.PipelineFlags()
B = 3
cams = [train_host.SyntheticCamera(scene, dev, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
gen = torch.Generator(device="cpu").manual_seed(7)
gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(dev) for _ in range(B)]
seen = {}
for rep in range(40):
    ma = train_host.GaussianParams(scene, dev); oa = train_host.make_optimizer(ma); sink = ma.grad_sink()
    ref = []
    for _ in range(2):
        for b in range(B):
            loss = fused_l1_ssim(render_raw(cams[b], ma, pipe, bg, grad_sink=sink, accumulate=b > 0)["render"], gts[b], 0.2)
            (loss / B).backward(); ref.append(round(float(loss.detach()), 6))
        oa.step()
    seen[tuple(ref)] = seen.get(tuple(ref), 0) + 1
for k, v in seen.items(): print(v, k)
PY
