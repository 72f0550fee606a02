"""Timing of the SH backward alone at C3: the per-view kernel (inside fdgs_rasterize_backward, stage table) against
fdgs_sh_backward_batch with 1, 2 and 4 views (dev tool; run through gpurun)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import _capi, synth, train_host
from fdgs.fused import raw_backward, raw_forward, raw_settings
from fdgs.gaussian_renderer.diff_gaussian_rasterization import _C

dev = torch.device("cuda:0")
scene = synth.make_scene(synth.CONFIGS["C3"], seed=0)
model = train_host.GaussianParams(scene, dev)
if len(sys.argv) > 1 and sys.argv[1] == "morton":
    train_host.spatial_sort(model, None)
pipe, bg = train_host.PipelineFlags(), scene["bg"].to(dev)
NV = 4
cams = [train_host.SyntheticCamera(scene, dev, timestamp=(b + 0.5) / NV * scene["time_duration"]) for b in range(NV)]
sets = [raw_settings(c, model, pipe, bg) for c in cams]
tens = sets[0][1]
(xyz, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv) = tens
H, W, P = scene["H"], scene["W"], model.P
up = (torch.randn(3, H, W) * 1e-2).to(dev)
fwd = [raw_forward(rs, *tens) for rs, _ in sets]
sink = model.grad_sink()
stage = torch.empty((NV, P, 8), device=dev)
gacc = torch.zeros((NV, P, 16), device=dev)
pend = []
for b, (rs, _) in enumerate(sets):
    (R, color, flow, depth, T, radii, geom, binb, img, _c, om) = fwd[b]
    pend.append(raw_backward(rs, xyz, om, radii, feats, opacity, ts, scaling, scaling_t, rotation, rotation_r, pv, geom, R, binb, img, up, None, None, None,
                             sink, b > 0, grad_accum=gacc[b], sh_stage=stage[b], begin_only=True))
torch.cuda.synchronize()
for nv in (1, 2, 4):
    for _ in range(3):
        _C.sh_backward_batch(pend[:nv])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        _C.sh_backward_batch(pend[:nv])
    torch.cuda.synchronize()
    print("sh_backward_batch, %d view(s): %.1f us per call" % (nv, (time.perf_counter() - t0) / 50 * 1e6))
