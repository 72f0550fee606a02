import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fdgs import synth, train_host, _capi
from fdgs.fused import render_raw
from fdgs.loss import fused_l1_ssim
from fdgs.pipeline import StepPipeline
dev = torch.device("cuda:0")
cfg = synth.SceneConfig("pipe", 6000, 208, 160, 3, 2, 0.03, 10.0, True, 4, False)
scene = synth.make_scene(cfg, seed=4)
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
pipe = train_host.PipelineFlags()
B = 3
cams = [train_host.SyntheticCamera(scene, dev, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
gen = torch.Generator(device="cpu").manual_seed(7)
gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(dev) for _ in range(B)]
ma = train_host.GaussianParams(scene, dev); oa = train_host.make_optimizer(ma); sink = ma.grad_sink()
ref = []
for _ in range(2):
    for b in range(B):
        loss = fused_l1_ssim(render_raw(cams[b], ma, pipe, bg, grad_sink=sink, accumulate=b > 0)["render"], gts[b], 0.2)
        (loss / B).backward(); ref.append(float(loss))
    oa.step()
print("ref", ref)
for rep in range(3):
    for overlap in (False, True):
        for batch in (True, False):
            for lazy in (True, False):
                mp = train_host.GaussianParams(scene, dev)
                sp = StepPipeline(mp, train_host.make_optimizer(mp), world_size=1, lambda_dssim=0.2, overlap=overlap, fuse_sh_adam=True, batch_views=batch, sh_group=1, lazy=lazy)
                got = []
                for _ in range(2):
                    results, losses = sp.step(cams, gts, pipe, bg)
                    got += [float(l) for l in losses]
                torch.cuda.synchronize()
                ok = np.allclose(got, ref, rtol=1e-5, atol=1e-6)
                print(rep, "overlap", overlap, "batch", batch, "lazy", lazy, "redone", sp.lazy_redone, "OK" if ok else "MISMATCH %s" % got, [r["num_rendered"] for r in results])
