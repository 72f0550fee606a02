import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from fdgs import synth, train_host
from fdgs.pipeline import StepPipeline
dev=torch.device('cuda:0')
cfg = synth.SceneConfig("ovb", 60012, 256, 192, 3, 2, 0.012, 10.0, True, 4, False)
scene = synth.make_scene(cfg, seed=6, pose="rig1")
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
pipe = train_host.PipelineFlags()
B, steps = 3, 6
cams = [train_host.SyntheticCamera(scene, dev, timestamp=(b + 0.5) / B * scene["time_duration"]) for b in range(B)]
gen = torch.Generator(device="cpu").manual_seed(9)
gts = [torch.rand(3, scene["H"], scene["W"], generator=gen).to(dev) for _ in range(B)]
for nofix in (False, True):
  for mode in ("plain","overlap"):
    m = train_host.GaussianParams(scene, dev)
    opt = train_host.make_optimizer(m)
    opt.set_lr("_features", 0.02, 0.02)
    sp = StepPipeline(m, opt, world_size=1, lambda_dssim=0.2, overlap_steps=mode=="overlap", batch_views=True)
    if nofix:
        # emulate the unfixed code: pretend no batching for the carried decision
        import fdgs.pipeline as P
    update = opt.step_sh_staged
    def slow_update(*a, _update=update, **k):
        torch.cuda._sleep(40_000_000)
        return _update(*a, **k)
    opt.step_sh_staged = slow_update
    losses=[]
    import time
    for k in range(steps):
        n = 1 if k % 2 == 0 else B
        t0=time.time()
        _res, ls = sp.step(cams[:n], gts[:n], pipe, bg)
        losses += [l.clone() for l in ls]
        print(mode, 'step',k,'n',n,'host ms %.2f'%((time.time()-t0)*1e3),'carried',sp.steps_carried, 'token', sp._carry is not None)
    torch.cuda.synchronize()
    print(mode, [float(l) for l in losses])
