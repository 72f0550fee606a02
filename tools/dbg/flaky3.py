"""Which tensor of the autograd reference loop goes wrong after the polluting tests?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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

def ref_loop(tag):
    ma = train_host.GaussianParams(scene, dev); oa = train_host.make_optimizer(ma); sink = ma.grad_sink()
    trace = {}
    for s in range(2):
        for b in range(B):
            pkg = render_raw(cams[b], ma, pipe, bg, grad_sink=sink, accumulate=b > 0)
            loss = fused_l1_ssim(pkg["render"], gts[b], 0.2)
            (loss / B).backward()
            torch.cuda.synchronize()
            trace["s%d v%d image" % (s, b)] = pkg["render"].detach().clone()
            trace["s%d v%d loss" % (s, b)] = loss.detach().clone().reshape(1)
            trace["s%d v%d viewspace" % (s, b)] = pkg["viewspace_points"].grad.detach().clone()
            for n in ma.NAMES:
                trace["s%d v%d grad %s" % (s, b, n)] = ma.params[n].grad.detach().clone()
        oa.step()
        torch.cuda.synchronize()
        for n in ma.NAMES:
            trace["s%d param %s" % (s, n)] = ma.params[n].detach().clone()
        trace["s%d exp_avg" % s] = oa.exp_avg.clone()
        trace["s%d exp_avg_sq" % s] = oa.exp_avg_sq.clone()
    return trace

def compare(a, b, tag):
    bad = 0
    for k in a:
        d = (a[k].float() - b[k].float()).abs()
        scale = max(1e-12, float(a[k].float().abs().max()))
        rel = float(d.max()) / scale
        if rel > 1e-3:
            bad += 1
            if bad <= 12:
                print("  %s: %-28s max diff %.3e (scale %.3e, rel %.2e, %d elements beyond 1e-3 of scale)" % (tag, k, float(d.max()), scale, rel, int((d > 1e-3 * scale).sum())))
    print("%s: %d tensors differ" % (tag, bad))

base = ref_loop("base")
base2 = ref_loop("base2")
compare(base, base2, "isolated repeat")
import test_gpu_api as T
T.test_gradient_accumulation_over_views(dev)
for D, D_t, sh3d in [(3, 2, False), (3, 1, False), (3, 0, False), (2, 0, True), (0, 0, True)]:
    T.test_deferred_sh_gradient_matches_accumulation(dev, D, D_t, sh3d)
after = ref_loop("after")
compare(base, after, "after grad-acc + deferred tests")
for args in [(True, True, 3, True, 1), (False, True, 3, True, 1), (True, False, 3, True, 1)]:
    try:
        T.test_step_pipeline_matches_autograd_step(dev, *args)
        print("pipeline variant", args, "passed")
    except AssertionError as e:
        print("pipeline variant", args, "FAILED", str(e)[:300].replace("\n", " "))
    again = ref_loop("again")
    compare(base, again, "after pipeline variant %s" % (args,))
