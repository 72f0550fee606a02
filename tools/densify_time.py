"""Timing of densify_and_prune at workload size against the same re-layout done the reference's way (boolean-mask
gathers + torch.cat per tensor and per Adam moment).  Dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdgs import synth, train_host, harness
from fdgs.densify import densify_and_prune
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
scene = synth.make_scene(synth.CONFIGS[name], seed=0)


def fresh():
    model = train_host.GaussianParams(scene, dev)
    opt = train_host.make_optimizer(model)
    stats = harness.DensificationStats(model.P, dev, 1)
    g = torch.Generator(device="cpu").manual_seed(1)
    stats.denom.copy_(torch.randint(1, 4, (model.P, 1), generator=g).float())
    stats.xyz_gradient_accum.copy_(torch.rand(model.P, 1, generator=g) * stats.denom.cpu() * 6e-4)
    return model, opt, stats


extent = float(torch.exp(train_host.GaussianParams(scene, dev)._scaling).max(1).values.median() / 0.01)
for trial in range(3):
    model, opt, stats = fresh()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rep = densify_and_prune(model, opt, stats, 2e-4, 0.005, extent, 20, generator=torch.Generator(device=dev).manual_seed(3))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fdgs densify_and_prune %s: %.2f ms  %s" % (name, dt * 1e3, rep))

# the reference's way on the same decisions: per tensor, per moment: mask gather + cat (gaussian_model.py:391-485)
model, opt, stats = fresh()
P = model.P
grads = stats.xyz_gradient_accum / stats.denom
scal = torch.exp(model._scaling).max(1).values
clone = (grads[:, 0] >= 2e-4) & (scal <= 0.01 * extent)
split = (grads[:, 0] >= 2e-4) & (scal > 0.01 * extent)
tensors = [model.params[n].detach() for n in model.NAMES]
moments = [[opt.exp_avg[slice(*model.offsets[n])].view(model.params[n].shape) for n in model.NAMES],
           [opt.exp_avg_sq[slice(*model.offsets[n])].view(model.params[n].shape) for n in model.NAMES]]
for trial in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = []
    for group in [tensors] + moments:
        new = []
        for t in group:
            a = torch.cat((t, t[clone]), 0)                                   # densify_and_clone + cat_tensors_to_optimizer
            sel = torch.cat((split, torch.zeros(int(clone.sum()), dtype=torch.bool, device=dev)))
            b = torch.cat((a, a[sel].repeat(2, *([1] * (t.dim() - 1)))), 0)   # densify_and_split
            keep = ~torch.cat((sel, torch.zeros(2 * int(sel.sum()), dtype=torch.bool, device=dev)))
            c = b[keep]                                                       # prune_points(prune_filter)
            new.append(c[torch.ones(c.shape[0], dtype=torch.bool, device=dev)])  # final prune_points(prune_mask)
        out.append(new)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("mask-gather + cat per tensor (reference structure, PyTorch): %.2f ms  -> %d Gaussians" % (dt * 1e3, out[0][0].shape[0]))
