"""Quick timing of the native forward / backward on a synthetic config (dev tool, not the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from fdgs import synth
from util import native_args_fwd, scene_to_device
from fdgs.gaussian_renderer.diff_gaussian_rasterization import _C

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
colour_only = len(sys.argv) > 3 and sys.argv[3] == "colour"
tile_cull = not (len(sys.argv) > 4 and sys.argv[4] == "reflists")   # fdgs_forward_out.tile_cull, as the training step runs
# depth / alpha / flow without upstream gradient (the training case)
dev = torch.device("cuda:0")
sc = scene_to_device(synth.make_scene(synth.CONFIGS[name], seed=0), dev)
g = {k: v.to(dev) for k, v in synth.make_upstream_grads(sc["W"], sc["H"], seed=1, scale=1e-2).items()}
e = torch.Tensor([])
gg = lambda k: sc[k] if sc.get(k) is not None else e
def fwd():
    return _C.rasterize_gaussians(*native_args_fwd(sc), tile_cull=tile_cull)
def bwd(res):
    (R, color, flow, depth, T, radii, geom, binb, img, covs_com, om) = res
    return _C.rasterize_gaussians_backward(sc["bg"], sc["means3D"], om, radii, gg("colors_precomp"), gg("flow_2d"), sc["opacities"],
        gg("ts"), gg("scales"), gg("scales_t"), gg("rotations"), gg("rotations_r"), 1.0, gg("cov3D_precomp"), -1.0,
        sc["world_view_transform"], sc["full_proj_transform"], sc["tanfovx"], sc["tanfovy"], g["grad_color"], None if colour_only else g["grad_depth"],
        None if colour_only else g["grad_alpha"], None if colour_only else g["grad_flow"], gg("shs"), sc["sh_degree"], sc["sh_degree_t"], sc["camera_center"], sc["timestamp"],
        sc["time_duration"], sc["rot_4d"], sc["gaussian_dim"], sc["force_sh_3d"], geom, R, binb, img, False)
for _ in range(3):
    r = fwd(); bwd(r)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters):
    r = fwd()
torch.cuda.synchronize()
t1 = time.time()
for _ in range(iters):
    bwd(r)
torch.cuda.synchronize()
t2 = time.time()
print("%s R=%d fwd %.3f ms  bwd %.3f ms" % (name, r[0], (t1 - t0) / iters * 1e3, (t2 - t1) / iters * 1e3))
