"""Golden fixtures for densification / pruning, produced by running the REFERENCE's own
``GaussianModel.densify_and_prune`` (scene/gaussian_model.py:584-610 and everything it calls) on the CPU of the
build container.  The module's CUDA-only imports are stubbed and ``device="cuda"`` is redirected to the CPU; the
normal samples ``densify_and_split`` draws are recorded so that the implementation under test can be fed the same ones.

    python tests/golden/make_golden_densify.py   ->   tests/golden/densify_*.npz
"""
import os, sys, types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests"))

for name in ("pointops2", "pointops2.functions", "pointops2.functions.pointops", "simple_knn", "simple_knn._C", "plyfile"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["pointops2.functions.pointops"].furthestsampling = None
sys.modules["pointops2.functions.pointops"].knnquery = None
sys.modules["simple_knn._C"].distCUDA2 = None
sys.modules["plyfile"].PlyData = None
sys.modules["plyfile"].PlyElement = None
sys.path.insert(0, "/root/reference")


def _cpu(fn):
    def wrapped(*a, **k):
        if k.get("device", None) is not None and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


for fname in ("zeros", "ones", "empty", "tensor", "full", "rand", "randn"):
    setattr(torch, fname, _cpu(getattr(torch, fname)))

import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_gaussian_model", "/root/reference/scene/gaussian_model.py")
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)   # the reference module itself (scene/__init__.py would pull in the dataset readers)
GaussianModel = _mod.GaussianModel
from fdgs import synth  # noqa: E402

RECORDED = []
_normal = torch.normal


def recording_normal(*a, **k):
    out = _normal(*a, **k)
    RECORDED.append(out.detach().clone())
    return out


torch.normal = recording_normal


def run(tag, cfg, seed, max_grad, min_opacity, extent, max_screen_size, max_grad_t, prune_only, warm_steps=2):
    scene = synth.make_scene(cfg, seed=seed)
    P, M = scene["means3D"].shape[0], scene["M"]
    g = GaussianModel(sh_degree=cfg.sh_degree, gaussian_dim=cfg.gaussian_dim, time_duration=[0.0, scene["time_duration"]],
                      rot_4d=cfg.rot_4d, force_sh_3d=cfg.force_sh_3d, sh_degree_t=cfg.sh_degree_t)
    nn = torch.nn
    inv_sig = lambda x: torch.log(x / (1 - x))
    g._xyz = nn.Parameter(scene["means3D"].clone().requires_grad_(True))
    g._features_dc = nn.Parameter(scene["shs"][:, :1].clone().contiguous().requires_grad_(True))
    g._features_rest = nn.Parameter(scene["shs"][:, 1:].clone().contiguous().requires_grad_(True))
    g._opacity = nn.Parameter(inv_sig(scene["opacities"].clamp(1e-6, 1 - 1e-6)).requires_grad_(True))
    g._scaling = nn.Parameter(torch.log(scene["scales"]).requires_grad_(True))
    g._rotation = nn.Parameter(scene["rotations"].clone().requires_grad_(True))
    if cfg.gaussian_dim == 4:
        g._t = nn.Parameter(scene["ts"].clone().requires_grad_(True))
        g._scaling_t = nn.Parameter(torch.log(scene["scales_t"]).requires_grad_(True))
        if cfg.rot_4d:
            g._rotation_r = nn.Parameter(scene["rotations_r"].clone().requires_grad_(True))
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                 position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3,
                                 position_t_lr_init=-1.0)
    g.spatial_lr_scale = 1.0
    g.training_setup(args)
    gen = torch.Generator().manual_seed(1000 + seed)
    for _ in range(warm_steps):  # populate Adam's exp_avg / exp_avg_sq
        for grp in g.optimizer.param_groups:
            p = grp["params"][0]
            p.grad = 1e-3 * torch.randn(p.shape, generator=gen)
        g.optimizer.step()
    g.max_radii2D = torch.randint(0, 40, (P,), generator=gen).float()
    g.denom = torch.randint(0, 5, (P, 1), generator=gen).float()
    g.xyz_gradient_accum = torch.rand(P, 1, generator=gen) * g.denom * 4 * max_grad
    if cfg.gaussian_dim == 4:
        g.t_gradient_accum = torch.rand(P, 1, generator=gen) * g.denom * 1e-4

    def snap(prefix, out):
        names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
                 "rotation": "_rotation", "t": "_t", "scaling_t": "_scaling_t", "rotation_r": "_rotation_r"}
        for grp in g.optimizer.param_groups:
            p = grp["params"][0]
            out[prefix + grp["name"]] = p.detach().numpy().copy()
            st = g.optimizer.state.get(p, None)
            if st is not None:
                out[prefix + grp["name"] + ".exp_avg"] = st["exp_avg"].numpy().copy()
                out[prefix + grp["name"] + ".exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
        out[prefix + "max_radii2D"] = g.max_radii2D.numpy().copy()
        out[prefix + "denom"] = g.denom.numpy().copy()
        out[prefix + "xyz_gradient_accum"] = g.xyz_gradient_accum.numpy().copy()
        if cfg.gaussian_dim == 4:
            out[prefix + "t_gradient_accum"] = g.t_gradient_accum.numpy().copy()

    out = {}
    snap("in.", out)
    RECORDED.clear()
    torch.manual_seed(4242 + seed)
    g.densify_and_prune(max_grad, min_opacity, extent, max_screen_size, max_grad_t, prune_only=prune_only)
    snap("out.", out)
    for i, s in enumerate(RECORDED):
        out["normal.%d" % i] = s.numpy().copy()
    out["args"] = np.array([max_grad, min_opacity, extent, -1.0 if max_screen_size is None else max_screen_size,
                            -1.0 if max_grad_t is None else max_grad_t, float(prune_only), args.percent_dense], dtype=np.float64)
    out["cfg"] = np.array([cfg.sh_degree, cfg.sh_degree_t, cfg.gaussian_dim, int(cfg.rot_4d)], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "densify_%s.npz" % tag), **out)
    print(tag, "P %d -> %d, %d recorded normal draws" % (P, out["out.xyz"].shape[0], len(RECORDED)))


SC = synth.SceneConfig
run("rot4d", SC("d", 300, 64, 48, 3, 2, 0.05, 10.0, True, 4, False), 1, 2e-4, 0.3, 6.0, 20, 2e-4 / 40, False)
run("rot4d_pruneonly", SC("d", 250, 64, 48, 3, 2, 0.05, 10.0, True, 4, False), 2, 2e-4, 0.3, 6.0, 20, 2e-4 / 40, True)
run("dim4_norot", SC("d", 500, 64, 48, 1, 0, 0.05, 1.0, False, 4, True), 3, 2e-4, 0.3, 6.0, None, 2e-4 / 40, False)
run("dim3", SC("d", 500, 64, 48, 2, 0, 0.05, 1.0, False, 3, False), 4, 2e-4, 0.3, 6.0, 20, None, False)
