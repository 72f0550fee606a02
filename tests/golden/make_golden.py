#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with the reference's OWN kernel source compiled for the
CPU (oracle/_ref/liboracle_ref.so, built by oracle/refbuild/build_ref.py from /root/reference).

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py [fixture names]
Each fixture is a compressed .npz holding the seeded inputs (regenerable from fdgs.synth with the stored
config / seed, kept anyway so the file is self-contained) and every forward intermediate and gradient the
reference produces: radii, tiles_touched, depths, means2D, conic_opacity, rgb, cov3D, out_means3D, clamped,
point_offsets, sorted keys, point_list, ranges, n_contrib, final T, images, and the 13 gradient tensors.
The reference ships no tests or golden vectors for this path: these files are the pin.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fdgs import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

SC = synth.SceneConfig
HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (config, seed, make_scene kwargs)
FIXTURES = {
    "rot4d_sh3t2": (SC("g", 400, 80, 64, 3, 2, 0.05, 10.0, True, 4, False), 11, dict(random_flow=True, bg=(0.2, 0.3, 0.4))),
    "rot4d_sh0": (SC("g", 900, 96, 96, 0, 0, 0.04, 1.0, True, 4, True), 12, dict()),
    "dim3_sh2": (SC("g", 700, 100, 60, 2, 0, 0.04, 1.0, False, 3, False), 13, dict(random_flow=True)),
    "dim4_norot_sh1": (SC("g", 700, 90, 70, 1, 0, 0.04, 1.0, False, 4, True), 14, dict(bg=(1.0, 1.0, 1.0))),
    # the two flags no other fixture sets: scale_modifier != 1 (forward.cu:418-434, backward.cu:911-916) and prefilter_var > 0
    # (forward.cu:333, 434, backward.cu:746), on the rot_4d path and on the 3D-covariance + temporal-marginal path
    "rot4d_sh1t1_mod07_pf02": (SC("g", 500, 88, 72, 1, 1, 0.05, 4.0, True, 4, False), 15, dict(random_flow=True, bg=(0.1, 0.0, 0.3))),
    "dim4_norot_sh2_mod16_pf005": (SC("g", 500, 96, 64, 2, 0, 0.03, 1.0, False, 4, True), 16, dict()),
    # round 5 -- general cameras (fdgs.synth.POSES: every entry of the view / projection matrices live: forward.cu:198-237,
    # backward.cu:486-617, 878-894) and M = 48 allocated with lower degrees active (scene/gaussian_model.py:65,92,253-257):
    # a rotated rig camera with the centre-shift projection; the slanted one (Gaussians behind z <= 0.2 and beyond the 1.3 tanfov
    # clamp, auxiliary.h:153, forward.cu:206-211); degrees (1, 0) and (3, 1) active of the allocated (3, 2), on rotated cameras
    "rot4d_sh3t2_rig2": (SC("g", 500, 96, 72, 3, 2, 0.05, 10.0, True, 4, False), 17, dict(random_flow=True, bg=(0.3, 0.2, 0.1), pose="rig2")),
    "rot4d_sh3t1_slant": (SC("g", 1500, 96, 72, 3, 1, 0.04, 4.0, True, 4, False), 18, dict(pose="slant")),
    "dim3_sh2_rig0": (SC("g", 700, 100, 60, 2, 0, 0.04, 1.0, False, 3, False), 19, dict(random_flow=True, pose="rig0")),
    "rot4d_alloc48_deg10_rig1": (SC("g", 500, 88, 72, 1, 0, 0.05, 4.0, True, 4, False), 20, dict(pose="rig1", alloc=(3, 2), bg=(0.1, 0.0, 0.3))),
    "rot4d_alloc48_deg31_rig3": (SC("g", 500, 88, 72, 3, 1, 0.05, 4.0, True, 4, False), 21, dict(pose="rig3", alloc=(3, 2), random_flow=True)),
    # round 6 -- ARBITRARY ORIENTATIONS: uniformly distributed unit quaternions for `rotations` (computeCov3D and its backward,
    # forward.cu:242-276, backward.cu:621-684) and for the pair (`rotations`, `rotations_r`) of the 4D rotation M_r M_l
    # (computeCov3D_conditional, forward.cu:279-352, backward.cu:689-834); every fixture before this one draws them within ~6 degrees
    # of identity.  On rotated cameras.  The rot_4d ones with scales_t x 2 and the pairs whose splat is wider than 40 px / more
    # elongated than 1:5 on screen redrawn (tests/util.py::bounded_footprint: the plain 1e-4 gradient bar applies to what is left)
    "rot4d_sh3t1_uniform_rig1": (SC("g", 600, 96, 72, 3, 1, 0.05, 2.0, True, 4, False), 22, dict(pose="rig1", rot_sigma="uniform", st_scale=2.0, random_flow=True, bg=(0.2, 0.1, 0.4))),
    "rot4d_sh3t2_uniform_rig0": (SC("g", 600, 96, 72, 3, 2, 0.05, 2.0, True, 4, False), 24, dict(pose="rig0", rot_sigma="uniform", st_scale=2.0)),
    "dim3_sh2_uniform_rig3": (SC("g", 700, 100, 60, 2, 0, 0.04, 1.0, False, 3, False), 23, dict(pose="rig3", rot_sigma="uniform", random_flow=True)),
    "dim4_norot_sh1_uniform_rig2": (SC("g", 700, 90, 70, 1, 0, 0.04, 1.0, False, 4, True), 25, dict(pose="rig2", rot_sigma="uniform", bg=(1.0, 1.0, 1.0))),
}
BOUNDED = ("rot4d_sh3t1_uniform_rig1", "rot4d_sh3t2_uniform_rig0")
# scene-dict overrides applied after make_scene (the flags travel in the fixture as sc_scale_modifier / sc_prefilter_var)
OVERRIDES = {
    "rot4d_sh1t1_mod07_pf02": dict(scale_modifier=0.7, prefilter_var=0.2),
    "dim4_norot_sh2_mod16_pf005": dict(scale_modifier=1.6, prefilter_var=0.05),
}

INPUT_KEYS = ("means3D", "ts", "scales", "scales_t", "rotations", "rotations_r", "opacities", "shs", "flow_2d", "bg",
              "world_view_transform", "full_proj_transform", "camera_center")
SCALAR_KEYS = ("W", "H", "sh_degree", "sh_degree_t", "timestamp", "time_duration", "rot_4d", "gaussian_dim",
               "force_sh_3d", "scale_modifier", "prefilter_var", "tanfovx", "tanfovy")


def scene_for(name):
    cfg, seed, kw = FIXTURES[name]
    sc = synth.make_scene(cfg, seed=seed, **kw)
    sc.update(OVERRIDES.get(name, {}))
    if name in BOUNDED:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import bounded_footprint
        bounded_footprint(sc)
    return sc


def main():
    if pyoracle.build_ref() is None:
        raise SystemExit("needs /root/reference to build oracle/_ref/liboracle_ref.so")
    only = sys.argv[1:]   # fixture names to (re)generate; default: all (gradients differ in the last bits from run to run: atomics)
    for name in FIXTURES:
        if only and name not in only:
            continue
        sc = scene_for(name)
        g = synth.make_upstream_grads(sc["W"], sc["H"], seed=1, scale=1e-2)
        o = pyoracle.Oracle(sc, kind="reference")
        out = dict(o.forward())
        gr = o.backward(g["grad_color"], g["grad_depth"], g["grad_alpha"], g["grad_flow"])
        data = {}
        for k in INPUT_KEYS:
            data["in_" + k] = sc[k].numpy()
        for k in SCALAR_KEYS:
            data["sc_" + k] = np.asarray(sc[k])
        for k, v in g.items():
            data["up_" + k] = v.numpy()
        for k, v in out.items():
            if k in ("border", "border_g"):
                continue
            data["fw_" + k] = v
        data["fw_R"] = np.asarray(o.R)
        for k, v in gr.items():
            data["bw_" + k] = v
        o.close()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print("%-16s R=%6d  %7.1f KiB" % (name, o.R, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
