#!/usr/bin/env python3
"""Golden vectors for the Python-side covariance of ``pipe.compute_cov3D_python`` (gaussian_renderer/__init__.py:73-81),
produced by the REFERENCE's own functions run on the CPU in the build container:

  * utils/general_utils.py (strip_symmetric, build_rotation, build_scaling_rotation, build_rotation_4d,
    build_scaling_rotation_4d) -- the file is read in place and executed with its hard-coded "cuda" device replaced by "cpu";
  * scene/gaussian_model.py: the methods setup_functions / get_scaling* / get_rotation* / get_t / get_cov_t / get_marginal_t /
    get_covariance / get_current_covariance_and_mean_offset are cut out of the class by their AST nodes (the module itself
    imports plyfile / simple_knn, which this image does not have) and compiled into a bare class.

No reference source is copied: this script reads /root/reference at run time.   python tests/golden/make_golden_pycov.py
Fixtures: tests/golden/pycov_{rot4d,dim4,dim3}.npz = raw parameters, scaling modifier, timestamp, prefilter_var and the
reference's covariance / mean offset / marginal.  tests/test_refmodel_host.py holds fdgs.train_host.ReferenceStyleModel to them.
"""
import ast
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
METHODS = ("setup_functions", "get_scaling", "get_scaling_t", "get_scaling_xyzt", "get_rotation", "get_rotation_r", "get_xyz", "get_t",
           "get_cov_t", "get_marginal_t", "get_covariance", "get_current_covariance_and_mean_offset")


def reference_model_class():
    ns = {"__name__": "ref_general_utils"}
    src = open(os.path.join(REF, "utils", "general_utils.py")).read().replace('"cuda"', '"cpu"').replace("'cuda'", "'cpu'")
    # pointops2 (a CUDA extension, out of scope) is imported at module level for two functions this path never calls
    import types
    stub = types.ModuleType("pointops2.functions.pointops")
    stub.furthestsampling = stub.knnquery = None
    for modname in ("pointops2", "pointops2.functions"):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    sys.modules.setdefault("pointops2.functions.pointops", stub)
    exec(compile(src, "general_utils.py", "exec"), ns)
    tree = ast.parse(open(os.path.join(REF, "scene", "gaussian_model.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianModel")
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS]
    cls.bases, cls.decorator_list = [], []
    mod = ast.Module(body=[cls], type_ignores=[])
    ns2 = dict(ns)
    ns2["torch"] = torch
    exec(compile(ast.fix_missing_locations(mod), "gaussian_model.py", "exec"), ns2)
    return ns2["GaussianModel"]


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference")
    Ref = reference_model_class()
    g = torch.Generator().manual_seed(5)
    n = 257
    for name, rot_4d, dim, mod, ts, pv in (("rot4d", True, 4, 0.8, 1.3, 0.15), ("dim4", False, 4, 1.5, 0.4, -1.0), ("dim3", False, 3, 1.0, 0.0, -1.0)):
        m = Ref.__new__(Ref)
        m.rot_4d, m.gaussian_dim, m.prefilter_var = rot_4d, dim, pv
        m._xyz = torch.randn(n, 3, generator=g)
        m._scaling = torch.randn(n, 3, generator=g) * 0.5 - 2.0
        m._scaling_t = torch.randn(n, 1, generator=g) * 0.5
        m._rotation = torch.randn(n, 4, generator=g)
        m._rotation_r = torch.randn(n, 4, generator=g)
        m._t = torch.randn(n, 1, generator=g) * 2.0
        m.setup_functions()
        d = {"scaling": m._scaling, "scaling_t": m._scaling_t, "rotation": m._rotation, "rotation_r": m._rotation_r, "t": m._t,
             "mod": torch.tensor(mod), "timestamp": torch.tensor(ts), "prefilter_var": torch.tensor(pv),
             "rot_4d": torch.tensor(rot_4d), "gaussian_dim": torch.tensor(dim)}
        if rot_4d:
            cov, off = m.get_current_covariance_and_mean_offset(mod, ts)
            d["cov"], d["mean_offset"] = cov, off
        else:
            d["cov"] = m.get_covariance(mod)
        if dim == 4:
            d["marginal_t"] = m.get_marginal_t(ts)
            d["cov_t"] = m.get_cov_t(mod)
        path = os.path.join(HERE, "pycov_%s.npz" % name)
        np.savez_compressed(path, **{k: v.numpy() for k, v in d.items()})
        print(path, {k: tuple(v.shape) for k, v in d.items() if v.dim()})


if __name__ == "__main__":
    sys.exit(main())
