"""CPU: the C-ABI library loads and exports everything include/fdgs.h declares; host-side argument handling
(no GPU compute is launched here)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "fdgs.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fdgs_[a-z_0-9]+)\s*\(", text)) - {"fdgs_alloc_fn"})


def test_library_exports_every_declared_symbol():
    from fdgs import _capi
    names = _declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(_capi.lib, n), "libfdgs.so does not export %s" % n
        assert n in _capi.EXPORTED or n in ("fdgs_alloc_fn",), "binding list misses %s" % n
    assert _capi.lib.fdgs_version() == _capi.FDGS_VERSION == 502
    with open(os.path.join(ROOT, "include", "fdgs.h")) as f:
        assert re.search(r"#define FDGS_VERSION (\d+)", f.read()).group(1) == str(_capi.FDGS_VERSION)


def test_scratch_sizes_are_monotone_and_aligned():
    from fdgs import _capi
    g1, g2 = _capi.lib.fdgs_geometry_bytes(1000), _capi.lib.fdgs_geometry_bytes(300000)
    assert 0 < g1 < g2 and g1 % 256 == 0 and g2 % 256 == 0
    assert g2 / 300000 < 100  # 89 B / Gaussian of scratch
    i = _capi.lib.fdgs_image_bytes(1352, 1014)
    assert i % 256 == 0 and i >= 1352 * 1014 * 8
    b = _capi.lib.fdgs_binning_bytes(3_000_000, 1352, 1014)
    assert b % 256 == 0 and 12 * 3_000_000 <= b < 13 * 3_000_000  # point_list 4 B + (depth bits, id) pairs 8 B per instance + cull planes
    # the cull planes (one bit per list entry and 8x8 block + one word per tile and plane): 4 x (R / 64 + T + 2) 64-bit words
    T = ((1352 + 15) // 16) * ((1014 + 15) // 16)
    assert b >= 12 * 3_000_000 + 32 * (3_000_000 // 64 + T + 2)
    small = _capi.lib.fdgs_binning_bytes(10, 1352, 1014)   # few instances on many tiles: the per-tile words dominate
    assert small >= 32 * (T + 2) and small < 64 * (T + 2) + 4096


def test_argument_errors_are_reported_without_touching_the_gpu():
    from fdgs import _capi
    scene = _capi.FdgsScene()
    scene.P, scene.W, scene.H = 10, 0, 16
    out = _capi.FdgsForwardOut()
    R = C.c_int32(0)
    cb = _capi.ALLOC_FN(lambda u, w, n: None)
    rc = _capi.lib.fdgs_rasterize_forward(C.byref(scene), C.byref(out), cb, None, None, C.byref(R))
    assert rc == 1 and "bad sizes" in _capi.last_error()
    scene.W = 16  # sizes fine, required pointers missing
    rc = _capi.lib.fdgs_rasterize_forward(C.byref(scene), C.byref(out), cb, None, None, C.byref(R))
    assert rc == 1 and "must not be NULL" in _capi.last_error()
    assert _capi.lib.fdgs_mark_visible(-1, None, None, None, None, None) == 1
    assert _capi.lib.fdgs_profile_read(99, None, None) == 1


def test_structs_carry_their_size_and_a_short_struct_is_rejected():
    """Every ABI struct starts with struct_size; a caller built against another header (here: a size that is off by one field) gets
    FDGS_ERR_INVALID_ARG instead of the library reading past the end of its struct."""
    from fdgs import _capi
    for cls in (_capi.FdgsScene, _capi.FdgsForwardOut, _capi.FdgsBackwardIn, _capi.FdgsBackwardOut, _capi.FdgsDebugView):
        assert cls().struct_size == C.sizeof(cls) and cls._fields_[0][0] == "struct_size"
    scene, out = _capi.FdgsScene(), _capi.FdgsForwardOut()
    scene.P, scene.W, scene.H = 10, 16, 16
    R = C.c_int32(0)
    cb = _capi.ALLOC_FN(lambda u, w, n: None)
    scene.struct_size -= 4
    assert _capi.lib.fdgs_rasterize_forward(C.byref(scene), C.byref(out), cb, None, None, C.byref(R)) == 1
    assert "struct_size" in _capi.last_error() and "fdgs_scene" in _capi.last_error()
    scene.struct_size += 4
    out.struct_size = 0
    assert _capi.lib.fdgs_rasterize_forward(C.byref(scene), C.byref(out), cb, None, None, C.byref(R)) == 1
    assert "fdgs_forward_out" in _capi.last_error()
    bi, bo = _capi.FdgsBackwardIn(), _capi.FdgsBackwardOut()
    bo.struct_size += 8
    assert _capi.lib.fdgs_rasterize_backward(C.byref(scene), C.byref(bi), C.byref(bo), None) == 1
    assert "fdgs_backward_out" in _capi.last_error()


def _settings(**kw):
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import GaussianRasterizationSettings
    base = dict(image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5, bg=torch.zeros(3), scale_modifier=1.0,
                viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, sh_degree_t=0, campos=torch.zeros(3),
                timestamp=0.0, time_duration=1.0, rot_4d=True, gaussian_dim=4, force_sh_3d=False, prefiltered=False,
                debug=False)
    base.update(kw)
    return GaussianRasterizationSettings(**base)


def test_rasterizer_keeps_the_reference_exceptions_and_signature():
    """gaussian_renderer/diff_gaussian_rasterization.py:263-280: same keyword names / defaults, same Exceptions."""
    import inspect
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
    assert list(GaussianRasterizationSettings._fields) == [
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "sh_degree_t", "campos", "timestamp", "time_duration", "rot_4d", "gaussian_dim", "force_sh_3d",
        "prefiltered", "debug"]
    sig = inspect.signature(GaussianRasterizer.forward)
    assert list(sig.parameters)[1:] == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "flow_2d", "ts",
                                        "scales", "scales_t", "rotations", "rotations_r", "cov3D_precomp",
                                        "prefilter_var"]
    assert sig.parameters["prefilter_var"].default == -1.0
    r = GaussianRasterizer(_settings())
    P = 4
    m, o = torch.zeros(P, 3), torch.ones(P, 1)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, shs=torch.zeros(P, 1, 3), colors_precomp=torch.zeros(P, 3))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, colors_precomp=torch.zeros(P, 3))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, colors_precomp=torch.zeros(P, 3), scales=torch.ones(P, 3), rotations=torch.ones(P, 4),
          cov3D_precomp=torch.zeros(P, 6))
    with pytest.raises(Exception, match="rotations_r and scales_t and ts"):
        r(m, m, o, colors_precomp=torch.zeros(P, 3), scales=torch.ones(P, 3), rotations=torch.ones(P, 4))


def test_no_cpu_fallback():
    """The product path must fail loudly, not silently fall back, when handed CPU tensors."""
    from fdgs.gaussian_renderer.diff_gaussian_rasterization import GaussianRasterizer
    r = GaussianRasterizer(_settings(rot_4d=False, gaussian_dim=3))
    P = 4
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1), colors_precomp=torch.zeros(P, 3),
          scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(torch.zeros(P, 3))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, include or load it."""
    pkg = os.path.join(ROOT, "4d-gaussian-splatting_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if not fn.endswith((".py", ".hip", ".h", ".sh")):
                continue
            with open(os.path.join(dirpath, fn)) as f:
                for ln in f:
                    low = ln.lower()
                    if "oracle" in low or "ref_hip" in low or "refhip" in low:
                        assert not any(tok in low for tok in ("import", "#include", "cdll", "dlopen", "liboracle", "ref_hip", "refhip")), \
                            "%s references the oracle: %s" % (fn, ln.strip())


def test_only_the_baseline_leg_of_the_bench_touches_the_oracle():
    """bench.py may use oracle/ in its baseline leg only (cpu_baseline and its GPU half, reference_kernels_on_this_gpu): every import of it
    sits inside those two functions."""
    import ast
    with open(os.path.join(ROOT, "bench.py")) as f:
        tree = ast.parse(f.read())
    allowed = {"cpu_baseline", "reference_kernels_on_this_gpu"}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            for sub in ast.walk(node):
                if isinstance(sub, ast.ImportFrom) and sub.module and sub.module.split(".")[0] == "oracle":
                    assert node.name in allowed, "bench.py::%s imports the oracle" % node.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any(getattr(n, "module", "") and n.module.split(".")[0] == "oracle" for n in top)


def test_fdgs_adam_rejects_what_it_does_not_implement():
    """fdgs.optim.Adam (the torch.optim.Optimizer of the drop-in path): no weight decay / amsgrad, no CPU parameters."""
    import pytest
    import torch
    from fdgs.optim import Adam
    p = torch.nn.Parameter(torch.zeros(4))
    with pytest.raises(ValueError, match="weight_decay"):
        Adam([{"params": [p], "name": "xyz"}], lr=0.0, weight_decay=0.1)
    opt = Adam([{"params": [p], "name": "xyz"}], lr=0.0)
    assert not opt.bucketable() and not opt.ensure_homed()
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="GPU"):
        opt.step()


def test_blend_bwd_inline_asm_register_allocation():
    """csrc/blend_bwd.hip's joint DPP reductions take b[] as plain inputs that are read after a[] has been written; that no b[k]
    shares a VGPR with an a[j] is checked in the gfx950 assembly the compiler produces (tools/check_reduce_regs.py), so that a
    compiler bump cannot break it silently."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_reduce_regs", os.path.join(ROOT, "tools", "check_reduce_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sizes = [r[0] for r in mod.check()]
    assert 9 in sizes and 12 in sizes


def test_register_budgets_of_the_built_kernels():
    """Occupancy is part of the design (DESIGN.md sections 4 and 8): the per-Gaussian kernels are built without the SLP vectorizer, whose
    packed operations on register pairs cost preprocess_fwd 62 VGPRs and preprocess_bwd 35; the SSIM forward filters four packed channels.
    The VGPR counts of the gfx950 code the compiler produces with build.sh's flags are checked here, so that a flag that goes missing or a
    compiler bump shows up as a failed test and not as a slower step."""
    import re
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(ROOT, "4d-gaussian-splatting_amd", "csrc")
    with open(os.path.join(csrc, "build.sh")) as f:
        sh = f.read()
    extra = dict(re.findall(r'\[(\w+)\]="([^"]*)"', re.search(r"declare -A EXTRA=\((.*?)\)\n", sh, re.S).group(1)))
    common = re.search(r'COMMON="([^"]*)"', sh).group(1).replace("$ARCH", "gfx950").split()
    budgets = {   # kernel name fragment -> (translation unit, most VGPRs, fewest waves per SIMD that means)
        "preprocess_fwd_kernelILi0E": ("preprocess_fwd", 184, 2), "preprocess_fwd_kernelILi2E": ("preprocess_fwd", 168, 3),
        "preprocess_bwd_kernelE": ("preprocess_bwd", 104, 4), "ssim_fwd_kernel": ("ssim", 96, 5), "ssim_bwd_kernel": ("ssim", 96, 5),
    }

    def vgprs(tu):
        cmd = ["/opt/rocm/bin/hipcc"] + common + extra.get(tu, "").split() + ["-S", "--cuda-device-only", "-o", "-", os.path.join(csrc, tu + ".hip")]
        asm = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
        return {m.group(1): (int(m.group(2)), int(m.group(3))) for m in
                re.finditer(r"\.name:\s+(\S+)\n.*?\.vgpr_count:\s+(\d+)\n.*?\.vgpr_spill_count:\s+(\d+)", asm, re.S)}
    tus = sorted({v[0] for v in budgets.values()})
    with ThreadPoolExecutor(len(tus)) as ex:
        got = dict(zip(tus, ex.map(vgprs, tus)))
    for frag, (tu, most, waves) in budgets.items():
        hits = {k: v for k, v in got[tu].items() if frag in k}
        assert hits, (frag, list(got[tu]))
        for name, (n, spills) in hits.items():
            assert n <= most and spills == 0, "%s: %d VGPRs, %d spills (budget %d = %d waves per SIMD)" % (name, n, spills, most, waves)
            assert 512 // ((n + 7) // 8 * 8) >= waves, (name, n)
