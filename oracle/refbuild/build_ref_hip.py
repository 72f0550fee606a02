#!/usr/bin/env python3
"""TEST / BASELINE INFRASTRUCTURE ONLY -- builds oracle/_ref/liboracle_ref_hip.so: the REFERENCE'S OWN CUDA rasterizer compiled for gfx950.

SURVEY.md section 8c row 3 ("secondary oracle, GPU box only") as a cross-compiled artefact: /root/reference does not exist on the GPU box
and no reference text may travel, but hipcc cross-compiles here, and a built .so under oracle/_ref/ travels like every other one.

Recipe: forward.cu / backward.cu / rasterizer_impl.cu and their headers are read in place from
/root/reference/diff-gaussian-rasterization/cuda_rasterizer into a TEMPORARY directory outside the repository, translated there by
``torch.utils.hipify`` (the tool PyTorch-ROCm itself uses for CUDA extensions: cuda.h / cuda_runtime.h / cub -> hip / hipcub), plus four
token-level edits of the temporary copy that the translator leaves undone:
  * the ``#include`` lines of <cooperative_groups/reduce.h> and "device_launch_parameters.h" are dropped (nothing of theirs is used:
    no cg::reduce call exists in the sources; the second is an nvcc IDE helper);
  * ``<< <`` / ``>> >`` (the launch brackets written with a space) -> ``<<<`` / ``>>>``;
  * ``__trap()`` -> ``__builtin_trap()`` (auxiliary.h:159, the debug-only bounds check).
The kernels -- every arithmetic statement of the reference -- are compiled as they are, with hipcc's DEFAULT floating-point contraction
(fused multiply-adds where the compiler sees fit: what nvcc does by default too), -O3, vendored GLM from the reference tree, hipcub
for the two CUB calls.  oracle/refbuild/ref_hip_api.cpp (ours) wraps Rasterizer::forward / backward in a C ABI.

Usage: python oracle/refbuild/build_ref_hip.py [--reference /root/reference];  exit code 3 when the reference tree is absent.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref")
OUT = os.path.join(OUT_DIR, "liboracle_ref_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    dgr = os.path.join(args.reference, "diff-gaussian-rasterization")
    cr = os.path.join(dgr, "cuda_rasterizer")
    if not os.path.isdir(cr):
        print("reference tree not found at %s" % cr, file=sys.stderr)
        return 3
    from torch.utils.hipify import hipify_python
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="fdgs_refhip_")
    try:
        src, hip = os.path.join(tmp, "src"), os.path.join(tmp, "hip")
        shutil.copytree(cr, src)
        with open(os.devnull, "w") as null:      # (the translator prints a line per file)
            old = sys.stdout
            sys.stdout = null
            try:
                hipify_python.hipify(project_directory=src, output_directory=hip, includes=["*"], extensions=(".cu", ".h", ".cuh"),
                                     show_detailed=False, is_pytorch_extension=True, hipify_extra_files_only=False)
            finally:
                sys.stdout = old
        for name in os.listdir(hip):
            path = os.path.join(hip, name)
            with open(path) as f:
                text = f.read()
            text = "\n".join(l for l in text.split("\n") if "cooperative_groups/reduce.h" not in l and "device_launch_parameters.h" not in l)
            text = re.sub(r"<<\s+<", "<<<", text)
            text = re.sub(r">>\s+>", ">>>", text)
            text = re.sub(r"\b__trap\(\)", "__builtin_trap()", text)
            with open(path, "w") as f:
                f.write(text)
        objs = []
        common = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-I", hip, "-I", os.path.join(dgr, "third_party", "glm")]
        procs = []
        for stem in ("forward", "backward", "rasterizer_impl"):
            obj = os.path.join(tmp, stem + ".o")
            objs.append(obj)
            procs.append(subprocess.Popen(common + ["-c", os.path.join(hip, stem + ".hip"), "-o", obj]))
        api = os.path.join(tmp, "ref_hip_api.o")
        objs.append(api)
        procs.append(subprocess.Popen(common + ["-x", "hip", "-c", os.path.join(HERE, "ref_hip_api.cpp"), "-o", api]))
        for p in procs:
            if p.wait() != 0:
                return 1
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
