#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds oracle/_ref/liboracle_ref.so.

Compiles the *reference's own* kernel source for the CPU: reads
forward.cu / backward.cu / rasterizer_impl.cu in place from
/root/reference/diff-gaussian-rasterization/cuda_rasterizer and simple_knn.cu from
/root/reference/simple-knn, cuts each file
just before its first host launcher (the only non-C++ syntax in them is the
`<<<...>>>` launch), drops the cut text into a *temporary* build directory
outside the repository, and compiles it with g++ -ffp-contract=off against the
shim headers in oracle/refbuild/shim plus the drivers in oracle/refbuild.
No reference source is copied into the repository; only the .so lands in
oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).

Usage: python oracle/refbuild/build_ref.py [--reference /root/reference]
Exit code 0 and prints the .so path on success; exit code 3 when the
reference tree is absent (e.g. on the GPU box -- the prebuilt .so is used).
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref")

CUTS = {
    "forward.cu": ("void FORWARD::render(", "forward_trunc.inc"),
    "backward.cu": ("void BACKWARD::preprocess(", "backward_trunc.inc"),
    "rasterizer_impl.cu": ("void CudaRasterizer::Rasterizer::markVisible(", "impl_trunc.inc"),
}
# simple-knn (distCUDA2): the device code up to the host function that needs cub / thrust (simple_knn.cu:193)
KNN_CUT = ("simple_knn.cu", "void SimpleKNN::knn(", "knn_trunc.inc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--opt", default="-O2")
    ap.add_argument("--contract", action="store_true",
                    help="ALSO build oracle/_ref/liboracle_ref_fma.so: the same source with -ffp-contract=fast -mfma, i.e. with the "
                         "compiler free to fuse a multiply and an add into one FMA the way nvcc does by default -- not nvcc's choices "
                         "(no CUDA toolchain exists here), but the same KIND of perturbation: tests/test_oracle_pin.py measures how far "
                         "the reference's own outputs move under it (what 'bit-exact' can and cannot mean against a CUDA device)")
    args = ap.parse_args()
    dgr = os.path.join(args.reference, "diff-gaussian-rasterization")
    cr = os.path.join(dgr, "cuda_rasterizer")
    if not os.path.isdir(cr):
        print("reference tree not found at %s" % cr, file=sys.stderr)
        return 3
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "liboracle_ref.so")
    tmp = tempfile.mkdtemp(prefix="fdgs_refbuild_")
    try:
        for name, (marker, inc) in CUTS.items():
            with open(os.path.join(cr, name)) as f:
                text = f.read()
            cut = text.find(marker)
            if cut < 0:
                raise RuntimeError("marker %r not found in %s" % (marker, name))
            with open(os.path.join(tmp, inc), "w") as f:
                f.write(text[:cut])
        knn = os.path.join(args.reference, "simple-knn")
        with open(os.path.join(knn, KNN_CUT[0])) as f:
            text = f.read()
        cut = text.find(KNN_CUT[1])
        if cut < 0:
            raise RuntimeError("marker %r not found in %s" % (KNN_CUT[1], KNN_CUT[0]))
        with open(os.path.join(tmp, KNN_CUT[2]), "w") as f:
            f.write(text[:cut])
        srcs = ["ref_emu.cpp", "ref_fwd.cpp", "ref_bwd.cpp", "ref_impl.cpp", "ref_api.cpp", "ref_knn.cpp"]
        cmd = ["g++", "-std=c++17", args.opt, "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-w",
               "-I", os.path.join(HERE, "shim"), "-I", HERE, "-I", tmp, "-I", cr,
               "-I", os.path.join(dgr, "third_party", "glm"), "-I", knn,
               "-o", out] + [os.path.join(HERE, s) for s in srcs]
        subprocess.check_call(cmd)
        if args.contract:
            out_fma = os.path.join(OUT_DIR, "liboracle_ref_fma.so")
            cmd_fma = [c for c in cmd if c != "-ffp-contract=off"]
            cmd_fma[cmd_fma.index(out)] = out_fma
            cmd_fma[4:4] = ["-ffp-contract=fast", "-mfma", "-DORACLE_KIND_NAME=\"reference_fma\""]
            subprocess.check_call(cmd_fma)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
