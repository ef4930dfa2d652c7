"""Builds the in-tree HIP library (texocr_amd/libtexocr_hip.so) for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtexocr_hip.so")
SOURCES = ["engine.hip"]
HEADERS = ["common.h", "conv.h", "gemm_big.h", "gemm_pp.h", "rows.h", "enc_attn.h", "dec_gemm.h", "dec_attn.h", "step.h", "persist.h"]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "texocr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs.  Every kernel here fits 256 registers, and the attention
    # kernels read their score tiles with the VALU right after the MFMA: with accumulators in AGPRs the compiler copied ~250
    # registers per 64-key stage between the two files (v_accvgpr_read / _write: 60 % of the encoder attention's VALU work)
    # -ffp-contract=on: a*b+c fuses only INSIDE one source expression (hipcc's default, "fast", lets the backend fuse across
    # statements, and it decides per surrounding code: the same inlined tile function then rounds differently in the
    # per-stage kernels and in the persistent decode kernel, which are required to give the same bits).
    extra = os.environ.get("TXO_HIPCC_FLAGS", "-mllvm -amdgpu-mfma-vgpr-form -ffp-contract=on").split()
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-result", *extra,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]
    if verbose:
        print("[texocr_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


def build_example(verbose: bool = True) -> str:
    """examples/generate_tiny: the torch-free C host program on the C ABI (plain gcc; HIP runtime API for device memory)."""
    root = os.path.dirname(HERE)
    src, out = os.path.join(root, "examples", "generate_tiny.c"), os.path.join(root, "examples", "generate_tiny")
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(OUT)):
        return out
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c99", "-O2", "-I" + os.path.join(root, "include"), "-I" + os.path.join(rocm, "include"), src,
           "-L" + HERE, "-ltexocr_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../texocr_amd", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", out]
    if verbose:
        print("[texocr_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_example()
