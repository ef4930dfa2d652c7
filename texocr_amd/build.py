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
HEADERS = ["common.h", "prefill.h", "conv.h", "gemm_big.h", "gemm_pp.h", "gemm_split.h", "rows.h", "enc_attn.h", "dec_gemm.h", "dec_attn.h", "lat_attn.h", "step.h", "persist.h"]


# Mandatory flags (never replaced by the environment):
# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs (see below); -ffp-contract=on: the bit-identity of the two decode
# paths depends on it (see below).  TXO_HIPCC_FLAGS only ADDS flags (e.g. -DTXO_XS_VE=4 for an experiment build).
# -fno-honor-nans (r05): without it hipcc canonicalises (v_max_f32 x, x) every fmaxf operand that comes out of an MFMA, a cross-lane move or
# a load -- about one instruction in eight of a softmax / arg-max / LayerNorm chain.  No kernel produces or tests a NaN (masks are
# large finite fills or -inf, which the flag leaves alone); tokens and logits are bit-identical with and without it on every test,
# the persistent decode launch is 1.6-2.0 % shorter (profiles/r05_no_honor_nans_ab.txt).
BASE_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-result",
              "-mllvm", "-amdgpu-mfma-vgpr-form", "-ffp-contract=on", "-fno-honor-nans"]


def _flags() -> list:
    return BASE_FLAGS + os.environ.get("TXO_HIPCC_FLAGS", "").split()


def _stamp(out: str) -> str:
    return out + ".flags"


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    # a library built with other flags is stale whatever its age (the flags of the last build are kept beside it; a library
    # that travelled to the GPU box keeps its stamp, so it is not rebuilt there)
    try:
        with open(_stamp(out)) as f:
            if f.read().split() != _flags():
                return True
    except OSError:
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "texocr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, out: str = "") -> str:
    out = out or os.environ.get("TXO_LIB_OUT", "") or OUT
    if not force and not _stale(out):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs.  Every kernel here fits 256 registers, and the attention
    # kernels read their score tiles with the VALU right after the MFMA: with accumulators in AGPRs the compiler copied ~250
    # registers per 64-key stage between the two files (v_accvgpr_read / _write: 60 % of the encoder attention's VALU work)
    # -ffp-contract=on: a*b+c fuses only INSIDE one source expression (hipcc's default, "fast", lets the backend fuse across
    # statements, and it decides per surrounding code: the same inlined tile function then rounds differently in the
    # per-stage kernels and in the persistent decode kernel, which are required to give the same bits).
    cmd = [hipcc, *_flags(), *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print("[texocr_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(_stamp(out), "w") as f:
        f.write(" ".join(_flags()) + "\n")
    return out


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


def build_test_hooks(verbose: bool = True) -> str:
    """tests/hooks/libtxo_testhooks.so: helpers of the GPU test suite that must not live in the product library (a kernel
    that holds compute units for the persistent launch's contention test)."""
    root = os.path.dirname(HERE)
    src, out = os.path.join(root, "tests", "hooks", "hold_cus.hip"), os.path.join(root, "tests", "hooks", "libtxo_testhooks.so")
    if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", src, "-o", out]
    if verbose:
        print("[texocr_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--example" in sys.argv:            # needs gcc and the HIP headers; __graft_entry__.build() builds it too
        build_example()
    if "--test-hooks" in sys.argv:
        build_test_hooks()
