"""No-GPU checks of the C-ABI boundary: the library loads, exports every symbol include/texocr.h declares,
argument validation answers without touching the GPU, and the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from texocr_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from texocr_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "texocr.h")).read()
    declared = set(re.findall(r"\b(txo_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.txo_version().decode().startswith("texocr-amd")


def _cfg(**over):
    from texocr_amd import _lib
    base = dict(canvas_h=224, canvas_w=224, embed=0, in_channels=3, embed_dim=256, enc_heads=8, enc_layers=4, dec_heads=8, dec_layers=4,
                enc_exp=4, dec_exp=4, vocab=1000, max_len=256, bos=998, eos=997, pad=999, dtype=0, max_batch=4,
                max_tokens=0)
    base.update(over)
    return _lib.TxoConfig(**base)


@pytest.mark.parametrize("over,frag", [
    (dict(canvas_h=100), "canvas"), (dict(embed=1, in_channels=3), "hybrid"), (dict(embed_dim=100), "embed_dim"), (dict(embed_dim=1024), "embed_dim"),
    (dict(dtype=7), "dtype"), (dict(max_batch=0), "max_batch"), (dict(bos=5000), "bos"),
    (dict(max_tokens=100000), "max_tokens"), (dict(enc_layers=0), "layers"),
])
def test_create_rejects_bad_config_without_gpu(lib, over, frag):
    from texocr_amd import _lib
    h = C.c_void_p()
    cfg = _cfg(**over)
    rc = lib.txo_engine_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.TXO_E_INVALID
    assert frag in lib.txo_last_error().decode()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_null_arguments(lib):
    from texocr_amd import _lib
    assert lib.txo_engine_create(None, None) == _lib.TXO_E_INVALID
    assert lib.txo_encode(None, None, 1, 3, 16, 16, None, None) == _lib.TXO_E_INVALID
    assert lib.txo_decode_step(None, None, 0, None, None, None) == _lib.TXO_E_INVALID
    assert lib.txo_profile_enable(None, 1) == _lib.TXO_E_INVALID


def test_no_cpu_fallback():
    """Without a GPU the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from texocr_amd.config import Dims
    from texocr_amd.model import HipEngine, model_from_dims
    with pytest.raises(RuntimeError, match="no GPU|HIP"):
        HipEngine(Dims(canvas=64, embed_dim=64, enc_heads=1, enc_layers=1, dec_heads=1, dec_layers=1, vocab=16,
                       max_len=8, bos=1, eos=2, pad=3))
    with pytest.raises(RuntimeError):
        model_from_dims(Dims(canvas=224))


def test_product_does_not_import_oracle():
    import ast
    pkg = os.path.join(ROOT, "texocr_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n.split(".")[0] == "oracle" for n in names), fn


def test_header_is_plain_c_and_a_c_program_links(lib, tmp_path):
    """The boundary is a C ABI: the header compiles as C99 and a C program links against the library and gets the documented
    status code + message for an invalid configuration (no GPU needed: validation precedes any HIP call)."""
    import shutil, subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(root, "include", "texocr.h")], check=True)
    exe = str(tmp_path / "abi_demo")
    libdir = os.path.join(root, "texocr_amd")
    subprocess.run([gcc, "-std=c99", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "abi_demo.c"),
                    "-L" + libdir, "-ltexocr_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "texocr-amd" in out and "-> -1" in out and "embed_dim" in out
