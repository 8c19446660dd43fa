"""The C-ABI library must load without a GPU and export every symbol include/*.h declares;
without a GPU its entry points must fail loudly, never compute on the CPU."""
import ctypes
import os
import re

import pytest

from mp3common import PRODUCT_SO, ROOT


def declared_symbols():
    syms = []
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        src = open(os.path.join(inc, fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M):
            syms.append(m.group(1))
    return sorted(set(syms))


def test_library_loads_and_exports_declared_symbols(product):
    syms = declared_symbols()
    assert "mp3mi_batch_encode" in syms and "iteration_loop" in syms, syms
    missing = [s for s in syms if not hasattr(product.lib, s)]
    assert not missing, missing


def test_no_oracle_or_cpu_fallback_linked():
    """the product must not reference the oracle"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", PRODUCT_SO], capture_output=True, text=True).stdout
    assert "mp3o_" not in out
    src_dir = os.path.join(ROOT, "mp3-enc-bsd_amd")
    for base, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith((".hip", ".cpp", ".c", ".h", ".py")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "mp3_oracle" not in text and "liboracle" not in text, os.path.join(base, f)


def test_fails_loudly_without_gpu(product):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    b = ctypes.c_void_p()
    rc = product.lib.mp3mi_batch_create(ctypes.byref(b), 4, 44100, 2, None, 128, 8)
    assert rc != 0 and not b.value


def test_argument_errors_do_not_need_a_gpu(product):
    b = ctypes.c_void_p()
    assert product.lib.mp3mi_batch_create(ctypes.byref(b), 4, 22050, 2, None, 128, 8) == -1
    assert product.lib.mp3mi_batch_create(ctypes.byref(b), 4, 44100, 3, None, 128, 8) == -1
    assert product.lib.mp3mi_batch_create(ctypes.byref(b), 0, 44100, 2, None, 128, 8) == -1
