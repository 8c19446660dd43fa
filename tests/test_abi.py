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
    """what the reference refuses the product refuses (MP3MI_ERR_ARG): the MPEG-2 LSF rates L3psycho_anal exits on
    (src/l3psy.c:170-176), bitrates outside the Layer III table (src/common.c:118-125, 460-481), channel counts"""
    b = ctypes.c_void_p()
    for rate in (22050, 24000, 16000, 8000, 96000):
        assert product.lib.mp3mi_batch_create(ctypes.byref(b), 4, rate, 2, None, 128, 8) == -1
    for kbps in (0, 16, 100, 384):
        assert product.lib.mp3mi_batch_create(ctypes.byref(b), 4, 44100, 2, None, kbps, 8) == -1
    assert product.lib.mp3mi_batch_create(ctypes.byref(b), 4, 44100, 3, None, 128, 8) == -1
    assert product.lib.mp3mi_batch_create(ctypes.byref(b), 0, 44100, 2, None, 128, 8) == -1


def test_options_struct(product):
    """mp3mi_batch_options: defaults, the environment overlay (the one place the library reads its environment), and
    a struct of another size refused"""
    import subprocess
    import sys
    o = product.options()
    assert o.struct_size == ctypes.sizeof(o) and o.scratch_mb == 0 and o.chunk_frames == 0 and o.test_flags == 0
    assert (o.call_overlap, o.gate, o.placement, o.y_after_loop, o.psy_beside, o.loop_part_streams, o.call_hold) == (-1, -1, -1, -1, -1, 0, -1)
    code = ("import sys, ctypes; sys.path.insert(0, %r); from mp3common import Mp3mi, BatchOptions; m = Mp3mi(); o = BatchOptions(); "
            "m.lib.mp3mi_batch_options_from_env(ctypes.byref(o)); "
            "print(o.chunk_frames, o.scratch_mb, o.test_flags, o.gate, o.placement, o.loop_part_streams, o.psy_beside, o.call_hold)" % os.path.join(ROOT, "tests"))
    env = dict(os.environ, MP3MI_CHUNK_FRAMES="7", MP3MI_SCRATCH_MB="100", MP3MI_PSY_EXACT="1", MP3MI_CW_EXACT="1", MP3MI_NO_GATE="1",
               MP3MI_NO_PLACE="1", MP3MI_CALL_HOLD="0", MP3MI_LOOP_PART_STREAMS="200", MP3MI_PSY_BESIDE="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [7, 100, 4 | 32, 0, 0, 192, 2, 0]
    b = ctypes.c_void_p()
    o.struct_size = 8
    assert product.lib.mp3mi_batch_create_ex(ctypes.byref(b), 4, 44100, 2, None, 128, 8, ctypes.byref(o)) == -1
    # ... and so is a struct of the right size whose layout is another header's (the size alone does not tell them apart)
    o = product.options()
    assert o.abi == 6
    o.abi = 5
    assert product.lib.mp3mi_batch_create_ex(ctypes.byref(b), 4, 44100, 2, None, 128, 8, ctypes.byref(o)) == -1


def test_product_reads_its_environment_in_one_place_only():
    """getenv appears in mp3mi_batch_options_from_env and nowhere else in the product sources"""
    src_dir = os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(src_dir)):
        if f.endswith((".hip", ".cpp", ".h")):
            text = open(os.path.join(src_dir, f), errors="replace").read()
            for m in re.finditer(r"getenv\s*\(", text):
                fn = re.findall(r"^[\w\" ].*?\b(\w+)\s*\([^;{}]*\)\s*\{", text[:m.start()], flags=re.M)
                hits.append((f, fn[-1] if fn else "?"))
    assert hits and all(h == ("batch.cpp", "mp3mi_batch_options_from_env") for h in hits), sorted(set(hits))
