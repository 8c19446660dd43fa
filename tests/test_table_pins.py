"""Init-time tables (windows, twiddles, spreading function, MDCT cosines, power tables) come out of glibc's libm in
the reference; one changed bit changes the bitstream.  The library does not compute them on the machine that encodes:
the generator build of csrc/tables_host.cpp (make -C csrc blob) computed them in the environment the golden vectors
come from and csrc/tables_blob.bin ships them.  Every member of mp3mi_tables is pinned per sampling rate by an FNV-1a
hash (tools/gen_table_pins.py -> csrc/tables_pins.h, tests/golden/table_pins.json); mp3mi_build_tables refuses tables
that differ, and no switch turns that off."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import pytest

from mp3common import ROOT


def digest(lib):
    lib.mp3mi_tables_digest.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    out = {}
    for ri, name in enumerate(("44100", "48000", "32000")):
        h = (ctypes.c_uint64 * 256)()
        names = (ctypes.c_char_p * 256)()
        n = lib.mp3mi_tables_digest(ri, h, names, 256)
        assert n > 0
        out[name] = {names[i].decode(): "%016x" % h[i] for i in range(n)}
    return out


def check(lib):
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "table_pins.json")))["rates"]
    got = digest(lib)
    for rate in pins:
        drift = [m for m in pins[rate] if got[rate].get(m) != pins[rate][m]]
        assert not drift and len(got[rate]) == len(pins[rate]), "rate %s: members differ from their pins: %s" % (rate, drift)


def test_tables_of_this_host_match_the_pins(product):
    check(product.lib)


def test_header_and_fixture_hold_the_same_pins():
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "table_pins.json")))["rates"]
    hdr = open(os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc", "tables_pins.h")).read()
    for rate in pins:
        for member, h in pins[rate].items():
            assert "0x%sull, /* %s */" % (h, member) in hdr


@pytest.mark.gpu
def test_tables_of_the_gpu_box_match_the_pins(product):
    """the same check on the machine that actually encodes (batch_create repeats it on every call)"""
    check(product.lib)


def test_generator_reproduces_the_committed_blob(tmp_path):
    """where this host IS the reference environment (the generator checks the pins and refuses otherwise), the blob it
    writes is the committed one, byte for byte"""
    csrc = os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "build/gen_table_blob"], check=True)
    out = str(tmp_path / "blob.bin")
    r = subprocess.run([os.path.join(csrc, "build", "gen_table_blob"), out], capture_output=True, text=True)
    if r.returncode != 0:
        assert "differs from its pin" in r.stderr, r.stderr
        pytest.skip("this host's libm is not the reference environment's: the generator refuses, as it must")
    assert open(out, "rb").read() == open(os.path.join(csrc, "tables_blob.bin"), "rb").read()


def test_tables_do_not_depend_on_this_hosts_libm(tmp_path):
    """The same encode under an LD_PRELOADed libm whose sin / cos / exp / log / pow / atan2 are off by one ulp
    (tests/libm_perturb.c): the CPU test build of the kernels (the only way to run them here) must still hash its
    tables to the pins and emit the golden's bytes.  The shim is proven to bite: the generator, which DOES call libm,
    refuses to write a blob under it."""
    shim = str(tmp_path / "libm_perturb.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-o", shim, os.path.join(ROOT, "tests", "libm_perturb.c"), "-ldl", "-lm"], check=True)
    env = dict(os.environ, LD_PRELOAD=shim)
    csrc = os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "build/gen_table_blob"], check=True)
    r = subprocess.run([os.path.join(csrc, "build", "gen_table_blob"), str(tmp_path / "x.bin")], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "differs from its pin" in r.stderr, "the shim does not perturb libm here"
    code = """
import sys, hashlib, json
sys.path.insert(0, %r)
from mp3common import Mp3mi, pad_frames
from golden_util import manifest, case_pcm
import test_table_pins
emu = Mp3mi(emu=True)
test_table_pins.check(emu.lib)
case = [c for c in manifest() if c["name"] == "x44_128_crc_dual"][0]
pcm, nf = pad_frames(case_pcm(case, emu.synth), 2)
from stage_check import run_batch_with_stages
got, _ = run_batch_with_stages(emu, pcm[None, :], case["rate"], 2, case["kbps"], nf, mode=case["mode"])
assert hashlib.md5(got[0]).hexdigest() == case["mp3_md5"], "bytes changed under the perturbed libm"
print("ok")
""" % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
