"""Init-time tables (windows, twiddles, spreading function, MDCT cosines, power tables) come out of the host's
libm; one changed bit changes the bitstream.  Every member of mp3mi_tables is pinned per sampling rate by an
FNV-1a hash generated in the environment the golden vectors come from (tools/gen_table_pins.py ->
csrc/tables_pins.h, tests/golden/table_pins.json); mp3mi_build_tables refuses tables that differ."""
import ctypes
import json
import os

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
