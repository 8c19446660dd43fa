"""Layers I and II: the CPU restatement (oracle/mp12_oracle.inc) against the golden vectors the
unmodified reference produced, and -- where oracle/_ref is present -- against the reference itself over a sample of the
layer x rate x mode x bitrate matrix (tools/l12_oracle_vs_ref.py runs all 504 cells)."""
import hashlib
import os
import random

import numpy as np
import pytest

import golden_l12
from mp3common import L12_BITRATES, L12_DT, REF_HARNESS_L12, Oracle, l12_signal, l12_spf, oracle_l12, ref_l12


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.mark.parametrize("name", sorted(golden_l12.MANIFEST))
def test_oracle_reproduces_the_golden_vectors(orc, name):
    meta, pcm, mpg, dumps = golden_l12.load(name)
    assert hashlib.md5(mpg).hexdigest() == meta["mpg_md5"] and len(mpg) == meta["mpg_len"]
    got, d = oracle_l12(orc, meta["layer"], meta["rate"], meta["kbps"], meta["mode"], pcm, dumps=meta["frames"])
    assert got == mpg
    assert not golden_l12.seams_equal(dumps, d, with_sb_frames=2)


@pytest.mark.skipif(not os.path.exists(REF_HARNESS_L12), reason="reference build absent (oracle/_ref)")
def test_oracle_vs_reference_sample_of_the_matrix(orc, tmp_path):
    cells = [(layer, rate, mode, kbps) for layer in (1, 2) for rate in (44100, 48000, 32000)
             for mode in ("s", "m", "j", "d", "se", "je") for kbps in L12_BITRATES[layer]]
    random.Random(20261004).shuffle(cells)
    for layer, rate, mode, kbps in cells[:36]:
        ch = 1 if mode[0] == "m" else 2
        pcm = l12_signal(l12_spf(layer) * 7 + 211, ch, hash((layer, rate, mode, kbps)) & 0xffff, rate)
        rb, rd = ref_l12(layer, rate, kbps, mode, pcm, str(tmp_path))
        ob, od = oracle_l12(orc, layer, rate, kbps, mode, pcm, dumps=len(rd))
        assert rb == ob, (layer, rate, mode, kbps)
        for name in L12_DT.names:
            assert np.array_equal(rd[name], od[name]), (layer, rate, mode, kbps, name)


def test_oracle_refuses_what_the_reference_refuses(orc):
    pcm = np.zeros(4000, np.int16)
    for args in ((2, 22050, 64, "s"), (2, 44100, 40, "s"), (1, 44100, 48, "s"), (3, 44100, 128, "s")):
        with pytest.raises(ValueError):
            oracle_l12(orc, *args, pcm)
