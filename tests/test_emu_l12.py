"""Layers I and II: the kernels' LOGIC on the CPU wave emulator (tests/hipemu; never a product path) against the golden
vectors of the unmodified reference and against the oracle -- bytes and seams, chunked == unchunked, exact tiers forced."""
import numpy as np
import pytest

import golden_l12
from mp3common import L12Run, Mp3mi, Oracle, l12_compare_seams, l12_signal, l12_spf, oracle_l12


@pytest.fixture(scope="module")
def emu():
    return Mp3mi(emu=True)


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.mark.parametrize("name", ["l2_j44_064", "l2_j48_096_crc", "l2_m44_032", "l2_s44_128_silence_noise", "l1_j32_096_crc",
                                  "l1_s44_256_silence_noise", "l1_m48_448", "l2_j44_032_square_crc", "l1_j44_032_square_crc", "l1_s44_032_fields_exceed_frame",
                                  "l1_d48_032_crc_fields_exceed_frame"])
def test_emulated_kernels_reproduce_golden_vectors(emu, name):
    meta, pcm, mpg, dumps = golden_l12.load(name)
    run = L12Run(emu, meta["layer"], meta["rate"], meta["kbps"], meta["mode"], [pcm], seams=True)
    try:
        got = run.encode()
        seams, f0 = run.seams()
        assert got[0] == mpg
        assert not l12_compare_seams(dumps, seams[0], f0)
    finally:
        run.close()


@pytest.mark.parametrize("layer,rate,kbps,mode", [(2, 44100, 128, "s"), (1, 32000, 192, "j"), (2, 48000, 56, "m")])
def test_chunks_tiers_and_ragged_streams(emu, orc, layer, rate, kbps, mode):
    """three streams of different lengths; one chunk, several chunks (1 MiB of scratch), every exact tier forced"""
    ch = 1 if mode[0] == "m" else 2
    spf = l12_spf(layer)
    nfr = 7 if layer == 2 else 17
    pcms = [l12_signal(spf * nfr - 173 * i, ch, 40 + i, rate) for i in range(3)]
    want = [oracle_l12(orc, layer, rate, kbps, mode, p)[0] for p in pcms]
    for scratch, flags in ((0, 0), (1, 0), (0, 2 | 4 | 32)):
        run = L12Run(emu, layer, rate, kbps, mode, pcms, scratch_mb=scratch, flags=flags)
        try:
            assert run.encode() == want, (scratch, flags)
        finally:
            run.close()


def test_mixed_bitrates_in_one_batch(emu, orc):
    """Layer II streams of one batch with different bitrates take different allocation tables (src/common.c:291-318)"""
    kb = [32, 96, 192, 384]
    pcms = [l12_signal(1152 * 4, 2, 60 + i) for i in range(4)]
    run = L12Run(emu, 2, 44100, kb, "s", pcms)
    try:
        got = run.encode()
    finally:
        run.close()
    for i in range(4):
        assert got[i] == oracle_l12(orc, 2, 44100, kb[i], "s", pcms[i])[0], kb[i]


@pytest.mark.parametrize("layer,rate,kbps,mode,pieces", [(2, 44100, 128, "j", [1, 3, 2]), (1, 48000, 192, "s", [1, 1, 5, 9, 2]), (2, 32000, 64, "m", [2, 1, 1, 2])])
def test_streaming_equals_one_call(emu, orc, layer, rate, kbps, mode, pieces):
    """the stream fed piece by piece (Layer I pieces shorter than the history a call needs) == the whole-file call == the
    oracle; then the same batch starts new streams"""
    ch = 1 if mode[0] == "m" else 2
    nfr = sum(pieces)
    pcms = [l12_signal(l12_spf(layer) * nfr, ch, 70 + i, rate) for i in range(2)]
    want = [oracle_l12(orc, layer, rate, kbps, mode, p)[0] for p in pcms]
    run = L12Run(emu, layer, rate, kbps, mode, pcms, scratch_mb=1)
    try:
        assert run.encode_streaming(pieces) == want
        assert run.encode() == want
        assert run.encode_streaming(pieces[::-1]) == want
    finally:
        run.close()
