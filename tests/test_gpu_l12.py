"""Layers I and II on the MI355X, through the C ABI of include/mp3mi_l12.h: the golden vectors of
the unmodified reference (bytes and seams), a sample of the layer x rate x mode x bitrate matrix against the oracle,
ragged and chunked batches, the exact tiers forced, mixed bitrates, and full-width batches (4096 streams) with a sample
of streams against the oracle and the reference binary (oracle/_ref/encode -l 1|2, carried to the GPU box)."""
import os
import random
import struct
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import golden_l12
from mp3common import (L12_BITRATES, REF_ENCODE, L12Run, l12_compare_seams, l12_signal, l12_spf, oracle_l12)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(golden_l12.MANIFEST))
def test_gpu_reproduces_reference_golden_l12(product, name):
    meta, pcm, mpg, dumps = golden_l12.load(name)
    run = L12Run(product, meta["layer"], meta["rate"], meta["kbps"], meta["mode"], [pcm], seams=True)
    try:
        got = run.encode()
        seams, f0 = run.seams()
        assert got[0] == mpg
        assert not l12_compare_seams(dumps, seams[0], f0)
    finally:
        run.close()


def test_matrix_sample_against_the_oracle(product, oracle):
    """84 cells of layer x rate x mode (with -e) x bitrate, 6 streams of different lengths each: a random 60, and every
    two-channel cell at the layers' lowest bitrate (tools/matrix_parity_l12.py runs all 504: profiles/*_parity_matrix_l12_*)"""
    cells = [(layer, rate, mode, kbps) for layer in (1, 2) for rate in (44100, 48000, 32000)
             for mode in ("s", "m", "j", "d", "se", "je") for kbps in L12_BITRATES[layer]]
    lowest = [c for c in cells if c[3] == 32 and c[2] != "m"]
    random.Random(4).shuffle(cells)
    bad = []
    for layer, rate, mode, kbps in cells[:60] + lowest:
        ch = 1 if mode[0] == "m" else 2
        spf = l12_spf(layer)
        nfr = 6 if layer == 2 else 15
        pcms = [l12_signal(spf * nfr - 97 * i, ch, (hash((layer, rate, mode, kbps)) + i) & 0xffff, rate) for i in range(6)]
        run = L12Run(product, layer, rate, kbps, mode, pcms)
        try:
            got = run.encode()
        finally:
            run.close()
        with ThreadPoolExecutor(max_workers=6) as ex:
            want = list(ex.map(lambda p: oracle_l12(oracle, layer, rate, kbps, mode, p)[0], pcms))
        if got != want:
            bad.append((layer, rate, mode, kbps, [i for i in range(6) if got[i] != want[i]]))
    assert not bad, bad[:5]


@pytest.mark.parametrize("layer,rate,kbps,mode", [(2, 44100, 128, "s"), (1, 32000, 192, "j"), (2, 48000, 64, "je")])
def test_chunks_tiers_and_ragged_streams_gpu(product, oracle, layer, rate, kbps, mode):
    ch = 1 if mode[0] == "m" else 2
    spf = l12_spf(layer)
    nfr = 9 if layer == 2 else 25
    pcms = [l12_signal(spf * nfr - 173 * i, ch, 40 + i, rate) for i in range(5)] + [np.zeros(0, np.int16)]
    want = [oracle_l12(oracle, layer, rate, kbps, mode, p)[0] for p in pcms]
    for scratch, flags in ((0, 0), (1, 0), (0, 2 | 4 | 32)):
        run = L12Run(product, layer, rate, kbps, mode, pcms, n_frames=nfr, scratch_mb=scratch, flags=flags)
        try:
            assert run.encode() == want, (scratch, flags)
        finally:
            run.close()


def test_mixed_bitrates_in_one_batch_gpu(product, oracle):
    kb = [L12_BITRATES[2][i % 14] for i in range(28)]
    pcms = [l12_signal(1152 * 5, 2, 60 + i) for i in range(28)]
    run = L12Run(product, 2, 44100, kb, "j", pcms)
    try:
        got = run.encode()
    finally:
        run.close()
    bad = [kb[i] for i in range(28) if got[i] != oracle_l12(oracle, 2, 44100, kb[i], "j", pcms[i])[0]]
    assert not bad, bad


def reference_binary_l12(pcm, layer, rate, ch, kbps, mode):
    with tempfile.TemporaryDirectory() as td:
        wav, out = os.path.join(td, "a.wav"), os.path.join(td, "a.mpg")
        data = np.ascontiguousarray(pcm, dtype="<i2").tobytes()
        with open(wav, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)
        subprocess.run([REF_ENCODE, "-l", str(layer), "-s", "%g" % (rate / 1000.0), "-b", str(kbps), "-m", mode[0]] +
                       ["-" + o for o in mode[1:]] + [wav, out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
        return open(out, "rb").read()


@pytest.mark.parametrize("layer,rate,kbps,mode,frames", [(2, 44100, 192, "s", 120), (2, 44100, 128, "j", 120), (1, 44100, 384, "s", 360),
                                                         (2, 32000, 64, "m", 120)])
def test_full_width_batch_l12(product, oracle, layer, rate, kbps, mode, frames):
    """4096 streams of bench-family PCM (mp3mi_synth_pcm_device): 48 streams spread over the batch against the oracle,
    6 of them against the reference binary"""
    S = 4096
    ch = 1 if mode[0] == "m" else 2
    run = L12Run(product, layer, rate, kbps, mode, n_frames=frames, synth=(S, 0))
    try:
        got = run.encode()
        sample = sorted(set(np.linspace(0, S - 1, 48).astype(int).tolist()))
        pcm = {s: run.pcm_of(s) for s in sample}
    finally:
        run.close()
    frame_bytes = len(got[0]) - 1
    assert all(len(g) == frame_bytes + 1 for g in got) and frame_bytes % frames == 0
    with ThreadPoolExecutor(max_workers=16) as ex:
        refs = dict(zip(sample, ex.map(lambda s: oracle_l12(oracle, layer, rate, kbps, mode, pcm[s])[0], sample)))
    bad = [s for s in sample if got[s] != refs[s]]
    assert not bad, bad[:8]
    if os.path.exists(REF_ENCODE):
        for s in sample[::8]:
            assert reference_binary_l12(pcm[s], layer, rate, ch, kbps, mode) == got[s], s


@pytest.mark.parametrize("layer,rate,kbps,mode,pieces", [(2, 44100, 160, "j", [1, 30, 2, 17]), (1, 44100, 288, "s", [1, 2, 40, 7, 100]), (2, 32000, 64, "m", [25, 25])])
def test_streaming_equals_one_call_gpu(product, oracle, layer, rate, kbps, mode, pieces):
    """256 streams fed piece by piece through mp3mi_l12_batch_encode_next + flush == the whole-file call == the oracle"""
    S, nfr = 256, sum(pieces)
    run = L12Run(product, layer, rate, kbps, mode, n_frames=nfr, synth=(S, 77))
    try:
        whole = run.encode()
        assert run.encode_streaming(pieces) == whole
        sample = list(range(0, S, 16))
        pcm = {s: run.pcm_of(s) for s in sample}
    finally:
        run.close()
    with ThreadPoolExecutor(max_workers=16) as ex:
        refs = dict(zip(sample, ex.map(lambda s: oracle_l12(oracle, layer, rate, kbps, mode, pcm[s])[0], sample)))
    assert all(whole[s] == refs[s] for s in sample)
