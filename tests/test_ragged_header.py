"""ragged inputs (per-stream sample counts, the zero-filled last frame of
src/encode.c:162-166) and the header bits of the reference driver's -c / -o / -d options."""
import os
import subprocess

import numpy as np
import pytest

from mp3common import ROOT, SEED, encode_host_ex
from test_dropin import write_wav

REF_ENCODE = os.path.join(ROOT, "oracle", "_ref", "encode")


def ragged_case(mp, oracle, rate, ch, kbps):
    nf = 6
    lens = [nf * 1152, 4 * 1152 + 300, 2 * 1152, 1, 5 * 1152 - 1, 0]
    pcm = np.zeros((len(lens), nf * 1152 * ch), np.int16)
    for s, n in enumerate(lens):
        pcm[s] = mp.synth(nf * 1152, ch, rate, 700 + s, SEED)  # garbage beyond n must be ignored
    got = encode_host_ex(mp, pcm, lens, rate, ch, kbps, nf)
    for s, n in enumerate(lens):
        if n == 0:
            assert got[s] == b""
            continue
        ref, _ = oracle.encode(pcm[s, :n * ch], rate, kbps, ch)
        assert got[s] == ref, "stream %d with %d samples" % (s, n)


def header_case(mp, tmp_path, rate, ch, kbps, cflag, oflag, emph):
    """against the reference CLI itself (the oracle has no header options)"""
    nf = 4
    pcm = mp.synth(nf * 1152, ch, rate, 800, SEED)
    write_wav(tmp_path / "h.wav", pcm, ch, rate)
    args = [REF_ENCODE, "-s", "%g" % (rate / 1000.0), "-b", str(kbps)]
    if cflag:
        args.append("-c")
    if oflag:
        args.append("-o")
    if emph:
        args += ["-d", {1: "5", 3: "c"}[emph]]
    subprocess.run(args + [str(tmp_path / "h.wav"), str(tmp_path / "h.mp3")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = open(tmp_path / "h.mp3", "rb").read()
    got = encode_host_ex(mp, pcm[None, :], None, rate, ch, kbps, nf, cflag, oflag, emph)[0]
    assert got == ref


def mode_crc_case(mp, tmp_path, rate, ch, kbps, dual, crc, nf=5):
    """-m d (dual channel) and -e (error protection: the reference's zero CRC word, src/l3bitstream.c:312, 338-342)
    against the reference CLI itself"""
    from mp3common import BatchRun
    pcm = mp.synth(nf * 1152, ch, rate, 810, SEED)
    write_wav(tmp_path / "m.wav", pcm, ch, rate)
    args = [REF_ENCODE, "-s", "%g" % (rate / 1000.0), "-b", str(kbps)]
    if ch == 1:
        args += ["-m", "m"]
    elif dual:
        args += ["-m", "d"]
    if crc:
        args.append("-e")
    subprocess.run(args + [str(tmp_path / "m.wav"), str(tmp_path / "m.mp3")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = open(tmp_path / "m.mp3", "rb").read()
    run = BatchRun(mp, 1, rate, ch, kbps, nf, pcm=pcm[None, :], mode=(2 if dual else None), crc=crc)
    try:
        out, lens = run.encode()
        assert out[0, :lens[0]].tobytes() == ref
    finally:
        run.close()


def test_mode_arguments(product):
    """joint stereo is refused as the reference refuses it for Layer III; the mode must fit the channel count"""
    import ctypes
    L = product.lib
    assert L.mp3mi_batch_set_mode(None, 0) == -1 and L.mp3mi_batch_set_error_protection(None, 1) == -1


@pytest.mark.skipif(not os.path.exists(REF_ENCODE), reason="oracle/_ref/encode not built")
@pytest.mark.parametrize("ch,dual,crc", [(2, 1, 0), (2, 0, 1), (1, 0, 1)])
def test_dual_channel_and_error_protection_emulated(emu, tmp_path, ch, dual, crc):
    mode_crc_case(emu, tmp_path, 44100, ch, 128 if ch == 2 else 64, dual, crc, nf=4)
    if ch == 2 and dual:
        import ctypes
        b = ctypes.c_void_p()
        assert emu.lib.mp3mi_batch_create(ctypes.byref(b), 1, 44100, 2, None, 128, 2) == 0
        assert emu.lib.mp3mi_batch_set_mode(b, 1) == -1 and emu.lib.mp3mi_batch_set_mode(b, 3) == -1
        emu.lib.mp3mi_batch_destroy(b)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_ENCODE), reason="oracle/_ref/encode not built")
@pytest.mark.parametrize("rate,ch,kbps,dual,crc", [(44100, 2, 128, 1, 0), (48000, 2, 192, 1, 1), (32000, 1, 64, 0, 1), (44100, 2, 320, 0, 1)])
def test_dual_channel_and_error_protection_gpu(product, tmp_path, rate, ch, kbps, dual, crc):
    mode_crc_case(product, tmp_path, rate, ch, kbps, dual, crc, nf=24)


def test_ragged_batch_emulated(emu, oracle):
    ragged_case(emu, oracle, 44100, 2, 128)


@pytest.mark.skipif(not os.path.exists(REF_ENCODE), reason="oracle/_ref/encode not built")
@pytest.mark.parametrize("c,o,e", [(1, 0, 0), (0, 1, 1), (1, 1, 3)])
def test_header_bits_emulated(emu, tmp_path, c, o, e):
    header_case(emu, tmp_path, 44100, 2, 128, c, o, e)


@pytest.mark.gpu
@pytest.mark.parametrize("rate,ch,kbps", [(44100, 2, 128), (32000, 1, 64)])
def test_ragged_batch_gpu(product, oracle, rate, ch, kbps):
    ragged_case(product, oracle, rate, ch, kbps)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_ENCODE), reason="oracle/_ref/encode not built")
def test_header_bits_gpu(product, tmp_path):
    header_case(product, tmp_path, 48000, 2, 192, 1, 1, 1)
