"""The deterministic PCM generator (mp3-enc-bsd_amd/csrc/pcm_synth_core.h): the bench and parity workloads are a
pure function of (seed, stream, sample index).  The md5s below pin the host generator; the device generator
(mp3mi_synth_pcm_device) must produce the same bytes (emulated here, on the MI355X under -m gpu)."""
import hashlib

import numpy as np
import pytest

from mp3common import SEED, DevMem

PINS = [
    # rate, channels, stream, samples per channel, md5 of the interleaved int16 bytes
    (44100, 2, 0, 441000, "7589dc2be844ca87a11943f2d2a1a27d"),  # = tests/golden s44_128_sweep10s
    (44100, 2, 4095, 383 * 1152, "eec6c80c120a8727c2e26ea3b9a611bd"),
    (48000, 2, 1234, 417 * 1152, "05595389de8c5dd951f42651e497883f"),
    (32000, 1, 16383, 278 * 1152, "9b9cb44ce6ab60a18dd027f050f4a077"),
]


@pytest.mark.parametrize("rate,ch,stream,n,md5", PINS)
def test_host_generator_md5_pinned(emu, rate, ch, stream, n, md5):
    assert hashlib.md5(emu.synth(n, ch, rate, stream).tobytes()).hexdigest() == md5


def device_equals_host(mp, rate, ch, stream0, S, n):
    mem = DevMem(mp)
    try:
        d = mem.alloc(S * n * ch * 2)
        assert mp.lib.mp3mi_synth_pcm_device(d, S, n, ch, rate, stream0, SEED) == 0
        got = mem.download(d, (S, n * ch), np.int16)
    finally:
        mem.free()
    for s in range(S):
        assert np.array_equal(got[s], mp.synth(n, ch, rate, stream0 + s)), "stream %d" % (stream0 + s)


def test_device_generator_equals_host_emulated(emu):
    device_equals_host(emu, 44100, 2, 7, 2, 700)
    device_equals_host(emu, 32000, 1, 16000, 1, 300)


@pytest.mark.gpu
@pytest.mark.parametrize("rate,ch,stream0", [(44100, 2, 0), (48000, 2, 4090), (32000, 1, 16380)])
def test_device_generator_equals_host_gpu(product, rate, ch, stream0):
    device_equals_host(product, rate, ch, stream0, 6, 120000)
