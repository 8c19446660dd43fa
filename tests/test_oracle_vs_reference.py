"""Where the compiled reference travels with the repo (oracle/_ref/ref_harness, built here from
/root/reference/src), run it live against the oracle on fresh random configurations."""
import os
import struct
import subprocess

import numpy as np
import pytest

from mp3common import REF_HARNESS, STAGE_DT

pytestmark = pytest.mark.skipif(not os.path.exists(REF_HARNESS), reason="oracle/_ref not built (no /root/reference)")


@pytest.mark.parametrize("rate,ch,kbps,seed", [(44100, 2, 128, 1), (44100, 2, 192, 2), (48000, 2, 96, 3), (48000, 1, 128, 4),
                                              (32000, 2, 64, 5), (32000, 1, 40, 6), (44100, 2, 56, 7), (48000, 2, 256, 8)])
def test_random_signals(oracle, tmp_path, rate, ch, kbps, seed):
    rng = np.random.default_rng(seed)
    n = 1152 * 9 + 321
    t = np.arange(n) / rate
    sig = 9000 * np.sin(2 * np.pi * (200 + 3000 * rng.random()) * t) + rng.normal(0, 10 ** rng.uniform(0.5, 3.5), n)
    sig[n // 2:n // 2 + 200] += rng.choice([-15000, 15000], 200)
    pcm = np.clip(np.round(np.stack([sig * (0.5 + 0.5 * c) for c in range(ch)], axis=1)), -32768, 32767).astype(np.int16).reshape(-1)
    wav = tmp_path / "a.wav"
    data = pcm.astype("<i2").tobytes()
    wav.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)
    subprocess.run([REF_HARNESS, str(wav), str(tmp_path / "a.mp3"), str(rate), str(kbps), "m" if ch == 1 else "s", str(tmp_path / "a.dump")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = (tmp_path / "a.mp3").read_bytes()
    rd = np.fromfile(tmp_path / "a.dump", dtype=STAGE_DT)
    got, dumps = oracle.encode(pcm, rate, kbps, ch, dumps=len(rd))
    assert got == ref
    for name in STAGE_DT.names:
        if name != "frame_index":
            assert np.array_equal(dumps[name], rd[name]), name
