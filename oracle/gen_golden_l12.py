#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Golden vectors for Layers I and II from the UNMODIFIED reference
(oracle/_ref/ref_harness_l12; a few also through oracle/_ref/encode -l N to pin the harness itself): per case
tests/golden/l12_<name>.npz = {pcm, mpg (the reference's bytes), dumps (stage_dump_l12 records; `sb` kept for the first
two frames only)}, listed in tests/golden/L12_MANIFEST.json with md5s.  Runs only where /root/reference exists."""
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import L12_DT, REF_ENCODE, l12_signal, l12_spf, ref_l12  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def silence_then_noise(n, ch, seed):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, ch))
    x[n // 2:] = rng.uniform(-20000, 20000, (n - n // 2, ch))
    return x.astype(np.int16).reshape(-1)


def square(n, ch, period):
    t = np.arange(n)
    x = np.where((t // period) % 2 == 0, 32767, -32768)
    return np.repeat(x[:, None], ch, 1).astype(np.int16).reshape(-1)


def two_tones(n, ch, rate):
    t = np.arange(n)
    x = np.zeros((n, ch))
    x[:, 0] = 32000 * np.sin(2 * np.pi * 3000 * t / rate)
    x[:, ch - 1] = 32000 * np.sin(2 * np.pi * 5000 * t / rate)
    return np.rint(x).astype(np.int16).reshape(-1)


def noise(n, ch, seed):
    return (6000 * np.random.default_rng(seed).standard_normal((n, ch))).clip(-32768, 32767).astype(np.int16).reshape(-1)


def tones(n, ch, rate):
    t = np.arange(n)
    x = np.zeros((n, ch))
    for c in range(ch):
        for k, f in enumerate((440.0, 1760.0 + 300 * c, 7040.0, 12000.0)):
            x[:, c] += 6000 / (k + 1) * np.sin(2 * np.pi * f * t / rate + c)
    return x.astype(np.int16).reshape(-1)


CASES = [  # name, layer, rate, kbps, mode, frames, signal
    ("l2_s44_192", 2, 44100, 192, "s", 8, "mix"),
    ("l2_j44_064", 2, 44100, 64, "j", 8, "mix"),
    ("l2_j48_096_crc", 2, 48000, 96, "jeco", 6, "mix"),
    ("l2_m32_048", 2, 32000, 48, "m", 8, "mix"),
    ("l2_m44_032", 2, 44100, 32, "m", 8, "tones"),
    ("l2_d48_384", 2, 48000, 384, "d", 5, "mix"),
    ("l2_s32_320_crc", 2, 32000, 320, "se", 5, "mix"),
    ("l2_s44_128_silence_noise", 2, 44100, 128, "s", 8, "silence"),
    ("l2_s44_256_square", 2, 44100, 256, "s", 5, "square"),
    ("l1_s44_128", 1, 44100, 128, "s", 20, "mix"),
    ("l1_j32_096_crc", 1, 32000, 96, "je", 20, "mix"),
    ("l1_m48_448", 1, 48000, 448, "m", 16, "mix"),
    ("l1_j44_192_tones", 1, 44100, 192, "j", 20, "tones"),
    ("l1_s44_256_silence_noise", 1, 44100, 256, "s", 24, "silence"),
    # coverage-guided (tools/ref_coverage_l12.py): joint stereo that does engage, with the CRC over the shared allocation
    # (src/common.c:1301); bit estimates that run out of steps (src/encode.c:837-843, 950); unallocated subbands under -e
    ("l2_j44_032_square_crc", 2, 44100, 32, "je", 6, "square30"),
    ("l2_j44_064_two_tones", 2, 44100, 64, "j", 6, "two_tones"),
    ("l2_s44_112_noise_crc", 2, 44100, 112, "se", 5, "noise"),
    ("l1_j44_032_square_crc", 1, 44100, 32, "je", 18, "square30"),
    # the reference's frames outgrow their slots: two-channel Layer I at 32 kbps, 44.1 / 48 kHz -- the header and the
    # allocation fields alone exceed the frame's 256 bits (found by tools/matrix_parity_l12.py on the device)
    ("l1_s44_032_fields_exceed_frame", 1, 44100, 32, "s", 12, "mix"),
    ("l1_d48_032_crc_fields_exceed_frame", 1, 48000, 32, "de", 12, "tones"),
]


def main():
    manifest = {}
    with tempfile.TemporaryDirectory() as wd:
        for i, (name, layer, rate, kbps, mode, frames, kind) in enumerate(CASES):
            ch = 1 if mode[0] == "m" else 2
            n = l12_spf(layer) * frames - 101  # a ragged end: the last frame is zero-filled
            pcm = {"mix": lambda: l12_signal(n, ch, 7000 + i, rate), "silence": lambda: silence_then_noise(n, ch, i),
                   "square": lambda: square(n, ch, 37), "tones": lambda: tones(n, ch, rate), "square30": lambda: square(n, ch, 30),
                   "two_tones": lambda: two_tones(n, ch, rate), "noise": lambda: noise(n, ch, 4)}[kind]()
            mpg, dumps = ref_l12(layer, rate, kbps, mode, pcm, wd)
            assert len(dumps) == frames
            if i % 4 == 0:  # the reference's own main(): raw PCM in, same bytes out
                raw = os.path.join(wd, "raw.wav")  # (a "WAVE" tag at offset 8 makes the driver skip 44 bytes and not swap: src/musicin.c:357-362)
                data = np.ascontiguousarray(pcm, "<i2").tobytes()
                with open(raw, "wb") as f:
                    f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) +
                            b"data" + struct.pack("<I", len(data)) + data)
                args = [REF_ENCODE, "-l", str(layer), "-b", str(kbps), "-s", "%g" % (rate / 1000.0), "-m", mode[0]]
                args += ["-" + o for o in mode[1:]]
                r = subprocess.run(args + [raw, os.path.join(wd, "cli.mpg")], capture_output=True, cwd=wd)
                assert r.returncode == 0, r.stderr[-300:]
                assert open(os.path.join(wd, "cli.mpg"), "rb").read() == mpg, name
            d = dumps.copy()
            d["sb"][2:] = 0
            np.savez_compressed(os.path.join(GOLD, "l12_%s.npz" % name), pcm=pcm, mpg=np.frombuffer(mpg, np.uint8), dumps=d)
            manifest[name] = {"layer": layer, "rate": rate, "kbps": kbps, "mode": mode, "frames": frames, "signal": kind,
                              "mpg_md5": hashlib.md5(mpg).hexdigest(), "mpg_len": len(mpg)}
            print(name, len(mpg), manifest[name]["mpg_md5"])
    json.dump(manifest, open(os.path.join(GOLD, "L12_MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
