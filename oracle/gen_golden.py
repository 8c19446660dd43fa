#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/* from the UNMODIFIED reference.

Runs only where /root/reference exists: builds oracle/_ref (reference objects + our harness),
synthesises the inputs with the product's deterministic PCM generator, encodes them with the
reference, and commits: the PCM (int16 .npy), the reference's MP3 md5/length, and the
reference's per-stage dumps (oracle/stage_dump.h records) of the first frames.  These files are
data (inputs + expected outputs); no reference source is copied.
"""
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
from mp3common import STAGE_DT, SEED  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [
    # name, rate, channels, kbps, stream, seconds, dumped frames
    ("s44_128_sweep10s", 44100, 2, 128, 0, 10.0, 4),      # BASELINE configs[0]: 383 frames
    ("s44_128_bursty", 44100, 2, 128, 5, 0.60, 23),        # every block type
    ("s48_064", 48000, 2, 64, 2, 0.36, 8),
    ("s48_320", 48000, 2, 320, 3, 0.36, 8),
    ("m32_064", 32000, 1, 64, 8, 0.55, 8),
    ("s44_032_starved", 44100, 2, 32, 14, 0.30, 6),
]


def wav_bytes(pcm, ch, rate):
    data = pcm.astype("<i2").tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
            struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    import ctypes
    synth_so = os.path.join(tempfile.mkdtemp(), "libsynth.so")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc"), "-o", synth_so,
                    os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc", "pcm_synth_host.cpp")], check=True)
    synth = ctypes.CDLL(synth_so)
    os.makedirs(GOLD, exist_ok=True)
    manifest = []
    for name, rate, ch, kbps, stream, secs, ndump in CASES:
        n = int(rate * secs)
        pcm = np.zeros(n * ch, np.int16)
        synth.mp3mi_synth_pcm(ctypes.c_void_p(pcm.ctypes.data), ctypes.c_long(n), ch, rate, ctypes.c_uint32(stream), ctypes.c_uint32(SEED))
        with tempfile.TemporaryDirectory() as td:
            wav, mp3, dump = (os.path.join(td, x) for x in ("a.wav", "a.mp3", "a.dump"))
            open(wav, "wb").write(wav_bytes(pcm, ch, rate))
            subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_harness"), wav, mp3, str(rate), str(kbps),
                            "m" if ch == 1 else "s", dump], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            data = open(mp3, "rb").read()
            d = np.fromfile(dump, dtype=STAGE_DT)
        entry = {"name": name, "rate": rate, "channels": ch, "kbps": kbps, "stream": stream, "n_samples_per_ch": n,
                 "frames": len(d), "pcm_md5": hashlib.md5(pcm.tobytes()).hexdigest(), "mp3_md5": hashlib.md5(data).hexdigest(),
                 "mp3_len": len(data), "dump_frames": ndump, "source": "reference via oracle/_ref/ref_harness"}
        if n * ch <= 60000:
            np.save(os.path.join(GOLD, name + ".pcm.npy"), pcm)
            open(os.path.join(GOLD, name + ".mp3"), "wb").write(data)
            entry["pcm_file"] = name + ".pcm.npy"
            entry["mp3_file"] = name + ".mp3"
        np.savez_compressed(os.path.join(GOLD, name + ".stages.npz"), dumps=d[:ndump])
        manifest.append(entry)
        print(entry)
    json.dump(manifest, open(os.path.join(GOLD, "MANIFEST.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
