#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- adds the direct FFT seam (oracle/fft_seam.h) to fixtures of tests/golden/.

Runs only where /root/reference exists.  oracle/_ref/ref_harness_fft is our driver around the UNMODIFIED reference
objects with the reference's own fft() (src/subs.c:38) intercepted at link time (-Wl,--wrap=fft); it writes what
fft() / enphinew() returned to L3psycho_anal -- energy, phi and the raw lines behind them, of the long and the three
short transforms of every call (src/l3psy.c:494, 527).  Stored per fixture as <name>.fft.npz: `seam`, records of dtype
FFT_SEAM_DT in call order [frame][gr][ch] for the first `frames` frames.  The files are data (expected outputs)."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from golden_util import GOLD, case_pcm, manifest  # noqa: E402
from mp3common import FFT_SEAM_DT, Mp3mi  # noqa: E402
from gen_golden import wav_bytes  # noqa: E402

# fixture -> frames kept: every block type and the attack frames (bursty); the energy floor of enphinew and r + |r'| = 0
# (faint tone after digital silence: src/subs.c:70-74); the mono / 32 kHz window placement
FIXTURES = {"s44_128_bursty": 12, "x44_128_faint_after_silence": 6, "m32_064": 4}


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    synth = Mp3mi(emu=True).synth
    for case in manifest():
        if case["name"] not in FIXTURES:
            continue
        nf = FIXTURES[case["name"]]
        pcm = case_pcm(case, synth)
        with tempfile.TemporaryDirectory() as td:
            wav, mp3, dump, fft = (os.path.join(td, x) for x in ("a.wav", "a.mp3", "a.dump", "a.fft"))
            open(wav, "wb").write(wav_bytes(pcm, case["channels"], case["rate"]))
            subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_harness_fft"), wav, mp3, str(case["rate"]), str(case["kbps"]),
                            case.get("mode", "m" if case["channels"] == 1 else "s"), dump, fft], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            data = open(mp3, "rb").read()
            seam = np.fromfile(fft, dtype=FFT_SEAM_DT)
        # the interception is a pass-through: the file is the one the manifest holds
        assert hashlib.md5(data).hexdigest() == case["mp3_md5"], case["name"]
        assert len(seam) == case["frames"] * 2 * case["channels"], (len(seam), case["frames"])
        np.savez_compressed(os.path.join(GOLD, case["name"] + ".fft.npz"), seam=seam[: nf * 2 * case["channels"]])
        print(case["name"], "frames", nf, "records", nf * 2 * case["channels"])


if __name__ == "__main__":
    main()
