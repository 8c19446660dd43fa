#!/usr/bin/env python3
"""Layers I and II: random configurations against the oracle on the device -- layer, rate, mode with random -e / -c / -o,
bitrate (also different per stream), ragged lengths from one sample up, one to many chunks.  TEST INFRASTRUCTURE.
    python3 tools/fuzz_l12.py [--cases 200] [--seed 1] [--out gpurun_out/....json]"""
import argparse
import json
import os
import random
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import L12_BITRATES, L12Run, Mp3mi, Oracle, l12_signal, l12_spf, oracle_l12  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--streaming", action="store_true", help="half of the cases through mp3mi_l12_batch_encode_next in random pieces")
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    mp, orc = Mp3mi(), Oracle()
    bad, frames = [], 0
    for case in range(a.cases):
        layer = rnd.choice((1, 2))
        rate = rnd.choice((44100, 48000, 32000))
        mode = rnd.choice("smjd") + "".join(o for o in "eco" if rnd.random() < 0.3)
        ch = 1 if mode[0] == "m" else 2
        S = rnd.choice((1, 2, 3, 7, 16))
        same = rnd.random() < 0.5
        kb = [rnd.choice(L12_BITRATES[layer]) for _ in range(S)]
        if same:
            kb = [kb[0]] * S
        spf = l12_spf(layer)
        nfr = rnd.choice((1, 2, 3, 5, 9, 20 if layer == 2 else 60))
        lens = [rnd.choice((1, 2, 31, spf - 1, spf, spf + 1, rnd.randint(1, spf * nfr), spf * nfr)) for _ in range(S)]
        lens = [min(n, spf * nfr) for n in lens]
        kind = rnd.choice(("mix", "silence", "loud"))
        pcms = []
        for i, n in enumerate(lens):
            if kind == "silence":
                p = np.zeros(n * ch, np.int16)
            elif kind == "loud":
                p = np.where(np.random.default_rng(case * 31 + i).random(n * ch) < 0.5, 32767, -32768).astype(np.int16)
            else:
                p = l12_signal(n, ch, case * 17 + i, rate)
            pcms.append(p)
        streaming = a.streaming and rnd.random() < 0.5
        if streaming:  # whole-length streams fed in random pieces (ragged batches and streaming do not combine)
            lens = [spf * nfr] * S
            pcms = [l12_signal(spf * nfr, ch, case * 17 + i, rate) for i in range(S)]
            pieces, left = [], nfr
            while left:
                p = rnd.randint(1, left)
                pieces.append(p)
                left -= p
        run = L12Run(mp, layer, rate, kb[0] if same else kb, mode, pcms, n_frames=nfr, scratch_mb=rnd.choice((0, 1)))
        try:
            got = run.encode_streaming(pieces) if streaming else run.encode()
        finally:
            run.close()
        with ThreadPoolExecutor(max_workers=16) as ex:
            want = list(ex.map(lambda t: oracle_l12(orc, layer, rate, t[1], mode, t[0])[0], zip(pcms, kb)))
        frames += sum((n + spf - 1) // spf for n in lens)
        if got != want:
            bad.append({"case": case, "layer": layer, "rate": rate, "mode": mode, "kbps": kb, "frames": nfr, "lens": lens, "signal": kind,
                        "streams": [i for i in range(S) if got[i] != want[i]]})
            print("MISMATCH", bad[-1], flush=True)
    rec = {"what": __doc__.split("\n\n")[0], "cases": a.cases, "seed": a.seed, "frames_total": frames, "mismatching_cases": bad, "bit_exact": not bad}
    print(json.dumps({k: v for k, v in rec.items() if k != "what"}), flush=True)
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
