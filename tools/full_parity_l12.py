#!/usr/bin/env python3
"""Parity on the POPULATION for Layers I and II: encodes a full-width batch on the GPU (the
deterministic bench PCM, mp3mi_synth_pcm_device) and compares EVERY stream byte for byte with the CPU oracle
(oracle/mp12_oracle.inc, one thread per host core) and every n-th with the unmodified reference binary
(oracle/_ref/encode -l N).  TEST INFRASTRUCTURE: the oracle is the checker here, never the thing shipped.

    python3 tools/full_parity_l12.py --layer 2 --rate 44100 --kbps 160 --mode s [--streams 4096] [--frames 383]
                                     [--stream0 0] [--ref-every 64] [--flags 0] --out profiles/....json
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import REF_ENCODE, L12Run, Mp3mi, Oracle, l12_spf, oracle_l12  # noqa: E402
from test_gpu_l12 import reference_binary_l12  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", type=int, default=2)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--kbps", type=int, default=160)
    ap.add_argument("--mode", default="s")
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--stream0", type=int, default=0)
    ap.add_argument("--ref-every", type=int, default=64)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    nf = a.frames or (1149 if a.layer == 1 else 383)
    ch = 1 if a.mode[0] == "m" else 2
    mp = Mp3mi()
    run = L12Run(mp, a.layer, a.rate, a.kbps, a.mode, n_frames=nf, synth=(a.streams, a.stream0), flags=a.flags)
    t0 = time.perf_counter()
    got = run.encode()
    t_gpu = time.perf_counter() - t0
    orc = Oracle()
    cores = max(1, min(os.cpu_count() or 1, 64))

    def check(s):
        pcm = run.pcm_of(s)
        ok = oracle_l12(orc, a.layer, a.rate, a.kbps, a.mode, pcm)[0] == got[s]
        rok = None
        if a.ref_every and s % a.ref_every == 0 and os.path.exists(REF_ENCODE):
            rok = reference_binary_l12(pcm, a.layer, a.rate, ch, a.kbps, a.mode) == got[s]
        return s, ok, rok

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(check, range(a.streams)))
    t_cpu = time.perf_counter() - t0
    run.close()
    bad = [s for s, ok, _ in res if not ok]
    rbad = [s for s, _, rok in res if rok is False]
    nref = sum(1 for _, _, rok in res if rok is not None)
    rec = {"what": "every stream of a Layer %s batch vs oracle/liboracle.so (mp12_oracle.inc), every %dth also vs oracle/_ref/encode -l %d"
                   % ("I" if a.layer == 1 else "II", a.ref_every, a.layer),
           "layer": a.layer, "rate": a.rate, "kbps": a.kbps, "mode": a.mode, "streams": a.streams, "frames": nf, "stream0": a.stream0,
           "test_flags": a.flags, "frames_total": a.streams * nf, "mismatching_streams": bad[:32], "n_mismatching": len(bad),
           "vs_reference_binary": nref, "mismatching_vs_reference": rbad[:32], "bit_exact": not bad and not rbad,
           "gpu_seconds_incl_download": round(t_gpu, 2), "cpu_seconds": round(t_cpu, 1), "host_cores": cores}
    print("layer %d %d Hz %d kbps -m %s stream0 %d: %d frames, %d mismatching, %d vs reference of %d"
          % (a.layer, a.rate, a.kbps, a.mode, a.stream0, a.streams * nf, len(bad), len(rbad), nref), flush=True)
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)
    return 1 if (bad or rbad) else 0


if __name__ == "__main__":
    sys.exit(main())
