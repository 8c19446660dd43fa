#!/usr/bin/env python3
"""Layers I and II: every layer x MPEG-1 rate x mode (s m j d, and s / j with -e) x bitrate combination -- 504 cells --
on the GPU against the CPU oracle: 32 ragged streams of 8 (Layer II) / 24 (Layer I) frames per cell, every stream
compared byte for byte.  TEST INFRASTRUCTURE.   python3 tools/matrix_parity_l12.py --out gpurun_out/....json"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import L12_BITRATES, L12Run, Mp3mi, Oracle, l12_signal, l12_spf, oracle_l12  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--streams", type=int, default=32)
    a = ap.parse_args()
    mp, orc = Mp3mi(), Oracle()
    t0 = time.time()
    rows, frames_total, bad_total = [], 0, 0
    for layer in (1, 2):
        for rate in (44100, 48000, 32000):
            for mode in ("s", "m", "j", "d", "se", "je"):
                for kbps in L12_BITRATES[layer]:
                    ch = 1 if mode[0] == "m" else 2
                    spf, nfr = l12_spf(layer), (8 if layer == 2 else 24)
                    pcms = [l12_signal(spf * nfr - 53 * i, ch, (layer * 1000003 + rate + kbps * 131 + i * 7 + ord(mode[0])) & 0xffffff, rate) for i in range(a.streams)]
                    run = L12Run(mp, layer, rate, kbps, mode, pcms)
                    try:
                        got = run.encode()
                    finally:
                        run.close()
                    with ThreadPoolExecutor(max_workers=16) as ex:
                        want = list(ex.map(lambda p: oracle_l12(orc, layer, rate, kbps, mode, p)[0], pcms))
                    bad = [i for i in range(a.streams) if got[i] != want[i]]
                    frames_total += a.streams * nfr
                    bad_total += len(bad)
                    rows.append({"layer": layer, "rate": rate, "mode": mode, "kbps": kbps, "mismatching_streams": len(bad)})
                    if bad:
                        print("MISMATCH", layer, rate, mode, kbps, bad, flush=True)
    rec = {"what": __doc__.split("\n\n")[0], "combinations": len(rows), "streams_per_combination": a.streams, "frames_total": frames_total,
           "mismatching_streams_total": bad_total, "bit_exact": bad_total == 0, "seconds": round(time.time() - t0, 1), "rows": rows}
    print(json.dumps({k: v for k, v in rec.items() if k != "rows"}), flush=True)
    if a.out:
        json.dump(rec, open(a.out, "w"))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
