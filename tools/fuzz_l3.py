#!/usr/bin/env python3
"""Layer III: random configurations against the oracle on the device -- rate, channels, mode (stereo / dual / mono) with
random -e / -c / -o, bitrates (also different per stream), ragged lengths from zero samples up, signals (mix, silence,
full-scale noise, a click behind silence), chunk length forced small or left alone.  Where the REFERENCE dies on an input
(oracle: ReferenceAborts) the product must void that stream and name the assertion.  TEST INFRASTRUCTURE.
    python3 tools/fuzz_l3.py [--cases 300] [--seed 1] [--out gpurun_out/....json]"""
import argparse
import ctypes
import json
import os
import random
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import DevMem, Mp3mi, Oracle, ReferenceAborts, l12_signal  # noqa: E402

BITRATES = [32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    mp, orc = Mp3mi(), Oracle()
    L = mp.lib
    L.mp3mi_batch_encode_ragged.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    bad, frames, aborts = [], 0, 0
    for case in range(a.cases):
        rate = rnd.choice((44100, 48000, 32000))
        ch = rnd.choice((1, 2))
        mode = ("m" if ch == 1 else rnd.choice("sd")) + "".join(o for o in "eco" if rnd.random() < 0.3)
        S = rnd.choice((1, 2, 3, 7, 16, 65))
        same = rnd.random() < 0.5
        kb = [rnd.choice(BITRATES) for _ in range(S)]
        if same:
            kb = [kb[0]] * S
        nfr = rnd.choice((1, 2, 3, 5, 9, 24))
        lens = [min(1152 * nfr, rnd.choice((0, 1, 2, 575, 576, 1151, 1152, 1153, rnd.randint(1, 1152 * nfr), 1152 * nfr))) for _ in range(S)]
        kind = rnd.choice(("mix", "mix", "silence", "loud", "click"))
        pcm = np.zeros((S, nfr * 1152 * ch), np.int16)
        for i, n in enumerate(lens):
            if kind == "loud":
                p = np.where(np.random.default_rng(case * 31 + i).random(n * ch) < 0.5, 32767, -32768).astype(np.int16)
            elif kind == "click":
                p = np.zeros(n * ch, np.int16)
                if n > 700:
                    p[600 * ch] = 20000
            elif kind == "silence":
                p = np.zeros(n * ch, np.int16)
            else:
                p = l12_signal(n, ch, case * 17 + i, rate)
            pcm[i, :n * ch] = p
        opt = mp.options(chunk_frames=rnd.choice((0, 0, 1, 2, 5)))
        b = ctypes.c_void_p()
        karr = np.ascontiguousarray(kb, dtype=np.int32)
        rc = L.mp3mi_batch_create_ex(ctypes.byref(b), S, rate, ch, None if same else karr.ctypes.data, kb[0] if same else 0, nfr, ctypes.byref(opt))
        assert rc == 0, rc
        assert L.mp3mi_batch_set_mode(b, {"s": 0, "d": 2, "m": 3}[mode[0]]) == 0
        assert L.mp3mi_batch_set_error_protection(b, int("e" in mode[1:])) == 0
        assert L.mp3mi_batch_set_header(b, int("c" in mode[1:]), int("o" in mode[1:]), 0) == 0
        mem = DevMem(mp)
        stride = L.mp3mi_batch_out_stride(b, nfr)
        d_pcm, d_out, d_len, d_ns = mem.alloc(pcm.nbytes), mem.alloc(S * stride), mem.alloc(4 * S), mem.alloc(4 * S)
        mem.upload(d_pcm, pcm)
        mem.upload(d_ns, np.ascontiguousarray(lens, dtype=np.int32))
        rc = L.mp3mi_batch_encode_ragged(b, d_pcm, d_ns, nfr, d_out, stride, d_len)
        assert rc == 0, rc
        src = L.mp3mi_batch_sync(b)
        out = mem.download(d_out, (S, stride), np.uint8)
        ol = mem.download(d_len, (S,), np.uint32)
        st = np.zeros(S, np.int32)
        L.mp3mi_batch_stream_status(b, st.ctypes.data)
        L.mp3mi_batch_destroy(b)
        mem.free()

        def ref(i):
            try:
                return orc.encode(pcm[i, :lens[i] * ch], rate, kb[i], ch, mode=mode)[0], 0
            except ReferenceAborts as e:
                return None, e.status
        with ThreadPoolExecutor(max_workers=16) as ex:
            want = list(ex.map(ref, range(S)))
        wrong = []
        for i in range(S):
            if lens[i] == 0:  # no samples: the reference dies in its flush (nothing was ever queued); the product's row is empty (include/mp3mi.h)
                if ol[i] != 0:
                    wrong.append(i)
            elif want[i][0] is None:
                aborts += 1
                if ol[i] != 0 or (st[i] & 255) != want[i][1]:
                    wrong.append(i)
            elif out[i, :ol[i]].tobytes() != want[i][0]:
                wrong.append(i)
        if (src == -6) != any(w[0] is None for w in want) or (src not in (0, -6)):
            wrong.append(-1)
        frames += sum((n + 1151) // 1152 for n in lens)
        if wrong:
            bad.append({"case": case, "rate": rate, "mode": mode, "kbps": kb, "frames": nfr, "lens": lens, "signal": kind,
                        "chunk_frames": int(opt.chunk_frames), "streams": wrong})
            print("MISMATCH", bad[-1], flush=True)
    rec = {"what": __doc__.split("\n\n")[0], "cases": a.cases, "seed": a.seed, "frames_total": frames, "streams_the_reference_dies_on": aborts,
           "mismatching_cases": bad, "bit_exact": not bad}
    print(json.dumps({k: v for k, v in rec.items() if k != "what"}), flush=True)
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
