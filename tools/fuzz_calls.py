#!/usr/bin/env python3
"""Random CALL CHAINS on one batch, on the device, against the oracle: whole-file calls, streaming calls piece by piece with a
flush, host-buffer calls -- issued back to back without a sync, or with a sync / a status query / a timing query thrown in
at random -- on batches of 1 .. 5000 streams (two parts per chunk above 4096) with chunk lengths forced small or left alone,
with the hold of a call's last k_loop (options.call_hold, csrc/batch.cpp) on and off.  What is checked: every call's bytes are
the oracle's, whatever was in flight around it.  TEST INFRASTRUCTURE.
    python3 tools/fuzz_calls.py [--cases 40] [--seed 1] [--out gpurun_out/....json]"""
import argparse
import ctypes
import json
import os
import random
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import BatchRun, Mp3mi, Oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--emu", action="store_true", help="the emulated CPU test build, tiny batches (a check of this script's own logic)")
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    mp, orc = Mp3mi(emu=a.emu), Oracle()
    L = mp.lib
    L.mp3mi_batch_encode_host_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.mp3mi_batch_total_timing.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
    bad, frames, calls, t0 = [], 0, 0, time.time()
    for case in range(a.cases):
        rate, ch = rnd.choice(((44100, 2), (44100, 2), (48000, 2), (32000, 1)))
        kbps = rnd.choice((64, 128, 128, 192)) if ch == 2 else rnd.choice((48, 64))
        S = rnd.choice((1, 2, 3)) if a.emu else rnd.choice((1, 3, 64, 700, 4096, 5000))
        nf = rnd.choice((3, 4)) if a.emu else (rnd.choice((6, 9, 14, 23)) if S <= 700 else rnd.choice((4, 6, 9)))
        hold = rnd.choice((-1, -1, 0))
        opt = mp.options(chunk_frames=rnd.choice((0, 1, 2, 3, 5)), call_hold=hold)
        run = BatchRun(mp, S, rate, ch, kbps, nf, stream0=rnd.randrange(1 << 20), options=opt)
        row = run.n_per_ch * ch
        whole = run.mem.download(run.d_pcm, (S, row), np.int16)
        sample = sorted(set([0, S - 1] + [rnd.randrange(S) for _ in range(4)]))
        with ThreadPoolExecutor(max_workers=8) as ex:
            refs = dict(zip(sample, ex.map(lambda s: orc.encode(whole[s], rate, kbps, ch)[0], sample)))
        pending = []  # (kind, [(d_out or host array, d_len or host array)]): outputs to check after the next sync
        ops = []

        def poke():
            r = rnd.random()
            if r < 0.15:
                assert L.mp3mi_batch_sync(run.b) == 0
                ops.append("sync")
            elif r < 0.22:
                run.status()
                ops.append("status")
            elif r < 0.28:
                x = (ctypes.c_double(), ctypes.c_double(), ctypes.c_long(), ctypes.c_long())
                assert L.mp3mi_batch_total_timing(run.b, *[ctypes.byref(v) for v in x]) == 0
                ops.append("timing")

        try:
            for _ in range(rnd.choice((2, 3, 4, 5))):
                kind = rnd.choice("WWSH")
                if kind == "W":
                    d_o, d_l = run.mem.alloc(S * run.stride), run.mem.alloc(4 * S)
                    assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, d_o, run.stride, d_l) == 0
                    pending.append(("W", [(d_o, d_l)]))
                    ops.append("W")
                    calls += 1
                    poke()
                elif kind == "H":
                    h_o, h_l = np.zeros((S, run.stride), np.uint8), np.zeros(S, np.uint32)
                    assert L.mp3mi_batch_encode_host_async(run.b, whole.ctypes.data, nf, h_o.ctypes.data, run.stride, h_l.ctypes.data) == 0
                    pending.append(("H", [(h_o, h_l)]))
                    ops.append("H")
                    calls += 1
                    poke()
                else:
                    pieces, left = [], nf
                    while left:
                        p = rnd.randint(1, left if pieces else left - 1)  # (at least two pieces)
                        pieces.append(p)
                        left -= p
                    outs, f0 = [], 0
                    for p in pieces:
                        n = p * 1152 * ch
                        piece = np.ascontiguousarray(whole[:, f0 * 1152 * ch: f0 * 1152 * ch + n])
                        d_p = run.mem.alloc(piece.nbytes)
                        run.mem.upload(d_p, piece)
                        d_o, d_l = run.mem.alloc(S * run.stride), run.mem.alloc(4 * S)
                        assert L.mp3mi_batch_encode_next(run.b, d_p, p, d_o, run.stride, d_l) == 0
                        outs.append((d_o, d_l))
                        f0 += p
                        calls += 1
                        poke()
                    d_o, d_l = run.mem.alloc(S * run.stride), run.mem.alloc(4 * S)
                    assert L.mp3mi_batch_flush(run.b, d_o, run.stride, d_l) == 0
                    outs.append((d_o, d_l))
                    pending.append(("S", outs))
                    ops.append("S%s" % pieces)
                    poke()
            assert L.mp3mi_batch_sync(run.b) == 0
            for k, (kind, outs) in enumerate(pending):
                got = {s: b"" for s in sample}
                for o, l in outs:
                    if kind == "H":
                        out, lens = o, l
                    else:
                        out = run.mem.download(o, (S, run.stride), np.uint8)
                        lens = run.mem.download(l, (S,), np.uint32)
                    for s in sample:
                        got[s] += out[s, :lens[s]].tobytes()
                for s in sample:
                    if got[s] != refs[s]:
                        bad.append({"case": case, "call": k, "kind": kind, "stream": s, "S": S, "nf": nf, "rate": rate, "ch": ch, "kbps": kbps,
                                    "chunk_frames": opt.chunk_frames, "call_hold": hold, "ops": ops})
            frames += S * nf * len(pending)
        finally:
            run.close()
        print("case %d: %d streams x %d frames, %s, chunk %d, hold %d: %s  %s" % (case, S, nf, "%d/%d/%d" % (rate, ch, kbps), opt.chunk_frames, hold, " ".join(ops),
                                                                                  "ok" if not [b for b in bad if b["case"] == case] else "MISMATCH"), flush=True)
    res = {"what": "random chains of whole-file / streaming / host-buffer calls issued back to back, sampled streams of every call against the oracle (tools/fuzz_calls.py)",
           "cases": a.cases, "seed": a.seed, "calls": calls, "frames": frames, "mismatches": bad, "seconds": round(time.time() - t0, 1)}
    L.mp3mi_source_hash.restype = ctypes.c_char_p
    res["source_hash"] = L.mp3mi_source_hash().decode()
    print(json.dumps({k: v for k, v in res.items() if k != "mismatches"}), len(bad), "mismatches")
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
