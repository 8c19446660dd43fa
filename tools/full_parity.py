#!/usr/bin/env python3
"""Parity on the POPULATION: encodes a full-size BASELINE workload on the GPU (bench.py's configs, the same
deterministic PCM) and compares EVERY stream byte for byte with the CPU oracle (oracle/_build/liboracle.so, one
thread per host core) and a subset with the unmodified reference binary (oracle/_ref/encode).  Streams that
differ are re-encoded alone with stage seams and traced to the first differing seam.  Writes
profiles/parity_config<N>.json.  TEST INFRASTRUCTURE: the oracle is the checker here, never the thing shipped.

    python3 tools/full_parity.py --config 1 [--ref-every 16] [--streams N] [--frames F] [--flags 0]
"""
import argparse
import importlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def first_difference(a, b):
    n = min(len(a), len(b))
    x = np.frombuffer(a[:n], np.uint8) != np.frombuffer(b[:n], np.uint8)
    return int(np.argmax(x)) if x.any() else n


def trace_seam(pcm, rate, ch, kbps, frame_hint):
    """Re-encode one stream alone up to a little past the differing frame and name the first seam that differs
    from the oracle's stage dump (tests/stage_check.py)."""
    from mp3common import Mp3mi, Oracle
    from stage_check import compare_stages, run_batch_with_stages
    nf = min(len(pcm) // (1152 * ch), frame_hint + 3)
    p = pcm[: nf * 1152 * ch]
    os.environ["MP3MI_CHUNK_FRAMES"] = str(nf)
    got, st = run_batch_with_stages(Mp3mi(emu=False), p[None, :], rate, ch, kbps, nf)
    ref, dumps = Oracle().encode(p, rate, kbps, ch, dumps=nf)
    bad = compare_stages(st, 0, dumps, ch)
    return {"frames_reencoded": nf, "bytes_equal_alone": got[0] == ref, "first_seams": [str(b) for b in bad[:6]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4])
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--ref-every", type=int, default=16, help="also compare every n-th stream with oracle/_ref/encode (0 = none)")
    ap.add_argument("--flags", type=int, default=0, help="MP3MI_TEST_* exact-tier bits for the GPU run")
    ap.add_argument("--stream0", type=int, default=0, help="first stream index of the synthetic batch (other streams = other inputs)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from mp3common import Oracle

    mp3 = importlib.import_module("mp3-enc-bsd_amd")
    cfg = dict(bench.CONFIGS[args.config])
    if args.streams:
        cfg["streams"] = args.streams
    if args.frames:
        cfg["frames"] = args.frames
    S, nf, C, rate = cfg["streams"], cfg["frames"], cfg["channels"], cfg["rate"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.Workload(mp3, cfg, dev, stream0=args.stream0)
    if args.flags:
        wl.batch.set_test_flags(args.flags)
    t0 = time.perf_counter()
    wl.step()
    t_gpu = time.perf_counter() - t0
    out_h = wl.out.cpu().numpy()
    len_h = wl.out_len.cpu().numpy()
    cores = max(1, os.cpu_count() or 1)  # all host cores
    orc = Oracle()

    def check(s):
        pcm = wl.pcm[s].cpu().numpy()
        ref = orc.encode(pcm, rate, wl.kbps[s], C)[0]
        got = out_h[s, : len_h[s]].tobytes()
        if got == ref:
            return None
        off = first_difference(got, ref)
        return {"stream": s, "kbps": wl.kbps[s], "gpu_len": len(got), "ref_len": len(ref), "first_differing_byte": off,
                "first_differing_frame": off // wl.frame_bytes[s]}

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(check, range(S)))
    t_cpu = time.perf_counter() - t0
    bad = [r for r in res if r is not None]

    ref_checked, ref_bad = 0, []
    if args.ref_every > 0:
        idx = list(range(0, S, args.ref_every))
        # one reference process per stream; grouped by bitrate because reference_baseline takes one list
        pcm_sample = [wl.pcm[i].cpu().numpy() for i in idx]
        r = bench.reference_baseline(pcm_sample, rate, [wl.kbps[i] for i in idx], C, cores)
        if r is not None:
            ref_checked = len(idx)
            ref_bad = [int(i) for k, i in enumerate(idx) if out_h[i, : len_h[i]].tobytes() != r[1][k]]

    for b in bad[:4]:
        try:
            b["trace"] = trace_seam(wl.pcm[b["stream"]].cpu().numpy(), rate, C, b["kbps"], b["first_differing_frame"])
        except Exception as e:  # the trace is diagnostic only
            b["trace"] = {"error": repr(e)}

    rec = {
        "config": args.config, "workload": cfg["name"], "streams": S, "frames_per_stream": nf, "frames_total": S * nf,
        "rate_hz": rate, "channels": C, "kbps": cfg["kbps"], "test_flags": args.flags,
        "pcm": "mp3mi_synth_pcm_device, seed 0x%08x, streams %d..%d" % (bench.SEED, args.stream0, args.stream0 + S - 1),
        "compared_with_oracle": S, "mismatching_streams": len(bad), "mismatches": bad[:32],
        "compared_with_reference_binary": ref_checked, "reference_binary_mismatches": ref_bad,
        "bit_exact": len(bad) == 0 and len(ref_bad) == 0,
        "gpu_seconds_one_pass": round(t_gpu, 3), "oracle_seconds": round(t_cpu, 1), "host_cores": cores,
        "device": torch.cuda.get_device_name(0), "library": mp3.lib().mp3mi_version().decode(),
    }
    dest = args.out or os.path.join(ROOT, "gpurun_out", "parity_config%d%s.json" % (args.config, "_flags%d" % args.flags if args.flags else ""))
    os.makedirs(os.path.dirname(dest), exist_ok=True)
    json.dump(rec, open(dest, "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "mismatches"}))
    wl.close()
    if not rec["bit_exact"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
