#!/usr/bin/env python3
"""Parity across the PARAMETER SPACE: every MPEG-1 sampling rate x channel count x Layer III bitrate the reference
accepts, a batch of streams each, every stream compared byte for byte with the CPU oracle (all host cores); one
combination in four also with error protection / dual-channel mode against the reference binary.  Writes
profiles-style JSON to gpurun_out/parity_matrix.json.  TEST INFRASTRUCTURE (the oracle is the checker).

    python3 tools/matrix_parity.py [--streams 192] [--frames 96]
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

BITRATES = [32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=192)
    ap.add_argument("--frames", type=int, default=96)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_matrix.json"))
    args = ap.parse_args()
    from mp3common import Oracle
    mp3 = importlib.import_module("mp3-enc-bsd_amd")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    orc = Oracle()
    cores = max(1, min(os.cpu_count() or 1, 64))
    S, nf = args.streams, args.frames
    rows, total_frames, total_bad, t0 = [], 0, 0, time.perf_counter()
    combo = 0
    for rate in (44100, 48000, 32000):
        for ch in (2, 1):
            pcm = torch.empty((S, nf * 1152 * ch), dtype=torch.int16, device=dev)
            mp3.synth_pcm_device(pcm, nf * 1152, ch, rate, stream0=1000 * combo, seed=bench.SEED)
            pcm_h = pcm.cpu().numpy()
            for kbps in BITRATES:
                combo += 1
                b = mp3.Batch(S, rate, ch, kbps, nf)
                out = torch.zeros((S, b.out_stride(nf)), dtype=torch.uint8, device=dev)
                ln = torch.zeros(S, dtype=torch.int32, device=dev)
                b.encode(pcm, nf, out, ln)
                b.sync()
                out_h, len_h = out.cpu().numpy(), ln.cpu().numpy()
                with ThreadPoolExecutor(max_workers=cores) as ex:
                    refs = list(ex.map(lambda s: orc.encode(pcm_h[s], rate, kbps, ch)[0], range(S)))
                bad = [s for s in range(S) if out_h[s, : len_h[s]].tobytes() != refs[s]]
                row = {"rate": rate, "channels": ch, "kbps": kbps, "streams": S, "frames": nf, "mismatching_streams": len(bad)}
                if combo % 4 == 0:  # the driver's -e (and -m d for stereo) against the reference binary, 16 streams
                    b.set_error_protection(1)
                    if ch == 2:
                        b.set_mode(2)
                    b.encode(pcm, nf, out, ln)
                    b.sync()
                    o2, l2 = out[:16].cpu().numpy(), ln[:16].cpu().numpy()
                    r = reference_opts([pcm_h[s] for s in range(16)], rate, kbps, ch, cores)
                    if r is not None:
                        row["with_crc%s_vs_reference_binary" % ("_dual" if ch == 2 else "")] = sum(
                            1 for s in range(16) if o2[s, : l2[s]].tobytes() != r[s])
                        bad += [("opts", s) for s in range(16) if o2[s, : l2[s]].tobytes() != r[s]]
                b.close()
                rows.append(row)
                total_frames += S * nf
                total_bad += len(bad)
                print(json.dumps(row), flush=True)
    rec = {"what": "every MPEG-1 rate x channels x Layer III bitrate, all streams vs oracle/liboracle.so", "combinations": len(rows),
           "frames_total": total_frames, "mismatching_streams_total": total_bad, "bit_exact": total_bad == 0,
           "seconds": round(time.perf_counter() - t0, 1), "device": torch.cuda.get_device_name(0), "rows": rows}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rec, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "rows"}))
    if total_bad:
        raise SystemExit(1)


def reference_opts(pcm_list, rate, kbps, ch, cores):
    """oracle/_ref/encode -e [-m d] on the given streams, or None where the binary is absent"""
    import shutil
    import struct
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "encode")
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="mp3refo_")

    def run(k):
        data = np.ascontiguousarray(pcm_list[k], dtype="<i2").tobytes()
        wav, mp3f = os.path.join(tmp, "%d.wav" % k), os.path.join(tmp, "%d.mp3" % k)
        with open(wav, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)
        a = [exe, "-s", "%g" % (rate / 1000.0), "-b", str(kbps), "-e"] + (["-m", "d"] if ch == 2 else ["-m", "m"])
        subprocess.run(a + [wav, mp3f], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(mp3f, "rb").read()

    with ThreadPoolExecutor(max_workers=cores) as ex:
        outs = list(ex.map(run, range(len(pcm_list))))
    shutil.rmtree(tmp, ignore_errors=True)
    return outs


if __name__ == "__main__":
    main()
