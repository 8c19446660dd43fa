"""The four BASELINE workloads at FULL size under the driver's -m gpu run (bench.py --config 1..4 are the same batches):
reservoir dynamics over 383 / 417 / 278 frames together with placement, pacing, chunking and -- for 8192 and 16 384
streams -- k_loop in parts.  One encode call each; 64 streams spread over the batch are compared byte for byte with the
oracle, 8 of them also with the unmodified reference binary (oracle/_ref/encode, built by __graft_entry__.build() and
carried to the GPU box).  tools/full_parity.py compares every stream (profiles/*_parity_config*.json)."""
import os
import struct
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from mp3common import ROOT, BatchRun

pytestmark = pytest.mark.gpu
REF_ENCODE = os.path.join(ROOT, "oracle", "_ref", "encode")
MIX48 = [64, 96, 128, 192, 256, 320]
CONFIGS = {  # bench.py CONFIGS / SURVEY.md 8(d)
    1: dict(streams=4096, frames=383, rate=44100, channels=2, kbps=128),
    2: dict(streams=8192, frames=383, rate=44100, channels=2, kbps=128),
    3: dict(streams=4096, frames=417, rate=48000, channels=2, kbps="mix48"),
    4: dict(streams=16384, frames=278, rate=32000, channels=1, kbps=64),
}


def reference_binary(pcm, rate, ch, kbps):
    with tempfile.TemporaryDirectory() as td:
        wav, mp3 = os.path.join(td, "a.wav"), os.path.join(td, "a.mp3")
        data = np.ascontiguousarray(pcm, dtype="<i2").tobytes()
        with open(wav, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)
        subprocess.run([REF_ENCODE, "-s", "%g" % (rate / 1000.0), "-b", str(kbps)] + (["-m", "m"] if ch == 1 else []) + [wav, mp3],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(mp3, "rb").read()


@pytest.mark.parametrize("cfg_id", [1, 2, 3, 4])
def test_baseline_config_at_full_size(product, oracle, cfg_id):
    c = CONFIGS[cfg_id]
    S, nf, rate, ch = c["streams"], c["frames"], c["rate"], c["channels"]
    kb = [MIX48[s % 6] for s in range(S)] if c["kbps"] == "mix48" else c["kbps"]
    run = BatchRun(product, S, rate, ch, kb, nf)  # PCM: mp3mi_synth_pcm_device, streams 0 .. S-1, as bench.py
    try:
        L = product.lib
        assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, run.d_out, run.stride, run.d_len) == 0
        assert L.mp3mi_batch_sync(run.b) == 0
        lens = run.mem.download(run.d_len, (S,), np.uint32)
        frame_bytes = np.array([int(1152 / (rate / 1000.0) * k / 8) for k in (kb if isinstance(kb, list) else [kb] * S)])
        # every stream: a whole file, the tail short of the last slot by the reservoir's leftover (<= 511 bytes + 1)
        assert np.all(lens <= nf * frame_bytes + 1) and np.all(lens + 512 > nf * frame_bytes)
        sample = sorted(set(np.linspace(0, S - 1, 64).astype(int).tolist()))
        pcm = {s: run.pcm_of(s) for s in sample}
        got = {s: run.mem.download(run.d_out + s * run.stride, (int(lens[s]),), np.uint8).tobytes() for s in sample}
        kof = (lambda s: kb[s]) if isinstance(kb, list) else (lambda s: kb)
        with ThreadPoolExecutor(max_workers=16) as ex:
            refs = dict(zip(sample, ex.map(lambda s: oracle.encode(pcm[s], rate, kof(s), ch)[0], sample)))
        bad = [s for s in sample if got[s] != refs[s]]
        assert not bad, "config %d: streams %s differ from the oracle" % (cfg_id, bad[:8])
        if os.path.exists(REF_ENCODE):
            eight = sample[::8]
            with ThreadPoolExecutor(max_workers=8) as ex:
                rb = dict(zip(eight, ex.map(lambda s: reference_binary(pcm[s], rate, ch, kof(s)), eight)))
            bad = [s for s in eight if got[s] != rb[s]]
            assert not bad, "config %d: streams %s differ from the reference binary" % (cfg_id, bad)
    finally:
        run.close()


def test_schedule_between_the_tuned_sizes(product, oracle):
    """The schedule (k_loop in parts, what runs beside which part: batch.cpp) is chosen from the stream count; the counts
    between the BASELINE sizes take the same code with ragged last parts: 6000 (two parts of 3008 / 2992), 12 288 (three
    of 4096) and 20 000 (five of 4032 / ... / 3872) streams, three chunks each, 48 streams per batch against the oracle."""
    for S, rate, ch, kbps, nf in ((6000, 44100, 2, 128, 12), (12288, 44100, 2, 128, 9), (20000, 32000, 1, 64, 9)):
        run = BatchRun(product, S, rate, ch, kbps, nf, stream0=7000, options=product.options(chunk_frames=(nf + 2) // 3))
        try:
            out_len = None
            L = product.lib
            assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, run.d_out, run.stride, run.d_len) == 0
            assert L.mp3mi_batch_sync(run.b) == 0
            lens = run.mem.download(run.d_len, (S,), np.uint32)
            sample = sorted(set(np.linspace(0, S - 1, 48).astype(int).tolist()))
            for s in sample:
                ref = oracle.encode(run.pcm_of(s), rate, kbps, ch)[0]
                got = run.mem.download(run.d_out + s * run.stride, (int(lens[s]),), np.uint8).tobytes()
                assert got == ref, "%d streams: stream %d differs from the oracle" % (S, s)
        finally:
            run.close()


@pytest.mark.parametrize("layer", [3, 2])
def test_two_ranks_share_the_gpu(layer):
    """bench.py's N > 1 path rehearsed on the one-GPU box, as the driver would start it: plain `python bench.py --gpus 2`.
    bench.py itself starts the two ranks (gloo for the barrier and the max-over-ranks time, both on device 0:
    MP3MI_BENCH_ONE_GPU=1); each encodes its own stream range and checks it against the oracle; the line says n_gpus 2,
    bit_exact, and names two disjoint ranges with different bytes.  A child process: this one is not replaced."""
    import json
    import sys
    env = dict(os.environ, MP3MI_BENCH_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--streams", "1024", "--frames", "48", "--layer", str(layer)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's), relayed by the launcher"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity_spot_check"]["bit_exact"] and d["value"] > 0
    ranks = sorted(d["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == [0, 1]
    assert ranks[0]["first_stream"] + ranks[0]["streams"] <= ranks[1]["first_stream"]  # disjoint stream ranges
    assert ranks[0]["sample_digest"] != ranks[1]["sample_digest"]
    # the CPU baseline is part of the line at every N (rank 0 times it after the timed region), and the roofline object
    # names the bound the kernel runs against beside its HBM figure
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["roofline"]["bound"] == "valu-issue" and d["roofline"]["unit"] == "GB/s" and 0 < d["roofline"]["frac"] < 1
    if layer == 3:  # the PCIe leg is part of the Layer III line at every N
        e = d["end_to_end"]
        assert e is not None and e["bytes_equal_resident_path"] and e["frames_per_s"] > 0


def test_four_ranks_rehearse_the_eight_gpu_command_on_one_gpu():
    """`python bench.py --gpus 8` is one command the day an 8-GPU node exists; what can be rehearsed on one device is: the launcher's
    ranks (four: the pool allows at most six processes on a card at once, and the children a rank forks for the CPU baseline
    count against it between fork and exec), their disjoint stream ranges, the control traffic over gloo
    (collective_backend says so), the all-or-none vote on page-locked buffers of the end_to_end leg, ONE JSON line -- and the
    launcher's own end: SIGTERM to it must take the ranks along (no orphan keeps the GPU)."""
    import json
    import signal
    import sys
    import time
    env = dict(os.environ, MP3MI_BENCH_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--streams", "512", "--frames", "48"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["parity_spot_check"]["bit_exact"] and d["value"] > 0
    ranks = sorted(d["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == list(range(4))
    assert all(ranks[k]["first_stream"] + ranks[k]["streams"] <= ranks[k + 1]["first_stream"] for k in range(3))
    assert len({x["sample_digest"] for x in ranks}) == 4
    assert d["collective_backend"].startswith("gloo") and isinstance(d["cpu_affinity_rank0"], str)
    assert d["end_to_end"] is not None and (d["end_to_end"].get("bytes_equal_resident_path") or "skipped" in d["end_to_end"])
    assert d["cpu_baseline"]["cores"] == (os.cpu_count() or 1) or d["cpu_baseline"]["cores"] == len(os.sched_getaffinity(0))
    # the launcher is stopped from outside: its ranks end with it
    p = subprocess.Popen(cmd[:-1] + ["383"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(20)  # (the ranks are up and encoding)
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) == 4, kids
    p.send_signal(signal.SIGTERM)
    p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM
    deadline = time.time() + 30
    while time.time() < deadline and any(os.path.exists("/proc/%s" % k) for k in kids):
        time.sleep(0.2)
    assert not any(os.path.exists("/proc/%s" % k) for k in kids), "ranks outlived their launcher"


def test_under_an_external_launcher_the_flag_must_match():
    """`--gpus` is what the line will say; a launcher that started another number of ranks is refused (before any GPU work),
    and so is asking for more ranks than the node has devices."""
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MP3MI_BENCH_ONE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--streams", "64", "--frames", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "one rank per GPU" in r.stderr and not r.stdout.strip()
