"""Parity on the POPULATION, under the driver: BASELINE configs[1] at full size (4096 streams x 383 frames, 44.1 kHz stereo,
128 kbps) encoded on the GPU, and EVERY stream compared byte for byte with the CPU oracle on all host cores, every 32nd
also with the unmodified reference binary (oracle/_ref/encode) -- tools/full_parity.py as a child process (the GPU-side
sample checks of tests/test_gpu_fullsize.py compare 64 streams per workload; this one compares all 1 568 768 frames)."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_every_stream_of_configs1_against_the_oracle():
    out = os.path.join(ROOT, "gpurun_out", "population_parity_config1.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "full_parity.py"), "--config", "1", "--ref-every", "32", "--out", out],
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rec = json.load(open(out))
    assert rec["streams"] == 4096 and rec["frames_per_stream"] == 383 and rec["compared_with_oracle"] == 4096
    assert rec["mismatching_streams"] == 0 and rec["bit_exact"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "encode")):  # (travels with the snapshot; absent only in a bare checkout)
        assert rec["compared_with_reference_binary"] == 128 and rec["reference_binary_mismatches"] == []
    print("population parity: %d frames, oracle %.0f s on %d cores, whole test %.0f s" % (rec["frames_total"], rec["oracle_seconds"], rec["host_cores"], time.time() - t0))
