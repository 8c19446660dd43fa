"""The schedule of a step rests on what fits where (DESIGN.md sections 4 and 5): four wavefronts of k_loop per SIMD (at most 88
registers each as the hardware allocates them, no scratch) leave 160 registers and ~31 KB of a CU's LDS to ONE wavefront of the next
chunk's kernels beside them (k_cw, k_part, k_psy, k_filter; k_mdct moves in as k_loop's wavefronts leave); the transforms
run three (long) and four (short) wavefronts per SIMD in one workgroup per CU.  A kernel that grows past its budget does not fail --
it silently no longer runs beside k_loop, and the pipeline settles in another order (EXPERIMENTS.md: +15 .. +70 ms per step).
This test compiles the three kernel files for gfx950 with the compiler's resource remarks and checks the budgets.  CPU only
(hipcc cross-compiles); the objects go to a temporary directory."""
import os
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def makefile_flags(stem):
    """FLAGS_<stem> of csrc/Makefile: what the product build adds for this file (k_loop: no machine-level hoisting)."""
    for line in open(os.path.join(CSRC, "Makefile")):
        m = re.match(r"FLAGS_%s\s*\?=\s*(.*)$" % stem, line)
        if m:
            return m.group(1).split()
    return []


def resources(stem, tmp):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-I.", "-I../../include"] + makefile_flags(stem) + [
           "-Rpass-analysis=kernel-resource-usage", "-c", stem + ".hip", "-o", os.path.join(tmp, stem + ".o")]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|SGPRs Spill): (\d+)", line)
        if m and name:
            out[name]["SGPRSpill" if m.group(1) == "SGPRs Spill" else m.group(1).split(" ")[0]] = int(m.group(2))
    return out


def the(kernels, pattern):
    hits = [v for k, v in kernels.items() if re.search(pattern, k)]
    assert len(hits) == 1, (pattern, [k for k in kernels if re.search(pattern, k)])
    return hits[0]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_the_kernels_fit_where_the_schedule_puts_them():
    tmp = tempfile.mkdtemp(prefix="mp3mi_budgets_")
    try:
        with ThreadPoolExecutor(4) as ex:
            loop, fbm, fft, psy = ex.map(lambda s: resources(s, tmp), ["k_loop", "k_fbmdct", "k_fft", "k_psy"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    k_loop = the(loop, r"^_Z6k_loopPK")
    assert k_loop["VGPRs"] <= 88 and k_loop["ScratchSize"] == 0, k_loop            # four per SIMD (allocated by eights), 160 registers left beside them
    assert k_loop["LDS"] <= 32 * 1024 + 256, k_loop                                # four 4-wavefront workgroups a CU (63 LDS granules of 512 bytes each), ~31 KB left
    # scalar registers the compiler parks in vector-register lanes: every access is a VECTOR instruction in an issue-bound kernel
    # (round 5: 131, 355 lane moves in the listing; round 6: the kernel's arguments are fetched where they are used: 86 .. 89)
    assert k_loop["SGPRSpill"] <= 90, k_loop
    beside = 512 - 4 * 88
    k_mdct, k_filter = the(fbm, r"^_Z6k_mdctPK"), the(fbm, r"^_Z8k_filterPK")
    assert k_mdct["VGPRs"] <= 192 and k_mdct["ScratchSize"] == 0, k_mdct           # two per SIMD when alone; beside k_loop as its wavefronts leave
    k_psy, k_part = the(psy, r"^_Z5k_psyILb1E"), the(psy, r"^_Z6k_partPK")
    assert k_psy["VGPRs"] <= beside and k_part["VGPRs"] <= beside, (k_psy, k_part)
    assert k_psy["ScratchSize"] == 0 and k_part["ScratchSize"] == 0, (k_psy, k_part)   # (round 5: k_psy<true> spilled two registers under a bound of 128)
    assert k_mdct["LDS"] <= 160 * 1024 // 8, k_mdct                                # eight one-wavefront workgroups a CU when alone
    assert k_filter["VGPRs"] <= beside and k_filter["ScratchSize"] == 0 and k_filter["LDS"] <= 10 * 1024, k_filter
    f_long, f_short = the(fft, r"^_Z5k_fftILi2ELi12ELb1E"), the(fft, r"^_Z5k_fftILi2ELi16ELb0E")
    assert f_long["VGPRs"] <= 168 and f_long["ScratchSize"] == 0 and f_long["LDS"] <= 160 * 1024, f_long    # 12 wavefronts: three per SIMD
    assert f_short["VGPRs"] <= 128 and f_short["ScratchSize"] == 0 and f_short["LDS"] <= 160 * 1024, f_short  # 16: four per SIMD
    k_cw = the(fft, r"^_Z4k_cwPK")
    assert k_cw["VGPRs"] <= 64, k_cw
