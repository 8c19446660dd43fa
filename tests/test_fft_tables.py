"""The FFT tables the kernels run from (csrc/tables_host.cpp, FftGen): the leaves k_fft finishes in registers cover every element
exactly once with whole pairs, no round of the program names an element twice, and the modelled LDS cycles of the program stay
where the annealed placement left them (profiles/r06_experiments.txt, F10 / F10b: long 457 of 336, short 381 of 264).  That the
tables compute the reference's transform is what the emulator and GPU parity tests check; this one pins their SHAPE.  CPU only:
tests/fft_tables_check.cpp is compiled against csrc/tables_host.cpp and the committed table blob."""
import os
import shutil
import subprocess
import tempfile

import pytest

from mp3common import ROOT

CSRC = os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("ld") is None, reason="needs g++ and ld")
def test_leaves_cover_the_transforms_and_the_program_keeps_its_cycles():
    tmp = tempfile.mkdtemp(prefix="mp3mi_ffttab_")
    try:
        blob_o = os.path.join(tmp, "tables_blob.o")
        r = subprocess.run(["ld", "-r", "-b", "binary", "-z", "noexecstack", "-o", blob_o, "tables_blob.bin"], cwd=CSRC, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        exe = os.path.join(tmp, "check")
        cmd = ["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-DMP3MI_EMU", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "tests", "hipemu"), os.path.join(ROOT, "tests", "fft_tables_check.cpp"),
               os.path.join(CSRC, "tables_host.cpp"), blob_o, "-o", exe, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = {l.split()[0]: l.split() for l in r.stdout.splitlines()}
        val = lambda kind, key: int(lines[kind][lines[kind].index(key) + 1])
        assert val("long", "rounds") == 9 and val("short", "rounds") == 8, r.stdout
        assert val("long", "leaves") == 64 and val("short", "leaves") == 48, r.stdout
        # the placement as annealed: a change of the generator that loses it shows here, not as a slower kernel three weeks later
        assert val("long", "cycles") <= 460 and val("short", "cycles") <= 388, r.stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
