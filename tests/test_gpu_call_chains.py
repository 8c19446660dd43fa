"""Random chains of calls issued back to back (tools/fuzz_calls.py) in the driver's -m gpu run: whole-file, streaming and
host-buffer calls on one batch without a sync between them -- or with a sync, a status or a timing query thrown in --, 1 to
5000 streams, chunk lengths forced small, the hold of a call's last k_loop (csrc/batch.cpp, k_hold) on and off; sampled
streams of every call against the oracle.  40 chains, a fixed seed: about ten seconds."""
import json
import os
import subprocess
import sys

import pytest

from mp3common import ROOT

pytestmark = pytest.mark.gpu


def test_call_chains_against_the_oracle(tmp_path):
    out = tmp_path / "chains.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_calls.py"), "--cases", "40", "--seed", "2025", "--out", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.load(open(out))
    assert d["cases"] == 40 and d["calls"] > 100 and not d["mismatches"]
