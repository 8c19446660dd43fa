"""Golden vectors of Layers I and II (tests/golden/l12_*.npz, written by oracle/gen_golden_l12.py from the unmodified
reference): loading and comparison helpers shared by the CPU and GPU tests."""
import json
import os

import numpy as np

from mp3common import L12_SEAMS, ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "L12_MANIFEST.json")))


def load(name):
    z = np.load(os.path.join(GOLD, "l12_%s.npz" % name))
    return MANIFEST[name], z["pcm"], z["mpg"].tobytes(), z["dumps"]


def seams_equal(ref_dumps, got_dumps, with_sb_frames=0):
    """[(frame, seam)] where stage_dump_l12 records differ"""
    bad = []
    for f in range(len(ref_dumps)):
        for name in L12_SEAMS + (["sb"] if f < with_sb_frames else []):
            if not np.array_equal(ref_dumps[f][name], got_dumps[f][name]):
                bad.append((f, name))
    return bad
