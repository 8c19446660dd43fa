import hashlib
import json
import os

import numpy as np

from mp3common import ROOT, SEED

GOLD = os.path.join(ROOT, "tests", "golden")


def manifest():
    return json.load(open(os.path.join(GOLD, "MANIFEST.json")))


def case_pcm(case, synth):
    """The case's PCM: from the committed file when small, else regenerated and md5-checked."""
    if "pcm_file" in case:
        pcm = np.load(os.path.join(GOLD, case["pcm_file"]))
    else:
        pcm = synth(case["n_samples_per_ch"], case["channels"], case["rate"], case["stream"], SEED)
    assert hashlib.md5(pcm.tobytes()).hexdigest() == case["pcm_md5"], "input PCM drifted for " + case["name"]
    return pcm


def case_stages(case):
    return np.load(os.path.join(GOLD, case["name"] + ".stages.npz"))["dumps"]
