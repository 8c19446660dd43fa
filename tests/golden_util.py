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
    """(stage dumps, the frame index of each): the first dump_frames frames, or the frames the case names"""
    d = np.load(os.path.join(GOLD, case["name"] + ".stages.npz"))["dumps"]
    return d, case.get("dump_frame_indices", list(range(len(d))))


def encoding_cases():
    """fixtures the reference encodes"""
    return [c for c in manifest() if not c.get("reference_aborts")]


def aborting_cases():
    """fixtures the reference dies on (an assertion fails): the expectation is the status"""
    return [c for c in manifest() if c.get("reference_aborts")]
