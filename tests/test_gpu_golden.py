"""HIP path against the committed golden vectors (reference output), no oracle in between."""
import hashlib

import numpy as np
import pytest

from golden_util import case_pcm, case_stages, manifest
from mp3common import pad_frames
from stage_check import compare_stages, run_batch_with_stages

pytestmark = pytest.mark.gpu
CASES = manifest()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_reproduces_reference_golden(product, case):
    pcm, nf = pad_frames(case_pcm(case, product.synth), case["channels"])
    assert nf == case["frames"]
    got, st = run_batch_with_stages(product, pcm[None, :], case["rate"], case["channels"], case["kbps"], nf)
    assert len(got[0]) == case["mp3_len"]
    assert hashlib.md5(got[0]).hexdigest() == case["mp3_md5"]
    gold = case_stages(case)
    if nf <= 64:  # the stage seams of the last chunk are fetchable when one chunk covers the stream
        bad = compare_stages(st, 0, gold, case["channels"])
        assert not bad, bad[:8]
