"""HIP path against the committed golden vectors (reference output), no oracle in between.

tests/golden/MANIFEST.json holds two kinds of fixtures: inputs the reference ENCODES (PCM, the md5 and length of its
MP3, its stage dumps of some frames) and inputs the reference DIES on (an assertion of its code fails; the expectation
is the status the product reports for the stream).  Together they execute every line and branch outcome of the
reference's Layer III code that any input can reach: profiles/r03_ref_coverage.json (tools/ref_coverage.py),
classification of the rest in tests/golden/coverage_notes.json."""
import hashlib

import pytest

import abort_cases
from golden_util import aborting_cases, case_pcm, case_stages, encoding_cases
from mp3common import pad_frames
from stage_check import compare_stages, run_batch_with_stages

pytestmark = pytest.mark.gpu
CASES = encoding_cases()
ABORTS = aborting_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_reproduces_reference_golden(product, case):
    pcm, nf = pad_frames(case_pcm(case, product.synth), case["channels"])
    assert nf == case["frames"]
    got, st = run_batch_with_stages(product, pcm[None, :], case["rate"], case["channels"], case["kbps"], nf, mode=case.get("mode"))
    assert len(got[0]) == case["mp3_len"]
    assert hashlib.md5(got[0]).hexdigest() == case["mp3_md5"]
    assert st["status"][0] == 0
    gold, frames = case_stages(case)
    if nf <= 64:  # the stage seams of the last chunk are fetchable when one chunk covers the stream
        bad = compare_stages(st, 0, gold, case["channels"], frames)
        assert not bad, bad[:8]


@pytest.mark.parametrize("case", ABORTS, ids=[c["name"] for c in ABORTS])
def test_gpu_reports_where_the_reference_dies(product, case):
    """the stream is voided (length 0), mp3mi_batch_sync says MP3MI_ERR_REFERENCE_ABORT, the status names the assertion
    and the frame"""
    pcm, nf = pad_frames(case_pcm(case, product.synth), case["channels"])
    got, st = run_batch_with_stages(product, pcm[None, :], case["rate"], case["channels"], case["kbps"], nf, mode=case.get("mode"),
                                    expect_abort=True)
    assert got[0] == b""
    assert st["status"][0] & 255 == case["reference_aborts"]["status"]
    assert st["status"][0] >> 8 == case["reference_aborts"]["frame"]


def test_gpu_an_aborting_stream_leaves_its_neighbours_alone(product, oracle):
    abort_cases.neighbours_case(product, oracle)


def test_gpu_host_wrapper_delivers_the_neighbours_of_an_aborting_stream(product, oracle):
    """mp3mi_encode_host (the host-buffer path: PCM up and bytes down beside the kernels) with a dying stream between two
    good ones: MP3MI_ERR_REFERENCE_ABORT AND the neighbours' files (csrc/batch.cpp, encode_host_impl)"""
    abort_cases.host_wrapper_case(product, oracle)


def test_gpu_a_streaming_call_reports_the_abort_when_it_happens(product, oracle):
    """k_stream_tail's bookkeeping on the device: the abort is reported by the sync after the call in which it happened,
    once; later calls and the flush deliver nothing for the stream; the status survives the flush (csrc/k_format.hip)"""
    abort_cases.streaming_case(product, oracle)
