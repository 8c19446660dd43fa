"""The direct FFT seam (fft / rsfft / enphinew, src/subs.c:38-534): `energy`, `phi` and the raw
spectrum lines as the reference's own fft() returned them to L3psycho_anal -- intercepted at link time in our driver
around the unmodified reference objects (oracle/ref_harness.c -DFFT_SEAM, oracle/gen_golden_fft.py) and committed for
three fixtures (tests/golden/*.fft.npz) -- against the oracle's restatement, the emulated kernels and, under -m gpu, the
device's k_fft.  A transform that drifts is named here, not three kernels later through a ratio."""
import os

import numpy as np
import pytest

from golden_util import GOLD, case_pcm, manifest
from mp3common import pad_frames
from stage_check import compare_fft_seam, run_batch_with_stages

FIXTURES = [c for c in manifest() if os.path.exists(os.path.join(GOLD, c["name"] + ".fft.npz"))]
IDS = [c["name"] for c in FIXTURES]


def golden_seam(case):
    return np.load(os.path.join(GOLD, case["name"] + ".fft.npz"))["seam"].reshape(-1, 2, case["channels"])


def test_fixtures_exist():
    assert len(FIXTURES) >= 2


@pytest.mark.parametrize("case", FIXTURES, ids=IDS)
def test_oracle_transforms_match_the_reference(oracle, emu, case):
    """every member, phi included, bit for bit"""
    gold = golden_seam(case)
    got = oracle.fft_seam(case_pcm(case, emu.synth), case["rate"], case["kbps"], case["channels"], len(gold))
    for name in gold.dtype.names:
        a, b = np.ascontiguousarray(got[name]), np.ascontiguousarray(gold[name])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name


def device_case(mp, case):
    gold = golden_seam(case)
    nf = len(gold)
    pcm, _ = pad_frames(case_pcm(case, mp.synth), case["channels"])
    pcm = pcm[: nf * 1152 * case["channels"]]
    _, st = run_batch_with_stages(mp, pcm[None, :], case["rate"], case["channels"], case["kbps"], nf, mode=case.get("mode"))
    bad = compare_fft_seam(st, 0, gold, case["channels"])
    assert not bad, bad[:8]
    # the floor is exercised where the fixture was made for it (src/subs.c:70-74)
    if case["name"] == "x44_128_faint_after_silence":
        assert np.any(gold["energy_l"] == np.float32(0.0005)) and np.any(gold["energy_l"] > np.float32(0.0005))


@pytest.mark.parametrize("case", FIXTURES, ids=IDS)
def test_emulated_k_fft_matches_the_reference(emu, case):
    device_case(emu, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FIXTURES, ids=IDS)
def test_gpu_k_fft_matches_the_reference(product, case):
    device_case(product, case)


@pytest.mark.gpu
def test_gpu_k_fft_matches_the_oracle_on_a_batch(product, oracle):
    """a batch wide enough that every wavefront slot of k_fft's persistent workgroups takes several tasks: 300 streams x
    4 frames, every 7th stream against the oracle's seam"""
    S, nf, rate, ch, kbps = 300, 4, 44100, 2, 128
    pcm = np.stack([product.synth(nf * 1152, ch, rate, 4000 + s) for s in range(S)])
    _, st = run_batch_with_stages(product, pcm, rate, ch, kbps, nf)
    for s in range(0, S, 7):
        bad = compare_fft_seam(st, s, oracle.fft_seam(pcm[s], rate, kbps, ch, nf), ch)
        assert not bad, bad[:8]
