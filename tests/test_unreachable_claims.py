"""Claims of tests/golden/coverage_notes.json ("no input can reach this line of the reference") that rest on a table
rather than on an argument: checked here against the tables themselves, as the oracle holds them."""
import json
import os
import re

from mp3common import ROOT


def c_array(src, name):
    m = re.search(name + r"\s*=\s*\{(.*?)\};", src, re.S)
    assert m, name
    return m.group(1)


def test_subdivide_never_corrects_its_table_lookup():
    """src/loop.c:1663-1666, 1673-1676: the while loops that walk region0_count / region1_count back only run when
    scalefac_band_long[index] > bigvalues_region, i.e. index >= the number of bands the big values reach (scfb_anz).
    subdv_table keeps region0_count + 1 and region0_count + region1_count + 2 below scfb_anz for every scfb_anz that
    has a non-zero count, so they never do -- for all three rates, every big_values."""
    src = open(os.path.join(ROOT, "oracle", "mp3_oracle.c")).read()
    subdv = [tuple(int(x) for x in p) for p in re.findall(r"\{\s*(\d+)\s*,\s*(\d+)\s*\}", c_array(src, r"SUBDV\[23\]\[2\]"))]
    assert len(subdv) == 23
    bands = re.findall(r"\{([0-9,\s]+)\}", c_array(src, r"SFB_L\[3\]\[23\]"))
    assert len(bands) == 3
    for row in bands:
        sfb = [int(x) for x in row.split(",")]
        assert len(sfb) == 23 and sfb[0] == 0 and sfb[22] == 576
        for big_values in range(1, 289):
            region = 2 * big_values
            anz = 0
            while sfb[anz] < region:
                anz += 1
            r0, r1 = subdv[anz]
            assert not (r0 and sfb[r0 + 1] > region), (big_values, anz)
            assert not (r1 and sfb[r0 + r1 + 2] > region), (big_values, anz)


def test_drain_is_never_a_multiple_of_32():
    """src/l3bitstream.c:507: `if ( remainingBits )` is never false -- the drain only exists for mono at 32 kHz with
    256 / 320 kbps and is the constant 2 * mean_bits - 2 * 4095 there (with and without -e)."""
    for rate_khz in (44.1, 48.0, 32.0):
        for kbps in (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320):
            for ch in (1, 2):
                for crc in (0, 16):
                    bits = 8 * int((1152.0 / rate_khz) * (kbps / 8.0))
                    payload = bits - (32 + (136 if ch == 1 else 256) + crc)
                    # stuffing = ResvSize + payload - sum(p23) - ResvMax <= payload - sum(p23) (reservoir.c:176-186), and plan b
                    # places up to 2 * ch * 4095 - sum(p23) of it (:199-214): the drain is at most payload - 2 * ch * 4095,
                    # and exactly that where it is positive -- there the frame is longer than 7680 bits, ResvMax = 0 (:81-82)
                    drain = payload - 2 * ch * 4095
                    if drain > 0:
                        assert bits > 7680 and (rate_khz, ch) == (32.0, 1) and kbps in (256, 320) and drain % 32 != 0, (rate_khz, kbps, ch, crc, drain)


def test_notes_name_existing_fixtures():
    notes = json.load(open(os.path.join(ROOT, "tests", "golden", "coverage_notes.json")))
    names = {c["name"] for c in json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST.json")))}
    for where, ab in notes["reference_aborts"].items():
        assert ab["fixture"] is None or ab["fixture"] in names, where
