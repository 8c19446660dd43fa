#!/usr/bin/env python3
"""Which lines and branches of the reference's Layer I / II code does the parity corpus reach?  TEST INFRASTRUCTURE,
the Layer I / II counterpart of tools/ref_coverage.py (same gcov build: make -C oracle ref_cov).

Runs the committed golden vectors of tests/golden/L12_MANIFEST.json (and, with --all, a cell of every layer x rate x
mode x bitrate combination) through oracle/_ref/cov/ref_harness_l12, checks the md5s on that build, and writes
profiles/<tag>_ref_coverage_l12.json: per function of the Layer I / II path the executed / total lines and branch
outcomes, and every line / outcome never executed.  tests/golden/coverage_notes_l12.json classifies the latter; the
script fails when a never-executed item is not classified, or a classified one is executed.

    python3 tools/ref_coverage_l12.py [--tag r03] [--all]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_coverage as rc  # noqa: E402
import golden_l12  # noqa: E402
from mp3common import L12_BITRATES, l12_signal, l12_spf  # noqa: E402

FUNCS = {  # reference file -> functions of the Layer I / II path (src/musicin.c:620-704 and what they call)
    "encode.c": ["encode_info", "mod", "I_combine_LR", "II_combine_LR", "I_scale_factor_calc", "II_scale_factor_calc", "pick_scale",
                 "put_scale", "II_transmission_pattern", "I_encode_scale", "II_encode_scale", "I_bits_for_nonoise", "II_bits_for_nonoise",
                 "I_main_bit_allocation", "II_main_bit_allocation", "I_a_bit_allocation", "II_a_bit_allocation",
                 "I_subband_quantization", "II_subband_quantization", "I_encode_bit_alloc", "II_encode_bit_alloc", "I_sample_encoding",
                 "II_sample_encoding", "encode_CRC", "get_audio", "read_samples", "window_subband", "filter_subband", "create_ana_filter"],
    "psy.c": ["psycho_anal"],
    "common.c": ["I_CRC_calc", "II_CRC_calc", "update_CRC", "pick_table", "read_bit_alloc", "js_bound", "hdr_to_frps"],
    "subs.c": ["fft", "enphinew"],
}


def run_one(item, tmp, env):
    name, pcm, layer, rate, kbps, mode, md5 = item
    tag = hashlib.md5(name.encode()).hexdigest()[:12]
    wd = os.path.join(tmp, tag)
    os.makedirs(wd)
    ch = 1 if mode[0] == "m" else 2
    wav, out = os.path.join(wd, "a.wav"), os.path.join(wd, "a.mpg")
    open(wav, "wb").write(rc.wav_bytes(pcm, ch, rate))
    subprocess.run([os.path.join(rc.COV, "ref_harness_l12"), wav, out, str(layer), str(rate), str(kbps), mode], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd=wd)
    got = hashlib.md5(open(out, "rb").read()).hexdigest()
    if md5 is not None and got != md5:
        raise SystemExit("coverage build disagrees with the golden md5 of %s" % name)
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--all", action="store_true")
    a = ap.parse_args()
    items = []
    for name in sorted(golden_l12.MANIFEST):
        meta, pcm, mpg, dumps = golden_l12.load(name)
        items.append(("golden:" + name, pcm, meta["layer"], meta["rate"], meta["kbps"], meta["mode"], meta["mpg_md5"]))
    if a.all:
        for layer in (1, 2):
            for rate in (44100, 48000, 32000):
                for mode in ("s", "m", "j", "d", "se", "je"):
                    for kbps in L12_BITRATES[layer]:
                        ch = 1 if mode[0] == "m" else 2
                        items.append(("matrix:%d/%d/%s/%d" % (layer, rate, mode, kbps), l12_signal(l12_spf(layer) * 6, ch, kbps + rate, rate),
                                      layer, rate, kbps, mode, None))
    run = rc.CovRun()
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(lambda it: run_one(it, run.dir, run.env), items))
    lines, branches, funcs = {}, {}, {}
    for src in ("encode.c", "psy.c", "common.c", "subs.c"):
        for base, rec in rc.gcov_json(src, run.obj).items():
            if base not in FUNCS:
                continue
            for ln in rec["lines"]:
                fn = ln.get("function_name")
                if fn not in FUNCS[base]:
                    continue
                key = (base, ln["line_number"])
                e = lines.setdefault(key, [0, fn])
                e[0] += ln["count"]
                for k, br in enumerate(ln["branches"]):
                    branches[key + (k,)] = branches.get(key + (k,), 0) + br["count"]
    run.close()
    notes_path = os.path.join(ROOT, "tests", "golden", "coverage_notes_l12.json")
    notes = json.load(open(notes_path)) if os.path.exists(notes_path) else {"lines": {}, "branches": {}}

    def note_for(table, key):
        for pat, why in table.items():
            f, rng = pat.split(":")
            lo, _, hi = rng.partition("-")
            if f == key[0] and int(lo) <= key[1] <= int(hi or lo):
                return why
        return None

    per_fn = {}
    never_l, never_b, wrongly = [], [], []
    for (f, l), (cnt, fn) in sorted(lines.items()):
        r = per_fn.setdefault(f + ":" + fn, {"lines": 0, "lines_hit": 0, "outcomes": 0, "outcomes_hit": 0})
        why = note_for(notes["lines"], (f, l))
        if why is None:
            r["lines"] += 1
            r["lines_hit"] += cnt > 0
            if cnt == 0:
                never_l.append("%s:%d" % (f, l))
        elif cnt > 0:
            wrongly.append("%s:%d classified unreachable but executed" % (f, l))
    for (f, l, k), cnt in sorted(branches.items()):
        fn = lines[(f, l)][1]
        r = per_fn[f + ":" + fn] if f + ":" + fn in per_fn else per_fn.setdefault(f + ":" + fn, {"lines": 0, "lines_hit": 0, "outcomes": 0, "outcomes_hit": 0})
        why = note_for(notes["lines"], (f, l)) or notes["branches"].get("%s:%d#%d" % (f, l, k))
        if why is None:
            r["outcomes"] += 1
            r["outcomes_hit"] += cnt > 0
            if cnt == 0:
                never_b.append("%s:%d#%d" % (f, l, k))
        elif cnt > 0 and note_for(notes["lines"], (f, l)) is None:
            wrongly.append("%s:%d#%d classified unreachable but executed" % (f, l, k))
    tot = {k: sum(r[k] for r in per_fn.values()) for k in ("lines", "lines_hit", "outcomes", "outcomes_hit")}
    out = {"what": "gcov of the unmodified reference (oracle/_ref/cov/ref_harness_l12) over %d inputs: %s" % (len(items), "golden vectors + the 504-cell matrix" if a.all else "the committed golden vectors tests/golden/l12_*"),
           "reachable_totals": tot, "per_function": per_fn, "never_executed_lines": never_l, "never_executed_outcomes": never_b,
           "classified_unreachable": {"lines": notes["lines"], "branches": notes["branches"]}}
    dest = os.path.join(ROOT, "profiles", "%s_ref_coverage_l12%s.json" % (a.tag, "_all" if a.all else ""))
    json.dump(out, open(dest, "w"), indent=1)
    print(json.dumps(tot))
    print("never executed lines:", never_l)
    print("never executed outcomes:", never_b)
    if wrongly:
        print("\n".join(wrongly))
    return 1 if (never_l or never_b or wrongly) else 0


if __name__ == "__main__":
    sys.exit(main())
