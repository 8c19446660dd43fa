#!/usr/bin/env python3
"""Which lines and branches of the REFERENCE does the parity corpus reach?  TEST INFRASTRUCTURE.

Runs only where /root/reference exists (the build container): `make -C oracle ref_cov` compiles the unmodified
reference sources with gcov instrumentation into oracle/_ref/cov/ (git-ignored, like oracle/_ref), this script runs
the whole committed corpus through that build --

  * the golden fixtures of tests/golden/MANIFEST.json (their md5s are checked on this build too),
  * one stream per cell of the 84-combination rate x channels x bitrate matrix (tools/matrix_parity.py), every
    fourth with the driver's -e / -m d,
  * streams of the four bench workloads at their full lengths (bench.py --config 1..4),
  * the edge inputs of tests/test_gpu_parity.py (silence, full-scale square, impulse, stationary tones),

-- then reads gcov's JSON and writes profiles/<tag>_ref_coverage.json: per Layer III source file and per function of
SURVEY.md section 8(a) the executed / total lines and branch outcomes, and every line / branch outcome that was never
executed.  tests/golden/coverage_notes.json classifies the never-executed ones (unreachable and why / reached by which
fixture), and the script fails when a line is neither executed nor classified.

    python3 tools/ref_coverage.py [--tag r03] [--quick]
"""
import argparse
import ctypes
import gzip
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import SEED  # noqa: E402
from golden_util import manifest, GOLD  # noqa: E402

COV = os.path.join(ROOT, "oracle", "_ref", "cov")
REFSRC = "/root/reference/src"
L3_FILES = ["l3psy.c", "subs.c", "mdct.c", "loop.c", "pow_nint.c", "reservoir.c", "l3bitstream.c", "formatBitstream.c",
            "huffman.c", "encode.c", "common.c", "l3side.h", "pow_nint.h", "huffcode.h"]
BITRATES = [32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]


def wav_bytes(pcm, ch, rate):
    data = np.ascontiguousarray(pcm, dtype="<i2").tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
            struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)


def load_synth():
    so = os.path.join(tempfile.mkdtemp(), "libsynth.so")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "mp3-enc-bsd_amd", "csrc", "pcm_synth_host.cpp")], check=True)
    lib = ctypes.CDLL(so)

    def f(n, ch, rate, stream, seed=SEED):
        out = np.zeros(n * ch, np.int16)
        lib.mp3mi_synth_pcm(ctypes.c_void_p(out.ctypes.data), ctypes.c_long(n), ch, rate, ctypes.c_uint32(stream), ctypes.c_uint32(seed))
        return out
    return f


def corpus(synth, quick):
    """yields (name, pcm int16 interleaved, rate, channels, kbps, harness mode string or None, expected md5 or None);
    the mode string is the driver's -m letter plus e / c / o for its -e, -c, -o options (oracle/ref_harness.c)"""
    for c in manifest():
        if "pcm_file" in c:
            pcm = np.load(os.path.join(GOLD, c["pcm_file"]))
        else:
            pcm = synth(c["n_samples_per_ch"], c["channels"], c["rate"], c["stream"])
        if c.get("reference_aborts"):
            continue
        yield ("golden:" + c["name"], pcm, c["rate"], c["channels"], c["kbps"], c.get("mode"), c["mp3_md5"])
    combo = 0
    for rate in (44100, 48000, 32000):
        for ch in (2, 1):
            for kbps in BITRATES:
                combo += 1
                pcm = synth(24 * 1152, ch, rate, 1000 * ((combo - 1) // 14) + kbps)
                yield ("matrix:%d/%d/%d" % (rate, ch, kbps), pcm, rate, ch, kbps, None, None)
                if combo % 4 == 0:
                    yield ("matrix:%d/%d/%d -e%s" % (rate, ch, kbps, " -m d" if ch == 2 else ""), pcm, rate, ch, kbps,
                           "de" if ch == 2 else "me", None)
    n = 2 if quick else 8
    for s in range(n):  # bench.py --config 1 / 2
        yield ("bench1:stream%d" % s, synth(383 * 1152, 2, 44100, s * 511), 44100, 2, 128, None, None)
    for s in range(6 if quick else 12):  # --config 3
        yield ("bench3:stream%d" % s, synth(417 * 1152, 2, 48000, s), 48000, 2, [64, 96, 128, 192, 256, 320][s % 6], None, None)
    for s in range(n):  # --config 4
        yield ("bench4:stream%d" % s, synth(278 * 1152, 1, 32000, s * 2047), 32000, 1, 64, None, None)
    # tests/test_gpu_parity.py: edge inputs
    nf, ch = 8, 2
    sil = np.zeros(nf * 1152 * ch, np.int16)
    sq = np.where((np.arange(nf * 1152 * ch) // 200) % 2 == 0, 32767, -32768).astype(np.int16)
    imp = np.zeros(nf * 1152 * ch, np.int16)
    imp[5000] = 30000
    for nm, p in (("silence", sil), ("square", sq), ("impulse", imp)):
        yield ("edge:" + nm, p, 44100, 2, 128, None, None)
    nf, rate, S = 40, 44100, 48
    t = np.arange(nf * 1152, dtype=np.float64) / rate
    rng = np.random.default_rng(5)
    for s in range(0, S, 4 if quick else 1):
        f1, f2 = 110.0 * 2 ** (s / 8.0), 997.0 + 371.0 * s
        a = [3000.0, 12000.0, 30000.0][s % 3]
        left = a * np.sin(2 * np.pi * f1 * t)
        right = a * np.sin(2 * np.pi * f1 * t + 0.5) if s % 2 else 0.5 * a * (np.sin(2 * np.pi * f1 * t) + np.sin(2 * np.pi * f2 * t))
        if s % 4 == 3:
            left = left + rng.normal(0.0, 2.0, left.shape)
        p = np.zeros((nf * 1152, 2), np.int16)
        p[:, 0] = np.clip(np.rint(left), -32768, 32767)
        p[:, 1] = np.clip(np.rint(right), -32768, 32767)
        yield ("edge:tonal%d" % s, p.reshape(-1), rate, 2, 128, None, None)


class CovRun:
    """One isolated set of gcov counters (GCOV_PREFIX redirects the .gcda files of the instrumented binaries)."""

    def __init__(self):
        self.dir = tempfile.mkdtemp(prefix="refcov_")
        self.obj = os.path.join(self.dir, "obj")
        os.makedirs(self.obj)
        for fn in os.listdir(os.path.join(COV, "obj")):
            if fn.endswith(".gcno"):
                os.symlink(os.path.join(COV, "obj", fn), os.path.join(self.obj, fn))
        strip = len(os.path.join(COV, "obj").strip("/").split("/"))
        self.env = dict(os.environ, GCOV_PREFIX=self.obj, GCOV_PREFIX_STRIP=str(strip))

    def run(self, items, workers=8):
        with ThreadPoolExecutor(max_workers=workers) as ex:
            return list(ex.map(lambda it: run_one(it, self.dir, self.env), items))

    def collect(self):
        """(lines {(file, line): [count, function]}, branch outcomes {(file, line, k): count}, functions {name: record});
        a header's counts are summed over the translation units that include it"""
        lines, branches, funcs = {}, {}, {}
        for src in SOURCES:
            for base, rec in gcov_json(src, self.obj).items():
                if base not in L3_FILES:
                    continue
                for fn in rec["functions"]:
                    f = funcs.setdefault((base, fn["name"]), {"file": base, "start": fn["start_line"], "end": fn["end_line"], "calls": 0})
                    f["calls"] += fn["execution_count"]
                for ln in rec["lines"]:
                    key = (base, ln["line_number"])
                    e = lines.setdefault(key, [0, ln.get("function_name")])
                    e[0] += ln["count"]
                    for k, br in enumerate(ln["branches"]):
                        branches[key + (k,)] = branches.get(key + (k,), 0) + br["count"]
        return lines, branches, funcs

    def close(self):
        shutil.rmtree(self.dir, ignore_errors=True)


SOURCES = ["l3psy.c", "subs.c", "mdct.c", "loop.c", "pow_nint.c", "reservoir.c", "l3bitstream.c", "formatBitstream.c", "encode.c", "common.c"]


def run_one(item, tmp, env):
    name, pcm, rate, ch, kbps, opts, md5 = item
    tag = hashlib.md5(name.encode()).hexdigest()[:12]
    wav, mp3f = os.path.join(tmp, tag + ".wav"), os.path.join(tmp, tag + ".mp3")
    with open(wav, "wb") as f:
        f.write(wav_bytes(pcm, ch, rate))
    cmd = [os.path.join(COV, "ref_harness"), wav, mp3f, str(rate), str(kbps), opts or ("m" if ch == 1 else "s")]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    data = open(mp3f, "rb").read()
    os.remove(wav)
    os.remove(mp3f)
    got = hashlib.md5(data).hexdigest()
    if md5 is not None and got != md5:
        raise SystemExit("coverage build disagrees with the golden md5 of %s" % name)
    return name, len(pcm) // ch // 1152


def gcov_json(src, obj):
    """gcov's JSON record of one translation unit of the coverage build (headers it includes come along)"""
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["gcov", "--json-format", "-b", "-c", "-o", obj, os.path.join(REFSRC, src)],
                       cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for fn in os.listdir(td):
            if fn.endswith(".gcov.json.gz"):
                j = json.load(gzip.open(os.path.join(td, fn)))
                for f in j["files"]:
                    out[os.path.basename(f["file"])] = f
        return out


# row -> functions
ROWS = {
    "a1 L3psycho_anal": ["L3psycho_anal"],
    "a2 L3para_read": ["L3para_read"],
    "a3 sprdngf/s3ind": ["sprdngf1", "sprdngf2", "s3ind"],
    "a4 fft": ["fft", "rsfft", "rsrec", "srrec", "BR_permute", "enphinew"],
    "a5 window_subband": ["window_subband", "read_ana_window"],
    "a6 filter_subband": ["filter_subband", "create_ana_filter"],
    "a7 mdct_sub": ["mdct_sub"],
    "a8 mdct": ["mdct"],
    "a9 iteration_loop": ["iteration_loop"],
    "a10 calc_xmin/scfsi/gr_deco/xr_max": ["calc_xmin", "calc_scfsi", "gr_deco", "xr_max"],
    "a11 quantanf_init": ["quantanf_init"],
    "a12 outer/bin_search/count_bits/inner": ["outer_loop", "bin_search_StepSize", "count_bits", "inner_loop"],
    "a13 quantize+pow_nint": ["quantize", "pow_nint", "init_pow_nint"],
    "a14 huffman counting": ["calc_runlen", "count1_bitcount", "subdivide", "bigv_tab_select", "new_choose_table", "choose_table",
                             "bigv_bitcount", "count_bit", "ix_max"],
    "a15 noise/amp/scale": ["calc_noise", "preemphasis", "amp_scalefac_bands", "loop_break", "scale_bitcount", "part2_length"],
    "a16 reservoir": ["ResvFrameBegin", "ResvMaxBits", "ResvAdjust", "ResvFrameEnd"],
    "a17 III_format_bitstream": ["III_format_bitstream", "encodeSideInfo", "encodeMainData", "Huffmancodebits",
                                 "L3_huffman_coder_count1", "HuffmanCode", "III_FlushBitstream", "abs_and_sign", "drain_into_ancillary_data",
                                 "putMyBits"],
    "a18 BF_BitstreamFrame": ["BF_BitstreamFrame", "WriteMainDataBits", "write_side_info", "store_side_info", "BF_FlushBitstream",
                              "main_data", "writePartMainData", "writePartSideInfo", "side_queue_elements", "free_side_queues",
                              "free_side_info", "get_side_info", "BF_PartLength", "BF_newPartHolder",
                              "BF_LoadHolderFromBitstreamPart", "BF_resizePartHolder", "BF_addElement", "BF_addEntry", "BF_freePartHolder"],
    "a19 putbits": ["putbits", "empty_buffer", "close_bit_stream_w"],
}
# The L3psycho_anal body holds the Layer I/II model as well (src/l3psy.c:284-437, `case 1: case 2:`): not Layer III code
NOT_L3_LINES = {("l3psy.c", l) for l in range(284, 440)}


def golden_items(synth, aborting=False):
    """the committed fixtures the reference encodes (aborting=False) / dies on (aborting=True)"""
    for c in manifest():
        if bool(c.get("reference_aborts")) != aborting:
            continue
        if "pcm_file" in c:
            pcm = np.load(os.path.join(GOLD, c["pcm_file"]))
        else:
            pcm = synth(c["n_samples_per_ch"], c["channels"], c["rate"], c["stream"])
        yield ("golden:" + c["name"], pcm, c["rate"], c["channels"], c["kbps"], c.get("mode"),
               c["reference_aborts"] if aborting else c["mp3_md5"])


def check_aborts(synth, notes):
    """Runs the fixtures on which the reference dies: gcov never sees those runs (abort() skips the counter dump),
    so the assertion message is the evidence.  Returns the outcomes (file:line#k) reached that way."""
    seen = []
    tmp = tempfile.mkdtemp(prefix="refcov_abort_")
    for name, pcm, rate, ch, kbps, opts, ab in golden_items(synth, aborting=True):
        wav = os.path.join(tmp, "a.wav")
        open(wav, "wb").write(wav_bytes(pcm, ch, rate))
        r = subprocess.run([os.path.join(COV, "ref_harness"), wav, os.path.join(tmp, "a.mp3"), str(rate), str(kbps), "m" if ch == 1 else "s"],
                           capture_output=True)
        msg = r.stderr.decode(errors="replace")
        where = ab["where"]  # "loop.c:358"
        if r.returncode == 0 or ("/" + where + ":") not in msg:
            raise SystemExit("%s: the reference was expected to die at %s, got rc %d: %s" % (name, where, r.returncode, msg[-200:]))
        keys = [k for k in notes.get("reference_aborts", {}) if k.split("#")[0] == where]
        if not keys:
            raise SystemExit("%s: %s is not listed under reference_aborts in coverage_notes.json" % (name, where))
        seen += keys
        print("reference dies as recorded: %s at %s" % (name, where))
    shutil.rmtree(tmp, ignore_errors=True)
    return seen


class Notes:
    """tests/golden/coverage_notes.json: which never-executed lines / outcomes no input can reach, and why"""

    def __init__(self, notes):
        self.lines, self.outcomes = {}, {}
        for r in notes.get("rules", []):
            for spec in r.get("lines", []):
                f, rng = spec.split(":")
                a, b = (rng.split("-") + [rng])[:2]
                for l in range(int(a), int(b) + 1):
                    self.lines[(f, l)] = r["why"]
            for o in r.get("outcomes", []):
                self.outcomes[o] = r["why"]
        self.aborts = notes.get("reference_aborts", {})
        self.used = set()

    def line(self, f, l):
        if (f, l) in self.lines:
            self.used.add((f, l))
            return self.lines[(f, l)]
        return None

    def outcome(self, f, l, k):
        key = "%s:%d#%d" % (f, l, k)
        if key in self.aborts:
            # reachable: the reference dies there (checked separately, gcov cannot see it) -- unless no input was found
            return None if self.aborts[key].get("fixture") else "not demonstrated: " + self.aborts[key]["why"]
        if key in self.outcomes:
            self.used.add(key)
            return self.outcomes[key]
        if "assert(" in src_line(f, l).replace(" ", ""):
            return "assertion holds"
        return None


def summarise(lines, branches, funcs, notes, aborts_seen=()):
    """rows of section 8(a) with executed / total lines and branch outcomes; the never-executed ones split into
    classified-unreachable and open.  aborts_seen: outcomes on which an abort fixture made the reference die."""
    N = notes if isinstance(notes, Notes) else Notes(notes)
    fn_row = {fn: row for row, fns in ROWS.items() for fn in fns}
    rows = {row: {"lines": 0, "lines_hit": 0, "lines_unreachable": 0, "branch_outcomes": 0, "branch_outcomes_hit": 0,
                  "branch_outcomes_unreachable": 0, "functions": {}} for row in ROWS}
    open_items = []
    for (f, l), (cnt, fn) in sorted(lines.items()):
        row = fn_row.get(fn)
        if row is None or (f, fn) not in funcs or (f, l) in NOT_L3_LINES:
            continue
        r = rows[row]
        ff = r["functions"].setdefault(fn if f.endswith(".c") and fn != "HuffmanCode" else "%s (%s)" % (fn, f),
                                       {"file": f, "calls": funcs[(f, fn)]["calls"], "never_executed": []})
        r["lines"] += 1
        key = "%s:%d" % (f, l)
        line_why = None
        if cnt > 0:
            r["lines_hit"] += 1
        else:
            line_why = N.line(f, l)
            ff["never_executed"].append(key)
            if line_why:
                r["lines_unreachable"] += 1
            else:
                open_items.append(key)
        k = 0
        while (f, l, k) in branches:
            r["branch_outcomes"] += 1
            bkey = "%s:%d#%d" % (f, l, k)
            if branches[(f, l, k)] > 0 or bkey in aborts_seen:
                r["branch_outcomes_hit"] += 1
            elif cnt > 0:
                ff["never_executed"].append(bkey)
                if N.outcome(f, l, k):
                    r["branch_outcomes_unreachable"] += 1
                else:
                    open_items.append(bkey)
            elif line_why:  # an outcome on a line that never ran: accounted with the line
                r["branch_outcomes_unreachable"] += 1
            k += 1
    return rows, open_items


def src_line(f, l, cache={}):
    if f not in cache:
        cache[f] = open(os.path.join(REFSRC, f), errors="replace").read().split("\n")
    return cache[f][l - 1].strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--corpus", default="golden", choices=["golden", "all"],
                    help="golden: the committed fixtures only (what tests/test_gpu_golden.py reproduces); all: plus matrix, bench streams, edge inputs")
    ap.add_argument("--quick", action="store_true", help="a thinner 'all' corpus (development)")
    ap.add_argument("--probe", nargs=4, metavar=("PCM.npy", "RATE", "CH", "KBPS"),
                    help="run one more input on top of the corpus and print which never-executed lines / outcomes it reaches")
    args = ap.parse_args()
    if not os.path.isdir(REFSRC):
        raise SystemExit("the reference sources are not here: coverage is measured in the build container only")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_cov"], check=True)
    synth = load_synth()
    items = list(golden_items(synth)) if args.corpus == "golden" else list(corpus(synth, args.quick))
    notes_path = os.path.join(GOLD, "coverage_notes.json")
    notes = json.load(open(notes_path)) if os.path.exists(notes_path) else {}
    aborts_seen = check_aborts(synth, notes)
    run = CovRun()
    done = run.run(items)
    lines, branches, funcs = run.collect()
    N = Notes(notes)
    rows, open_items = summarise(lines, branches, funcs, N, aborts_seen)
    if args.probe:
        pcm = np.load(args.probe[0]).reshape(-1)
        run.run([("probe", pcm, int(args.probe[1]), int(args.probe[2]), int(args.probe[3]), None, None)])
        l2, b2, f2 = run.collect()
        _, open2 = summarise(l2, b2, f2, notes, aborts_seen)
        for it in sorted(set(open_items) - set(open2)):
            print("probe reaches", it, "|", src_line(it.split(":")[0], int(it.split(":")[1].split("#")[0])))
        run.close()
        return
    run.close()
    frames = sum(n for _, n in done)
    tot = {k: sum(r[k] for r in rows.values()) for k in ("lines", "lines_hit", "lines_unreachable", "branch_outcomes", "branch_outcomes_hit",
                                                            "branch_outcomes_unreachable")}
    # a rule that names something the corpus DID execute is wrong
    stale = [("%s:%d" % k) for k in N.lines if lines.get(k, [0])[0] > 0]
    stale += [k for k in N.outcomes if branches.get((k.split(":")[0], int(k.split(":")[1].split("#")[0]), int(k.split("#")[1])), 0) > 0]
    rec = {"what": "gcov line / branch-outcome coverage of the unmodified reference (oracle/_ref/cov: the reference sources, -O0 --coverage) "
                   "over " + ("the committed golden fixtures (tests/golden/MANIFEST.json), each of which tests/test_gpu_golden.py "
                              "reproduces on the GPU" if args.corpus == "golden" else "the whole parity corpus (fixtures, matrix cells, bench streams, edge inputs)"),
           "corpus_inputs": len(items), "corpus_frames": frames, "corpus": [n for n, _ in done], "totals_section_8a": tot,
           "reachable_and_never_executed": open_items, "classified_unreachable_but_executed": stale,
           "reached_only_by_killing_the_reference": aborts_seen,
           "classification": "tests/golden/coverage_notes.json", "rows": rows}
    out = os.path.join(ROOT, "profiles", "%s_ref_coverage%s.json" % (args.tag, "" if args.corpus == "golden" else "_all"))
    json.dump(rec, open(out, "w"), indent=1)
    print("inputs %d, frames %d" % (len(items), frames))
    print("section 8(a): lines %d / %d (+%d unreachable), branch outcomes %d / %d (+%d unreachable)" % (
        tot["lines_hit"], tot["lines"], tot["lines_unreachable"], tot["branch_outcomes_hit"], tot["branch_outcomes"], tot["branch_outcomes_unreachable"]))
    for row, r in rows.items():
        print("  %-40s lines %4d/%4d (+%d)  outcomes %4d/%4d (+%d)" % (row, r["lines_hit"], r["lines"], r["lines_unreachable"],
                                                                      r["branch_outcomes_hit"], r["branch_outcomes"], r["branch_outcomes_unreachable"]))
    print("reachable (not classified unreachable) and never executed: %d" % len(open_items))
    for u in open_items:
        print("   ", u, "|", src_line(u.split(":")[0], int(u.split(":")[1].split("#")[0]))[:100])
    print("wrote", out)
    if stale:
        print("CLASSIFIED UNREACHABLE BUT EXECUTED:", stale)
    sys.exit(1 if open_items or stale else 0)


if __name__ == "__main__":
    main()
