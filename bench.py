#!/usr/bin/env python3
"""Headline benchmark: stereo 44.1 kHz Layer III frames/s at 128 kbps (bit-exact), MI355X.

One "step" = one pass of the whole hot path (psy FFTs, thresholds, filterbank+MDCT, iteration
loop, bitstream formatting) over one batch of synthetic PCM that is already resident in HBM.
The workloads are BASELINE.json's configs, selected with --config:

  1  4096 streams x 383 frames, 44.1 kHz stereo, 128 kbps            (configs[1]; the default, the metric's config)
  2  8192 streams x 383 frames per GPU, 44.1 kHz stereo, 128 kbps    (configs[2]: 65 536 streams on 8 GPUs, plain
                                                                      stereo -- the reference refuses joint stereo)
  3  4096 streams x 417 frames, 48 kHz stereo, stream s at {64,96,128,192,256,320}[s mod 6] kbps   (configs[3])
  4  16384 streams x 278 frames, 32 kHz mono, 64 kbps                (configs[4])

With N GPUs every rank owns its own streams (no collective on the data path: streams are independent), so
scaling is weak and `value` is the whole-job frames/s.  The PCM is the deterministic generator of
csrc/pcm_synth_core.h run on the device (md5-pinned by tests/test_synth.py).  A sample of the timed batch's
streams is compared byte for byte with the CPU oracle and with the unmodified reference binary; a mismatch
makes the run FAIL (exit 1, value null).  tools/full_parity.py compares every stream.

--layer 1|2 runs the same contract over the Layer I / Layer II path (include/mp3mi_l12.h) at the
reference's default bitrate for the layer (src/musicin.c:371-372: 288 / 160 kbps): 4096 streams x 10 s of 44.1 kHz stereo.
The default (--layer 3) is the metric of BASELINE.json.

The default line (N = 1, configs[1], Layer III) also carries `end_to_end` -- the same K steps with PCM and bytes in page-locked
host memory, crossing PCIe beside the kernels; `value` stays the resident rate -- and `other_workloads`:
two steps each of configs[2], configs[3], configs[4] and of the Layer II batch and one Layer I step, each with its own oracle
spot check.  `collective_backend` says what carried the ranks' barrier / maximum / votes (RCCL, or gloo where RCCL did not come up or
MP3MI_BENCH_BACKEND=gloo asked for it); `roofline.clock_ghz_measured` is the engine clock the driver reported during the timed region.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, one per GPU, and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

torch = None  # loaded by the processes that run a rank (load_torch): the launcher of `--gpus N` never imports it


def load_torch():
    global torch
    if torch is None:
        import torch as _t
        torch = _t
    return torch


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Coll:
    """The ranks' control traffic: a barrier, a maximum, a vote, a gather -- eight bytes each; the data path has no collective
    (streams are independent: the reference's loop is one stream in one process, /root/reference/src/musicin.c:585).
    RCCL (torch.distributed backend "nccl") carries it where it comes up; a rank whose RCCL does not -- or
    MP3MI_BENCH_BACKEND=gloo, or the one-GPU rehearsal -- falls back to gloo with CPU tensors FOR ALL RANKS: the ranks first
    meet on gloo (which needs no GPU), try RCCL on a group of its own, and vote.  `backend` says what carried the run."""

    def __init__(self, world, dev, want_nccl):
        self.world, self.dev, self.dist, self.group, self.backend = world, dev, None, None, "none (one rank)"
        if world <= 1:
            return
        import datetime
        import torch.distributed as dist
        self.dist = dist
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
        self.backend = "gloo (asked for)"
        if not want_nccl:
            return
        ok, why = 1, ""
        try:
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            t = torch.ones(1, dtype=torch.int32, device=dev)
            dist.all_reduce(t, group=g)
            torch.cuda.synchronize()
            if int(t.item()) != world:
                ok, why = 0, "a sum over %d ranks came back as %d" % (world, int(t.item()))
        except Exception as e:  # noqa: BLE001 -- whatever RCCL raises here, the run goes on over gloo
            ok, why, g = 0, str(e).splitlines()[0][:200], None
        v = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        if int(v.item()) == 1:
            self.group, self.backend = g, "nccl (RCCL)"
        else:
            self.backend = "gloo (RCCL did not come up on every rank%s)" % ((": " + why) if why else "")

    def _t(self, values, dtype):
        return torch.tensor(values, dtype=dtype, device=self.dev if self.group is not None else "cpu")

    def barrier(self):
        if self.dist is not None:
            self.dist.all_reduce(self._t([0], torch.int32), group=self.group)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def reduce(self, value, op, dtype=None):
        if self.dist is None:
            return value
        dtype = dtype or (torch.float64 if isinstance(value, float) else torch.int64)
        t = self._t([value], dtype)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op], group=self.group)
        return t.item()

    def gather(self, values):
        """every rank's list of int64 values, by rank"""
        mine = self._t(values, torch.int64)
        if self.dist is None:
            return [mine.tolist()]
        per = [mine.clone() for _ in range(self.world)]
        self.dist.all_gather(per, mine, group=self.group)
        return [t.tolist() for t in per]

    def close(self):
        if self.dist is not None:
            self.barrier()
            self.dist.destroy_process_group()


def gpu_numa_cpus(dev_index):
    """(cpus of the NUMA node the GPU `dev_index` hangs on, description) from the KFD topology and sysfs -- no GPU call, no torch.
    The ranks of a node then pin themselves to the cores next to their GPU: page-locked buffers (7.9 GB a rank on the end_to_end
    leg) are allocated there, and eight ranks do not pile onto one socket.  None when the topology does not say."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for n in sorted(os.listdir(base), key=int):
            props = dict(l.split() for l in open(os.path.join(base, n, "properties")) if len(l.split()) == 2)
            if int(props.get("simd_count", "0")) > 0 and int(props.get("vendor_id", "0")) == 0x1002:
                gpus.append(props)
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis:
            gpus = [gpus[int(i)] for i in vis.split(",") if i.strip().isdigit() and int(i) < len(gpus)]
        pr = gpus[dev_index]
        loc, dom = int(pr["location_id"]), int(pr.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None, "GPU %s: no NUMA node recorded" % bdf
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        return (cpus or None), "GPU %s on NUMA node %d" % (bdf, node)
    except Exception as e:  # noqa: BLE001
        kfd_why = type(e).__name__
    # The KFD topology is not readable for everyone (ordinary users on some hosts).  Second source: the AMD accelerators / display
    # controllers on the PCI bus, in address order -- the order the HIP runtime lists them in unless *_VISIBLE_DEVICES says otherwise.
    try:
        devs = []
        for d in sorted(os.listdir("/sys/bus/pci/devices")):
            b = "/sys/bus/pci/devices/" + d
            if open(b + "/vendor").read().strip() != "0x1002":
                continue
            cls = int(open(b + "/class").read().strip(), 16) >> 8
            if cls in (0x0300, 0x0302, 0x0380, 0x1200) and os.path.exists(b + "/numa_node") and os.path.isdir(b + "/drm"):
                devs.append(d)
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis:
            devs = [devs[int(i)] for i in vis.split(",") if i.strip().isdigit() and int(i) < len(devs)]
        bdf = devs[dev_index]
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None, "GPU %s: no NUMA node recorded" % bdf
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b2 = part.partition("-")
            cpus.update(range(int(a), int(b2 or a) + 1))
        cpus &= os.sched_getaffinity(0)
        return (cpus or None), "GPU %s (PCI order) on NUMA node %d" % (bdf, node)
    except Exception as e:  # noqa: BLE001
        return None, "topology not readable (kfd: %s, pci: %s)" % (kfd_why, type(e).__name__)


def pin_to_gpu_numa(dev_index, local_rank, local_world):
    """Before torch is imported: this rank's threads onto the cores next to its GPU; ranks that share a NUMA node share its
    cores (not split: the CPU baseline of rank 0 widens its own mask again).  Returns what was done, for the bench line."""
    if os.environ.get("MP3MI_BENCH_NO_PIN") == "1":
        return "unchanged (MP3MI_BENCH_NO_PIN=1)"
    cpus, what = gpu_numa_cpus(dev_index)
    if not cpus:
        return "unchanged (%s)" % what
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return "unchanged (sched_setaffinity: %s)" % e
    return "%d cpus of %s" % (len(cpus), what)


class ClockSampler:
    """The engine clock while the timed region runs, read from the driver's sysfs table (pp_dpm_sclk: the starred line) every
    20 ms by a side thread of this process.  `effective_clock_ghz` of the issue roofline is DERIVED (wave cycles over time, pulled
    down by the loop kernel's tail); this is the clock the hardware reports.  None where the file is not there or not readable."""

    def __init__(self, dev_index):
        import glob
        self.samples, self.stop_flag, self.thread, self.path = [], False, None, None
        try:
            want = None
            pr = torch.cuda.get_device_properties(dev_index)
            if hasattr(pr, "pci_bus_id"):
                want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
            for c in cards:
                if want and want in os.path.realpath(os.path.dirname(c)):
                    self.path = c
            if self.path is None and len(cards) == 1:
                self.path = cards[0]
        except Exception:  # noqa: BLE001
            self.path = None

    def _run(self):
        import re
        while not self.stop_flag:
            try:
                for line in open(self.path):
                    if "*" in line:
                        m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                        if m:
                            self.samples.append(int(m.group(1)))
            except OSError:
                return
            time.sleep(0.02)

    def start(self):
        if self.path is not None:
            import threading
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()
        return self

    def stop(self):
        self.stop_flag = True
        if self.thread is not None:
            self.thread.join(timeout=1)
        if not self.samples:
            return None
        a = np.array(self.samples, dtype=np.float64)
        return {"mean": round(float(a.mean()) / 1e3, 3), "min": round(float(a.min()) / 1e3, 3), "max": round(float(a.max()) / 1e3, 3), "samples": int(a.size),
                "source": self.path}

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured copy
SEED = 0x6D70336D
MIX48 = [64, 96, 128, 192, 256, 320]

CONFIGS = {
    1: dict(streams=4096, frames=383, rate=44100, channels=2, kbps=128, name="BASELINE configs[1]"),
    2: dict(streams=8192, frames=383, rate=44100, channels=2, kbps=128, name="BASELINE configs[2] (8192 streams per GPU, plain stereo)"),
    3: dict(streams=4096, frames=417, rate=48000, channels=2, kbps="mix48", name="BASELINE configs[3] (stream s at {64..320}[s mod 6] kbps)"),
    4: dict(streams=16384, frames=278, rate=32000, channels=1, kbps=64, name="BASELINE configs[4]"),
}


def kbps_of(cfg, stream):
    return MIX48[stream % 6] if cfg["kbps"] == "mix48" else cfg["kbps"]


class Workload:
    """A config's batch on one GPU: encoder, deterministic PCM in HBM, output buffers."""

    def __init__(self, mp3, cfg, dev, stream0):
        load_torch()  # (tools/full_parity.py builds a Workload without going through main())
        self.cfg, self.dev, self.stream0 = cfg, dev, stream0
        S, nf, C, rate = cfg["streams"], cfg["frames"], cfg["channels"], cfg["rate"]
        self.kbps = [kbps_of(cfg, stream0 + s) for s in range(S)]
        uniform = len(set(self.kbps)) == 1
        self.batch = mp3.Batch(S, rate, C, self.kbps[0] if uniform else self.kbps, nf)
        self.pcm = torch.empty((S, nf * 1152 * C), dtype=torch.int16, device=dev)
        mp3.synth_pcm_device(self.pcm, nf * 1152, C, rate, stream0=stream0, seed=SEED)
        self.out = torch.zeros((S, self.batch.out_stride(nf)), dtype=torch.uint8, device=dev)
        self.out_len = torch.zeros(S, dtype=torch.int32, device=dev)
        self.frame_bytes = [mp3.frame_bytes(rate, k) for k in self.kbps]
        torch.cuda.synchronize()

    def encode(self):
        self.batch.encode(self.pcm, self.cfg["frames"], self.out, self.out_len)

    def step(self):
        self.encode()
        self.batch.sync()

    def alg_bytes_per_frame(self):
        """PCM in + bitstream out, averaged over the batch's bitrates"""
        return 1152 * self.cfg["channels"] * 2 + float(np.mean(self.frame_bytes))

    def close(self):
        self.batch.close()


# What bounds each kernel (measured stand-alone durations against bytes moved and instructions
# issued).  A label, not a measurement: the measurements are in the profile the bench line names.
KERNEL_BOUND = {
    "k_loop": "valu+salu issue (4 wavefronts per SIMD, serial per stream)",
    "k_fft": "lds pipe and valu issue, about even (the butterfly program of the blocks of 16 to 128 points; larger and smaller ones run in registers)", "k_cw": "valu issue (f64)", "k_cw_fix": "valu issue (f64)",
    "k_part": "hbm (one lane per record, 2 KB rows)", "k_psy": "latency (one wavefront per track, serial over granules)",
    "k_filter": "valu issue (f64) + hbm", "k_mdct": "hbm + valu issue (the loop's stateless head in its tail)", "k_prep": "idle (the records k_mdct lists: none)", "k_format": "latency (bit scatter)",
}


def current_profile(source_hash, streams, frames):
    """The committed counter profile this build may quote: profiles/CURRENT names it (written when a profile is
    committed: tools/gpu_round_profile.sh, tools/make_profile_json.py); it is used only if it was taken on the same
    sources (mp3mi_source_hash) and the same workload.  Returns (profile dict or None, reason)."""
    ptr = os.path.join(ROOT, "profiles", "CURRENT")
    if not os.path.exists(ptr):
        return None, "no profiles/CURRENT"
    name = open(ptr).read().strip()
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception as e:
        return None, "profiles/%s unreadable: %s" % (name, e)
    if d.get("source_hash") != source_hash:
        return None, "profiles/%s was taken on sources %s, this library is %s: counters not quoted" % (name, d.get("source_hash"), source_hash)
    if d.get("streams") != streams or d.get("frames") != frames:
        return None, "profiles/%s is of %s x %s, this run of %d x %d" % (name, d.get("streams"), d.get("frames"), streams, frames)
    d["file"] = name
    return d, None


def current_profile_l12(layer, source_hash, streams, frames):
    """The same for `--layer N`: profiles/CURRENT_LAYER<N> names a profile of tools/gpu_l12_profile.sh."""
    ptr = os.path.join(ROOT, "profiles", "CURRENT_LAYER%d" % layer)
    if not os.path.exists(ptr):
        return None, "no profiles/CURRENT_LAYER%d" % layer
    name = open(ptr).read().strip()
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception as e:
        return None, "profiles/%s unreadable: %s" % (name, e)
    if d.get("source_hash") != source_hash:
        return None, "profiles/%s was taken on sources %s, this library is %s: counters not quoted" % (name, d.get("source_hash"), source_hash)
    if d.get("streams") != streams or d.get("frames") != frames:
        return None, "profiles/%s is of %s x %s, this run of %d x %d" % (name, d.get("streams"), d.get("frames"), streams, frames)
    d["file"] = name
    return d, None


def issue_roofline(k, streams, kernel_s_per_launch):
    """The bound the kernel actually runs against: VALU issue.  From the SQ counter passes of the profile (per-wave
    quad-cycles, MI355X_MICROARCH.md):
      achieved = SQ_ACTIVE_INST_VALU  -- quad-cycles in which a wavefront of the kernel was issuing a vector instruction
      peak     = SQ_WAVE_CYCLES / waves resident per SIMD -- the quad-cycles the SIMDs were held by the kernel (a SIMD's
                 vector port serves one of its resident wavefronts at a time)
    so frac is the share of the SIMDs' vector issue capacity the launch used; what is left is waitcnt / issue stalls.
    The instruction mix per launch goes along (what there is to remove: the kernel gets faster by executing fewer
    instructions, not by moving fewer bytes)."""
    if not k or "SQ_ACTIVE_INST_VALU" not in k:
        return None
    n = float(k["dispatches"])
    waves_per_simd = max(1.0, min(4.0, streams / 1024.0))
    achieved, peak = k["SQ_ACTIVE_INST_VALU"] / n, k["SQ_WAVE_CYCLES"] / n / waves_per_simd
    f64 = sum(k.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")) / n
    f32 = sum(k.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32")) / n
    vmem = (k.get("SQ_INSTS_VMEM_RD", 0) + k.get("SQ_INSTS_VMEM_WR", 0)) / n
    return {"bound": "valu-issue", "achieved": int(achieved), "peak": int(peak), "unit": "SIMD quad-cycles per launch", "frac": round(achieved / peak, 4),
            "waves_per_simd": waves_per_simd,
            "wave_time_split": {c: round(k[c] / k["SQ_WAVE_CYCLES"], 4) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if c in k},
            "instructions_per_launch": {"valu": int(k["SQ_INSTS_VALU"] / n), "valu_f64": int(f64), "valu_f32": int(f32),
                                        "valu_int32": int(k.get("SQ_INSTS_VALU_INT32", 0) / n), "salu": int(k["SQ_INSTS_SALU"] / n),
                                        "branch": int(k.get("SQ_INSTS_BRANCH", 0) / n), "lds": int(k["SQ_INSTS_LDS"] / n), "vmem": int(vmem)},
            "effective_clock_ghz": round(4.0 * peak / 1024.0 / kernel_s_per_launch / 1e9, 3) if kernel_s_per_launch > 0 else None}


def cpu_baseline(pcm_sample, rate, kbps_list, channels, cores):
    """Oracle (CPU restatement of the reference) on a bounded sample of the same workload."""
    from mp3common import Oracle
    orc = Oracle()
    orc.encode(pcm_sample[0][: 1152 * channels * 8], rate, kbps_list[0], channels)  # warm up tables/page cache
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        outs = list(ex.map(lambda a: orc.encode(a[0], rate, a[1], channels)[0], zip(pcm_sample, kbps_list)))
    dt = time.perf_counter() - t0
    frames = sum(len(p) // (1152 * channels) for p in pcm_sample)
    return frames / dt, outs


def without_private_bit(data, frame_bytes):
    """The header's private bit is an uninitialised automatic of the reference's main(): not a
    function of the input.  0 in every run observed; masked anyway before the reference BINARY's bytes are compared."""
    b = bytearray(data)
    for p in range(0, len(b) - 3, frame_bytes):
        if b[p] == 0xff and (b[p + 1] & 0xf0) == 0xf0:
            b[p + 2] &= 0xfe
    return bytes(b)


def reference_baseline(pcm_sample, rate, kbps_list, channels, cores, layer=3):
    """The UNMODIFIED reference encoder (oracle/_ref/encode, compiled from /root/reference/src by
    oracle/Makefile where the sources exist; the binary travels with the repository) on the same
    sample, one process per stream.  Returns (frames/s, outputs) or None when the binary is absent."""
    import shutil
    import struct
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "encode")
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="mp3ref_")
    for k, p in enumerate(pcm_sample):
        data = np.ascontiguousarray(p, dtype="<i2").tobytes()
        with open(os.path.join(tmp, "%d.wav" % k), "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                    struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * 2, channels * 2, 16) + b"data" +
                    struct.pack("<I", len(data)) + data)

    def run(k):
        args = [exe, "-l", str(layer), "-s", "%g" % (rate / 1000.0), "-b", str(kbps_list[k])] + (["-m", "m"] if channels == 1 else [])
        subprocess.run(args + [os.path.join(tmp, "%d.wav" % k), os.path.join(tmp, "%d.mp3" % k)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)  # (psycho_anal writes "out.dat" where it runs)
        return open(os.path.join(tmp, "%d.mp3" % k), "rb").read()

    run(0)  # warm the page cache
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        outs = list(ex.map(run, range(len(pcm_sample))))
    dt = time.perf_counter() - t0
    spf = 384 if layer == 1 else 1152
    frames = sum(len(p) // (spf * channels) for p in pcm_sample)
    shutil.rmtree(tmp, ignore_errors=True)
    return frames / dt, outs


L12_KERNEL_BOUND = {
    "k12_alloc": "valu+salu issue (a wave minimum per granted step, serial per frame)", "k12_psy": "latency + hbm (serial partition / spreading sums, 14 KB of rows per record)",
    "k_filter": "valu issue (f64) + hbm", "k_fft12": "lds pipe (butterfly program) with the phases' f64 chain in its shadow",
}


def all_host_cores():
    """The CPU baselines run on ALL host cores: the rank widens the mask it narrowed for its GPU's NUMA node."""
    n = os.cpu_count() or 1
    try:
        os.sched_setaffinity(0, range(n))
        n = len(os.sched_getaffinity(0))
    except OSError:
        n = len(os.sched_getaffinity(0))
    return max(1, n)


def main_l12(args, mp3, dev, coll, rank, world, affinity):
    """The Layer I / II path under the same contract: a step = one mp3mi_l12_batch_encode call over the batch."""
    import hashlib
    from mp3common import Oracle, oracle_l12
    layer = args.layer
    spf = 384 if layer == 1 else 1152
    rate, C, mode = 44100, 2, "s"
    kbps = 288 if layer == 1 else 160  # the driver's default: bitrate[version][lay - 1][9], src/musicin.c:371-372
    S = args.streams or 4096
    nf = args.frames or (1149 if layer == 1 else 383)  # 10 s of audio, as BASELINE configs[0] / [1]
    batch = mp3.BatchL12(layer, S, rate, C, kbps, nf)
    pcm = torch.empty((S, nf * spf * C), dtype=torch.int16, device=dev)
    mp3.synth_pcm_device(pcm, nf * spf, C, rate, stream0=rank * S, seed=SEED)
    out = torch.zeros((S, batch.out_stride(nf)), dtype=torch.uint8, device=dev)
    out_len = torch.zeros(S, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    barrier = coll.barrier
    for _ in range(args.warmup):
        batch.encode(pcm, nf, out, out_len)
        batch.sync()
    barrier()
    k_before = batch.kernel_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.encode(pcm, nf, out, out_len)
    batch.sync()
    barrier()
    dt = time.perf_counter() - t0
    k_after = batch.kernel_timing()
    dt = float(coll.reduce(dt, "max"))

    timed_baseline = rank == 0 and not args.no_cpu_baseline
    cores = all_host_cores() if timed_baseline else 8
    n_sample = max(4, cores * 2) if timed_baseline else 8
    idx = sorted(set(np.linspace(0, S - 1, n_sample).astype(int).tolist()))
    pcm_sample = [pcm[i].cpu().numpy() for i in idx]
    out_h, len_h = out[idx].cpu().numpy(), out_len[idx].cpu().numpy()
    got = [out_h[k, : len_h[k]].tobytes() for k in range(len(idx))]
    orc = Oracle()
    oracle_l12(orc, layer, rate, kbps, mode, pcm_sample[0][: spf * C * 4])
    t1 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        refs = list(ex.map(lambda p: oracle_l12(orc, layer, rate, kbps, mode, p)[0], pcm_sample))
    fps = len(idx) * nf / (time.perf_counter() - t1)
    bad = [int(idx[k]) for k in range(len(idx)) if got[k] != refs[k]]
    cpu = None
    if timed_baseline:
        cpu = {"value": round(fps, 1), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "%d of this batch's streams x %d frames, oracle/liboracle.so (mp12_oracle.inc), one thread per stream" % (len(idx), nf)}
        ref = reference_baseline(pcm_sample, rate, [kbps] * len(idx), C, cores, layer=layer)
        if ref is not None:
            rfps, routs = ref
            fbl = mp3.frame_bytes_l12(layer, rate, kbps)
            bad += [int(idx[k]) for k in range(len(idx)) if without_private_bit(got[k], fbl) != without_private_bit(routs[k], fbl) and int(idx[k]) not in bad]
            cpu = {"value": round(rfps, 1), "unit": "frames/s", "cores": cores, "kind": "reference",
                   "sample": "%d of this batch's streams x %d frames, oracle/_ref/encode -l %d (unmodified reference, gcc -O2), one process per stream" % (len(idx), nf, layer),
                   "port_value": round(fps, 1)}
    n_bad = int(coll.reduce(len(bad), "sum"))
    per_rank = coll.gather([rank, rank * S, S, int(hashlib.md5(b"".join(got)).hexdigest()[:15], 16)])
    parity_ok = n_bad == 0
    ranks = [{"rank": int(t[0]), "first_stream": int(t[1]), "streams": int(t[2]), "sample_digest": "%015x" % int(t[3])} for t in per_rank]
    if rank == 0:
        fb = mp3.frame_bytes_l12(layer, rate, kbps)
        alg = spf * C * 2 + fb  # PCM in + frame out
        kt = {k: (k_after[k][0] - k_before[k][0], k_after[k][1] - k_before[k][1]) for k in k_after}
        dom = max(kt, key=lambda k: kt[k][0])
        dom_ms, dom_n = kt[dom]
        avg_launch_s = dom_ms / 1e3 / max(dom_n, 1)
        frames_per_launch = S * nf * args.steps / max(dom_n, 1)
        achieved = alg * frames_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM bytes of the dominant kernel per launch, from the committed counter profile of this build (if there is one)
        prof, prof_why = current_profile_l12(layer, mp3.lib().mp3mi_source_hash().decode(), S, nf)
        traffic = pipeline = None
        if prof is not None:
            pk = [v for k, v in prof["kernels"].items() if k.split("<")[0] == dom]
            if pk and "hbm_read_GB_per_step" in pk[0]:
                traffic = int((pk[0]["hbm_read_GB_per_step"] + pk[0]["hbm_write_GB_per_step"]) * 1e9 / max(pk[0].get("launches_per_step", 1), 1))
            own = [v for k, v in prof["kernels"].items() if k.split("<")[0] in kt]
            hbm = sum(v.get("hbm_read_GB_per_step", 0) + v.get("hbm_write_GB_per_step", 0) for v in own) * 1e9
            pipeline = {"alg_bytes_per_step": alg * S * nf, "hbm_bytes_per_step": int(hbm), "ratio": round(hbm / (alg * S * nf), 2)}
        result = {
            "metric": "Layer %s stereo 44.1 kHz frames/s @%d kbps (bit-exact), MI355X" % ("I" if layer == 1 else "II", kbps),
            "value": round(S * nf * args.steps * world / dt, 1) if parity_ok else None, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "timing": "the K timed calls are issued back to back; ms_per_step = wall time / K",
            "config": {"workload": "batch of %d synthetic 44.1 kHz stereo streams x %d Layer %s frames (%d samples) per GPU, %d kbps CBR, "
                                   "psychoacoustic model 2 (SURVEY 8(f) row 4; not a BASELINE.json config)" % (S, nf, "I" if layer == 1 else "II", spf, kbps),
                       "layer": layer, "streams_per_gpu": S, "frames_per_stream": nf,
                       "pcm": "mp3mi_synth_pcm_device, seed 0x%08x, streams rank*S .." % SEED,
                       "parallelism": "streams sharded across GPUs, no collective"},
            "roofline": {"bound": "valu-issue", "roofline_of": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE), the step's bytes of the kernel over its launches",
                         "traffic_source": ("profiles/" + prof["file"]) if prof is not None else None, "traffic_unavailable_because": prof_why,
                         "pipeline": pipeline,
                         "algorithmic_bytes_per_frame": alg, "kernel_ms_per_launch": round(avg_launch_s * 1e3, 3),
                         "launches_per_step": dom_n // max(args.steps, 1),
                         "limited_by": "instruction issue and latency of the frame kernels, not HBM (kernels: what bounds each)",
                         "kernels": {k: {"bound": L12_KERNEL_BOUND.get(k, "?"), "launches_per_step": kt[k][1] // max(args.steps, 1),
                                         "avg_ms_per_launch": round(kt[k][0] / max(kt[k][1], 1), 3)} for k in sorted(kt)},
                         "source_hash": mp3.lib().mp3mi_source_hash().decode()},
            "cpu_baseline": cpu, "ranks": ranks, "collective_backend": coll.backend, "cpu_affinity_rank0": affinity,
            "parity_spot_check": {"streams_per_rank": len(idx), "bit_exact": parity_ok, "mismatching_streams_rank0": bad,
                                  "witness": "oracle/liboracle.so" + (" + oracle/_ref/encode" if cpu and cpu["kind"] == "reference" else "")},
        }
        print(json.dumps(result), flush=True)
    batch.close()
    coll.close()
    if not parity_ok:
        raise SystemExit("bench.py: PARITY FAILURE -- the GPU bitstream differs from the reference on %d sampled streams" % n_bad)


def other_workloads(mp3, dev, steps=2):
    """configs[2] (8192 streams: the per-GPU share of the 65 536-stream config, the loop kernel in two parts), configs[3],
    configs[4] (Layer III) and the Layer II and Layer I batches, `steps` timed steps each (Layer I: one) after one warm-up,
    every one with its own oracle spot check (8 streams spread over the batch, whole files byte for byte).  Short, so that
    the default bench run carries driver-visible figures for them; tools/full_parity.py and the -m gpu tests hold their
    parity proper."""
    from mp3common import Oracle, oracle_l12
    orc = Oracle()
    rows = []
    for cid in (2, 3, 4):
        cfg = CONFIGS[cid]
        S, nf, C, rate = cfg["streams"], cfg["frames"], cfg["channels"], cfg["rate"]
        wl = Workload(mp3, cfg, dev, stream0=0)
        wl.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.encode()
        wl.batch.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        idx = sorted(set(np.linspace(0, S - 1, 8).astype(int).tolist()))
        out_h, len_h = wl.out[idx].cpu().numpy(), wl.out_len[idx].cpu().numpy()
        pcm_h = [wl.pcm[i].cpu().numpy() for i in idx]
        with ThreadPoolExecutor(max_workers=len(idx)) as ex:
            refs = list(ex.map(lambda k: orc.encode(pcm_h[k], rate, wl.kbps[idx[k]], C)[0], range(len(idx))))
        ok = all(out_h[k, : len_h[k]].tobytes() == refs[k] for k in range(len(idx)))
        rows.append({"config_id": cid, "workload": "%d x %d frames, %.1f kHz %s, %s kbps (%s)" % (S, nf, rate / 1000.0, "stereo" if C == 2 else "mono",
                                                                                                 "64-320 mixed" if cfg["kbps"] == "mix48" else cfg["kbps"], cfg["name"]),
                     "value": round(S * nf * steps / dt, 1) if ok else None, "unit": "frames/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
                     "algorithmic_bytes_per_frame": round(wl.alg_bytes_per_frame(), 1), "bit_exact": ok, "streams_checked": len(idx)})
        wl.close()
        del wl
        torch.cuda.empty_cache()
    for layer, kbps, nf, lsteps in ((2, 160, 383, steps), (1, 288, 1149, 1)):  # (the driver's default bitrates, src/musicin.c:371-372; 10 s of audio)
        rate, C, S, spf = 44100, 2, 4096, (384 if layer == 1 else 1152)
        batch = mp3.BatchL12(layer, S, rate, C, kbps, nf)
        pcm = torch.empty((S, nf * spf * C), dtype=torch.int16, device=dev)
        mp3.synth_pcm_device(pcm, nf * spf, C, rate, stream0=0, seed=SEED)
        out = torch.zeros((S, batch.out_stride(nf)), dtype=torch.uint8, device=dev)
        out_len = torch.zeros(S, dtype=torch.int32, device=dev)
        batch.encode(pcm, nf, out, out_len)
        batch.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(lsteps):
            batch.encode(pcm, nf, out, out_len)
        batch.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        idx = sorted(set(np.linspace(0, S - 1, 8).astype(int).tolist()))
        out_h, len_h = out[idx].cpu().numpy(), out_len[idx].cpu().numpy()
        pcm_h = [pcm[i].cpu().numpy() for i in idx]
        with ThreadPoolExecutor(max_workers=len(idx)) as ex:
            refs = list(ex.map(lambda p: oracle_l12(orc, layer, rate, kbps, "s", p)[0], pcm_h))
        ok = all(out_h[k, : len_h[k]].tobytes() == refs[k] for k in range(len(idx)))
        rows.append({"layer": layer, "workload": "%d x %d Layer %s frames (%d samples), 44.1 kHz stereo, %d kbps, psychoacoustic model 2 (SURVEY 8(f) row 4)"
                                                 % (S, nf, "I" if layer == 1 else "II", spf, kbps),
                     "value": round(S * nf * lsteps / dt, 1) if ok else None, "unit": "frames/s", "steps": lsteps, "ms_per_step": round(dt / lsteps * 1e3, 3),
                     "algorithmic_bytes_per_frame": spf * C * 2 + mp3.frame_bytes_l12(layer, rate, kbps), "bit_exact": ok, "streams_checked": len(idx)})
        batch.close()
        del batch, pcm, out, out_len
        torch.cuda.empty_cache()
    return rows


def launch_ranks(n):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: start N copies of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them), relay rank 0's
    standard output -- the one JSON line -- and return non-zero if any rank does.  The launcher makes no GPU call and no
    torch.cuda query (it does not even import torch) and is never replaced by another program; a rank that fails takes
    the others down with it (by their exact PIDs), so that nobody waits at a barrier for ever -- and so does the launcher's own
    end: SIGTERM / SIGHUP (a timeout around the command) stop the ranks before the launcher leaves.
    The serial loop this replaces: /root/reference/src/musicin.c:585 (one stream, one process)."""
    import signal
    import socket
    import subprocess
    import tempfile

    class Stopped(Exception):
        pass

    def on_signal(signum, frame):  # the launcher is told to stop (a driver's or a test's timeout): the ranks go with it
        raise Stopped(signum)
    old_handlers = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGHUP)}
    procs, rc = [], 0
    markdir = tempfile.mkdtemp(prefix="mp3mi_bench_")
    try:
        for attempt in range(3):
            # a free port is only known to be free NOW: if the rendezvous fails before every rank has joined (somebody took the port in
            # between), the ranks are started again on another one
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            mark = os.path.join(markdir, "joined%d" % attempt)
            base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                        MP3MI_BENCH_RENDEZVOUS_MARK=mark, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs = []
            for r in range(n):
                env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
                # rank 0's stdout is this process's stdout; the other ranks print nothing there, and what they might goes to stderr
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                              stdout=None if r == 0 else sys.stderr, stderr=None))
            rc, alive = 0, set(range(n))
            while alive:
                for r in sorted(alive):
                    code = procs[r].poll()
                    if code is None:
                        continue
                    alive.discard(r)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 128 - code
                        sys.stderr.write("bench.py: rank %d ended with %d: stopping the other ranks\n" % (r, code))
                        for o in alive:
                            procs[o].send_signal(signal.SIGTERM)
                if alive:
                    time.sleep(0.05)
            # once more only if every rank got as far as the rendezvous and not every rank got through it
            at = all(os.path.exists(mark + ".at.%d" % r) for r in range(n))
            joined = all(os.path.exists(mark + ".in.%d" % r) for r in range(n))
            if rc == 0 or joined or not at or attempt == 2:
                break
            sys.stderr.write("bench.py: the ranks did not all join the rendezvous on port %d: once more on another port\n" % port)
    except KeyboardInterrupt:
        rc = 130
    except Stopped as e:
        rc = 128 + int(e.args[0])
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
        deadline = time.time() + 10
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        import shutil
        shutil.rmtree(markdir, ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4],
                    help="BASELINE.json workload (see the module docstring); the default is configs[1] for every N, so that the "
                         "1 / 2 / 4 / 8-GPU series is one per-GPU workload")
    ap.add_argument("--layer", type=int, default=3, choices=[1, 2, 3],
                    help="3: the Layer III path (BASELINE.json's metric, the default); 1 / 2: the Layer I / II path of SURVEY 8(f) row 4")
    ap.add_argument("--streams", type=int, default=0, help="override: streams per GPU")
    ap.add_argument("--frames", type=int, default=0, help="override: frames per stream")
    ap.add_argument("--kbps", type=int, default=0, help="override: one bitrate for every stream (measurements of the schedule at other rates)")
    ap.add_argument("--host-io", action="store_true",
                    help="after the resident measurement (which stays `value`), time the same K steps through "
                         "mp3mi_batch_encode_host_async -- PCM in page-locked host memory, file bytes back to it, both crossing PCIe "
                         "chunk by chunk beside the kernels -- and add an `end_to_end` object to the line (SURVEY 8(d)).  On by default "
                         "for the Layer III line; this flag forces it where --no-cpu-baseline would drop it")
    ap.add_argument("--no-host-io", action="store_true", help="skip the end_to_end leg")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short runs of configs[3], configs[4] and Layer II that the default line (N = 1, configs[1] at its own "
                         "size) carries as `other_workloads`")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="profiling pass: skip the timed CPU baseline, the end_to_end leg and the other workloads (only the "
                         "headline pipeline's kernels are launched); a small oracle parity check remains")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    # `--gpus N` without a launcher around it: this process becomes the launcher of N ranks and never touches the GPU itself
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to report a line whose n_gpus "
                         "is not what was asked for" % (args.gpus, world))
    distributed = world > 1
    # (test hook: MP3MI_BENCH_ONE_GPU=1 runs every rank on device 0 over gloo, so that the multi-rank code path -- stream
    # ranges per rank, barriers, max-over-ranks time, parity vote -- can be exercised on a one-GPU box)
    one_gpu = os.environ.get("MP3MI_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    # the rank's threads next to its GPU -- before torch (and with it the HIP runtime's helper threads) is loaded
    affinity = pin_to_gpu_numa(dev_index, local_rank, world) if distributed else "unchanged (one rank)"
    load_torch()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the encoder has no CPU path")
    if not one_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but this node shows %d device(s): one rank per GPU, no sharing (MP3MI_BENCH_ONE_GPU=1 is the "
                         "one-GPU rehearsal hook of the tests)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    mark = os.environ.get("MP3MI_BENCH_RENDEZVOUS_MARK") if distributed else None  # (the launcher's retry looks for these two files)
    if mark:
        open(mark + ".at.%d" % rank, "w").close()
    coll = Coll(world, dev, want_nccl=not one_gpu and os.environ.get("MP3MI_BENCH_BACKEND", "nccl") != "gloo")
    if mark:
        open(mark + ".in.%d" % rank, "w").close()

    mp3 = importlib.import_module("mp3-enc-bsd_amd")
    if args.layer != 3:
        return main_l12(args, mp3, dev, coll, rank, world, affinity)
    cfg_id = args.config
    cfg = dict(CONFIGS[cfg_id])
    default_size = not (args.streams or args.frames or args.kbps)
    if args.kbps:
        cfg["kbps"] = args.kbps
    if args.streams:
        cfg["streams"] = args.streams
    if args.frames:
        cfg["frames"] = args.frames
    S, nf, C, rate = cfg["streams"], cfg["frames"], cfg["channels"], cfg["rate"]
    wl = Workload(mp3, cfg, dev, stream0=rank * S)

    barrier = coll.barrier

    # A step = one mp3mi_batch_encode call over the whole batch.  The K timed calls are issued back to back, as a
    # service encoding batch after batch would issue them: a call's feed-forward kernels may start beside the last
    # loop kernel of the call before (batch.cpp, encode_impl); the timed region is closed by a sync of the batch,
    # the barrier and a device-wide synchronize, and the HIP-event timing of the K calls is read after it.
    # (MP3MI_BENCH_SYNC_EACH=1: a sync after every call, as rounds 1-2 measured.)
    sync_each = os.environ.get("MP3MI_BENCH_SYNC_EACH", "0") == "1"
    for _ in range(args.warmup):
        wl.step()
    barrier()
    t_before = wl.batch.total_timing()
    clock = ClockSampler(dev_index).start() if rank == 0 else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if sync_each:
            wl.step()
        else:
            wl.encode()
    wl.batch.sync()
    barrier()
    dt = time.perf_counter() - t0
    clock_ghz = clock.stop() if clock is not None else None
    t_after = wl.batch.total_timing()
    loop_ms, all_ms, launches = t_after[0] - t_before[0], t_after[1] - t_before[1], t_after[2] - t_before[2]
    assert t_after[3] - t_before[3] == args.steps
    dt = float(coll.reduce(dt, "max"))

    # End to end with PCIe (--host-io): the same K steps with the PCM in page-locked host memory and the bytes delivered to it.
    end_to_end = None
    if args.host_io or not (args.no_host_io or args.no_cpu_baseline):
        # (every rank page-locks its own copy -- 7.9 GB at the default size; a rank that cannot says so, and ALL ranks then leave the
        # leg out together: a rank missing at the barriers below would hang the others)
        h_pcm = h_out = h_len = None
        try:
            h_pcm = torch.empty(wl.pcm.shape, dtype=torch.int16, pin_memory=True)  # page-locked from the start: no staging copy
            h_out = torch.empty(wl.out.shape, dtype=torch.uint8, pin_memory=True)
            h_len = torch.empty(S, dtype=torch.int32, pin_memory=True)
        except RuntimeError as e:
            print("bench.py: rank %d could not page-lock the host buffers of the end-to-end leg (%s)" % (rank, str(e).splitlines()[0]), file=sys.stderr)
            h_pcm = None
        if int(coll.reduce(1 if h_pcm is not None else 0, "min")) == 0:
            end_to_end = {"skipped": "a rank could not page-lock its host buffers"}
            h_pcm = None
    else:
        h_pcm = None
    if h_pcm is not None:
        h_pcm.copy_(wl.pcm)
        for _ in range(max(args.warmup, 1)):
            wl.batch.encode_host_async(h_pcm, nf, h_out, h_len)
            wl.batch.sync()
        barrier()
        st0 = wl.batch.host_io_stats()
        th0 = time.perf_counter()
        for _ in range(args.steps):
            wl.batch.encode_host_async(h_pcm, nf, h_out, h_len)  # back to back: call n + 1's PCM goes up beside call n's last kernels
        wl.batch.sync()
        barrier()
        dth = time.perf_counter() - th0
        st1 = wl.batch.host_io_stats()
        dth = float(coll.reduce(dth, "max"))
        same = bool(torch.equal(h_len, wl.out_len.cpu())) and all(
            torch.equal(h_out[i, : int(h_len[i])], wl.out[i, : int(h_len[i])].cpu()) for i in range(0, S, max(1, S // 256)))
        up_b, dn_b = st1["h2d_bytes"] - st0["h2d_bytes"], st1["d2h_bytes"] - st0["d2h_bytes"]
        up_ms, dn_ms = st1["h2d_ms"] - st0["h2d_ms"], st1["d2h_ms"] - st0["d2h_ms"]
        copy_ms_per_step = (up_ms + dn_ms) / args.steps
        extra_ms = max(0.0, (dth - dt) / args.steps * 1e3)
        end_to_end = {"frames_per_s": round(S * nf * args.steps * world / dth, 1), "ms_per_step": round(dth / args.steps * 1e3, 3),
                      "ratio_to_resident": round(dt / dth, 4),
                      "h2d_GBs": round(up_b / up_ms / 1e6, 2) if up_ms > 0 else None, "d2h_GBs": round(dn_b / dn_ms / 1e6, 2) if dn_ms > 0 else None,
                      "h2d_GB_per_step": round(up_b / args.steps / 1e9, 3), "d2h_GB_per_step": round(dn_b / args.steps / 1e9, 3),
                      "copy_ms_per_step": round(copy_ms_per_step, 2),
                      "overlap": round(1.0 - min(1.0, extra_ms / copy_ms_per_step), 4) if copy_ms_per_step > 0 else None,
                      "bytes_equal_resident_path": same,
                      "how": "mp3mi_batch_encode_host_async on page-locked buffers: the PCM goes up as a 2-D copy per chunk on a copy stream of its "
                             "own (the chunk's first kernel waits for it), the call's bytes come down in one copy behind its last formatter, beside "
                             "the next call's kernels; GB/s = bytes / summed copy durations (HIP events on the copy streams); overlap = share of the "
                             "copy time that did not lengthen the step"}
        del h_pcm, h_out, h_len

    # parity spot check on the exact device bytes of THIS rank (every rank checks its own streams; the CPU baseline is
    # timed on rank 0 -- at every N, AFTER the timed region: the other ranks do their small check and wait at the
    # parity vote below -- so that every bench line carries it)
    timed_baseline = rank == 0 and not args.no_cpu_baseline
    cores = all_host_cores() if timed_baseline else 8  # (all of the host's cores, whatever the rank was pinned to for its GPU)
    n_sample = max(4, cores * 2) if timed_baseline else 8
    idx = sorted(set(np.linspace(0, S - 1, n_sample).astype(int).tolist()))
    pcm_sample = [wl.pcm[i].cpu().numpy() for i in idx]
    kb_sample = [wl.kbps[i] for i in idx]
    out_h, len_h = wl.out[idx].cpu().numpy(), wl.out_len[idx].cpu().numpy()
    got = [out_h[k, : len_h[k]].tobytes() for k in range(len(idx))]
    fps, refs = cpu_baseline(pcm_sample, rate, kb_sample, C, cores)
    bad = [int(idx[k]) for k in range(len(idx)) if got[k] != refs[k]]
    if end_to_end is not None and not end_to_end["bytes_equal_resident_path"]:
        bad.append(-1)  # the host path delivered other bytes than the device path: a parity failure like any other
    cpu = None
    if timed_baseline:
        cpu = {"value": round(fps, 1), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "%d of this batch's streams x %d frames, oracle/liboracle.so, one thread per stream" % (len(idx), nf)}
        ref = reference_baseline(pcm_sample, rate, kb_sample, C, cores)
        if ref is not None:  # the reference binary itself: the baseline proper, and a second parity witness
            rfps, routs = ref
            fb3 = [mp3.frame_bytes(rate, k) for k in kb_sample]
            bad += [int(idx[k]) for k in range(len(idx)) if without_private_bit(got[k], fb3[k]) != without_private_bit(routs[k], fb3[k]) and int(idx[k]) not in bad]
            cpu = {"value": round(rfps, 1), "unit": "frames/s", "cores": cores, "kind": "reference",
                   "sample": "%d of this batch's streams x %d frames, oracle/_ref/encode (unmodified reference, gcc -O2), one process per stream" % (len(idx), nf),
                   "port_value": round(fps, 1)}
    # which streams each rank encoded, and a digest of what it produced (the ranks' ranges must be disjoint, their bytes differ)
    import hashlib
    n_bad = int(coll.reduce(len(bad), "sum"))
    per_rank = coll.gather([rank, rank * S, S, int(hashlib.md5(b"".join(got)).hexdigest()[:15], 16)])
    parity_ok = n_bad == 0
    ranks = [{"rank": int(t[0]), "first_stream": int(t[1]), "streams": int(t[2]), "sample_digest": "%015x" % int(t[3])} for t in per_rank]

    frames_total = S * nf * args.steps * world
    if rank == 0:
        # dominant kernel: k_loop; algorithmic bytes of one launch / its average duration (HIP events on its stream)
        alg = wl.alg_bytes_per_frame()
        frames_per_launch = S * nf * args.steps / max(launches, 1)
        avg_launch_s = loop_ms / 1e3 / max(launches, 1)
        achieved = alg * frames_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        lps = launches // max(args.steps, 1)
        # (a batch of more streams than the kernel holds resident -- 16 wavefronts per CU -- goes through it in parts, one launch each)
        kname = "k_loop"
        # Counter figures cannot be read from inside the timed run: they come from the committed profile that
        # profiles/CURRENT names -- if, and only if, it was taken on the sources this library was built from.
        src_hash = mp3.lib().mp3mi_source_hash().decode()
        prof, prof_why = current_profile(src_hash, S, nf)
        traffic = issue = pipeline = kernels = None
        if prof is not None:
            pk = prof["kernels"]
            if kname in pk and pk[kname].get("dispatches") == lps:
                traffic = pk[kname].get("hbm_bytes_per_launch")
                issue = issue_roofline(pk[kname], S, avg_launch_s)
            hbm_step = sum(v.get("hbm_bytes_per_launch", 0) * v.get("dispatches", 0) for v in pk.values())
            pipeline = {"alg_bytes_per_step": int(alg * S * nf), "hbm_bytes_per_step": int(hbm_step),
                        "ratio": round(hbm_step / (alg * S * nf), 2), "hbm_gbs_over_the_step": round(hbm_step / (dt / args.steps) / 1e9, 1)}
            kernels = {k: {"bound": KERNEL_BOUND.get(k.split("<")[0], "?"), "launches_per_step": v.get("dispatches"),
                           "avg_ms_per_launch": v.get("avg_ms"), "hbm_bytes_per_launch": v.get("hbm_bytes_per_launch")}
                       for k, v in sorted(pk.items())}
        result = {
            "metric": "stereo 44.1 kHz frames/s @128 kbps (bit-exact), 1/2/4/8 MI355X + %HBM roofline",
            "value": round(frames_total / dt, 1) if parity_ok else None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "timing": "sync_each" if sync_each else "pipelined (the K timed calls are issued back to back; ms_per_step = wall time / K)",
            "config": {"workload": "batch of %d synthetic %.1f kHz %s streams x %d frames per GPU, %s kbps CBR (%s%s)"
                       % (S, rate / 1000.0, "stereo" if C == 2 else "mono", nf,
                          "64-320 mixed" if cfg["kbps"] == "mix48" else str(cfg["kbps"]), cfg["name"],
                          "" if default_size else ", size overridden"),
                       "config_id": cfg_id, "streams_per_gpu": S, "frames_per_stream": nf,
                       "pcm": "mp3mi_synth_pcm_device, seed 0x%08x, streams rank*S .." % SEED,
                       "parallelism": "streams sharded across GPUs, no collective"},
            # bound: what the dominant kernel runs against; achieved / peak / frac: its algorithmic
            # bytes against the HBM roofline, as the contract asks -- tiny, because the kernel is an instruction stream
            "roofline": {"bound": "valu-issue", "roofline_of": "hbm", "kernel": kname, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE)",
                         "traffic_source": ("profiles/" + prof["file"]) if prof is not None else None,
                         "traffic_unavailable_because": prof_why,
                         "algorithmic_bytes_per_frame": round(alg, 1),
                         "kernel_ms_per_launch": round(avg_launch_s * 1e3, 3), "launches_per_step": lps,
                         # (HIP-event span of a call from its first to its last kernel; calls overlap when pipelined, so
                         # it is only reported where every call was synchronised)
                         "all_kernels_ms_per_step": round(all_ms / max(args.steps, 1), 3) if sync_each else None,
                         "limited_by": "instruction issue of the dominant kernel, not HBM (issue: share of the SIMDs' VALU issue capacity in use; kernels: what bounds each)",
                         "issue": issue, "clock_ghz_measured": clock_ghz, "pipeline": pipeline, "kernels": kernels, "source_hash": src_hash},
            "cpu_baseline": cpu,
            "end_to_end": end_to_end,
            "ranks": ranks,
            "collective_backend": coll.backend,
            "cpu_affinity_rank0": affinity,
            "parity_spot_check": {"streams_per_rank": len(idx), "bit_exact": parity_ok, "mismatching_streams_rank0": bad,
                                  "witness": "oracle/liboracle.so" + (" + oracle/_ref/encode" if cpu and cpu["kind"] == "reference" else "")},
        }
    wl.close()
    del wl
    others_ok = True
    if rank == 0:
        # The other workloads, driver-visible (short runs; the headline above is untouched by them: its batch is closed).
        # Only on the default line: one GPU, configs[1] at its own size.
        if world == 1 and cfg_id == 1 and default_size and not (args.no_other_workloads or args.no_cpu_baseline):
            # (whatever happens in here -- no memory for a batch of 16 384 streams, an oracle error --, the finished headline
            # measurement is printed; the failure is recorded in the line and in the exit code)
            try:
                torch.cuda.empty_cache()
                result["other_workloads"] = other_workloads(mp3, dev)
                others_ok = all(o["bit_exact"] for o in result["other_workloads"])
                if not others_ok:
                    result["value"] = None
            except Exception as e:  # noqa: BLE001
                result["other_workloads"] = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}
                others_ok = False
        print(json.dumps(result), flush=True)
    coll.close()
    if not parity_ok:
        raise SystemExit("bench.py: PARITY FAILURE -- the GPU bitstream differs from the reference on %d sampled streams" % n_bad)
    if not others_ok:
        raise SystemExit("bench.py: other_workloads failed (see the line's other_workloads)")


if __name__ == "__main__":
    main()
