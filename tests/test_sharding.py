"""Multi-GPU path without GPUs: streams shard across ranks with no data-path collective
(SURVEY.md 8(e)).  Two gloo ranks each take their slice of a batch through the emulated
library; rank 0 gathers only the timing-style scalar and the per-stream md5s for checking."""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def shard(n_streams, rank, world):
    """contiguous stream ranges, n_streams/world per rank (remainder to the first ranks)"""
    base, rem = divmod(n_streams, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _worker(rank, world, port, n_streams, nf, ret):
    from mp3common import Mp3mi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    emu = Mp3mi(emu=True)
    lo, hi = shard(n_streams, rank, world)
    pcm = np.stack([emu.synth(nf * 1152, 2, 44100, 900 + s) for s in range(lo, hi)])
    outs = emu.encode_host(pcm, 44100, 2, 128, nf)
    md5s = [hashlib.md5(o).hexdigest() for o in outs]
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, md5s))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks timing reduction
    if rank == 0:
        ret["md5"] = gathered
        ret["tmax"] = float(t.item())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_streams_without_exchange(oracle):
    from mp3common import Mp3mi
    Mp3mi(emu=True)  # make sure the emulated library is built before forking
    world, n_streams, nf = 2, 5, 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29731, n_streams, nf, ret), nprocs=world, join=True)
    assert ret["tmax"] == 2.0
    seen = {}
    for lo, md5s in ret["md5"]:
        for i, m in enumerate(md5s):
            seen[lo + i] = m
    assert sorted(seen) == list(range(n_streams))
    emu = Mp3mi(emu=True)
    for s in range(n_streams):
        ref, _ = oracle.encode(emu.synth(nf * 1152, 2, 44100, 900 + s), 44100, 128, 2)
        assert seen[s] == hashlib.md5(ref).hexdigest()


def test_shard_ranges_cover_everything():
    for n, w in ((4096, 8), (65536, 8), (5, 2), (7, 4)):
        r = [shard(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))


def _bench(args, **envkw):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(envkw)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_launches_its_own_ranks_and_fails_with_them():
    """`python bench.py --gpus 2` starts two ranks itself (RANK / WORLD_SIZE / MASTER_* per child).  Without a GPU every
    rank refuses to run (the encoder has no CPU path), and the launcher -- which makes no GPU call of its own -- must come
    back non-zero with nothing on standard output, not hang at a barrier."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--streams", "64", "--frames", "2"])
    assert r.returncode != 0 and not r.stdout.strip()
    # (one refusal at least: the launcher takes the other rank down as soon as the first one fails, possibly before it spoke)
    assert 1 <= r.stderr.count("bench.py needs a GPU") <= 2 if not torch.cuda.is_available() else True


def test_bench_refuses_a_world_size_other_than_the_flag():
    r = _bench(["--gpus", "4"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = _bench(["--gpus", "1"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def _coll_worker(rank, world, port, ret):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.load_torch()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    c = bench.Coll(world, torch.device("cpu"), want_nccl=False)
    c.barrier()
    out = (c.backend, c.reduce(float(rank + 1), "max"), c.reduce(rank, "min"), c.reduce(rank + 1, "sum"), c.gather([rank, 10 * rank]))
    if rank == 0:
        ret["out"] = out
    c.close()


def test_the_bench_ranks_control_traffic_over_gloo():
    """bench.py's Coll: barrier, maximum, vote and gather of the ranks over gloo with CPU tensors -- what carries a run where RCCL
    does not come up on every rank (or MP3MI_BENCH_BACKEND=gloo asks for it); three ranks, no GPU."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_coll_worker, args=(3, 29757, ret), nprocs=3, join=True)
    backend, tmax, vmin, vsum, per = ret["out"]
    assert backend.startswith("gloo") and tmax == 3.0 and vmin == 0 and vsum == 6
    assert per == [[0, 0], [1, 10], [2, 20]]


def test_numa_pinning_never_raises_and_says_what_it_did():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    before = os.sched_getaffinity(0)
    cpus, what = bench.gpu_numa_cpus(0)
    assert isinstance(what, str) and (cpus is None or cpus <= before)
    said = bench.pin_to_gpu_numa(0, 0, 2)
    assert isinstance(said, str) and said
    os.sched_setaffinity(0, before)
    assert bench.all_host_cores() >= 1
    os.sched_setaffinity(0, before)
