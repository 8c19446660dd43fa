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
