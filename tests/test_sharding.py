"""Multi-GPU path on CPU: world_size-2 gloo processes shard the streams exactly like bench.py /
a multi-GPU deployment does (no data-path collective)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpcnet_amd import shard, synth  # noqa: E402


def test_partition_properties():
    for n in (0, 1, 7, 8, 1024, 8192, 8193):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                first, count = shard.partition(n, r, world)
                seen += list(range(first, first + count))
                assert all(shard.owner(s, n, world) == r for s in range(first, first + count))
            assert seen == list(range(n))
            sizes = [shard.partition(n, r, world)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.partition(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_streams, T, q, engine=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = shard.partition(n_streams, rank, world)
        blob = synth.blob_bytes(synth.make_model())
        feats = np.stack([synth.make_features(1000 + s, T) for s in range(first, first + count)])
        if engine:                                  # the HIP engine itself, one batch per rank (all ranks on device 0 here)
            from lpcnet_amd import api
            b = api.LPCNetBatch(count, blob, device=0)
            pcm = b.synthesize(feats)
            b.close()
        else:
            from oracle import orc                  # CPU-only run: the oracle stands in for the per-rank engine
            om = orc.OracleModel(blob)
            pcm = np.stack([om.new_state().synthesize(f) for f in feats])
        # control plane only: barrier + max of a local time + total of produced samples
        dist.barrier()
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total = torch.tensor([pcm.size], dtype=torch.int64)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        # optional result gather (not on the data path of the engine)
        gathered = [None] * world
        dist.all_gather_object(gathered, (first, pcm))
        if rank == 0:
            full = np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])])
            q.put((float(t), int(total), full))
    finally:
        dist.destroy_process_group()


def _two_ranks(n_streams, T, engine):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, T, q, engine)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, total, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == float(world) and total == n_streams * T * 160
    from oracle import orc
    om = orc.OracleModel(synth.blob_bytes(synth.make_model()))
    want = np.stack([om.new_state().synthesize(synth.make_features(1000 + s, T)) for s in range(n_streams)])
    assert np.array_equal(full, want)


def test_two_rank_sharding_reproduces_single_process():
    _two_ranks(3, 4, engine=False)


@pytest.mark.gpu
def test_two_rank_sharding_on_the_engine():
    """the same two gloo ranks with the HIP ENGINE as the per-rank worker (both on device 0: the boxes have one GPU),
    7 streams so that the shards are uneven; checked against the oracle"""
    _two_ranks(7, 6, engine=True)
