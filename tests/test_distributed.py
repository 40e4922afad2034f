"""Multi-process sharding + gather (world_size 2, gloo, CPU).  The compute on each rank is a stand-in
(the product has no CPU path); what is under test is the partition of B alignments over ranks and
the single all-gather of the 16-double result rows."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    batch = importlib.import_module("slam-eds_amd.batch")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = batch.shard_range(total, world, rank)
    # stand-in "solve": row b carries its global index so the gathered table can be verified
    local = np.zeros((count, batch.RESULT_WIDTH))
    for i in range(count):
        b = first + i
        local[i] = batch.pack_result([b, 2 * b, 3 * b], [0, 0, 0, 1], np.arange(6) + b, 0.5 * b, 10, True)
    table = batch.gather_results(local, total)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, first, count, table))


@pytest.mark.parametrize("total", [8, 7, 1])
def test_shard_and_gather_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = []
    for rank, first, count, table in res:
        covered += list(range(first, first + count))
        assert table.shape == (total, 16)
        for b in range(total):
            assert np.array_equal(table[b, 0:3], [b, 2 * b, 3 * b]) and table[b, 13] == 0.5 * b and table[b, 15] == 1.0
    assert sorted(covered) == list(range(total))            # every alignment solved exactly once


def test_shard_range_properties():
    batch = importlib.import_module("slam-eds_amd.batch")
    for total in (0, 1, 5, 64, 1000):
        for world in (1, 2, 4, 8):
            spans = [batch.shard_range(total, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for f, c in spans:
                assert f == min(pos, total) or c == 0
                pos = f + c
    assert batch.shard_range(64, 8, 3) == (24, 8)           # config 5: 64 alignments, 8 per GPU


def test_gather_is_identity_without_process_group():
    batch = importlib.import_module("slam-eds_amd.batch")
    local = np.arange(32.0).reshape(2, 16)
    assert np.array_equal(batch.gather_results(local, 2), local)
