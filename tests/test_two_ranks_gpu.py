"""Two ranks, one GPU, the REAL shards (Next #3a).  RCCL refuses two ranks on one device, so the ranks rendezvous over gloo
and share device 0 — what runs on the GPU is exactly what two ranks of an 8-GPU job run: `BatchTracker` solving its shard of BASELINE.json's
configs[4] (64 alignments, seeds 5000 + b, 640x480, 2 000 points, 10 LM6 iterations) with the kernel the library picks for 32 alignments,
then the one all-gather of 16 doubles per alignment.  The gathered table must equal the unsharded table row for row, and the oracle.
The children are started before they touch the GPU (a process that has initialised HIP is never re-used as a rank)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, json, os, sys
import numpy as np
import torch                                   # before libeds_hip.so (capi.torch_loaded_first)
import torch.distributed as dist
root, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
sys.path.insert(0, root)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
batch = importlib.import_module("slam-eds_amd.batch")
TOTAL, N, H, W = 64, 2000, 480, 640
first, count = batch.shard_range(TOTAL, world, rank)
als = [synth.make_alignment(5000 + b, H=H, W=W, N=N) for b in range(first, first + count)]
cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10)      # device 0 for every rank
bt = batch.BatchTracker(cfg, TOTAL, N, H, W, rank=rank, world_size=world)
bt.load(als)
tables = []
for step in range(3):                           # the bench's step, three times: reset, solve the shard, gather
    bt.reset_states(als)
    bt.solve(sync=True)
    tables.append(bt.gather(device=None))       # gloo: host tensors
li = bt.handle.last_launch()
dist.barrier()
if rank == 0:
    np.save(out, np.stack(tables))
print("RANK_RESULT " + json.dumps({"rank": rank, "first": first, "count": count, "kernel": li["kernel"], "cus": li["cus_per_alignment"],
                                   "same_every_step": bool(all(np.array_equal(t, tables[0]) for t in tables))}))
bt.close()
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_and_gather_config4(gpu, capi, synth, po, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "table.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, ROOT, str(r), "2", str(port), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    res = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        line = [l for l in so.splitlines() if l.startswith("RANK_RESULT ")]
        assert p.returncode == 0 and line, f"rc {p.returncode}\n{so[-1500:]}\n{se[-3000:]}"
        res.append(json.loads(line[-1][len("RANK_RESULT "):]))
    res.sort(key=lambda r: r["rank"])
    assert [(r["first"], r["count"]) for r in res] == [(0, 32), (32, 32)]
    assert all(r["same_every_step"] for r in res)
    # 32 alignments per rank: teams of 4 CUs, in two candidate groups (round 5: eds_launch_rule.hpp) — 8 CUs each
    assert all(r["kernel"].startswith("eds_fused6_kernel<0, 1, 512,") and r["kernel"].endswith(", 4, 2>") and r["cus"] == 8 for r in res), res
    tables = np.load(out)
    table = tables[0]
    assert table.shape == (64, 16) and table[:, 15].min() == 1.0
    # the unsharded solve of the same 64 alignments in THIS process: teams of 4 either way, so the rows are bit-identical
    als = [synth.make_alignment(5000 + b) for b in range(64)]
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), 64, 2000, 480, 640)
    for b, a in enumerate(als):
        h.set_alignment(b, a)
    h.set_states(0, np.stack([a.p0 for a in als]), np.stack([a.q0 for a in als]), np.stack([a.v0 for a in als]))
    h.optimize_batch(0, 0, 64)
    want = h.results(0, 64)
    h.close()
    assert np.array_equal(table[:, :15], want[:, :15]), np.abs(table[:, :15] - want[:, :15]).max()
    # ... and the oracle, on rows of both shards
    for b in (0, 7, 31, 32, 40, 63):
        ref = po.Oracle(als[b]).pose6_lm(als[b].p0, als[b].q0, als[b].v0, iters=10, lambda0=0.01)
        assert table[b, 14] == ref["iterations"]
        assert po.se3_distance(table[b, 0:3], table[b, 3:7], ref["p"], ref["q"]) <= 1e-6
