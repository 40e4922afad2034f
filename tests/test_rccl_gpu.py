"""First contact with RCCL on the one GPU a test box has (VERDICT r2 "next" #2): a process group of ONE rank
over the `nccl` backend (= RCCL on ROCm), the sharded step of bench.py — solve of step k on the library's stream while
the all-gather of step k - 1 runs on torch's stream — and `all_gather_into_tensor` on a DEVICE tensor, checked bit for
bit against the unsharded path.  Runs in a child process: torch has to be imported before libeds_hip.so in the process
that uses both (capi.torch_loaded_first), and a process that has touched the GPU must not be re-used as a rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, json, os, socket, sys
import numpy as np
import torch                                   # before libeds_hip.so
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
batch = importlib.import_module("slam-eds_amd.batch")
B, N, H, W = 48, 600, 120, 160
world = dist.get_world_size()
first, count = batch.shard_range(B, world, dist.get_rank())
als = [synth.make_alignment(5000 + b, H=H, W=W, N=N) for b in range(first, first + count)]
cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=8)
bt = batch.BatchTracker(cfg, B, N, H, W, rank=dist.get_rank(), world_size=world)
bt.load(als)
dev = torch.device("cuda", 0)
# unsharded reference: plain synchronous solve, host-side table
bt.reset_states(als); bt.solve(sync=True)
want = bt.local_results().copy()
# the sharded step of bench.py: launch step k (library stream), all-gather step k - 1 (torch stream, device tensor), then wait
# (bench.py's step_sharded: a ResultGatherer with its own stream, pinned staging and event, started behind the launch of step k)
tables, prev = [], None
g = batch.ResultGatherer(B, device=dev, to_host=True, force=True)
for k in range(4):
    bt.reset_states(als)
    bt.solve(sync=False)
    if prev is not None:
        g.start(prev)
    bt.handle.sync()
    if prev is not None:
        tables.append(g.finish())
    prev = bt.local_results()
tables.append(batch.gather_results(prev, B, device=dev, to_host=True, force=True))      # and the one-shot form
torch.cuda.synchronize()
# the collective on a device tensor directly (what gather_results wraps), and a barrier
x = torch.arange(16, dtype=torch.float64, device=dev).reshape(1, 16)
y = torch.empty((world, 16), dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(y, x)
dist.barrier()
ok_tables = all(np.array_equal(t, want) for t in tables)
out = {"world": world, "backend": dist.get_backend(), "tables": len(tables), "bit_identical": bool(ok_tables),
       "device_gather_ok": bool(torch.equal(y.cpu(), x.cpu())), "iterations": float(want[:, 14].mean()), "success": float(want[:, 15].mean()),
       "torch_loaded_first": capi.torch_loaded_first}
bt.close()
dist.destroy_process_group()
print("RCCL_RESULT " + json.dumps(out))
'''


@pytest.mark.gpu
def test_rccl_world1_sharded_step_and_device_gather():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=600, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_RESULT ")]
    assert r.returncode == 0 and line, f"rc {r.returncode}\nstdout: {r.stdout[-2000:]}\nstderr: {r.stderr[-3000:]}"
    out = json.loads(line[-1][len("RCCL_RESULT "):])
    assert out["backend"] == "nccl" and out["world"] == 1           # n_gpus from the collective, not from the environment
    assert out["tables"] == 4 and out["bit_identical"], out         # sharded + overlapped path == unsharded path, bit for bit
    assert out["device_gather_ok"] and out["success"] == 1.0 and out["iterations"] > 0, out
    assert out["torch_loaded_first"] is True
