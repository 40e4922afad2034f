"""Batched and multi-GPU alignment driver.

Independent (keyframe, event-frame) alignments shard embarrassingly: alignment
``b`` of ``B`` goes to rank ``b // ceil(B / G)``; every rank solves its shard on its own GPU with
no communication, then ONE small collective gathers the per-alignment results
(p[3], q[4], v[6], cost, iterations, status = 16 doubles) — ``torch.distributed.all_gather`` which
is RCCL over xGMI with the ``nccl`` backend and gloo in the CPU tests.  The reference has no
distributed layer at all (one Tracker per process, Tracker.hpp:40-58), so there is nothing to
translate here.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

RESULT_WIDTH = 16      # p3 q4 v6 cost iters status


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [first, first + count) of `total` alignments for `rank`."""
    per = -(-total // world_size)
    first = min(rank * per, total)
    return first, max(0, min(per, total - first))


def pack_result(p, q, v, cost, iterations, success) -> np.ndarray:
    out = np.zeros(RESULT_WIDTH)
    out[0:3], out[3:7], out[7:13] = p, q, v
    out[13], out[14], out[15] = cost, iterations, 1.0 if success else 0.0
    return out


def gather_results(local: np.ndarray, total: int, device=None, to_host: bool = True, force: bool = False):
    """All-gather the per-rank result rows into the global [total, 16] table (every rank gets it; with
    ``to_host=False`` a rank only takes part in the collective and returns None — e.g. every rank but 0 in bench.py).

    `local` is this rank's [count, 16] block.  Without an initialised process group this is the
    identity (single process); with a group of ONE rank too, unless ``force`` asks for the collective
    all the same (the single-GPU box's only way to put RCCL's all-gather of a device tensor under
    test: tests/test_rccl_gpu.py, ``EDS_BENCH_FORCE_DIST=1 python bench.py``).  Shards may be ragged; rows are padded to the common shard size for
    the collective and trimmed afterwards.
    """
    from . import capi
    if device is not None and capi.torch_loaded_first is False:
        raise RuntimeError("a GPU collective needs torch.cuda, and PyTorch-ROCm only works on its own HIP runtime: import torch before the "
                           "first slam-eds_amd call in this process (libeds_hip.so was loaded first and brought /opt/rocm's libamdhip64.so.7)")
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1, RESULT_WIDTH)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return local[:total].copy()
    world = dist.get_world_size()
    per = -(-total // world)
    buf = torch.zeros((per, RESULT_WIDTH), dtype=torch.float64)
    buf[: local.shape[0]] = torch.from_numpy(local)
    if device is not None:
        buf = buf.to(device)
    out = torch.empty((world * per, RESULT_WIDTH), dtype=torch.float64, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    if not to_host:
        return None
    table = out.cpu().numpy()
    rows = []
    for r in range(world):
        first, count = shard_range(total, world, r)
        rows.append(table[r * per: r * per + count])
    return np.concatenate(rows, axis=0) if rows else np.zeros((0, RESULT_WIDTH))


class BatchTracker:
    """Solves this rank's shard of `total` alignments on one GPU handle."""

    def __init__(self, cfg, total: int, max_points: int, H: int, W: int, rank: int = 0, world_size: int = 1):
        if world_size > 1:
            import torch  # noqa: F401  (before libeds_hip.so: see capi.torch_loaded_first)
        from . import capi
        self.capi = capi
        self.total, self.rank, self.world_size = total, rank, world_size
        self.first, self.count = shard_range(total, world_size, rank)
        self.handle = capi.Handle(cfg, max(1, self.count), max_points, H, W)

    def load(self, alignments: Sequence) -> None:
        """`alignments[i]` is the global alignment ``first + i`` of this shard."""
        assert len(alignments) == self.count
        for i, al in enumerate(alignments):
            self.handle.set_alignment(i, al)

    def reset_states(self, alignments: Sequence) -> None:
        for i, al in enumerate(alignments):
            self.handle.set_state(i, al.p0, al.q0, al.v0)

    def solve(self, level: int = 0, sync: bool = True) -> None:
        if self.count > 0:
            self.handle.optimize_batch(level, 0, self.count, sync=sync)

    def local_results(self) -> np.ndarray:
        if self.count == 0:
            return np.zeros((0, RESULT_WIDTH))
        return self.handle.results(0, self.count)

    def gather(self, device=None) -> np.ndarray:
        return gather_results(self.local_results(), self.total, device=device)

    def close(self):
        self.handle.close()
