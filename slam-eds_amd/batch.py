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


class ResultGatherer:
    """The one collective of the path, with everything it needs allocated ONCE: pinned host staging in and out, the device tensors,
    a stream of its own and an event.  ``start(local)`` queues copy-in -> all_gather_into_tensor -> copy-out on that stream and
    returns at once (the solve of the next step runs meanwhile on the library's stream: neither is the legacy default stream, so
    they do not serialise); ``finish()`` waits for the event and returns the [total, 16] table (rank 0 / to_host ranks) or None.
    Per-step tensor allocations are what this class exists to avoid: with torch.zeros / .to(device) / .cpu() in every step the
    caching allocator and the pageable copies stalled single steps by 10-70 ms (measured on one MI355X with RCCL, world size 1)."""

    def __init__(self, total: int, device=None, to_host: bool = True, force: bool = False):
        import torch
        import torch.distributed as dist
        self.total, self.device, self.to_host = int(total), device, bool(to_host)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self._pending = None
        if not self.active:
            return
        self.world = dist.get_world_size()
        self.per = -(-self.total // self.world)
        gpu = device is not None
        self.h_in = torch.zeros((self.per, RESULT_WIDTH), dtype=torch.float64, pin_memory=gpu)
        self.h_out = torch.zeros((self.world * self.per, RESULT_WIDTH), dtype=torch.float64, pin_memory=gpu)
        if gpu:
            self.d_in = torch.zeros((self.per, RESULT_WIDTH), dtype=torch.float64, device=device)
            self.d_out = torch.zeros((self.world * self.per, RESULT_WIDTH), dtype=torch.float64, device=device)
            self.stream = torch.cuda.Stream(device=device)
            self.stream.wait_stream(torch.cuda.current_stream(device))      # the zero-fills above ran on the current stream: order them first
            self.event = torch.cuda.Event()
        self._in_np = self.h_in.numpy()

    def start(self, local: np.ndarray) -> None:
        import torch
        import torch.distributed as dist
        local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1, RESULT_WIDTH)
        if not self.active:
            self._pending = local[: self.total].copy()
            return
        self._in_np[: local.shape[0]] = local
        self._in_np[local.shape[0]:] = 0.0
        if self.device is not None:
            with torch.cuda.stream(self.stream):
                self.d_in.copy_(self.h_in, non_blocking=True)
                dist.all_gather_into_tensor(self.d_out, self.d_in)
                if self.to_host:
                    self.h_out.copy_(self.d_out, non_blocking=True)
                self.event.record(self.stream)
        else:
            dist.all_gather_into_tensor(self.h_out, self.h_in)
        self._pending = True

    def finish(self):
        if self._pending is None:
            return None
        if not self.active:
            out, self._pending = self._pending, None
            return out
        self._pending = None
        if self.device is not None:
            self.event.synchronize()
        if not self.to_host:
            return None
        table = self.h_out.numpy()
        rows = []
        for r in range(self.world):
            first, count = shard_range(self.total, self.world, r)
            rows.append(table[r * self.per: r * self.per + count])
        return np.concatenate(rows, axis=0) if rows else np.zeros((0, RESULT_WIDTH))


_gatherers = {}


def gather_results(local: np.ndarray, total: int, device=None, to_host: bool = True, force: bool = False):
    """All-gather the per-rank result rows into the global [total, 16] table (every rank gets it; with
    ``to_host=False`` a rank only takes part in the collective and returns None — e.g. every rank but 0 in bench.py).

    `local` is this rank's [count, 16] block.  Without an initialised process group this is the
    identity (single process); with a group of ONE rank too, unless ``force`` asks for the collective
    all the same (the single-GPU box's only way to put RCCL's all-gather of a device tensor under
    test: tests/test_rccl_gpu.py, ``EDS_BENCH_FORCE_DIST=1 python bench.py``).  Shards may be ragged; rows are padded to the
    common shard size for the collective and trimmed afterwards.  (A ResultGatherer per (total, device, ...) is kept and reused;
    callers that want the collective to overlap other work use the class directly.)
    """
    from . import capi
    if device is not None and capi.torch_loaded_first is False:
        raise RuntimeError("a GPU collective needs torch.cuda, and PyTorch-ROCm only works on its own HIP runtime: import torch before the "
                           "first slam-eds_amd call in this process (libeds_hip.so was loaded first and brought /opt/rocm's libamdhip64.so.7)")
    import torch.distributed as dist
    gen = id(dist.group.WORLD) if (dist.is_available() and dist.is_initialized()) else 0
    key = (int(total), str(device), bool(to_host), bool(force), gen)
    g = _gatherers.get(key)
    if g is None:
        for k in [k for k in _gatherers if k[4] != gen]:
            del _gatherers[k]                   # buffers of a process group that no longer exists
        g = _gatherers[key] = ResultGatherer(total, device=device, to_host=to_host, force=force)
    g.start(local)
    return g.finish()


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
