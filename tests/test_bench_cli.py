"""bench.py's launcher contract (VERDICT r1 #1d / ADVICE r1): `--gpus N` is honoured or the run fails loudly — it never
silently runs one GPU and prints n_gpus 1.  CPU-only: no GPU call is made on these paths."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_gpus_n_spawns_ranks_and_fails_without_n_devices():
    """No launcher, --gpus 2: bench.py starts two ranks itself (torch.distributed.run, before any GPU call); with fewer than two
    visible GPUs every rank refuses, and the parent exits non-zero without printing a result line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=_env(), timeout=600)
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and '"n_gpus"' not in r.stdout
    r = subprocess.run([sys.executable, BENCH, "--gpus", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode != 0
