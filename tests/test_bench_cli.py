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


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_physical_roofline_never_exceeds_one_and_matches_profiler_names(tmp_path, monkeypatch):
    """bench.py's roofline block: `frac` is physical — corrected bytes of the committed PMC passes / kernel time / 8 TB/s —,
    found under the profiler's full template names (which print defaulted arguments the library's own name omits), and falls back to the
    must-move bytes when no pass of the workload is committed."""
    import json
    import types
    b = _load_bench()
    a = types.SimpleNamespace(batch=4096, points=2000, iters=10, solver="lm6", sampling="bicubic", exec_="device", height=480, width=640)
    t = b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 1, 1>", a)
    assert t and t["source"].startswith("profiles/traffic_r") and 1.5e10 < t["bytes"] < 2.1e10 and abs(t["read_requests"] * 128 / t["bytes"] - 1.0) < 0.05
    assert b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 1, 1>", a)["bytes"] != b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 3, 1>", a)["bytes"]
    assert b.pmc_traffic("eds_fused6_kernel<9, 9, 9, 9, 9>", a) is None
    r = b.physical_roofline("eds_fused6_kernel<0, 4, 512, 1, 1>", 2.6, 4096 * 2000 * 11, 140, 84, a)
    assert r["basis"].startswith("physical") and 0.7 < r["frac"] <= 1.0 and abs(r["achieved"] - r["traffic"] / 2.6e-3 / 1e9) < 1e-6
    assert abs(r["frac_must_move"] - 4096 * 2000 * 11 * 84 / 2.6e-3 / 1e9 / 8000) < 1e-9 and r["frac_credit_8d"] > r["frac_must_move"]
    r2 = b.physical_roofline("eds_no_such_kernel", 1.0, 1000, 140, 84, a)
    assert r2["traffic"] is None and r2["frac"] == r2["frac_must_move"] and "must-move" in r2["basis"]
    other = types.SimpleNamespace(**{**a.__dict__, "points": 1999})
    assert b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 1, 1>", other) is None         # counters of another workload are never used


def test_usable_cpus_honours_the_cgroup_quota(monkeypatch):
    b = _load_bench()
    n, quota = b._usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) and (quota is None or n <= int(quota + 0.5) or n == 1)
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            import io
            return io.StringIO("400000 100000\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    n2, q2 = b._usable_cpus()
    assert q2 == 4.0 and n2 == min(4, len(os.sched_getaffinity(0)))
