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


def test_roofline_block_follows_the_contract_and_matches_profiler_names():
    """bench.py's roofline block: `achieved` / `frac` = SURVEY 8d's ALGORITHMIC bytes per launch / kernel time / 8 TB/s (the contract's
    recipe); `traffic` / `frac_physical` = corrected bytes of the committed PMC passes (never above 1), found under the profiler's full
    template names (which print defaulted arguments the library's own name omits); null when no pass of the workload is committed."""
    import types
    b = _load_bench()
    a = types.SimpleNamespace(batch=4096, points=2000, iters=10, solver="lm6", sampling="bicubic", exec_="device", height=480, width=640)
    t = b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 1, 1>", a)
    assert t and t["source"].startswith("profiles/traffic_r") and 1.5e10 < t["bytes"] < 2.1e10 and abs(t["read_requests"] * 128 / t["bytes"] - 1.0) < 0.05
    assert b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 1, 1>", a)["bytes"] != b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 3, 1>", a)["bytes"]
    assert b.pmc_traffic("eds_fused6_kernel<9, 9, 9, 9, 9>", a) is None
    units = 4096 * 2000 * 11
    r = b.roofline_block("eds_fused6_kernel<0, 4, 512, 1, 1>", 2.6, units, 140, 84, a)
    assert r["algorithmic_bytes_per_launch"] == units * 140 and abs(r["achieved"] - units * 140 / 2.6e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert 0.7 < r["frac_physical"] <= 1.0 and abs(r["achieved_physical"] - r["traffic"] / 2.6e-3 / 1e9) < 1e-6
    assert abs(r["frac_must_move"] - units * 84 / 2.6e-3 / 1e9 / 8000) < 1e-9 and r["frac"] > r["frac_must_move"]
    assert r["traffic_over_algorithmic"] > 1.0                                          # wasted re-reads are visible at a glance
    r2 = b.roofline_block("eds_no_such_kernel", 1.0, 1000, 140, 84, a)
    assert r2["traffic"] is None and r2["frac_physical"] is None and r2["frac"] == 1000 * 140 / 1e-3 / 1e9 / 8000.0
    other = types.SimpleNamespace(**{**a.__dict__, "points": 1999})
    assert b.pmc_traffic("eds_fused6_kernel<0, 4, 512, 1, 1>", other) is None         # counters of another workload are never used


def test_compact_record_fits_the_drivers_tail():
    """BENCH_r05.json.parsed was null because the one JSON line had grown to 20 KB and the driver keeps an 8 KB tail.
    The last stdout line is now `compact_record(full)`: built here from a canned full record (a real run's, tests/golden/), it must
    stay under 4 KB, parse, and carry the contract's keys with `roofline` and `cpu_baseline`; a record bloated by an unforeseen long
    string sheds its digests rather than outgrow the limit."""
    import json
    b = _load_bench()
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_record.json")))
    assert len(json.dumps(full)) > 15000
    rec, line = b.compact_record(full)
    assert len(line) < b.COMPACT_LIMIT <= 4096 and "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity_max_se3", "reference_problem", "latency", "configs"):
        assert k in back, k
    assert back["config"]["workload"] and "model" not in back["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "frac_physical", "frac_must_move"):
        assert k in back["roofline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-4
    assert abs(back["roofline"]["achieved"] - back["roofline"]["algorithmic_bytes_per_launch"] / (back["roofline"]["kernel_ms"] * 1e-3) / 1e9) < 1.0
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert abs(back["value"] - full["value"]) / full["value"] < 1e-4 and abs(back["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-4
    bloated = dict(full, latency=dict(full["latency"], slice_ms=1.0), configs={f"config{i}": full["configs"]["config2"] for i in range(40)})
    _, line2 = b.compact_record(bloated)
    assert len(line2) < b.COMPACT_LIMIT and json.loads(line2)["roofline"]["kernel"] == back["roofline"]["kernel"]


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


def test_every_leg_of_the_record_has_committed_pmc_passes():
    """no leg's roofline may be arithmetic only.  Every workload bench_detail.py asks physical counters for
    (`roofline_block(..., workload=...)`) is a leg tools/profile_legs.sh profiles by default (`bench_detail.LEGS`), and the committed
    profiles/traffic_*.json holds FETCH_SIZE / WRITE_SIZE / request counts of it, taken at the bench's iteration count and sampler."""
    import json
    import re
    import types
    b = _load_bench()
    import bench_detail as bd
    src = open(os.path.join(ROOT, "bench_detail.py")).read()
    asked = set(re.findall(r'workload="([a-z0-9_]+)"', src)) | {f"dist_{k}" for k in ("uniform", "edges")} | set(re.findall(r'roof\("([a-z0-9_]+)"', src))
    asked.discard("dist_")               # (the distribution block builds its two names: "dist_" + layout)
    assert {"config2", "config2_resident", "config3", "config4_one_gpu", "b1_lm6", "b1_ref12", "dist_uniform", "dist_edges"} <= asked
    assert asked <= set(bd.LEGS), asked - set(bd.LEGS)
    script = open(os.path.join(ROOT, "tools", "profile_legs.sh")).read()
    default_legs = set(re.search(r"LEGS=\$\{LEGS:-([^}]*)\}", script).group(1).split())
    assert default_legs == set(bd.LEGS), default_legs ^ set(bd.LEGS)
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"traffic_r\d+\.json", f))[-1]
    w = json.load(open(os.path.join(ROOT, "profiles", newest)))["workloads"]
    a = types.SimpleNamespace(batch=4096, points=2000, iters=10, solver="lm6", sampling="bicubic", exec_="device", height=480, width=640)
    for leg in sorted(asked):
        assert leg in w and w[leg]["iterations"] == 10 and w[leg]["sampling"] == "bicubic" and w[leg]["bytes_per_step"] > 0, leg
        for k, v in w[leg]["kernels"].items():
            assert v["fetch_kb"] > 0 and v["write_kb"] > 0 and v["l2"].get("TCC_EA0_RDREQ_sum", 0) > 0 and v["avg_us"] > 0, (leg, k)
            if leg != "config3":          # (one kernel per step: the block reads it by name; config3's four levels are read as the step's total)
                r = b.roofline_block(k, v["avg_us"] * 1e-3, 1000, 140, 84, a, workload=leg)
                assert r["traffic"] and r["traffic_source"] == "profiles/" + newest and 0.0 < r["frac_physical"] <= 1.0, (leg, k)
    assert bd._step_traffic("config3", a)[0] == w["config3"]["bytes_per_step"]
    # the headline's own passes, and the reference problem's kernel of this round
    t = json.load(open(os.path.join(ROOT, "profiles", newest)))
    assert any(k.startswith("eds_fused6_kernel<0, 4, 512, 1, 1") for k in t["kernels"]) and any("256, 736" in k for k in t["kernels"])
