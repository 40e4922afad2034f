cd /root/repo
python -m pytest tests -q -m gpu 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -2
SECONDS=0; python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; rc=$?; echo "bench rc $rc (${SECONDS} s)"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype")})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "frac_must_move", "frac_credit_8d", "traffic", "kernel_ms")})
print("cpu_baseline", d["cpu_baseline"])
print("resident frames", d.get("value_resident_frames"))
print("ref12", d["reference_problem"]["lm_iterations_per_s"], d["reference_problem"]["resident_frames"]["lm_iterations_per_s"])
print("resjac", d["roofline_resjac"]["kernel_ms"], d["roofline_resjac"]["frac"], "reduce", d["roofline_reduce"]["kernel_ms"], d["roofline_reduce"]["frac"])
print("strong", d["strong_scaling_config4"]["ms_per_step"], "gen", d["input_generation_s"], "parity", d["parity"]["parity_max_se3"], d["parity"]["rows_checked"])
print("configs", {k: (v["iterations_per_s"], v["ms_per_step"]) for k, v in d["configs"].items()})
PY
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print("host buffers inclusive", d.get("host_buffers_inclusive"))
PY
