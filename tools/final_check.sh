cd /root/repo
python -m pytest tests -q -m gpu 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -2
( time python bench.py > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err ) 2>&1 | grep real; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r04_final.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype")})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "frac_must_move", "traffic", "kernel_ms")})
print("cpu_baseline", d["cpu_baseline"])
print("new frame", d["value_new_frame_per_solve"], d["roofline_new_frame_per_solve"]["frac"], d["roofline_new_frame_per_solve"]["frac_must_move"])
print("ref12", d["reference_problem"]["lm_iterations_per_s"], d["reference_problem"]["new_frame_per_solve"]["lm_iterations_per_s"])
print("resjac", d["roofline_resjac"]["kernel_ms"], d["roofline_resjac"]["frac"], d["roofline_resjac"]["resjac_plus_reduce_ms"])
print("strong", d["strong_scaling_config4"]["ms_per_step"], "gen", d["input_generation_s"], "parity", d["parity"]["parity_max_se3"], d["parity"]["rows_checked"])
print("configs", {k: (v["iterations_per_s"], v["ms_per_step"]) for k, v in d["configs"].items()})
PY
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r04_final.json').read().strip().splitlines()[-1])
print("host buffers inclusive", d.get("host_buffers_inclusive"))
PY
