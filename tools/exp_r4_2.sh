cd /root/repo
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err; echo "bench rc $?"; tail -c 600 gpurun_out/bench_r04a.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r04a.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "kernel", d["roofline"]["kernel"], d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
print("gen s", d["input_generation_s"], "distinct", d["config"]["distinct_alignments"], "parity", d["parity"])
print("new frame", d.get("value_new_frame_per_solve"), d.get("roofline_new_frame_per_solve"))
print("strong", d.get("strong_scaling_config4"))
print("resjac", d["roofline_resjac"])
print("ref12", {k: d["reference_problem"][k] for k in ("lm_iterations_per_s", "kernel_ms", "kernel", "new_frame_per_solve")})
print("config4", d["configs"]["config4_one_gpu"]["ms_per_step"], d["latency"])
PY
python -m pytest tests/test_batch_configs_gpu.py -q -m gpu -k "bench_shape or ref12_batch" 2>&1 | tail -3
