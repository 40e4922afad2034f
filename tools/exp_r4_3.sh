cd /root/repo
( time python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r04b.json 2> gpurun_out/bench_r04b.err ) 2>&1 | grep real; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r04b.json').read().strip().splitlines()[-1])
print("value", d["value"], "gen s", d["input_generation_s"], "parity", d["parity"]["parity_max_se3"], d["parity"]["iteration_count_mismatches"])
print("strong", d["strong_scaling_config4"]["ms_per_step"])
PY
EDS_BENCH_BACKEND=gloo EDS_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 5 --warmup 1 --batch 512 --distinct 64 --no-cpu > gpurun_out/bench_r04_2rank.json 2> gpurun_out/bench_r04_2rank.err; echo "2-rank rc $?"; tail -3 gpurun_out/bench_r04_2rank.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r04_2rank.json').read().strip().splitlines()[-1])
print("2 ranks (gloo, shared GPU): n_gpus", d["n_gpus"], "value", d["value"], "strong", d["strong_scaling_config4"])
PY
