cd /root/repo
( time python bench.py --steps 10 --warmup 2 --no-cpu --no-configs --no-shared --no-ref12 > gpurun_out/bench_r04c.json 2> gpurun_out/bench_r04c.err ) 2>&1 | grep real; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r04c.json').read().strip().splitlines()[-1])
print("value", d["value"], "gen s", d["input_generation_s"], "parity", d["parity"]["parity_max_se3"], d["parity"]["iteration_count_mismatches"])
PY
