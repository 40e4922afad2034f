#!/bin/bash
cd $GRAFT_REPO_ROOT
{
echo "=== pytest frames batch"
timeout 900 python -m pytest tests/test_frames_batch_gpu.py tests/test_groups_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "=== bench quick"
timeout 1200 python bench.py --steps 10 --warmup 2 --cpu-seconds 12 > gpurun_out/bench_r5_b.json 2> gpurun_out/bench_r5_b.err; echo bench rc=$?
tail -5 gpurun_out/bench_r5_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r5_b.json').read().strip().splitlines()[-1])
print("value", d["value"], "resident", d.get("value_resident_frames"))
print("host_buffers", json.dumps(d.get("host_buffers_inclusive"), indent=0)[:1500])
for k in ("cpu_baseline","cpu_baseline_fast","cpu_baseline_ref12"): print(k, json.dumps(d.get(k))[:900])
print("prep", d["resident_frames"]["frame_layout_prep"])
PY
} > gpurun_out/r5_step3.log 2>&1
tail -60 gpurun_out/r5_step3.log
