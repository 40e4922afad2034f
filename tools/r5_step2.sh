#!/bin/bash
cd $GRAFT_REPO_ROOT
{
echo "=== stamps G=8 / G=1 (B1 lm6)"
for G in 1 2 8; do
echo "--- EDS_LM6_GROUPS=$G stamps1"; EDS_LM6_GROUPS=$G EDS_HIP_LIB=$PWD/slam-eds_amd/csrc/libeds_hip_stamps.so python tools/bench_b1.py lm6 30 2>&1 | grep -E "stamps|kernel median" | tail -3
echo "--- EDS_LM6_GROUPS=$G stamps2"; EDS_LM6_GROUPS=$G EDS_HIP_LIB=$PWD/slam-eds_amd/csrc/libeds_hip_stamps2.so python tools/bench_b1.py lm6 30 2>&1 | grep -E "stamps|kernel median" | tail -3
done
echo "=== pytest groups + launch info"
timeout 600 python -m pytest tests/test_groups_gpu.py tests/test_launch_info_gpu.py tests/test_team_timeout_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "=== bench quick"
timeout 900 python bench.py --steps 10 --warmup 2 --cpu-seconds 4 > gpurun_out/bench_r5_a.json 2> gpurun_out/bench_r5_a.err; echo bench rc=$?
tail -5 gpurun_out/bench_r5_a.err
} > gpurun_out/r5_step2.log 2>&1
tail -60 gpurun_out/r5_step2.log
