#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "=== normal lib"; 
python tools/bench_b1.py lm6 300
python tools/bench_b1.py ref12 300
echo "=== stamps lib";
EDS_HIP_LIB=$PWD/slam-eds_amd/csrc/libeds_hip_stamps.so python tools/bench_b1.py lm6 40 2>&1 | tail -12
EDS_HIP_LIB=$PWD/slam-eds_amd/csrc/libeds_hip_stamps.so python tools/bench_b1.py ref12 10 2>&1 | grep -v "^$" | tail -14
echo "=== B8/B16/B64 team sizes"
python tools/bench_b64.py 8 16 64
} > gpurun_out/r5_base.log 2>&1
tail -60 gpurun_out/r5_base.log
