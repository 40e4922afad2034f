#!/bin/bash
cd $GRAFT_REPO_ROOT
{
echo "=== pytest -m gpu"
SECONDS=0; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12; echo "(${SECONDS} s)"
echo "=== smoke"; python __graft_entry__.py smoke 2>&1 | tail -2
echo "=== bench default"
SECONDS=0; timeout 1500 python bench.py > gpurun_out/bench_full_check.json 2> gpurun_out/bench_full_check.err; echo "bench rc=$? (${SECONDS} s)"
tail -3 gpurun_out/bench_full_check.err
} > gpurun_out/full_check.log 2>&1
tail -40 gpurun_out/full_check.log
