#!/bin/bash
# A/B of the one-alignment latency between two builds inside ONE gpurun call: tools/ab_latency.sh lib_a.so lib_b.so [ref12|lm6] [reps]
cd "$(dirname "$0")/.." || exit 1
A=${1:-slam-eds_amd/csrc/libeds_hip_prev.so}; B=${2:-slam-eds_amd/csrc/libeds_hip.so}; W=${3:-ref12}; R=${4:-300}
for rep in 1 2 3; do
for L in "$A" "$B"; do
    printf "%-44s " "$L"; EDS_HIP_LIB=$PWD/$L python tools/bench_b1.py $W $R
done
done
