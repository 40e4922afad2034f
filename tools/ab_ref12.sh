#!/bin/bash
# A/B of the REF12 shapes between two builds inside ONE gpurun call: tools/ab_ref12.sh [base.so] [new.so] [sizes...]
cd "$(dirname "$0")/.." || exit 1
A=${1:-slam-eds_amd/csrc/libeds_hip_base.so}; B=${2:-slam-eds_amd/csrc/libeds_hip.so}; shift 2 2>/dev/null
for rep in 1 2; do for L in "$A" "$B"; do echo "== $L (rep $rep)"; EDS_HIP_LIB=$PWD/$L python tools/bench_ref12_shapes.py "$@" 2>&1 | grep "B="; done; done
