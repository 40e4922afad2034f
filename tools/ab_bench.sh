#!/bin/bash
# A/B of the headline bench between two builds of the library inside ONE gpurun call (boxes differ by up to +-10 %):
#   tools/ab_bench.sh [lib_a.so] [lib_b.so] [extra bench.py flags]
# default: the round-2 binary kept beside the current one.  Prints it/s, kernel ms, shared-frame kernel ms, parity for each.
cd "$(dirname "$0")/.." || exit 1
A=${1:-slam-eds_amd/csrc/libeds_hip_r02.so}; B=${2:-slam-eds_amd/csrc/libeds_hip.so}; shift 2 2>/dev/null
mkdir -p gpurun_out
for rep in 1 2; do
for L in "$A" "$B"; do
    EDS_HIP_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --no-cpu --no-ref12 "$@" > gpurun_out/ab_$(basename $L .so)_$rep.json 2> gpurun_out/ab_$(basename $L .so)_$rep.err
    python - "$L" gpurun_out/ab_$(basename $L .so)_$rep.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:50s} {d['value']/1e6:7.3f} M it/s  kernel {d['roofline']['kernel_ms']:.3f} ms  shared {d.get('shared_frames',{}).get('kernel_ms',float('nan')):.3f} ms  "
          f"parity {d['parity_max_se3']:.2e} mism {d['parity']['iteration_count_mismatches']}  B1 {d['latency']['B1_lm6_ms']:.3f} B64 {d['latency']['B64_ms']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[2].replace('.json', '.err')).read()[-1500:])
PY
done
done
