#!/bin/bash
# A/B of the batch kernels between two builds inside ONE gpurun call: headline (strips), first-solve (tiles), REF12 on both layouts.
#   tools/ab_layouts.sh a.so b.so
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for rep in 1 2; do for L in "$@"; do
    EDS_HIP_LIB=$PWD/$L python bench.py --steps 10 --warmup 2 --no-cpu --no-configs --no-shared --distinct 256 > gpurun_out/ab_$(basename $L .so)_$rep.json 2> gpurun_out/ab_$(basename $L .so)_$rep.err
    python - "$L" gpurun_out/ab_$(basename $L .so)_$rep.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    rp = d.get("reference_problem", {})
    print(f"{sys.argv[1]:44s} strips {d['value']/1e6:6.2f} M it/s (kernel {d['roofline']['kernel_ms']:.3f} ms)  tiles {d['value_new_frame_per_solve']/1e6:6.2f} M ({d['roofline_new_frame_per_solve']['kernel_ms']:.3f} ms)  "
          f"REF12 {rp.get('lm_iterations_per_s', 0)/1e6:6.2f} M ({rp.get('kernel_ms', 0):.3f} ms) / tiles {rp.get('new_frame_per_solve', {}).get('lm_iterations_per_s', 0)/1e6:6.2f} M ({rp.get('new_frame_per_solve', {}).get('kernel_ms', 0):.3f} ms)  parity {d['parity_max_se3']:.1e}")
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[2].replace('.json', '.err')).read()[-1500:])
PY
done; done
