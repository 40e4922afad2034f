#!/bin/bash
# Runs on the GPU box (via gpurun): for every leg of bench_detail.LEGS (the BASELINE configs beside the headline, the one-alignment
# latency shapes) a rocprofv3 kernel trace and three SEPARATE PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass;
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), each of the bare leg runner tools/run_leg.py.  tools/summarise_legs.py folds the result
# into profiles/traffic_<tag>.json ("workloads") — what bench_detail.roofline_block reads for the legs' frac_physical.
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
TAG=${1:-r06}
LEGS=${LEGS:-config2 config2_resident config3 config4_one_gpu b1_lm6 b1_ref12 dist_uniform dist_edges}
OUT=gpurun_out/prof_legs_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for leg in $LEGS; do
  steps=6; case $leg in b1_*|config4*) steps=40;; esac
  d=$OUT/$leg; mkdir -p "$d"
  python3 tools/run_leg.py $leg --steps $steps > "$d/plain.json" 2> "$d/plain.err"
  rocprofv3 --kernel-trace --stats -d "$d/trace" --output-format csv -- python3 tools/run_leg.py $leg --steps $steps > "$d/trace.json" 2> "$d/trace.err"
  rocprofv3 --pmc FETCH_SIZE -d "$d/pmc_fetch" --output-format csv -- python3 tools/run_leg.py $leg --steps $steps > "$d/fetch.json" 2> "$d/fetch.err"
  rocprofv3 --pmc WRITE_SIZE -d "$d/pmc_write" --output-format csv -- python3 tools/run_leg.py $leg --steps $steps > "$d/write.json" 2> "$d/write.err"
  rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d "$d/pmc_l2" --output-format csv -- python3 tools/run_leg.py $leg --steps $steps > "$d/l2.json" 2> "$d/l2.err"
  tail -n 2 "$d"/*.err | tail -n 12
done
python3 tools/summarise_legs.py "$TAG" > "$OUT/summarise.log" 2>&1; tail -n 30 "$OUT/summarise.log"
mkdir -p "gpurun_out/profiles_$TAG"; cp profiles/traffic_$TAG.json profiles/${TAG}_legs_* "gpurun_out/profiles_$TAG/" 2>/dev/null
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*_counter_collection.csv" -size +4M -delete
du -sh "$OUT"
