#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then two
# separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Everything lands under gpurun_out/ (scratch); tools/summarise_profile.py turns it into profiles/*.
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_${1:-r02}
rm -rf "$OUT"; mkdir -p "$OUT"
# (--gen-workers 0: a process the profiler has attached to must not start children; 256 distinct alignments keep the in-process generation short)
# (--gen-workers 0 is ALWAYS appended — also behind a BENCH_ARGS override; bench.py forces it by itself when it sees the profiler)
ARGS="${BENCH_ARGS:---steps 10 --warmup 2 --no-cpu --no-shared --no-configs --distinct 256} --gen-workers 0"
python3 bench.py --steps 10 --warmup 2 --no-cpu --no-shared --no-configs > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" --output-format csv -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" --output-format csv -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
# L2 behaviour of the gather: requests, hits, misses and what goes on to the fabric (4 TCC slots = one pass)
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d "$OUT/pmc_l2" --output-format csv -- python3 bench.py $ARGS > "$OUT/bench_l2.json" 2> "$OUT/l2.err"
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
# gpurun copies back at most 64 MiB: summarise here, keep the digests (they go to profiles/ by hand afterwards), drop the per-dispatch tables
python3 tools/summarise_profile.py "${1:-r02}" > "$OUT/summarise.log" 2>&1
mkdir -p "gpurun_out/profiles_${1:-r02}"
cp profiles/${1:-r02}_* profiles/traffic_${1:-r02}.json "gpurun_out/profiles_${1:-r02}/" 2>/dev/null
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*_counter_collection.csv" -size +4M -delete
du -sh "$OUT"
