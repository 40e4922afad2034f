#!/bin/bash
# The three REF12 batch shapes (tools/bench_ref12_onecu.py) under rocprofv3: kernel trace, fabric read requests + L2 counters, SQ counters.
# Output: gpurun_out/ref12_shapes_<tag>.txt (copied to profiles/ by hand).
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"; R=$PWD
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=$R/gpurun_out/ref12_shapes_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
python3 tools/bench_ref12_onecu.py 4096 6 > "$OUT/plain.txt" 2> "$OUT/plain.err"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 tools/bench_ref12_onecu.py 4096 3 noparity > "$OUT/trace.txt" 2> "$OUT/trace.err"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d "$OUT/l2" --output-format csv -- python3 tools/bench_ref12_onecu.py 4096 3 noparity > "$OUT/l2.txt" 2> "$OUT/l2.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- python3 tools/bench_ref12_onecu.py 4096 3 noparity > "$OUT/fetch.txt" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- python3 tools/bench_ref12_onecu.py 4096 3 noparity > "$OUT/write.txt" 2> "$OUT/write.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OUT/sq1" --output-format csv -- python3 tools/bench_ref12_onecu.py 4096 3 noparity > "$OUT/sq1.txt" 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/sq2" --output-format csv -- python3 tools/bench_ref12_onecu.py 4096 3 noparity > "$OUT/sq2.txt" 2> "$OUT/sq2.err"
python3 - "$OUT" <<'PY' > gpurun_out/ref12_shapes_${TAG}.txt
import csv, glob, collections, sys
out = sys.argv[1]
print(open(f"{out}/plain.txt").read())
def short(n):
    return "eds_fused12_kernel" + n[n.find("<"):n.find(">") + 1] if "eds_fused12_kernel" in n else None
fs = glob.glob(f"{out}/trace/*/*kernel_stats.csv")
if fs:
    print("== kernel trace (rocprofv3 --kernel-trace --stats)")
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Name"])
        if k: print(f"{k:52s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
for p in ("l2", "fetch", "write", "sq1", "sq2"):
    fs = glob.glob(f"{out}/{p}/*/*counter_collection.csv")
    if not fs: print(p, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== pass {p} (per launch, chip-wide)")
    for k, d in sorted(acc.items()):
        print(f"{k:52s} launches {len(next(iter(d.values()))):4d}  " + "  ".join(f"{c} {sum(v)/len(v):.5g}" for c, v in sorted(d.items())))
PY
find "$OUT" -name "*.csv" -size +1M -delete
cat gpurun_out/ref12_shapes_${TAG}.txt
