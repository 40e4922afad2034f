#!/bin/bash
# SQ counter passes of the bench (per-kernel, per-launch averages) -> gpurun_out/sq_<tag>.txt; two separate --pmc passes.
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
R=$PWD
export TMPDIR=/tmp
TAG=${1:-r04}
OUT=$R/gpurun_out/sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 3 --warmup 2 --no-cpu --no-shared --no-configs --distinct 256 --gen-workers 0"
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OUT/p1" --output-format csv -- python3 $R/bench.py $ARGS > "$OUT/b1.json" 2> "$OUT/b1.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/p2" --output-format csv -- python3 $R/bench.py $ARGS > "$OUT/b2.json" 2> "$OUT/b2.err"
cd $R
python3 - "$OUT" <<'PY' > gpurun_out/sq_${TAG}.txt
import csv, glob, collections, sys
out = sys.argv[1]
def short(n):
    for k in ("eds_fused6_kernel", "eds_fused12_kernel", "eds_resjac_kernel", "eds_reduce_kernel"):
        if k in n: return k + n[n.find("<"):n.find(">") + 1]
    return None
for p in ("p1", "p2"):
    fs = glob.glob(f"{out}/{p}/*/*counter_collection.csv")
    if not fs: print(p, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== pass {p}")
    for k, d in sorted(acc.items()):
        print(f"{k:52s} launches {len(next(iter(d.values()))):4d}  " + "  ".join(f"{c} {sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
find "$OUT" -name "*.csv" -size +1M -delete
cat gpurun_out/sq_${TAG}.txt
