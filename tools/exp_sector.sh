cd /root/repo
R=$PWD
./tools/ubench_sector.bin 8 64 2>&1 | tail -16
export TMPDIR=/tmp
mkdir -p gpurun_out/sector
cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum --kernel-trace --stats -d $R/gpurun_out/sector/req --output-format csv -- $R/tools/ubench_sector.bin 8 32 > /dev/null 2> $R/gpurun_out/sector/req.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/sector/fetch --output-format csv -- $R/tools/ubench_sector.bin 8 32 > /dev/null 2> $R/gpurun_out/sector/fetch.err
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("req", "fetch"):
    fs = glob.glob(f"gpurun_out/sector/{tag}/*/*counter_collection.csv")
    if not fs: print(tag, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        # the timed launches are the larger ones (iters = 32): take the max per kernel
        print(tag, k[:40], {c: max(v) for c, v in d.items()})
PY
