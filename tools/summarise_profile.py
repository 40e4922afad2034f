#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile.sh (gpurun_out/prof_<tag>/) into the tracked summaries
under profiles/: the kernel-stats CSV as rocprofv3 wrote it, the per-kernel PMC means, a traffic JSON
that bench.py reads for its `roofline.traffic` field, and a short markdown digest.

    python tools/summarise_profile.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    for k in ("eds_fused6_kernel", "eds_fused12_kernel", "eds_stream6_kernel", "eds_resjac_kernel", "eds_reduce_kernel", "eds_gram_kernel", "eds_model_kernel"):
        if k in name:
            return k + name[name.find("<"):name.find(">") + 1] if "<" in name else k
    return name[:40]


def newest(pattern):
    # gpurun merges results back without deleting what an earlier call left: take the latest file only
    return max(glob.glob(pattern), key=os.path.getmtime)


stats_csv = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = {short(r["Name"]): r for r in csv.DictReader(open(stats_csv))}

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("fetch", "write"):
    for f in [newest(os.path.join(src, f"pmc_{which}", "*", "*_counter_collection.csv"))]:
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))

l2 = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(newest(os.path.join(src, "pmc_l2", "*", "*_counter_collection.csv")))):
        l2[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
except ValueError:
    pass

bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
cfg = bench["config"]
out = {"tag": tag, "bench_config": cfg, "bench_value": bench["value"], "kernels": {}}
lines = [f"# rocprofv3 digest {tag}", "",
         f"command: `python3 bench.py {os.environ.get('BENCH_ARGS', '--steps 10 --warmup 2 --no-cpu --no-shared --no-configs')}` "
         f"({cfg['alignments_per_gpu']} alignments x {cfg['points']} points, {cfg['iterations']} {cfg['solver']} iterations, {cfg['sampling']})",
         f"bench value (un-profiled run): {bench['value']:.4g} {bench['unit']}", "",
         "| kernel | calls | avg us (kernel-trace) | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch | raw bytes/launch | corrected bytes/launch (2 x FETCH + WRITE) | corrected GB/s | of 8 TB/s |",
         "|---|---|---|---|---|---|---|---|---|"]
for k, r in stats.items():
    if not k.startswith("eds_"):
        continue
    avg_us = float(r["AverageNs"]) / 1e3
    f = pmc.get(k, {}).get("FETCH_SIZE", [])
    w = pmc.get(k, {}).get("WRITE_SIZE", [])
    fkb = sum(f) / len(f) if f else None
    wkb = sum(w) / len(w) if w else None
    raw = (fkb + wkb) * 1024 if (fkb is not None and wkb is not None) else None
    corr = (2.0 * fkb + wkb) * 1024 if (fkb is not None and wkb is not None) else None      # MI355X_MICROARCH.md: FETCH_SIZE tallies a 128-byte request at 64 B on gfx950
    out["kernels"][k] = {"calls": int(r["Calls"]), "avg_us": avg_us, "fetch_kb": fkb, "write_kb": wkb, "raw_hbm_bytes": raw, "hbm_bytes_corrected": corr,
                         "corrected_GBps": None if corr is None else corr / avg_us / 1e3, "frac_of_8TBps": None if corr is None else corr / avg_us / 1e3 / 8000.0}
    if k in l2:
        m = {c: sum(v) / len(v) for c, v in l2[k].items()}
        if m.get("TCC_REQ_sum"):
            m["hit_fraction"] = m.get("TCC_HIT_sum", 0.0) / m["TCC_REQ_sum"]
        out["kernels"][k]["l2"] = m
    lines.append(f"| {k} | {r['Calls']} | {avg_us:.1f} | {fkb if fkb is None else round(fkb, 1)} | "
                 f"{wkb if wkb is None else round(wkb, 1)} | {raw if raw is None else int(raw)} | {corr if corr is None else int(corr)} | "
                 f"{'' if corr is None else round(corr / avg_us / 1e3, 1)} | {'' if corr is None else round(corr / avg_us / 1e3 / 8000.0, 3)} |")
lines += ["", "L2 (separate `--pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum` pass; per launch, chip-wide sums):", "",
          "| kernel | TCC_REQ | TCC_HIT | TCC_MISS | TCC_EA0_RDREQ | hit fraction |", "|---|---|---|---|---|---|"]
for k, v in out["kernels"].items():
    if "l2" in v:
        m = v["l2"]
        lines.append(f"| {k} | {m.get('TCC_REQ_sum', 0):.4g} | {m.get('TCC_HIT_sum', 0):.4g} | {m.get('TCC_MISS_sum', 0):.4g} | "
                     f"{m.get('TCC_EA0_RDREQ_sum', 0):.4g} | {m.get('hit_fraction', float('nan')):.3f} |")
lines += ["",
          "FETCH_SIZE / WRITE_SIZE were collected in two separate `--pmc` passes (they do not fit one pass on gfx950).",
          "Raw = (FETCH_SIZE + WRITE_SIZE) x 1024.  Calibration on known byte counts (MI355X_MICROARCH.md asks for it):",
          "* streaming reads are tallied at exactly 1/2 — `eds_reduce_kernel<6>` reads 28 B/point (57.3 MB/launch) and FETCH_SIZE",
          "  says 28.3 MB — so the coalesced SoA part of every kernel here must be doubled;",
          "* scattered 16..64-byte loads are tallied at 64 B per distinct 64-B sector (tools/ubench_gather.hip under",
          "  `--pmc FETCH_SIZE`: 64 Mi random locations -> 4.32 GB whether 16, 32 or 64 B are read per location), i.e. correctly,",
          "  except that two sectors of one 128-B line fetched together are tallied once;",
          "* WRITE_SIZE of the residual/Jacobian kernel is exactly 28 B/point (r + six Jacobian planes).",
          "So raw is a lower bound of the true traffic through the fabric and (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — the guide's gfx950 correction, what",
          "bench.py's `roofline.traffic` / `roofline.frac` use — an upper bound; for the gather kernels the two bounds meet the request counts:",
          "TCC_EA0_RDREQ x 128 B = 2 x FETCH_SIZE to within a percent (every request of these kernels is a 128-byte line fill).",
          "`eds_fused12_kernel` (bench.py's REF12 measurement) writes the candidate residuals of every evaluation (8 KB per alignment and",
          "evaluation) plus the accepted copies: that is its WRITE_SIZE; `eds_fused6_kernel` (the headline) keeps them in registers and writes",
          "the residuals once."]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
tpath = os.path.join(dst, f"traffic_{tag}.json")
if os.path.exists(tpath):                       # (the legs' section, written by tools/summarise_legs.py, survives a re-run of the headline's profile)
    try:
        prev = json.load(open(tpath))
        if prev.get("workloads"):
            out["workloads"] = prev["workloads"]
    except Exception:
        pass
json.dump(out, open(tpath, "w"), indent=1)
print("\n".join(lines))
