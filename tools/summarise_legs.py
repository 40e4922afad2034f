#!/usr/bin/env python3
"""Folds the rocprofv3 output of tools/profile_legs.sh (gpurun_out/prof_legs_<tag>/<leg>/) into profiles/traffic_<tag>.json's
"workloads" section (per leg: per-kernel calls, average duration, FETCH_SIZE / WRITE_SIZE per launch, TCC counters, and the STEP's
totals over all its solver launches) and writes profiles/<tag>_legs_summary.md.

    python3 tools/summarise_legs.py r06
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", f"prof_legs_{tag}")
dst = os.path.join(ROOT, "profiles")
SOLVERS = ("eds_fused6_kernel", "eds_fused12_kernel", "eds_stream6_kernel")


def short(name):
    for k in SOLVERS + ("eds_resjac_kernel", "eds_reduce_kernel", "eds_gram_kernel"):
        if k in name:
            return k + name[name.find("<"):name.find(">") + 1] if "<" in name else k
    return name[:40]


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def counters(path):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


tpath = os.path.join(dst, f"traffic_{tag}.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {"tag": tag}
traffic.setdefault("workloads", {})
lines = [f"# rocprofv3 digest of the legs beside the headline ({tag})", "",
         "one step of each leg = what `tools/run_leg.py <leg>` repeats (bench_detail.LEGS); bytes are corrected for gfx950 (2 x FETCH_SIZE + WRITE_SIZE)", "",
         "| leg | kernel | launches / step | avg us | corrected bytes / launch | read requests / launch | corrected GB/s | of 8 TB/s |", "|---|---|---|---|---|---|---|---|"]
for d in sorted(glob.glob(os.path.join(src, "*"))):
    if not os.path.isdir(d):
        continue
    leg = os.path.basename(d)
    try:
        run = json.loads(open(os.path.join(d, "plain.json")).read().strip().splitlines()[-1])
        stats = {short(r["Name"]): r for r in csv.DictReader(open(newest(os.path.join(d, "trace", "*", "*_kernel_stats.csv"))))}
        fetch = counters(newest(os.path.join(d, "pmc_fetch", "*", "*_counter_collection.csv")))
        write = counters(newest(os.path.join(d, "pmc_write", "*", "*_counter_collection.csv")))
    except (ValueError, OSError, IndexError) as ex:
        print(f"{leg}: incomplete ({ex})")
        continue
    try:
        l2 = counters(newest(os.path.join(d, "pmc_l2", "*", "*_counter_collection.csv")))
    except ValueError:
        l2 = {}
    w = {"iterations": run["iterations"], "sampling": run["sampling"], "steps": run["steps"], "kernels": {}}
    step_bytes, step_req, step_us = 0.0, 0.0, 0.0
    for k, r in stats.items():
        if not k.startswith(SOLVERS):
            continue
        f, wr = fetch.get(k, {}).get("FETCH_SIZE", []), write.get(k, {}).get("WRITE_SIZE", [])
        if not f or not wr:
            continue
        fkb, wkb, avg_us, calls = sum(f) / len(f), sum(wr) / len(wr), float(r["AverageNs"]) / 1e3, int(r["Calls"])
        m = {c: sum(v) / len(v) for c, v in l2.get(k, {}).items()}
        if m.get("TCC_REQ_sum"):
            m["hit_fraction"] = m.get("TCC_HIT_sum", 0.0) / m["TCC_REQ_sum"]
        corr = (2.0 * fkb + wkb) * 1024.0
        per_step = calls / run["steps"]
        w["kernels"][k] = {"calls": calls, "launches_per_step": per_step, "avg_us": avg_us, "fetch_kb": fkb, "write_kb": wkb, "hbm_bytes_corrected": corr, "l2": m}
        step_bytes += corr * per_step; step_us += avg_us * per_step; step_req += (m.get("TCC_EA0_RDREQ_sum") or 0.0) * per_step
        lines.append(f"| {leg} | {k} | {per_step:g} | {avg_us:.1f} | {int(corr)} | {int(m.get('TCC_EA0_RDREQ_sum') or 0)} | {corr / avg_us / 1e3:.1f} | {corr / avg_us / 1e3 / 8000.0:.3f} |")
    w.update({"bytes_per_step": step_bytes, "read_requests_per_step": step_req or None, "solver_kernel_us_per_step": step_us})
    traffic["workloads"][leg] = w
    print(f"{leg}: {len(w['kernels'])} solver kernels, {step_bytes / 1e6:.2f} MB per step, {step_us:.1f} us per step")
json.dump(traffic, open(tpath, "w"), indent=1)
open(os.path.join(dst, f"{tag}_legs_summary.md"), "w").write("\n".join(lines) + "\n")
