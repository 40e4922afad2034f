#!/usr/bin/env python3
"""Runs ONE leg of bench_detail.py (a BASELINE config or a latency shape) bare: the leg's own set-up, then `--steps` passes of its
step — what tools/profile_legs.sh puts under rocprofv3 (kernel trace + separate PMC passes) so that profiles/traffic_*.json holds the
physical bytes of exactly the shapes bench.py reports.  No child processes (a profiled process must not start any), no oracle.

    python3 tools/run_leg.py config2 --steps 6
"""
import argparse
import importlib
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_detail as bd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("leg", choices=sorted(bd.LEGS))
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")
a = types.SimpleNamespace(iters=args.iters, lambda0=0.01, sampling="bicubic", solver="lm6", exec_="device", height=480, width=640, points=2000)
L = bd.LEGS[args.leg](capi, synth, a)
for _ in range(args.steps):
    L.step()
out = {"leg": args.leg, "steps": args.steps, "iterations": a.iters, "sampling": a.sampling}
if hasattr(L, "h"):
    out["kernel"] = L.h.last_launch()["kernel"]
    out["kernel_us_last"] = L.h.info(0)["device_time_us"]
L.finish()
print(json.dumps(out))
