#!/usr/bin/env python3
"""profiles/r02_rows_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/profile_rows.py) -> profiles/r02_rows_summary.md:
per kernel the call count, average / min / max duration and the algorithmic bytes one launch moves.

    python tools/summarise_rows.py [tag]
"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rows = {}
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_rows_kernel_stats.csv"))):
    n = r["Name"]
    key = n[n.find("k_"):].split("(")[0] if "k_" in n else n.split("(")[0]
    rows[key] = r
notes = [
    ("k_store_rowmajor", 0.62, "a band of 120 rows: 0.31 MB fp32 read from pinned host memory over PCIe, 0.31 MB of tiles out (set_event_frame: 4 bands per frame, behind the host's narrowing)"),
    ("k_vote", 3.70, "100 k events: 5 B each read from pinned host memory over PCIe, 4 fp64 atomic adds each (32 B)"),
    ("k_blur3<true, 1>", 4.92, "fp64 vote image in, level-0 plane out, sum of squares on the way (one plain level: no k_levels launch)"),
    ("k_blur3<false, 1>", 4.92, "fp64 image in + out"),
    ("k_levels", 4.92, "per level: fp64 image in + plane out (+ (2i+1)^2 window re-reads from L2 for level i >= 1); level 0 also clears the vote image"),
    ("k_store_levels", 3.72, "per level: fp64 plane in, tiled fp32 frame out"),
    ("k_vote_batch", 5.76, "32 slices of 20 k events: 5 B each from pinned host memory over PCIe (3.2 MB), 4 fp64 atomic adds each"),
    ("k_blur3<true, 4>", 157.3, "32 images: fp64 vote images in, level-0 planes out, sums of squares on the way; 4 rows per thread"),
    ("k_store_tiles_batch", 119.1, "32 images: fp64 planes in, tiled fp32 frames out; a 16-byte tile row per thread"),
    ("k_update_points", 0.22, "2 000 points: 9 planes in/out in HBM; coordinates, tracks, kept indices (36 B/point) to pinned host memory"),
    ("k_loss_param", 0.51, "64 alignments x 2 000 residuals: radix select of the median, then of the MAD (residual plane re-read per pass, L2)"),
    ("k_select", 3.69, "fp64 magnitude in, candidates out: one bitonic sort per 20 x 20 cell in LDS"),
    ("k_minmax", 0.31, "image in"),
    ("k_nearest_part", 0.06, "886 candidates x 3 000 depth points, 16 chunks staged through LDS"),
    ("k_nearest_merge", 0.25, "16 partial winners per candidate in, inverse depth + distance out"),
    ("k_log", 2.76, "u8 image in, fp64 log image out"),
    ("k_pyr_down", 1.58, "level l-1 tiled fp32 in, level l out (first level: 1.27 MB + 0.33 MB)"),
    ("k_sobel", 9.83, "fp64 in, 3 fp64 planes out"),
    ("k_prepare<unsigned char>", 3.99, "960x1280x3 u8 in, 640x480 u8 out"),
    ("k_weights_clean", None, ""), ("k_emit", None, ""), ("k_fill_slot", None, ""), ("k_scan_cells", None, ""), ("k_mirror_rows", None, ""),
]
out = [f"# rocprofv3 kernel-trace of the rows around the solve (SURVEY 8f, configs[3]) — {tag}", "",
       "command: `rocprofv3 --kernel-trace --stats -- python3 tools/profile_rows.py` (640x480; 100 k events; 64 alignments of 2 000 points, 64 slices of 20 k events in one batched call; one VGA keyframe",
       f"with a 3 000-point depth map, once from a 960x1280 RGB image; a 4-level pyramid).  Full CSV: `{tag}_rows_kernel_stats.csv`.", "",
       "| kernel | calls | avg us | min us | max us | algorithmic bytes per launch | GB/s at the average |", "|---|---|---|---|---|---|---|"]
for k, mb, note in notes:
    if k not in rows:
        continue
    r = rows[k]; avg = float(r["AverageNs"]) / 1e3
    out.append("| %s | %s | %.1f | %.1f | %.1f | %s | %s |" % (k, r["Calls"], avg, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
               ("%.2f MB (%s)" % (mb, note)) if mb else "", ("%.0f" % (mb * 1e6 / avg / 1e3)) if mb else ""))
tail = open(os.path.join(ROOT, "profiles", f"{tag}_rows_notes.md")).read() if os.path.exists(os.path.join(ROOT, "profiles", f"{tag}_rows_notes.md")) else ""
open(os.path.join(ROOT, "profiles", f"{tag}_rows_summary.md"), "w").write("\n".join(out) + "\n\n" + tail)
print("\n".join(out))
