"""The batch frame hand-over (eds_trk_set_event_frames) over thread counts and both transports (kernels reading pinned memory / copy engine),
against one eds_trk_set_event_frame per frame; plus what the box lets this process use of its CPUs.   python tools/bench_upload_batch.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "-")
H, W, B = 480, 640, 256
rng = np.random.default_rng(0)
base = [rng.standard_normal((H, W)) * 1e-2 for _ in range(8)]
for dt in (np.float64, np.float32):
    fr = [np.ascontiguousarray(base[i % 8] + i, dtype=dt) for i in range(B)]
    h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE), B, 64, H, W)
    def timed(f):
        f(); h.sync(); ts = []
        for _ in range(5):
            t = time.perf_counter(); f(); h.sync(); ts.append(time.perf_counter() - t)
        return 1e3 * float(np.median(ts))
    one = timed(lambda: [h.set_event_frame(b, fr[b]) for b in range(B)])
    print(f"{np.dtype(dt).name}: one call per frame {one:.2f} ms per {B} frames")
    for dma, streams in (("0", "2"), ("0", "1"), ("1", "1")):
        for T in (1, 2, 3, 4, 8):
            h.set_knob("EDS_UPLOAD_DMA", dma); h.set_knob("EDS_UPLOAD_THREADS", str(T)); h.set_knob("EDS_UPLOAD_STREAMS", streams)
            ms = timed(lambda: h.set_event_frames(0, fr))
            print(f"   batch call dma={dma} streams={streams} threads={T:2d}: {ms:6.2f} ms  ({B * fr[0].nbytes / ms / 1e6:.1f} GB/s of host frames, {B * H * W * 4 / ms / 1e6:.1f} GB/s over PCIe)", flush=True)
    h.close()
