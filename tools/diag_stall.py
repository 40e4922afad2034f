"""Looks for the rare long step of a 64-alignment team launch (strong_scaling_config4 of bench.py showed one ~75 ms step in 2 of 5 runs):
times set_states / optimize_batch / results separately over many steps and prints every step above 1 ms with the library's own
diagnostics (info flags, device time, kernel)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = 64
als = [synth.make_alignment(5000 + b) for b in range(B)]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
for i, x in enumerate(als): h.set_alignment(i, x)
P0 = np.stack([x.p0 for x in als]); Q0 = np.stack([x.q0 for x in als]); V0 = np.stack([x.v0 for x in als])
long_steps = 0
tt = []
for k in range(steps):
    t0 = time.perf_counter(); h.set_states(0, P0, Q0, V0)
    t1 = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True)
    t2 = time.perf_counter(); tab = h.results(0, B)
    t3 = time.perf_counter()
    tt.append(t3 - t0)
    if t3 - t0 > 1e-3:
        inf = h.info(0); long_steps += 1
        print(f"step {k}: set_states {1e3*(t1-t0):.3f} ms optimize {1e3*(t2-t1):.3f} ms results {1e3*(t3-t2):.3f} ms  flags {inf['flags']} device_time_us {inf['device_time_us']:.1f} "
              f"kernel {h.last_launch()['kernel']} flags of all slots {sorted(set(h.info(b)['flags'] for b in range(B)))}", flush=True)
print(f"{steps} steps, median {1e6*np.median(tt):.1f} us, {long_steps} above 1 ms")
h.close()
