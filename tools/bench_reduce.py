"""eds_reduce_kernel<6, PPL>: time of the reduction alone (resjac + reduce minus resjac, HIP events inside eds_trk_bench_eval) for 4 and 8
points per lane, on the bench's batch shape with the strip copies made; and the sums of one slot against each other."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
als = [synth.make_alignment(5000 + i) for i in range(8)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_HOST, num_blocks=1, max_num_iterations=3), B, 2000, 480, 640)
for b in range(B):
    a = als[b % 8]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 8])
h.prepare_frames(0, B)
p0 = np.stack([als[b % 8].p0 for b in range(B)]); q0 = np.stack([als[b % 8].q0 for b in range(B)]); v0 = np.stack([als[b % 8].v0 for b in range(B)])
h.set_states(0, p0, q0, v0)
ref = None
for ppl in ("4", "8", "4", "8"):
    h.set_knob("EDS_REDUCE_PPL", ppl)
    rj = min(h.bench_eval(0, B, ncols=6, with_reduction=False, reps=30) for _ in range(3))
    both = min(h.bench_eval(0, B, ncols=6, with_reduction=True, reps=30) for _ in range(3))
    g = h.eval(3, als[3].p0, als[3].q0, als[3].v0, ncols=6)
    if ref is None: ref = g["JtJ"]
    print(f"PPL={ppl}: resjac {rj*1e3:7.1f} us ({B*2000/rj/1e6:6.2f} G point-evals/s)  + reduce {both*1e3:7.1f} us  -> reduce {1e3*(both-rj):6.1f} us = {B*2000*28/(both-rj)/1e9:6.2f} TB/s"
          f"   |JtJ - JtJ(first)|/|JtJ| = {np.linalg.norm(g['JtJ']-ref)/np.linalg.norm(ref):.1e}", flush=True)
h.close()
