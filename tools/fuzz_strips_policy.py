"""Random sequences of frame changes, shared frames, explicit conversions and solves of random slot ranges on a batch handle: whatever
layout the library's reuse rule picks for a launch (eds_strips.hip: tiles for first solves, strip copies for frames solved again, a few
new frames converted at once), every solve must agree with the same solve on a mirror handle that is forced onto the 4x4 tiles
(EDS_FUSED_LAYOUT=tiles is read per solve) — i.e. the copies always hold what the frames hold NOW.

    python tools/fuzz_strips_policy.py [seed] [steps]
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
H, W = 120, 160
bad = 0
for trial in range(3):
    B = int(rng.choice([48, 96, 200])); N = int(rng.choice([700, 900, 1800]))
    solver = capi.SOLVER_REF12 if rng.random() < 0.3 else capi.SOLVER_LM6
    kw = dict(solver=solver, exec=capi.EXEC_DEVICE, sampling=int(rng.integers(0, 2)) if solver == capi.SOLVER_LM6 else 0, max_num_iterations=4,
              huber_tau=float(rng.choice([0.0, 0.02])) if solver == capi.SOLVER_LM6 else 0.0)
    als = [synth.make_alignment(7000 + 10 * trial + i, H=H, W=W, N=N) for i in range(6)]
    f32 = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    h = capi.Handle(capi.default_config(**kw), B, N, H, W); m = capi.Handle(capi.default_config(**kw), B, N, H, W)
    m.set_knob("EDS_FUSED_LAYOUT", "tiles")       # the mirror: always the tile kernels (a knob of THAT handle; h keeps the rule)
    ps, qs = np.array([1e-3, -2e-3, 5e-4]), synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    for b in range(B):
        for x in (h, m):
            x.set_alignment(b, als[b % 6])
    P, Q, V = np.stack([ps] * B), np.stack([qs] * B), np.stack([als[b % 6].v0 for b in range(B)])
    layouts = {1: 0, 2: 0}
    hist = {}
    for s in range(steps):
        op = rng.random()
        if op < 0.25:                                   # new frames in k slots
            k = int(rng.choice([1, 2, max(2, B // 12), B // 2]))
            for b in rng.choice(B, k, replace=False):
                f = f32[int(b) % 6] * np.float32(rng.uniform(0.8, 1.2))      # the slot's own scene, rescaled: new content, still a well-posed solve
                h.set_event_frame(int(b), f); m.set_event_frame(int(b), f); hist.setdefault(int(b), []).append((s, 'frame'))
        elif op < 0.32:                                 # a slot samples another slot's frame
            a_ = int(rng.integers(0, B)); b_ = (a_ + 6 * int(rng.integers(1, B // 6))) % B      # (a slot of the same scene: a frame of another scene makes an
            if b_ == a_: continue                                                             # ill-posed solve, on which two fp32 kernels may part by 1e-3)
            try:
                h.share_event_frame(a_, b_); m.share_event_frame(a_, b_); hist.setdefault(a_, []).append((s, 'shares', b_)); hist.setdefault(b_, []).append((s, 'shared by', a_))
            except capi.EdsError:
                pass
        elif op < 0.40:
            f0 = int(rng.integers(0, B)); c = int(rng.integers(1, B - f0 + 1))
            h.prepare_frames(f0, c); [hist.setdefault(b, []).append((s, 'prep')) for b in range(f0, f0 + c)]
        else:                                           # solve a range
            f0 = int(rng.integers(0, B // 2)); c = int(rng.integers(1, B - f0 + 1))
            h.set_states(0, P, Q, V); m.set_states(0, P, Q, V)
            h.optimize_batch(0, f0, c)
            li = h.last_launch(); [hist.setdefault(b, []).append((s, 'solve', li['layout'])) for b in range(f0, f0 + c)]
            m.optimize_batch(0, f0, c)          # (the mirror handle carries EDS_FUSED_LAYOUT=tiles as a knob of its own)
            assert m.last_launch()["layout"] == 1
            layouts[li["layout"]] = layouts.get(li["layout"], 0) + 1
            th, tm = h.results(f0, c), m.results(f0, c)
            d = np.abs(th[:, :13] - tm[:, :13]).max()
            tol = 1e-6 if solver == capi.SOLVER_LM6 else 1e-5
            if (d > tol or not np.array_equal(th[:, 14:16], tm[:, 14:16])) and kw["sampling"] == 1:
                # The bilinear sampler's derivative is one-sided: a point that lands within an fp32 ulp of a pixel boundary (seed 4 of this
                # tool: column 116.99999986) gets the left cell's slope from one kernel and the right cell's from the other — two
                # correct answers, one step apart by a per cent.  Such a coincidence does not survive a start 1e-6 away.
                P2 = P + 1e-6
                h.set_states(0, P2, Q, V); m.set_states(0, P2, Q, V)
                h.optimize_batch(0, f0, c)
                m.optimize_batch(0, f0, c)          # (the mirror handle carries EDS_FUSED_LAYOUT=tiles as a knob of its own)
                th, tm = h.results(f0, c), m.results(f0, c)
                d2 = np.abs(th[:, :13] - tm[:, :13]).max()
                print(f"trial {trial} step {s}: bilinear, differs by {d:.2e}; from a start 1e-6 away by {d2:.2e}", flush=True)
                d = d2
            if d > tol or not np.array_equal(th[:, 14:16], tm[:, 14:16]):
                print(f"trial {trial} step {s}: B={B} N={N} kw={kw} range [{f0}, {f0 + c}) layout {li['layout']} {li['kernel']}: differs by {d:.2e}", flush=True)
                rows = np.nonzero(np.abs(th[:, :13] - tm[:, :13]).max(axis=1) > tol)[0]
                print("    slots", [(int(f0 + r), hist.get(int(f0 + r), [])[-4:]) for r in rows[:6]], flush=True)
                bad += 1
    print(f"trial {trial}: B={B} N={N} solver={solver} sampling={kw['sampling']}: launches on tiles {layouts.get(1, 0)}, on strips {layouts.get(2, 0)}", flush=True)
    h.close(); m.close()
print(f"{bad} disagreements")
sys.exit(1 if bad else 0)
