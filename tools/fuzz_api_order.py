"""Random orders of the calls whose device work is no longer waited for (set_event_frame, set_idepth, build_event_frame, the
residual mirror, getCoord through mapped memory, the pyramid's event-ordered levels): after any sequence, a solve on the long-lived
handle must be BIT-IDENTICAL to the same solve on a fresh handle that was given the same inputs once (LM6: no atomics, the result is
a pure function of the inputs), and everything read back (frame, residuals, coordinates) must match too."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")

H, W, N, B = 240, 320, 1500, 3
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
als = [synth.make_alignment(9000 + i, H=H, W=W, N=N) for i in range(6)]
cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=6)
h = capi.Handle(cfg, B, N, H, W)
state = []                                  # per slot: (alignment index of the keyframe, frame array, idp array)
for b in range(B):
    a = als[b]
    h.set_alignment(b, a)
    state.append([b, np.ascontiguousarray(a.frame, dtype=np.float64), np.array(a.idp)])


def fresh_solve(slot):
    k, frame, idp = state[slot]
    a = als[k]
    g = capi.Handle(cfg, 1, N, H, W)
    g.set_keyframe(0, a.norm_coord, a.grad, idp, a.weights, a.fx, a.fy, a.cx, a.cy)
    g.set_event_frame(0, frame)
    g.set_state(0, a.p0, a.q0, a.v0)
    g.optimize_batch(0, 0, 1)
    out = (g.results(0, 1)[0].copy(), g.residuals(0).copy(), g.get_event_frame(0).copy())
    g.close()
    return out


bad = 0
for it in range(steps):
    op = rng.integers(0, 7)
    slot = int(rng.integers(0, B))
    if op == 0:                             # a new host frame
        src = als[int(rng.integers(0, len(als)))]
        fr = np.ascontiguousarray(src.frame * rng.uniform(0.5, 1.5), dtype=np.float64)
        h.set_event_frame(slot, fr); state[slot][1] = fr
    elif op == 1:                           # the depths moved
        idp = state[slot][2] * rng.uniform(0.9, 1.1, size=N)
        h.set_idepth(slot, idp); state[slot][2] = idp
    elif op == 2:                           # a frame built on the device from events: read it back, it becomes the slot's host frame
        n = int(rng.integers(100, 20000))
        x = rng.integers(0, W, n).astype(np.uint16); y = rng.integers(0, H, n).astype(np.uint16); pol = rng.integers(0, 2, n).astype(np.uint8)
        h.build_event_frame(slot, x, y, pol)
        state[slot][1] = h.get_event_frame(slot)
    elif op == 3:                           # another keyframe
        k = int(rng.integers(0, len(als)))
        a = als[k]
        h.set_keyframe(slot, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        state[slot][0] = k; state[slot][2] = np.array(a.idp)
    elif op == 4:                           # getCoord without culling: must not disturb anything
        h.set_state(slot, als[state[slot][0]].p0, als[state[slot][0]].q0, als[state[slot][0]].v0)
        h.update_points(slot, False)
    else:                                   # solve (one slot or a range) and compare with fresh handles
        first = slot if op == 5 else 0
        count = 1 if op == 5 else B
        for s in range(first, first + count):
            a = als[state[s][0]]
            h.set_state(s, a.p0, a.q0, a.v0)
        h.optimize_batch(0, first, count)
        tab = h.results(first, count)
        for i, s in enumerate(range(first, first + count)):
            ref_tab, ref_r, ref_frame = fresh_solve(s)
            ok = np.array_equal(tab[i], ref_tab) and np.array_equal(h.residuals(s), ref_r) and np.array_equal(h.get_event_frame(s), ref_frame)
            if not ok:
                bad += 1
                print(f"step {it}: slot {s} differs from a fresh handle (op {op})", flush=True)
print(f"{steps} steps, {bad} disagreements")
sys.exit(1 if bad else 0)
