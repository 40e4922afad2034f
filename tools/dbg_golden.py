import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/oracle")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po
name = sys.argv[1] if len(sys.argv) > 1 else "full_1234.npz"
g = np.load(ROOT+"/tests/golden/"+name)
al = synth.make_alignment(int(g["seed"]), H=int(g["H"]), W=int(g["W"]), N=int(g["N"]))
nb = int(g["num_blocks"])
for sampling, tag in ((0, "bc"), (1, "bl")):
    cfg = capi.default_config(sampling=sampling, num_blocks=nb, exec=capi.EXEC_HOST, solver=capi.SOLVER_LM6)
    h = capi.Handle(cfg, 1, al.N, al.H, al.W); h.set_alignment(0, al)
    for ex in (capi.EXEC_HOST, capi.EXEC_DEVICE):
        h.set_config(capi.default_config(sampling=sampling, num_blocks=nb, exec=ex, solver=capi.SOLVER_LM6, max_num_iterations=10))
        p, q, v, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
        tr = h.trace(0)
        errs = [np.linalg.norm(tr["increments"][k]-g[tag+"_lm6_inc"][k])/np.linalg.norm(g[tag+"_lm6_inc"][k]) for k in range(len(tr["increments"]))]
        print(name, tag, "exec", ex, "acc", tr["accepted"].tolist(), "gold", g[tag+"_lm6_acc"].tolist(), "errs", np.array2string(np.array(errs), precision=1, max_line_width=200))
        h.set_config(capi.default_config(sampling=sampling, num_blocks=nb, exec=ex, solver=capi.SOLVER_GN6, max_num_iterations=2))
        p, q, v, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    h.close()
