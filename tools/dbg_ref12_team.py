"""REF12 team kernels vs oracle on the cases of tests/test_batch_configs_gpu.py::test_ref12_team_kernel_vs_oracle, per team size."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth"); import pyoracle as po
start = sys.argv[1] if len(sys.argv) > 1 else "ctor"
als = [synth.make_alignment(7300 + b, H=240, W=320, N=n, start=start) for b, n in enumerate((600, 1024, 1500, 2000))]
for nb, loss, sampling in ((1, 0, 0), (4, 1, 0), (3, 2, 1)):
    refs = [po.Oracle(a, num_blocks=nb, loss_type=loss, loss_param=0.3, max_num_iterations=10, sampling=sampling).solve_lm(a.p0, a.q0, a.v0) for a in als]
    for team in (1, 2, 4):
        os.environ["EDS_REF12_TEAM"] = str(team)
        cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb, loss_type=loss, loss_param=0.3, sampling=sampling)
        h = capi.Handle(cfg, len(als), 2048, 240, 320)
        for b, a in enumerate(als): h.set_alignment(b, a)
        h.optimize_batch(0, 0, len(als)); tab = h.results(0, len(als))
        print(f"nb={nb} loss={loss} s={sampling} team={team}:", " ".join(f"{po.se3_distance(tab[b,0:3], tab[b,3:7], refs[b]['p'], refs[b]['q']):.1e}/{h.info(b)['num_iterations']}v{refs[b]['num_iterations']}" for b in range(len(als))))
        h.close()
