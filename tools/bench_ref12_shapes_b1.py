import importlib, os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tools") else ".")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(5000)
for nb in (1, 4):
  for team, g in (("8","4"),("4","4"),("4","2"),("8","2"),("2","4"),("16","2")):
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb), 1, 2000, 480, 640)
    h.set_knob("EDS_REF12_TEAM", team); h.set_knob("EDS_REF12_GROUPS", g)
    h.set_alignment(0, al)
    ker = []
    for rep in range(60):
        p, q, v, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0); ker.append(info["device_time_us"])
    print(f"nb={nb} TEAM={team} GROUPS={g}: kernel {np.median(ker[10:]):.1f} us {h.last_launch()['kernel']} steps {info['num_successful_steps']}/{info['num_unsuccessful_steps']}", flush=True)
    h.close()
