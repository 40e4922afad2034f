import importlib, sys, numpy as np
sys.path.insert(0,'.')
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B=4096; als=[synth.make_alignment(5000+i) for i in range(8)]
h=capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE),B,2000,480,640)
fr=[np.ascontiguousarray(a.frame,dtype=np.float32) for a in als]
for b in range(B):
    a=als[b%8]; h.set_keyframe(b,a.norm_coord,a.grad,a.idp,a.weights,a.fx,a.fy,a.cx,a.cy); h.set_event_frame(b,fr[b%8])
for rep in range(3):
    print("cold resjac", h.bench_kernel_cold(0,B,6,0,10), "cold reduce", h.bench_kernel_cold(0,B,6,1,10), "warm rj", h.bench_eval(0,B,6,False,20), "pair", h.bench_eval(0,B,6,True,20), h.hbm_probe())
