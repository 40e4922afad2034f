"""One REF12 alignment (640x480, 2 000 points, 8 CUs x 4 candidate groups) with 1 residual block and with 4 blocks + Huber, for the
diagnostic build's in-kernel stamps (where a lone solve spends its time):

    make -C slam-eds_amd/csrc libeds_hip_stamps.so
    EDS_HIP_LIB=$PWD/slam-eds_amd/csrc/libeds_hip_stamps.so python3 tools/diag_stamps_blocks.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(5000)
for nb, loss in ((1, capi.LOSS_NONE), (4, capi.LOSS_HUBER)):
    cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb, loss_type=loss, loss_param=0.3)
    h = capi.Handle(cfg, 1, 2000, 480, 640)
    h.set_alignment(0, al)
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")
    print(f"==== num_blocks {nb} loss {loss}", flush=True)
    for _ in range(3):
        h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    print("kernel us", h.info(0)["device_time_us"], h.last_launch()["kernel"], "iterations", h.info(0)["num_iterations"], flush=True)
    h.close()
