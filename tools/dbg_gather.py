"""Where does a sharded bench step spend its time on one GPU?  (process group of one rank over RCCL)"""
import importlib, os, socket, sys, time
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
batch = importlib.import_module("slam-eds_amd.batch")
dev = torch.device("cuda", 0)
local = np.random.rand(4096, 16)
for rep in range(3):
    t0 = time.perf_counter(); tab = batch.gather_results(local, 4096, device=dev, to_host=True, force=True); t1 = time.perf_counter()
    print(f"gather_results to_host: {1e3*(t1-t0):.3f} ms")
for rep in range(3):
    t0 = time.perf_counter(); batch.gather_results(local, 4096, device=dev, to_host=False, force=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"gather_results no host: {1e3*(t1-t0):.3f} ms")
buf = torch.zeros((4096, 16), dtype=torch.float64)
for rep in range(3):
    t0 = time.perf_counter(); b = torch.from_numpy(local); t1 = time.perf_counter(); d = b.to(dev); torch.cuda.synchronize(); t2 = time.perf_counter()
    out = torch.empty((4096, 16), dtype=torch.float64, device=dev); dist.all_gather_into_tensor(out, d); torch.cuda.synchronize(); t3 = time.perf_counter()
    c = out.cpu(); t4 = time.perf_counter()
    print(f"from_numpy {1e3*(t1-t0):.3f}  to(dev) {1e3*(t2-t1):.3f}  all_gather {1e3*(t3-t2):.3f}  cpu() {1e3*(t4-t3):.3f} ms")
dist.destroy_process_group()  # (end of part)
# ---- the bench's sharded step, timed piece by piece
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = 4096
als = [synth.make_alignment(5000 + i) for i in range(16)]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
for b in range(B):
    a = als[b % 16]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 16])
p0 = np.stack([als[b % 16].p0 for b in range(B)]); q0 = np.stack([als[b % 16].q0 for b in range(B)]); v0 = np.stack([als[b % 16].v0 for b in range(B)])
h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B, sync=True)
prev = h.results(0, B)
for rep in range(6):
    t0 = time.perf_counter(); h.set_states(0, p0, q0, v0); t1 = time.perf_counter()
    h.optimize_batch(0, 0, B, sync=False); t2 = time.perf_counter()
    tab = batch.gather_results(prev, B, device=dev, to_host=True, force=True); t3 = time.perf_counter()
    h.sync(); t4 = time.perf_counter()
    prev = h.results(0, B); t5 = time.perf_counter()
    print(f"set_states {1e3*(t1-t0):.3f}  launch {1e3*(t2-t1):.3f}  gather {1e3*(t3-t2):.3f}  sync {1e3*(t4-t3):.3f}  results {1e3*(t5-t4):.3f}  total {1e3*(t5-t0):.3f} ms")
dist.destroy_process_group()  # (end of part)
