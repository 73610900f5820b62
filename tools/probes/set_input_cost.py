"""What a fresh batch costs on the device and on the host: set_input alone (host time to enqueue / wall time with a final synchronise),
and the kernels it launches.   python tools/probes/set_input_cost.py        (rocprofv3 --kernel-trace --stats around it lists the kernels)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch, bench
from vts import ops
model, opt = bench.build_model(1024, 4, "skitG")
sd = opt.style_code_dim
b = [bench.make_batch(1024, 4, r, sd, quantize8=True) for r in (0, 1)]
b = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in x.items()} for x in b]
for i in range(6):
    model.set_input(b[i % 2], phase="train"); model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for i in range(n):
    model.set_input(b[i % 2], phase="train")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("set_input alone: host enqueue %.3f ms, wall incl. device %.3f ms per call" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
calls = {}
ops.TIMER = None
import vts.lib as L
lib = L.load()
t0 = time.perf_counter()
for i in range(n):
    model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
print("replay only: %.3f ms per step" % ((time.perf_counter() - t0) / n * 1e3))
t0 = time.perf_counter()
for i in range(n):
    model.set_input(b[i % 2], phase="train"); model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
print("fresh loop: %.3f ms per step" % ((time.perf_counter() - t0) / n * 1e3))
