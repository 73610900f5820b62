"""where set_input's time goes: per-call host time and synchronised time of its pieces"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch, bench
from vts import ops
model, opt = bench.build_model(1024, 4, "skitG")
sd = opt.style_code_dim
b = [bench.make_batch(1024, 4, r, sd) for r in (0, 1)]
b = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in x.items()} for x in b]
for i in range(4):
    model.set_input(b[i % 2], phase="train"); model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
def T(fn, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t; torch.cuda.synchronize()
    return h / n * 1e3, (time.perf_counter() - t) / n * 1e3
print("set_input            host %.2f ms  synced %.2f ms" % T(lambda: model.set_input(b[0], phase="train")))
x = b[0]
print("load S,I,M           host %.2f ms  synced %.2f ms" % T(lambda: (model._load("t_S", x["S"]), model._load("t_I", x["I"]), model._load("t_M", x["M"]))))
print("patch_set train      host %.2f ms  synced %.2f ms" % T(lambda: model._patch_set("t_tr", x["T_images"], x["I_masks"], x["T_coords"])))
M = model.M
n, _, h, w = M.shape
cand, pre = model._buf("cand", (n, h - 14, w - 14), torch.uint8), model._buf("cand_prefix", (n, h - 14 + 1), torch.int32)
print("mask_candidates      host %.2f ms  synced %.2f ms" % T(lambda: ops.mask_candidates(M, cand, pre)))
print("mask_mul x2          host %.2f ms  synced %.2f ms" % T(lambda: (ops.mask_mul(model._bufs["t_S"], M), ops.mask_mul(model._bufs["t_I"], M))))
for alt in (0, 1):
    for i in range(3):
        model.set_input(b[(i * alt) % 2], phase="train"); model.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(10):
        model.set_input(b[(i * alt) % 2], phase="train"); model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    print("fresh-input step (alternating=%d): %.2f ms, graphs alive: %s" % (alt, (time.perf_counter() - t) * 100, model._graphs is not None))
t = time.perf_counter()
for i in range(10):
    model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
print("resident step: %.2f ms" % ((time.perf_counter() - t) * 100))
import random
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.set_input(b[0], phase="train")
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    model._prepare_ranks()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    model.optimize_parameters(epoch=1)
    t5 = time.perf_counter(); torch.cuda.synchronize(); t6 = time.perf_counter()
    print("set_input host %.2f sync %.2f | prepare_ranks host %.2f sync %.2f | optimize host %.2f sync %.2f" % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)))
ts = []
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.set_input(b[0], phase="train")
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-iteration ms:", " ".join("%.1f" % v for v in ts))
orig = model._load
import types
def fast_load(self, name, host, dtype=torch.float32):
    if name in ("train_S", "train_I", "train_M"):
        return self._bufs[name]
    return orig(name, host, dtype)
model._load = types.MethodType(fast_load, model)
ts = []
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.set_input(b[0], phase="train")
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("without the S/I/M uploads:", " ".join("%.1f" % v for v in ts))
model._load = orig
import gc
rows = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.set_input(b[0], phase="train"); t1 = time.perf_counter()
    model._prepare_ranks(); t2 = time.perf_counter()
    model.optimize_parameters(epoch=1); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    rows.append("%.1f/%.1f/%.1f/%.1f" % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)))
print("host set_input / prepare_ranks / optimize / final sync:", " ".join(rows))
gc.disable()
rows = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.set_input(b[0], phase="train"); model.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); rows.append("%.1f" % ((time.perf_counter() - t0) * 1e3))
print("gc disabled:", " ".join(rows))
