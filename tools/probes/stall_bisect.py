"""which part of set_input makes every ~3rd graph-replayed step stall ~70 ms (measured round 2)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch, bench
from vts import ops
model, opt = bench.build_model(1024, 4, "skitG")
b = bench.make_batch(1024, 4, 0, opt.style_code_dim)
b = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()}
for i in range(4):
    model.set_input(b, phase="train"); model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
pin = torch.empty(4, dtype=torch.int32).pin_memory()
hp = torch.zeros(1 << 20).pin_memory(); dp = torch.zeros(1 << 20, device="cuda")
M = model.M; n, _, h, w = M.shape
cand, pre = model._bufs["cand"], model._bufs["cand_prefix"]
variants = {
    "step only": lambda: None,
    "+ H2D 4MB pinned": lambda: dp.copy_(hp, non_blocking=True),
    "+ D2H 16B pinned": lambda: pin.copy_(pre[:, -1], non_blocking=True),
    "+ D2H 16B contiguous": lambda: pin.copy_(pre[:, -1].contiguous(), non_blocking=True),
    "+ mask_candidates": lambda: ops.mask_candidates(M, cand, pre),
    "+ mask_mul": lambda: ops.mask_mul(model._bufs["train_S"], M, out=model.real_S),
    "+ event record+query spin": None,
    "+ set_input": lambda: model.set_input(b, phase="train"),
}
evt = torch.cuda.Event()
def spin():
    evt.record()
    while not evt.query():
        pass
variants["+ event record+query spin"] = spin
for name, fn in variants.items():
    ts = []
    for i in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); model.optimize_parameters(epoch=1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-28s" % name, " ".join("%.0f" % v for v in ts))
import types
orig_ps = model._patch_set
cache = {}
def cached_ps(self, tag, *a):
    if tag not in cache:
        cache[tag] = orig_ps(tag, *a)
    return cache[tag]
model._patch_set = types.MethodType(cached_ps, model)
ts = []
for i in range(15):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.set_input(b, phase="train"); model.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("%-28s" % "+ set_input w/o patch_set", " ".join("%.0f" % v for v in ts))
model._patch_set = orig_ps
orig_load = model._load
def load_nostyle(self, name, host, dtype=torch.float32):
    if name in self._bufs and name.endswith(("_S", "_I", "_M", "_style")):
        return self._bufs[name]
    return orig_load(name, host, dtype)
model._load = types.MethodType(load_nostyle, model)
ts = []
for i in range(15):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.set_input(b, phase="train"); model.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("%-28s" % "+ set_input w/o big loads", " ".join("%.0f" % v for v in ts))
