"""Host-side timeline of the fresh-input loop (no synchronisation inside): where the host spends an iteration and how far it runs ahead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch, bench
model, opt = bench.build_model(1024, 4, "skitG")
sd = opt.style_code_dim
b = [bench.make_batch(1024, 4, r, sd) for r in (0, 1)]
b = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in x.items()} for x in b]
for i in range(6):
    model.set_input(b[i % 2], phase="train"); model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
import types
_q = [0.0, 0.0]
class EvtProbe:
    def __init__(self, ev, slot): self.ev, self.slot = ev, slot
    def query(self):
        return self.ev.query()
    def synchronize(self):
        t = time.perf_counter(); self.ev.synchronize(); _q[self.slot] += time.perf_counter() - t
    def record(self, *a): return self.ev.record(*a)
orig_pr = model._prepare_ranks
acc = {"set_input": 0.0, "prepare_ranks": 0.0, "optimize_rest": 0.0}
def timed_pr():
    t = time.perf_counter()
    orig_pr(); acc["prepare_ranks"] += time.perf_counter() - t
model._prepare_ranks = timed_pr
parts = {}
def wrap(obj, name, key=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            kk = key or name
            if name == "_load":
                kk = "_load:" + str(a[0])
            parts[kk] = parts.get(kk, 0.0) + time.perf_counter() - t
    setattr(obj, name, g)
from vts import ops as _ops
wrap(model, "_load"); wrap(model, "_patch_set"); wrap(_ops, "mask_candidates"); wrap(_ops, "mask_mul"); wrap(_ops, "avgpool"); wrap(model, "_spe")
n = 50
t0 = time.perf_counter()
for i in range(n):
    t = time.perf_counter(); model.set_input(b[i % 2], phase="train"); t1 = time.perf_counter()
    pr0 = acc["prepare_ranks"]
    model.optimize_parameters(epoch=1); t2 = time.perf_counter()
    acc["set_input"] += t1 - t
    acc["optimize_rest"] += (t2 - t1) - (acc["prepare_ranks"] - pr0)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("per iteration: host loop %.2f ms, with final sync %.2f ms" % (th / n * 1e3, tt / n * 1e3))
print({k: round(v / n * 1e3, 3) for k, v in sorted(parts.items(), key=lambda kv: -kv[1])[:8]})
print({k: round(v / n * 1e3, 3) for k, v in acc.items()}, "cand wait %.3f ms, ranks_evt wait %.3f ms" % (_q[0] / n * 1e3, _q[1] / n * 1e3))
