"""per-iteration host timing of the fresh-input loop of bench.py (no sync inside the loop)"""
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch, bench
model, opt = bench.build_model(1024, 4, "skitG")
pin = lambda b: {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()}
batches = [pin(bench.make_batch(1024, 4, s, opt.style_code_dim)) for s in (0, 1)]
for i in range(6):
    model.set_input(batches[i % 2], phase="train"); model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
orig = model._prepare_ranks
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    return w
model._prepare_ranks = timed("prepare_ranks", orig)
rows = []
t_all = time.perf_counter()
for i in range(30):
    acc.clear()
    t0 = time.perf_counter()
    model.set_input(batches[i % 2], phase="train")
    t1 = time.perf_counter()
    model.optimize_parameters(epoch=1)
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, acc.get("prepare_ranks", 0) * 1e3))
torch.cuda.synchronize()
print("avg %.2f ms/step" % ((time.perf_counter() - t_all) / 30 * 1e3))
print("set_input     ", " ".join("%.1f" % r[0] for r in rows))
print("optimize      ", " ".join("%.1f" % r[1] for r in rows))
print(" prepare_ranks", " ".join("%.1f" % r[2] for r in rows))
