"""which part of _patch_set makes every ~3rd graph-replayed step stall 70..90 ms (stall_bisect.py narrowed it to _patch_set)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import numpy as np
import torch, bench
from vts import ops
from vts import lib as L
model, opt = bench.build_model(1024, 4, "skitG")
b = bench.make_batch(1024, 4, 0, opt.style_code_dim)
b = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()}
for i in range(4):
    model.set_input(b, phase="train"); model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
tag = "train_tr"
pin, dev = model._pins[tag], model._bufs[tag + "_block"]
words = pin.numel()
evt = model._pin_evt[tag]
T = torch.as_tensor(b["T_images"]); masks = torch.as_tensor(b["I_masks"])
P = T.shape[0] * T.shape[1]
raw_v = dev[:P * 2 * 1024].view(torch.float32).view(P, 2, 32, 32)
o_m = (P * 2 * 1024 + 63) // 64 * 64
mask_v = dev[o_m:o_m + P * 1024].view(torch.float32).view(P, 1, 32, 32)
realT = model._bufs[tag + "_real_T"]
src_raw = T.reshape(P, 2, 32, 32).to(torch.float32)

def cpu_fill():
    pin[:P * 2 * 1024].view(torch.float32).view(P, 2, 32, 32).copy_(src_raw)

def cpu_fill_numpy():
    np.copyto(pin[:P * 2 * 1024].view(torch.float32).numpy(), src_raw.reshape(-1).numpy())

def kcopy():
    L.check(L.load().vts_copy_words(pin.data_ptr(), dev.data_ptr(), words, L.stream()), "vts_copy_words")

def blit():
    dev.copy_(pin, non_blocking=True)

def to_f32():
    return T.reshape(P, 2, 32, 32).to(torch.float32)

def arange():
    return torch.arange(4, dtype=torch.int32).repeat_interleave(64)

def offsets():
    return model._patch_offsets(torch.as_tensor(b["T_coords"]).numpy())

variants = {
    "step only": lambda: None,
    "cpu fill pinned (torch copy_)": cpu_fill,
    "cpu fill pinned (numpy)": cpu_fill_numpy,
    "T.to(float32) (cpu)": to_f32,
    "arange.repeat_interleave (cpu)": arange,
    "patch offsets (numpy)": offsets,
    "kernel reads pinned": kcopy,
    "blit from pinned": blit,
    "mask_mul views": lambda: ops.mask_mul(raw_v, mask_v, out=realT),
    "fill + kernel copy": lambda: (cpu_fill(), kcopy()),
    "fill + blit": lambda: (cpu_fill(), blit()),
    "full _patch_set": lambda: model._patch_set(tag, b["T_images"], b["I_masks"], b["T_coords"]),
}
torch.set_num_threads(torch.get_num_threads())
print("torch threads", torch.get_num_threads())
for name, fn in variants.items():
    ts = []
    for i in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); model.optimize_parameters(epoch=1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-32s" % name, " ".join("%.0f" % v for v in ts))
torch.set_num_threads(1)
for name in ("cpu fill pinned (torch copy_)", "T.to(float32) (cpu)", "full _patch_set"):
    fn = variants[name]
    ts = []
    for i in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); model.optimize_parameters(epoch=1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("1 thread: %-22s" % name, " ".join("%.0f" % v for v in ts))
