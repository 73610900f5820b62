"""Inference forward of the generator at N = 16, 1024 x 1024: the Python schedule (two decoder lanes) against the network-level C entry
vts_unet_forward (one stream), both replayed as HIP graphs.  python tools/mb_unet_c.py [N] [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from tests.test_network_abi_gpu import generator  # noqa: E402
from vts import engine, lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G, _ = generator(size, n)
dev = torch.device("cuda:0")
s = torch.rand(n, 1, size, size, device=dev) * 2 - 1
grid = torch.rand(n, 8, size, size, device=dev) * 2 - 1
out = torch.empty(n, 5, size, size, device=dev)
lib = L.load()
d = engine.unet_desc(G, (s, grid), out)
ws = torch.empty(int(lib.vts_unet_forward_ws_floats(C.byref(d))), device=dev)


def run_c():
    L.check(lib.vts_unet_forward(C.byref(d), ws.data_ptr(), ws.numel(), L.stream()), "vts_unet_forward")


lane = torch.cuda.Stream()
d2 = engine.unet_desc(G, (s, grid), out, side_stream=lane)


def run_c2():
    L.check(lib.vts_unet_forward(C.byref(d2), ws.data_ptr(), ws.numel(), L.stream()), "vts_unet_forward")


def run_py():
    engine.unet_forward(G, (s, grid), keep=False)


for name, fn in (("python schedule (2 lanes)", run_py), ("vts_unet_forward (1 stream)", run_c), ("vts_unet_forward (2 lanes)", run_c2)):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        fn()
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("%-30s N %d %dx%d: %.3f ms per forward = %.4f ms per image" % (name, n, size, size, ms, ms / n))
