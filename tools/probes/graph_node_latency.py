"""Per-node cost of a DEPENDENT chain inside a replayed HIP graph: our tiny kernel, a torch elementwise kernel, a contiguous copy_
(memcpy node), a strided copy_, a fill_; and the same chains launched eagerly.  python tools/probes/graph_node_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd")); sys.path.insert(0, ROOT)
import torch
from vts import lib as L, ops

dev = torch.device("cuda:0")
lib = L.load()
a = torch.zeros(4096, device=dev); b = torch.zeros(4096, device=dev)
big = torch.zeros(4, 7, 256, 256, device=dev); src1 = torch.zeros(4, 1, 256, 256, device=dev)
ai = torch.zeros(1024, dtype=torch.int32, device=dev); bi = torch.zeros(1024, dtype=torch.int32, device=dev)
K = 200
chains = {
    "vts kernel (copy_words 4 KB)": lambda: L.check(lib.vts_copy_words(ai.data_ptr(), bi.data_ptr(), 1024, L.stream()), "cw"),
    "torch add_ (4096 floats)": lambda: a.add_(1.0),
    "torch contiguous copy_ (memcpy node)": lambda: b.copy_(a),
    "torch strided copy_ (channel slice)": lambda: big[:, 2:3].copy_(src1),
    "torch fill_": lambda: a.fill_(1.0),
    "vts mask_mul 1 MB": lambda: ops.mask_mul(src1, src1, out=src1),
}
for name, fn in chains.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream()):
        for _ in range(K):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    gr = (time.perf_counter() - t0) / 5 / K * 1e6
    print("%-42s eager %6.2f us/launch   graph %6.2f us/node" % (name, eager, gr))
