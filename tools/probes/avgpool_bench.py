"""avgpool3s2 on the pyramid levels of the headline step: us and achieved GB/s (algorithmic bytes = input + output)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch
from vts import ops
dev = torch.device("cuda:0")
for n, c, h in ((8, 4, 1024), (8, 4, 512), (4, 7, 1024), (4, 3, 1024), (4, 2, 512)):
    x = torch.randn(n, c, h, h, device=dev)
    for _ in range(3):
        y = ops.avgpool(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.avgpool(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("avgpool N%d C%d %d^2: %.1f us  %.0f GB/s" % (n, c, h, us, 4.0 * (x.numel() + y.numel()) / us / 1e3))
