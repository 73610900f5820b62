"""Measured ceilings of the box next to the datasheet figures used for the roofline: HBM streaming bandwidth
(device-to-device copy, read-only reduction, write-only fill on 1 GiB buffers) and the fp32 matrix rate (square fp32 GEMMs through
rocBLAS / hipBLASLt as torch.matmul dispatches them).  Plumbing only: torch ops, not part of the product path.
python tools/ceilings.py"""
import json
import time

import torch


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    n = 1 << 28   # 1 GiB of fp32
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    out = {"device": torch.cuda.get_device_name(0)}
    t = timed(lambda: y.copy_(x))
    out["hbm_copy_GBps"] = 2 * 4 * n / t / 1e9
    t = timed(lambda: x.sum())
    out["hbm_read_GBps"] = 4 * n / t / 1e9
    t = timed(lambda: y.fill_(1.0))
    out["hbm_write_GBps"] = 4 * n / t / 1e9
    del x, y
    for m in (4096, 8192):
        a = torch.randn(m, m, device=dev)
        b = torch.randn(m, m, device=dev)
        t = timed(lambda: torch.matmul(a, b), iters=10)
        out["fp32_gemm_%d_TFLOPs" % m] = 2.0 * m ** 3 / t / 1e12
    out["datasheet"] = {"hbm_GBps": 8000.0, "fp32_matrix_TFLOPs": 157.3}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
