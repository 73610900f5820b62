"""Portable deterministic pseudo-random tensors (TEST INFRASTRUCTURE ONLY).

A counter-based integer hash (splitmix64 finaliser) mapped to float32 in [-1, 1).
Used to regenerate identical test weights in this container (when the golden
vectors are made from the reference) and on the GPU box (when the HIP path is
checked), without committing megabytes of weights and without depending on any
library's RNG stream.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def uniform(shape, seed, name=""):
    """float32 tensor of `shape`, uniform in [-1, 1), a pure function of (shape, seed, name)."""
    n = int(np.prod(shape)) if len(shape) else 1
    salt = np.uint64(zlib.crc32(name.encode()) | (int(seed) << 32))
    with np.errstate(over="ignore"):
        h = _mix(np.arange(n, dtype=np.uint64) ^ _mix(salt))
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 random bits
    return torch.from_numpy((u * 2.0 - 1.0).astype(np.float32).reshape(shape))


def test_weights(shapes, seed, bn_keys=()):
    """Weights at 'kaiming-like' scale so activations stay O(1) through the stack
    (a xavier(0.02) net has ~0 tanh outputs, a weak parity signal)."""
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            if "up" in k:  # ConvTranspose2d weight [Cin, Cout, k, k]: each output sums Cin*4 taps
                fan_in = shp[0] * 4
            sd[k] = uniform(shp, seed, k) * float(np.sqrt(3.0 / fan_in))
        elif len(shp) == 2:         # Linear weight [out, in]
            sd[k] = uniform(shp, seed, k) * float(np.sqrt(3.0 / shp[1]))
        elif k.endswith("weight"):  # BatchNorm gamma
            sd[k] = 1.0 + 0.1 * uniform(shp, seed, k)
        else:  # biases
            sd[k] = 0.1 * uniform(shp, seed, k)
    return sd


def probe(t, name="probe"):
    """(sum, l2 norm, dot with a fixed pseudo-random vector) of a tensor, float64."""
    x = t.detach().double().reshape(-1)
    r = uniform((x.numel(),), 7, name).double()
    return np.array([x.sum().item(), x.norm().item(), (x * r).sum().item()], dtype=np.float64)
