#!/usr/bin/env python
"""What the memory system gives plain streaming kernels at the byte counts of the Q|K|V projection (context for DESIGN.md section 4):
a pure write of 3.07 GB, a copy 2 GB -> 2 GB, and 1 GB read + 3 GB written (x broadcast into three column blocks)."""
import torch

dev = "cuda:0"
N, d = 1_000_000, 256


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


x = torch.randn(N, d, device=dev)
q, k, v = (torch.empty(N, d, device=dev) for _ in range(3))
big = torch.empty(N, 3 * d, device=dev)
a2, b2 = torch.randn(N, 2 * d, device=dev), torch.empty(N, 2 * d, device=dev)
ms = timeit(lambda: big.fill_(1.0))
print("fill 3.07 GB: %.3f ms = %.2f TB/s" % (ms, big.numel() * 4 / ms / 1e9))
ms = timeit(lambda: b2.copy_(a2))
print("copy 2.05 GB -> 2.05 GB: %.3f ms = %.2f TB/s" % (ms, 2 * a2.numel() * 4 / ms / 1e9))
ms = timeit(lambda: (q.copy_(x), k.copy_(x), v.copy_(x)))
print("3 x copy 1.02 GB -> 1.02 GB (x read three times): %.3f ms = %.2f TB/s of 6.1 GB" % (ms, 6 * x.numel() * 4 / ms / 1e9))
ms = timeit(lambda: torch.mul(x, 2.0, out=q))
print("scale 1.02 GB -> 1.02 GB: %.3f ms = %.2f TB/s" % (ms, 2 * x.numel() * 4 / ms / 1e9))
