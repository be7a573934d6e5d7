"""tools/lab/pack_time.py: the halo pack kernels alone (5 M random rows of 256 floats out of 1 M): hgt_gather_rows_c24, hgt_gather_rows
(both one row per wavefront), and a device copy of the packed bytes (what the emulated exchange adds to the `pack` stage)."""
import torch
from pyhgt_amd import _lib

lib = _lib.load()
dev = "cuda:0"
N, n, d = 1_000_000, 5_000_000, 256
x = torch.randn(N, d, device=dev)
idx = torch.randint(0, N, (n,), device=dev, dtype=torch.int32)
wire = torch.empty(n, 3 * d, dtype=torch.uint8, device=dev)
wire2 = torch.empty_like(wire)
full = torch.empty(n, d, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


t = timeit(lambda: lib.hgt_gather_rows_c24(x.data_ptr(), d, idx.data_ptr(), n, d, wire.data_ptr(), st))
print("gather_rows_c24      : %.3f ms = %.2f TB/s of %.1f GB" % (t, (n * d * 7) / t / 1e9, n * d * 7 / 1e9))
t = timeit(lambda: lib.hgt_gather_rows(x.data_ptr(), d, idx.data_ptr(), n, d, full.data_ptr(), st))
print("gather_rows (fp32)   : %.3f ms = %.2f TB/s of %.1f GB" % (t, (n * d * 8) / t / 1e9, n * d * 8 / 1e9))
a, b = wire2.view(torch.int32).reshape(-1), wire.view(torch.int32).reshape(-1)
t = timeit(lambda: torch.bitwise_or(b, 0, out=a))
print("device copy of the packed rows: %.3f ms" % t)
