#!/usr/bin/env python
"""Latency-regime typed linear alone, under rocprofv3 --kernel-trace: per iteration one cache-flushing elementwise kernel, then the
SAME linear three times back to back (cold / warm / warm), tile kernel and slab kernels.  `show DIR` prints mean durations per position.
   python tools/lab/tile_gemm.py run [rows k n_out]     python tools/lab/tile_gemm.py show DIR"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(N=3200, k=256, n_out=768):
    import ctypes as C
    import torch
    from pyhgt_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    T = 4
    x = torch.randn(N, k, device=dev)
    W = torch.randn(T, n_out, k, device=dev) / k ** 0.5
    b = torch.randn(T, n_out, device=dev)
    nt = torch.randint(0, T, (N,), device=dev).sort().values
    rows = torch.arange(N, device=dev, dtype=torch.int32)
    off = torch.searchsorted(nt, torch.arange(T + 1, device=dev)).int()
    nblk = 3 if n_out % 3 == 0 else 1
    bc = n_out // nblk
    outs = [torch.empty(N, bc, device=dev) for _ in range(nblk)]
    optr = [o.data_ptr() for o in outs] + [0, 0]
    st = torch.cuda.current_stream().cuda_stream
    nb = C.c_uint64()
    lib.hgt_split_weights_bytes(T, k, n_out, C.byref(nb))
    ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    assert lib.hgt_split_weights(W.data_ptr(), n_out * k, T, k, n_out, ws.data_ptr(), st) == 0
    big = torch.empty(128 << 20, device=dev)
    for it in range(30):
        for sel in (0, _lib.HGT_LINEAR_NO_TILE):
            big.add_(1.0)
            for _ in range(3):
                assert lib.hgt_typed_linear_bf16x3(x.data_ptr(), k, rows.data_ptr(), off.data_ptr(), T, N, k, n_out, ws.data_ptr(),
                                                   b.data_ptr(), n_out, optr[0], optr[1], optr[2], bc, 0, sel, st) == 0
    torch.cuda.synchronize()


def show(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-8 * 20:]
    for pos in range(8):
        sel = rows[pos::8]
        us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
        print("%8.2f us  %s" % (sum(us) / len(us), sel[0]["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*[int(v) for v in sys.argv[2:5]])
    else:
        show(sys.argv[2])
