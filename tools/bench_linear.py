#!/usr/bin/env python
"""Micro-benchmark of the typed linear kernels alone (development aid, not the judged bench)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhgt_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--n-out", type=int, default=768)
    ap.add_argument("--types", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--which", default="both")
    ap.add_argument("--bypos", type=int, default=0)
    ap.add_argument("--c24", action="store_true", help="also time the split kernel reading 24-bit wire rows (prologue 2: the halo projections "
                    "of the multi-GPU path) next to the unpack + project pair it replaces")
    args = ap.parse_args()
    lib = _lib.load()
    dev = "cuda:0"
    N, k, n_out, T = args.rows, args.k, args.n_out, args.types
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

    def run_fp32():
        assert lib.hgt_typed_linear(x.data_ptr(), k, rows.data_ptr(), off.data_ptr(), T, N, k, n_out, W.data_ptr(), n_out * k,
                                    b.data_ptr(), n_out, optr[0], optr[1], optr[2], bc, 0, 0, 0, st) == 0

    def run_split():
        assert lib.hgt_split_weights(W.data_ptr(), n_out * k, T, k, n_out, ws.data_ptr(), st) == 0
        assert lib.hgt_typed_linear_bf16x3(x.data_ptr(), k, rows.data_ptr(), off.data_ptr(), T, N, k, n_out, ws.data_ptr(),
                                           b.data_ptr(), n_out, optr[0], optr[1], optr[2], bc, args.bypos, 0, st) == 0

    variants = [("fp32", run_fp32), ("bf16x3", run_split)]
    if args.c24:
        idx = torch.arange(N, dtype=torch.int32, device=dev)
        wire = torch.empty(N, 3 * k, dtype=torch.uint8, device=dev)
        xu = torch.empty_like(x)
        assert lib.hgt_gather_rows_c24(x.data_ptr(), k, idx.data_ptr(), N, k, wire.data_ptr(), st) == 0
        assert lib.hgt_split_weights(W.data_ptr(), n_out * k, T, k, n_out, ws.data_ptr(), st) == 0

        def run_c24():
            assert lib.hgt_typed_linear_bf16x3(wire.data_ptr(), 3 * k // 4, rows.data_ptr(), off.data_ptr(), T, N, k, n_out, ws.data_ptr(),
                                               b.data_ptr(), n_out, optr[0], optr[1], optr[2], bc, args.bypos, 2, st) == 0

        def run_unpack_then_split():
            assert lib.hgt_unpack_rows_c24(wire.data_ptr(), N, k, xu.data_ptr(), k, st) == 0
            assert lib.hgt_typed_linear_bf16x3(xu.data_ptr(), k, rows.data_ptr(), off.data_ptr(), T, N, k, n_out, ws.data_ptr(),
                                               b.data_ptr(), n_out, optr[0], optr[1], optr[2], bc, args.bypos, 0, st) == 0

        def run_pack():
            assert lib.hgt_gather_rows_c24(x.data_ptr(), k, idx.data_ptr(), N, k, wire.data_ptr(), st) == 0
        variants += [("c24", run_c24), ("unpack+x", run_unpack_then_split), ("pack_c24", run_pack)]
    for name, fn in variants:
        if args.which not in ("both", name) and not (args.c24 and name in ("c24", "unpack+x", "pack_c24")):
            continue
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        gb = (N * k * 4 + N * n_out * 4) / 1e9
        if name == "bf16x3" and hasattr(lib, "hgt_debug_pc_trace"):   # development builds with -DPC_TRACE=1 only
            buf = (C.c_uint64 * 16)()
            lib.hgt_debug_pc_trace(buf, 1)
            fn()
            torch.cuda.synchronize()
            lib.hgt_debug_pc_trace(buf, 1)
            v = list(buf)
            tiles = max(1, v[7])
            names = {0: "cons barrier", 1: "cons drain", 2: "cons mfma loop", 3: "cons epilogue", 8: "prod wait loads",
                     9: "prod commit", 10: "prod issue", 11: "prod barriers"}
            print("  trace (cycles per tile, wave 0 / wave 8): " + ", ".join("%s=%d" % (names[i], v[i] // tiles) for i in sorted(names)))
        print("%-7s rows=%d k=%d n_out=%d: %.3f ms  %.1f TFLOP/s(fp32-equivalent)  %.2f TB/s(min traffic)" % (
            name, N, k, n_out, ms, 2.0 * N * k * n_out / ms / 1e9, gb / ms))


if __name__ == "__main__":
    main()
