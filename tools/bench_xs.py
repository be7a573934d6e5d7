#!/usr/bin/env python
"""The x-stationary split typed linear (csrc/hgt_gemm_xs.hip) against the persistent slab kernel it replaces on large inputs:
bit-identity on ragged / permuted / multi-group inputs in every instantiated form, then timings at the c2 Q|K|V shape and the
halo K|V shape (development aid, not the judged bench).  The kernel is selected through the prologue bits of the C ABI
(HGT_LINEAR_FORCE_XS / HGT_LINEAR_NO_XS), not through the environment.   python tools/bench_xs.py [--quick]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhgt_amd import _lib  # noqa: E402

DEV = "cuda:0"


def setup(lib, N, k, n_out, T, f16, seed, ragged):
    g = torch.Generator().manual_seed(seed)
    gd = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(N, k, generator=gd, device=DEV)
    W = (torch.randn(T, n_out, k, generator=g) / k ** 0.5).to(DEV)
    b = torch.randn(T, n_out, generator=g).to(DEV)
    if ragged:      # uneven groups (one of them tiny, one empty when T >= 4), rows addressed through a permutation
        cuts = sorted(torch.randint(0, N, (T - 1,), generator=g).tolist())
        if T >= 4:
            cuts[1] = cuts[0]
            cuts[2] = min(N, cuts[1] + 37)
            cuts = sorted(cuts)
        off = torch.tensor([0] + cuts + [N], dtype=torch.int32, device=DEV)
        rows = torch.randperm(N, generator=g).to(torch.int32).to(DEV)
    else:
        nt = torch.randint(0, T, (N,), generator=g).sort().values
        off = torch.searchsorted(nt, torch.arange(T + 1)).int().to(DEV)
        rows = torch.arange(N, dtype=torch.int32, device=DEV)
    nb = C.c_uint64()
    assert lib.hgt_split_weights_bytes(T, k, n_out, C.byref(nb)) == 0
    ws = torch.empty(nb.value, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    split = lib.hgt_split_weights_f16 if f16 else lib.hgt_split_weights
    assert split(W.data_ptr(), n_out * k, T, k, n_out, ws.data_ptr(), st) == 0
    return x, W, b, rows, off, ws


def run(lib, f16, xptr, ldx, rows, off, T, N, k, n_out, ws, b, outs, bc, bypos, prologue, sel=0):
    """sel: 0 = the library's own choice, _lib.HGT_LINEAR_FORCE_XS / HGT_LINEAR_NO_XS (bits of the prologue argument)"""
    prologue = prologue | sel
    lin = lib.hgt_typed_linear_f16x3 if f16 else lib.hgt_typed_linear_bf16x3
    st = torch.cuda.current_stream().cuda_stream
    optr = [o.data_ptr() for o in outs] + [None, None]
    rc = lin(xptr, ldx, rows.data_ptr(), off.data_ptr(), T, N, k, n_out, ws.data_ptr(), b.data_ptr(), n_out, optr[0], optr[1], optr[2], bc,
             bypos, prologue, st)
    assert rc == 0, rc


def check(lib, N, k, n_out, T, f16, c24, bypos, ragged, seed=1):
    x, W, b, rows, off, ws = setup(lib, N, k, n_out, T, f16, seed, ragged)
    nblk = 3 if n_out % 3 == 0 and n_out >= 192 else (2 if n_out % 2 == 0 and n_out >= 128 else 1)
    bc = n_out // nblk
    st = torch.cuda.current_stream().cuda_stream
    if c24:
        idx = torch.arange(N, dtype=torch.int32, device=DEV)
        wire = torch.empty(N, 3 * k, dtype=torch.uint8, device=DEV)
        assert lib.hgt_gather_rows_c24(x.data_ptr(), k, idx.data_ptr(), N, k, wire.data_ptr(), st) == 0
        xptr, ldx, prologue = wire.data_ptr(), 3 * k // 4, 2
    else:
        xptr, ldx, prologue = x.data_ptr(), k, 0
    res = []
    for sel in (_lib.HGT_LINEAR_NO_XS, _lib.HGT_LINEAR_FORCE_XS):
        outs = [torch.full((N, bc), float("nan"), device=DEV) for _ in range(nblk)]
        run(lib, f16, xptr, ldx, rows, off, T, N, k, n_out, ws, b, outs, bc, bypos, prologue, sel)
        torch.cuda.synchronize()
        res.append(torch.cat(outs, 1))
    same = torch.equal(res[0].nan_to_num(nan=12345.0), res[1].nan_to_num(nan=12345.0))
    nbad = int((res[0].nan_to_num(nan=12345.0) != res[1].nan_to_num(nan=12345.0)).sum())
    dmax = float((res[0] - res[1]).nan_to_num(nan=1e30).abs().max()) if not same else 0.0
    print("check N=%d k=%d n_out=%d T=%d f16=%d c24=%d bypos=%d ragged=%d : %s (differing %d, max |d| %.3e)" % (
        N, k, n_out, T, f16, c24, bypos, ragged, "BIT-IDENTICAL" if same else "DIFFERENT", nbad, dmax), flush=True)
    return same


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench(lib, N, k, n_out, T, f16, c24, iters=5, reps=4):
    x, W, b, rows, off, ws = setup(lib, N, k, n_out, T, f16, 3, False)
    nblk = 3 if n_out % 3 == 0 else 2
    bc = n_out // nblk
    outs = [torch.empty(N, bc, device=DEV) for _ in range(nblk)]
    st = torch.cuda.current_stream().cuda_stream
    if c24:
        idx = torch.arange(N, dtype=torch.int32, device=DEV)
        wire = torch.empty(N, 3 * k, dtype=torch.uint8, device=DEV)
        assert lib.hgt_gather_rows_c24(x.data_ptr(), k, idx.data_ptr(), N, k, wire.data_ptr(), st) == 0
        xptr, ldx, prologue = wire.data_ptr(), 3 * k // 4, 2
    else:
        xptr, ldx, prologue = x.data_ptr(), k, 0
    variants = [("slab", _lib.HGT_LINEAR_NO_XS), ("xs", _lib.HGT_LINEAR_FORCE_XS)]
    best = {n: 1e9 for n, _ in variants}
    for rep in range(reps):      # interleaved repetitions, minimum per variant (the order of the variants and the clocks matter)
        for name, sel in (variants if rep % 2 == 0 else variants[::-1]):
            best[name] = min(best[name], timeit(lambda: run(lib, f16, xptr, ldx, rows, off, T, N, k, n_out, ws, b, outs, bc, 0, prologue, sel), iters))
    gb = (N * k * (3 if c24 else 4) + N * n_out * 4) / 1e9
    print("time  N=%d k=%d n_out=%d f16=%d c24=%d :" % (N, k, n_out, f16, c24) + "".join("  %s %.3f" % (n, best[n]) for n, _ in variants) +
          "   [ms; %.2f GB]" % gb, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--threshold", action="store_true", help="only: slab vs xs at row counts around the dispatch threshold")
    args = ap.parse_args()
    lib = _lib.load()
    ok = True
    if args.threshold:
        for N in (40000, 66000, 100000, 131072, 160000, 200000, 262144, 330000, 400000, 524288):
            bench(lib, N, 256, 768, 4, 0, 0, iters=8, reps=3)
        for N in (66000, 131072, 200000, 330000):
            bench(lib, N, 256, 512, 4, 0, 1, iters=8, reps=3)
            bench(lib, N, 512, 1536, 4, 0, 0, iters=8, reps=3)
        return
    if not args.no_check:
        # one round per workgroup, several rounds, ragged groups with an empty and a tiny group, both output addressings, both formats
        for (N, k, n_out, T, f16, c24, bypos, ragged) in [
            (1000, 256, 768, 3, 0, 0, 0, 0), (70001, 256, 768, 4, 0, 0, 0, 1), (300007, 256, 768, 4, 0, 0, 1, 1),
            (300007, 256, 768, 4, 1, 0, 0, 1), (200003, 256, 512, 3, 0, 1, 0, 1), (200003, 256, 512, 3, 1, 1, 1, 1),
            (150001, 128, 384, 2, 0, 0, 0, 1), (150001, 64, 192, 5, 1, 0, 0, 1), (90001, 256, 200, 3, 0, 0, 0, 1),
            (1000000, 256, 768, 4, 0, 0, 0, 0), (1000000, 256, 768, 4, 1, 0, 0, 0),
            (200003, 512, 1536, 3, 0, 0, 0, 1), (200003, 512, 1536, 4, 1, 0, 1, 1), (120001, 512, 512, 3, 0, 0, 0, 1), (3000, 512, 1536, 3, 0, 0, 0, 0),
        ]:
            ok &= check(lib, N, k, n_out, T, f16, c24, bypos, ragged)
        print("ALL BIT-IDENTICAL" if ok else "MISMATCH", flush=True)
    for (N, k, n_out, f16, c24) in [(500000, 512, 1536, 0, 0), (500000, 512, 1536, 1, 0), (500000, 512, 512, 0, 0), (1000000, 256, 768, 0, 0), (1000000, 256, 768, 1, 0), (1000000, 256, 512, 0, 1), (1000000, 256, 512, 0, 0),
                                    (625000, 256, 512, 0, 1), (1000000, 128, 384, 0, 0)]:
        bench(lib, N, k, n_out, 4, f16, c24)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
