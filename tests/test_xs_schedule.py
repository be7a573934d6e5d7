"""Host logic of the x-stationary typed linear (csrc/hgt_gemm_xs.hip): its work decomposition, enumerated on the CPU by
hgt_typed_linear_xs_schedule with the kernel's own scheduling helpers.  No GPU, no compute."""
import ctypes as C

import numpy as np
import pytest

from pyhgt_amd import _lib


def _schedule(group_sizes, k, n_cu):
    lib = _lib.load()
    off = np.concatenate([[0], np.cumsum(group_sizes)]).astype(np.int32)
    n_rows = int(off[-1])
    n = C.c_int64()
    rc = lib.hgt_typed_linear_xs_schedule(off.ctypes.data, len(group_sizes), n_rows, k, n_cu, None, 0, C.byref(n))
    assert rc in (0, -4)
    items = np.zeros((max(n.value, 1), 6), dtype=np.int32)
    assert lib.hgt_typed_linear_xs_schedule(off.ctypes.data, len(group_sizes), n_rows, k, n_cu, items.ctypes.data, n.value, C.byref(n)) == 0
    return off, items[:n.value]


@pytest.mark.parametrize("k", [256, 512])
@pytest.mark.parametrize("group_sizes,n_cu", [
    ([1000, 0, 37, 5000, 1], 256),            # an empty and two tiny groups, fewer units than workgroups
    ([250000, 250001, 249999, 250000], 256),  # the benchmark shape: ~15 rounds per workgroup
    ([128 * 7 + 5], 3),                       # one group, odd unit count, a few workgroups
    ([31, 33, 127, 129, 255, 257], 8),
    ([300007], 256),
])
def test_every_row_is_covered_exactly_once_and_rounds_never_mix_groups(group_sizes, k, n_cu):
    off, items = _schedule(group_sizes, k, n_cu)
    n_rows = int(off[-1])
    cover = np.zeros(n_rows, dtype=np.int32)
    nw = 4 if k == 512 else 8
    for wg, rnd, wave, g, row0, rows in items:
        assert 0 <= wave < nw and 1 <= rows <= 32
        assert off[g] <= row0 and row0 + rows <= off[g + 1], "an item stays inside its group"
        cover[row0:row0 + rows] += 1
    assert (cover == 1).all()
    # all items of one (workgroup, round) belong to ONE group: the ring holds one group's weights per step
    key = items[:, 0].astype(np.int64) * (1 << 32) + items[:, 1]
    for kk in np.unique(key):
        assert len(set(items[key == kk][:, 3].tolist())) == 1
    # a wavefront has at most one item per round, and the rounds of a workgroup walk the row list in ascending order
    trip = set()
    for wg, rnd, wave, g, row0, rows in items:
        assert (wg, rnd, wave) not in trip
        trip.add((wg, rnd, wave))
    for wg in np.unique(items[:, 0]):
        mine = items[items[:, 0] == wg]
        firsts = [mine[mine[:, 1] == r][:, 4].min() for r in sorted(set(mine[:, 1].tolist()))]
        assert firsts == sorted(firsts)


def test_work_is_balanced_over_the_workgroups():
    off, items = _schedule([250000] * 4, 256, 256)
    rows_per_wg = np.bincount(items[:, 0], weights=items[:, 5], minlength=256)
    assert rows_per_wg.max() - rows_per_wg.min() <= 128          # contiguous unit ranges differ by at most one 128-row unit
    rounds = [items[items[:, 0] == w][:, 1].max() + 1 for w in range(256)]
    assert max(rounds) - min(rounds) <= 1
