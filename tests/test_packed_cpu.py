"""Host side of the dead-row compaction: the row plan (rowpack.pack_batch -> plan_packed_rows). The kernels are tested on the GPU
(tests/test_packed_gpu.py)."""
import numpy as np

from pianobart_amd.rowpack import plan_packed_rows


def test_plan_keeps_every_live_row_and_fills_whole_tiles():
    rng = np.random.default_rng(0)
    for B, S, tile in ((32, 1024, 256), (8, 1024, 256), (6, 256, 256), (4, 64, 64), (1, 1024, 256)):
        for lo in (0, S // 2, S - 3, S):
            live = rng.integers(lo, S + 1, size=B)
            Tp, off, length = plan_packed_rows(live, S, tile)
            assert Tp % tile == 0 and live.sum() <= Tp <= B * S
            assert Tp - live.sum() < tile or Tp == tile
            assert (length >= live).all() and (length <= S).all() and length.sum() == Tp
            assert off[0] == 0 and (np.diff(off) == length[:-1]).all()


def test_plan_edge_cases():
    Tp, off, length = plan_packed_rows([0, 0, 0, 0], 64, 64)            # nothing live: one tile of dead rows
    assert Tp == 64 and length.sum() == 64 and (length <= 64).all()
    Tp, off, length = plan_packed_rows([64, 64, 64, 64], 64, 64)        # nothing to drop
    assert Tp == 256 and length.tolist() == [64] * 4
    Tp, off, length = plan_packed_rows([64, 1, 64, 64], 64, 64)         # the slack goes to the sequence that has dead rows
    assert Tp == 256 and length.tolist() == [64, 64, 64, 64]
    Tp, off, length = plan_packed_rows([60, 10, 10, 10], 64, 64)        # 90 live -> 128 rows; 38 fillers: 4 to seq 0, 34 to seq 1
    assert Tp == 128 and length.tolist() == [64, 44, 10, 10] and off.tolist() == [0, 64, 108, 118]


def test_dispatch_order_host_side():
    """rowpack.dispatch_order: a permutation of the pairs, all heads of a sequence adjacent in cost order, the 8 XCD columns of the dealt
    table carry equal sums to within one pair."""
    import numpy as np
    from pianobart_amd.rowpack import dispatch_order
    rng = np.random.default_rng(0)
    B, H = 32, 12
    cost = rng.integers(512, 1025, B).astype(np.float64) ** 2
    o = dispatch_order(cost, H)
    assert sorted(o.tolist()) == list(range(B * H))
    per_xcd = np.array([cost[o[x::8] // H].sum() for x in range(8)])
    assert per_xcd.max() - per_xcd.min() <= cost.max()
    assert cost[o[0] // H] == cost.max()
