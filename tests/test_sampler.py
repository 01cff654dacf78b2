"""Native MT19937 sampler == CPython's random.seed / random.shuffle, bit for bit
(model_handler.py:29-30,314,341).  Golden sequences were captured from CPython 3.10."""
import random

import numpy as np
import pytest

from conftest import load_golden
from ggad_amd.sampler import PyCompatRandom


def test_shuffle_matches_golden_sequences():
    g = load_golden("sampler_shuffle.npz")
    for seed, sizes in ((72, (1, 2, 3, 7, 64, 1000, 55275)), (0, (10, 4097)), (2 ** 40 + 5, (33,))):
        r = PyCompatRandom(seed)
        for s in sizes:
            a = np.arange(s, dtype=np.int64)
            r.shuffle(a)
            r.shuffle(a)
            assert np.array_equal(a, g[f"seed{seed}_n{s}"]), (seed, s)
        tail = [r.getrandbits32() for _ in range(4)]
        assert tail == g[f"seed{seed}_tail"].tolist()


@pytest.mark.parametrize("seed", [1, 72, 123456789, 2 ** 63 + 11])
def test_shuffle_matches_live_cpython(seed):
    random.seed(seed)
    r = PyCompatRandom(seed)
    for n in (5, 200, 3001):
        lst = list(range(n))
        random.shuffle(lst)
        a = np.arange(n, dtype=np.int64)
        r.shuffle(a)
        assert a.tolist() == lst


def test_state_exchange_with_cpython():
    random.seed(99)
    random.random()
    r = PyCompatRandom.from_python_state(random.getstate())
    lst = list(range(777))
    random.shuffle(lst)
    a = np.arange(777, dtype=np.int64)
    r.shuffle(a)
    assert a.tolist() == lst
    # and back
    random.setstate(r.to_python_state())
    assert random.getrandbits(32) == r.getrandbits32()


def test_native_batch_stream_equals_per_batch_path():
    """`BatchSchedule.next_batches` (one native call, generator walk pipelined with the swaps) produces exactly the batches,
    the list states and the generator state of the batch-by-batch path -- across epoch boundaries, for 1 and 3 ranks."""
    import numpy as np
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule
    n = 5000
    labels = np.zeros(n, dtype=np.int64)
    pool = np.arange(100, 700)
    labels[pool] = 1
    train = np.arange(1000, 4333)

    def make():
        return BatchSchedule(train.copy(), pool.copy(), labels, 150, PyCompatRandom(72), n_pseudo=50, batches_per_epoch=7)
    a, b = make(), make()
    ref = [a.next_batch() for _ in range(40)]
    got_n, got_l = [], []
    for k in (1, 5, 13, 21):                              # uneven call sizes: 40 batches in total
        nn, ll = b.next_batches(k)
        got_n += nn; got_l += ll
    assert len(got_n) == 40
    for (rn, rl), gn, gl in zip(ref, got_n, got_l):
        assert np.array_equal(rn, gn) and np.array_equal(rl, gl)
    assert np.array_equal(a.train, b.train) and np.array_equal(a.pool, b.pool)
    assert a.rng.to_python_state() == b.rng.to_python_state() and a._in_epoch == b._in_epoch
    # a long call: more pool shuffles in flight than the pipeline's ring holds (128), dozens of epoch shuffles (train double buffer)
    ref_long = [a.next_batch() for _ in range(300)]
    nn, ll = b.next_batches(300)
    for (rn, rl), gn, gl in zip(ref_long, nn, ll):
        assert np.array_equal(rn, gn) and np.array_equal(rl, gl)
    assert np.array_equal(a.train, b.train) and np.array_equal(a.pool, b.pool)
    assert a.rng.to_python_state() == b.rng.to_python_state() and a._in_epoch == b._in_epoch
    # three ranks: rank r of step s gets global batch 3 s + r
    c = make()
    ref3 = [c.next_batch() for _ in range(12)]
    for r in range(3):
        d = make()
        nn, _ = d.next_batches(4, rank=r, world=3)
        for s_ in range(4):
            assert np.array_equal(nn[s_], ref3[3 * s_ + r][0])


def test_native_batch_stream_rejects_ids_beyond_int32_and_handles_tiny_lists():
    """The pipelined scheduler keeps the pool as int32 while it runs: ids that do not fit are refused (no silent truncation);
    one-element lists (no swap to draw) pass through unchanged and consume nothing of the generator."""
    import ctypes
    import numpy as np
    from ggad_amd import _lib
    from ggad_amd.sampler import PyCompatRandom
    lib = _lib.load()
    r = PyCompatRandom(5)
    train = np.arange(10, dtype=np.int64)
    pool = np.array([1, 2, 1 << 31], dtype=np.int64)
    ie = ctypes.c_int32(3)
    out = np.zeros((4, 5), dtype=np.int64)
    lens = np.zeros(4, dtype=np.int32)
    rc = lib.ggad_sched_batches(r._h, train.ctypes.data, 10, pool.ctypes.data, 3, 3, 2, 3, ctypes.byref(ie), 2, out.ctypes.data,
                                lens.ctypes.data)
    assert rc == -1
    before = r.to_python_state()
    one_t, one_p = np.array([7], dtype=np.int64), np.array([9], dtype=np.int64)
    ie = ctypes.c_int32(1)
    rc = lib.ggad_sched_batches(r._h, one_t.ctypes.data, 1, one_p.ctypes.data, 1, 1, 1, 1, ctypes.byref(ie), 3, out.ctypes.data,
                                lens.ctypes.data)
    assert rc == 0 and r.to_python_state() == before
    assert lens.tolist()[:3] == [2, 2, 2] and out.reshape(-1)[:6].tolist() == [7, 9, 7, 9, 7, 9] and one_t[0] == 7 and one_p[0] == 9


def test_schedule_rows_are_recognised_as_one_flat_block():
    """`BatchChunk.build` takes the batches `BatchSchedule.next_batches` hands out -- consecutive rows of one contiguous int64
    matrix -- as ONE flat view (no concatenation on the critical path of a one-chunk run); anything else takes the copying path."""
    from ggad_amd.minibatch import _rows_as_flat
    m = np.empty((20, 200), dtype=np.int64)
    m[:] = np.arange(4000).reshape(20, 200)
    rows = list(m)
    flat = _rows_as_flat(rows)
    assert flat is not None and np.shares_memory(flat, m) and np.array_equal(flat, m.reshape(-1))
    assert np.array_equal(_rows_as_flat(rows[3:9]), m[3:9].reshape(-1))
    for other in ([m[0], m[2], m[4]], [m[0], m[5], m[2]], [m[0], m[2], m[2]], [np.arange(5), np.arange(5)],
                  [m[0], m[3], m[2], m[1], m[4]], [m[0], m[1], m[3], m[2], m[4], m[5], m[6]],      # first / middle / last in place, others permuted
                  [m[0], m[1].copy(), m[2]],
                  [m[0, :100], m[1, :100]], list(m[1::2][:5]), [list(range(5))], [m[0].astype(np.int32), m[1].astype(np.int32)]):
        assert _rows_as_flat(other) is None


def test_row_list_remembers_its_matrix_until_it_is_changed():
    """`BatchSchedule.next_batches` hands its batches out as `RowList`s: `BatchChunk.build` takes their matrix as it is (no per-row
    work on the critical path of a one-chunk run).  Slices stay RowLists over the sliced matrix; every in-place change makes the list
    an ordinary list again -- rows permuted by a hook must not be built in matrix order."""
    from ggad_amd.minibatch import RowList, _rows_as_flat
    m = np.arange(60, dtype=np.int64).reshape(6, 10)
    r = RowList(m)
    assert isinstance(r, list) and len(r) == 6 and np.array_equal(r[4], m[4])
    assert np.shares_memory(_rows_as_flat(r), m) and np.array_equal(_rows_as_flat(r), m.reshape(-1))
    s = r[2:5]
    assert type(s) is RowList and np.array_equal(_rows_as_flat(s), m[2:5].reshape(-1)) and r[0:6] is r and type(r[::2]) is list
    assert len(r[3:3]) == 0 and _rows_as_flat(r[3:3]) is None
    for change in (lambda t: t.reverse(), lambda t: t.__setitem__(0, t[1]), lambda t: t.append(m[0]), lambda t: t.pop(), lambda t: t.sort(key=lambda a: -a[0]),
                   lambda t: t.extend([m[1]]), lambda t: t.insert(0, m[5]), lambda t: t.__delitem__(0)):
        t = r[1:6]
        assert t.matrix is not None
        change(t)
        assert t.matrix is None
        flat = _rows_as_flat(t)                          # (now checked row by row like any list: in matrix order or not at all)
        assert flat is None or np.array_equal(flat, np.concatenate(t))
    swapped = r[0:5]
    swapped[1], swapped[2] = swapped[2], swapped[1]
    assert _rows_as_flat(swapped) is None
    assert RowList(m.astype(np.int32)).matrix is None and RowList(m[:, ::2]).matrix is None


def test_schedule_hands_out_row_lists():
    import random as pyrandom
    from ggad_amd.minibatch import RowList
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule
    labels = np.zeros(5000, dtype=np.int64)
    labels[4000:] = 1
    pyrandom.seed(3)
    sched = BatchSchedule(np.arange(4000), np.arange(4000, 5000), labels, 150, PyCompatRandom.from_python_state(pyrandom.getstate()))
    bn, bl = sched.next_batches(7)
    assert type(bn) is RowList and type(bl) is RowList and bn.matrix.shape == (7, 200) and bl.matrix.shape == (7, 200)
    assert all(np.array_equal(labels[a], b) for a, b in zip(bn, bl))
    b2, l2 = sched.next_batches(3, rank=1, world=2)      # (a rank's rows of a shared stream: their own contiguous matrix)
    assert type(b2) is RowList and b2.matrix is not None and b2.matrix.flags.c_contiguous and len(b2) == 3
