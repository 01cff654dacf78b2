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
