"""The C-ABI library loads on a CPU-only host and exports every symbol include/ggad_hip.h declares."""
import ctypes
import os
import re

import numpy as np

import pytest

from conftest import ROOT
from ggad_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ggad_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ggad_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_loads():
    from ggad_amd.build import build
    path = build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    assert lib.ggad_abi_version() == _lib.ABI_VERSION == 10


def test_only_the_c_abi_is_exported():
    """The dynamic symbol table holds the ggad_* entry points of include/ggad_hip.h and nothing else: no C++-mangled internal
    (csrc/libggad_hip.map)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert len(names) > 100
    assert [n for n in names if not n.startswith("ggad_")] == []
    assert set(names) == set(_declared_symbols())


def test_integration_doc_stub_runs_against_this_library():
    """The ctypes stub INTEGRATION.md shows a maintainer (section 2): its first lines are executed as written -- the library
    loads, the ABI version it asserts is the one the library reports, the structures it imports exist."""
    import re
    text = open(os.path.join(os.path.dirname(__file__), "..", "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes, numpy as np, torch.*?)\n```", text, re.S).group(1)
    head = "\n".join(block.splitlines()[:6])
    assert "ggad_abi_version() ==" in head and "argtypes" in head
    ns = {}
    exec(compile(head, "INTEGRATION.md", "exec"), ns)
    assert ns["lib"].ggad_abi_version() == _lib.ABI_VERSION
    assert f"ABI version {_lib.ABI_VERSION}" in text


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) > 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ggad_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in ggad_amd/_lib.py"
    for name in _lib.SIGNATURES:
        assert name in declared, f"{name} bound in _lib.py but not declared in ggad_hip.h"


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.GgadLibraryError):
        _lib.load(str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.load()


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # null pointers are rejected before any launch
    assert lib.ggad_exclusive_scan_i32(None, None, 4, None, None) == -1
    assert lib.ggad_mb_params_sync(None, 64, 17, None) == -1
    assert lib.ggad_mb_param_count(64, 17) == 5248
    assert lib.ggad_max_embed_dim() == 64


def test_csr_cache_round_trip(tmp_path):
    """Binary CSR cache of the reference's dict-of-sets adjacency (host side; no GPU needed)."""
    from ggad_amd import synth
    from ggad_amd.graph import DeviceGraph
    rowptr, col = synth.make_graph(500, 4000, 3, kind="powerlaw", max_degree=40)
    adj = synth.csr_to_adj_lists(rowptr, col)
    path = str(tmp_path / "adj.csr.npz")
    src = str(tmp_path / "adj_list.pickle")
    open(src, "wb").write(b"x" * 100)                                        # stands for the pickle the dict came from
    g1 = DeviceGraph.from_adj_lists_cached(adj, 500, "cpu", path, source_path=src)
    g2 = DeviceGraph.from_adj_lists_cached(None, 500, "cpu", path, source_path=src)   # served from the cache: the dict is not touched
    assert np.array_equal(g1.rowptr_host, rowptr) and np.array_equal(g1.col_host, col)
    assert np.array_equal(g2.rowptr_host, rowptr) and np.array_equal(g2.col_host, col)
    g3 = DeviceGraph.from_adj_lists_cached(adj, 400 + 100, "cpu", str(tmp_path / "missing" / "x.npz"))   # unwritable: still works
    assert g3.nnz == g1.nnz
    # a regenerated adjacency with the same node count must not be served from the stale cache (fingerprint of the source)
    rowptr2, col2 = synth.make_graph(500, 3000, 4, kind="powerlaw", max_degree=40)
    adj2 = synth.csr_to_adj_lists(rowptr2, col2)
    open(src, "wb").write(b"y" * 137)
    g4 = DeviceGraph.from_adj_lists_cached(adj2, 500, "cpu", path, source_path=src)
    assert np.array_equal(g4.rowptr_host, rowptr2) and np.array_equal(g4.col_host, col2)
    g5 = DeviceGraph.from_adj_lists_cached(adj, 500, "cpu", path)             # no source file: keyed by the dict's counts
    assert np.array_equal(g5.col_host, col)
    # a truncated cache file (a rank killed mid-write under the old fixed temporary name) is ignored, not fatal
    open(path, "wb").write(open(path, "rb").read()[:200])
    g6 = DeviceGraph.from_adj_lists_cached(adj, 500, "cpu", path)
    assert np.array_equal(g6.col_host, col)


def test_load_mat_split_equals_reference(tmp_path, capsys):
    """`load_mat` (reference utils.py:66-141): same index lists and the same consumption of python's `random` stream as the
    imported reference produced on a .mat written from the same seeded inputs (tests/golden/ingest_mat.npz)."""
    import random

    import scipy.io as sio
    import scipy.sparse as sp
    from conftest import load_golden
    from ggad_amd import synth
    from ggad_amd.utils import load_mat
    g = load_golden("ingest_mat.npz")
    n, ne, f, seed = int(g["n"]), int(g["n_entries"]), int(g["f"]), int(g["seed"])
    rowptr, col = synth.make_graph(n, ne, seed, kind="powerlaw", max_degree=60)
    adj = synth.csr_to_scipy(rowptr, col)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, 0.08, seed)
    assert synth.crc_of(rowptr, col, feat, ano) == int(g["inputs_crc"])
    for name, keys in (("Amazon", ("Network", "Attributes", "Label")), ("tiny", ("A", "X", "gnd"))):
        path = str(tmp_path / (name + ".mat"))
        sio.savemat(path, {keys[0]: sp.csr_matrix(adj), keys[1]: sp.csr_matrix(feat), keys[2]: ano.reshape(-1, 1)})
        random.seed(seed)
        r = load_mat(name, path=path)
        assert len(r) == 12 and abs(r[0] - adj).nnz == 0 and np.allclose(np.asarray(r[1].todense()), feat)
        assert np.array_equal(r[2], ano) and r[8] is None and r[9] is None
        for k, idx in (("all_idx", 3), ("idx_train", 4), ("idx_val", 5), ("idx_test", 6), ("normal_idx", 10), ("abnormal_idx", 11)):
            assert np.array_equal(np.asarray(r[idx], dtype=np.int64), g[f"{name}.{k}"]), (name, k)
        assert [random.getrandbits(32) for _ in range(3)] == g[name + ".tail"].tolist()
    capsys.readouterr()


def test_host_utilities_of_src_utils(tmp_path):
    """`sparse_to_adjlist` (src/utils.py:96-112), `pos_neg_split` (:115-130), `pick_step` (:133-137) against their plain
    python statements."""
    import pickle
    import random
    from collections import defaultdict

    import scipy.sparse as sp
    from ggad_amd.dgraph import load_dgraphfin, sparse_to_adjlist
    from ggad_amd.sage_utils import pick_step, pos_neg_split
    rng = np.random.default_rng(0)
    m = sp.random(60, 60, density=0.06, random_state=1, format="csr")
    ref = defaultdict(set)
    r, c = m.nonzero()
    for a, b in zip(r, c):
        ref[a].add(b)
        ref[b].add(a)
    path = str(tmp_path / "adj_list")
    got = sparse_to_adjlist(m, path)
    assert dict(got) == dict(ref) and isinstance(got, defaultdict)
    with open(path, "rb") as fh:
        assert dict(pickle.load(fh)) == dict(ref)
    np.savez(str(tmp_path / "d.npz"), x=rng.random((60, 3)), y=(np.arange(60) % 7 == 0).astype(np.int64))
    homo, feat, labels = load_dgraphfin(str(tmp_path / "d.npz"), path)
    assert dict(homo) == dict(ref) and feat.dtype == np.float32 and labels.sum() == 9
    nodes = [5, 9, 2, 7, 11]
    assert pos_neg_split(nodes, [0, 1, 0, 1, 0]) == ([9, 7], [5, 2, 11])
    assert pos_neg_split([4, 4, 3], [1, 0, 0]) == ([4], [4, 3])
    adj = {n: set(range(n % 4 + 1)) for n in nodes}
    y = np.array([0, 1, 0, 0, 1])
    random.seed(3)
    a = pick_step(nodes, y, adj, 6)
    random.seed(3)
    deg = [len(adj[n]) for n in nodes]
    lf = (y.sum() - len(y)) * y + len(y)
    assert a == random.choices(nodes, weights=np.array(deg) / lf, k=6)
