"""Synthetic full-graph configs at the published sizes, and their measurement (shared by run.py, bench.py and the full-size tests).

No dataset ships with this repo (reference `README.md:45-48`): `make_dataset` builds, from the seed, a symmetric power-law graph with
EXACTLY the published number of directed entries (`synth.make_graph(exact=True)`), U[0,1) features, Bernoulli anomaly labels at the
published rate, and then follows `run.py:84-110` of the reference: `split_nodes` (python `random`), `preprocess_features` for the
datasets the reference normalises (`run.py:87`: not Photo, and -- a typo there -- not T-Finance), the noise parameters of `run.py:61-66`.

`bench_fullgraph` times the training epoch of `run.py:142-214` (forward, loss block, backward, Adam) per config and the two kernels that
bound it, with their roofline fractions (CSR SpMM bytes = 8 nnz + 4 (N + 1) + 8 N H, flops = 2 nnz H; f32 MFMA
157.3 TF; HBM 8 TB/s).
"""
from __future__ import annotations

import os
import random
import time
import types
from typing import Dict, Optional

import numpy as np

from . import synth
from .utils import normalize_adj, preprocess_features, split_nodes

# published sizes (reference README.md:53-58): nodes, directed entries, features, anomaly rate
SIZES = {"reddit": (10984, 168016, 64, 0.033), "Amazon": (11944, 4398392, 25, 0.069), "photo": (7535, 119043, 745, 0.092),
         "t_finance": (39357, 21222543, 10, 0.046), "elliptic": (46564, 73248, 93, 0.098)}
EPOCHS = {"photo": 100, "elliptic": 150, "reddit": 300, "t_finance": 500, "Amazon": 800}
NORMALISED = ("Amazon", "tf_finace", "reddit", "elliptic")            # run.py:87 (typo kept: never T-Finance)
HBM_PEAK = 8.0e12
F32_PEAK = 157.3e12
LDS_PEAK = 150.0e12             # ds_read_b64 / ds_read_b128 with every CU streaming (MI355X_MICROARCH.md, LDS section: 256 B/clk/CU at ~2.4 GHz)


def make_dataset(name: str, seed: int = 0, verbose: bool = False) -> Dict[str, object]:
    """The inputs of `run.py` for config `name`: scipy adjacency, dense feature matrix (normalised as the reference would),
    labels, index lists, noise parameters.  Consumes python's `random` stream like `load_mat` does (seed it first)."""
    import scipy.sparse as sp
    n, ne, f, rate = SIZES[name]
    rowptr, col = synth.make_graph(n, ne, seed, kind="powerlaw", max_degree=max(64, n // 8), exact=True)
    adj = synth.csr_to_scipy(rowptr, col, n)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, rate, seed)
    all_idx, idx_train, idx_val, idx_test, normal_idx, abn_idx = split_nodes(ano, name, verbose=verbose)
    features = preprocess_features(sp.lil_matrix(feat)) if name in NORMALISED else np.asarray(feat)
    mean, var = (0.02, 0.01) if name in ("reddit", "photo") else (0.0, 0.0)                     # run.py:61-66
    return dict(name=name, n=n, f=f, adj=adj, rowptr=rowptr, col=col, features=np.asarray(features, dtype=np.float32), ano=ano,
                idx_test=idx_test, normal_idx=normal_idx, abn_idx=abn_idx, mean=mean, var=var)


def build_model(ds, dev, h: int = 300, seed: int = 0):
    """FullGraphAdj + Model + FlatAdam as `run.py` builds them (`:98-118`)."""
    import scipy.sparse as sp
    import torch
    from .fullgraph import FlatAdam, FullGraphAdj
    from .model import Model
    n = ds["n"]
    full = FullGraphAdj(normalize_adj(ds["adj"]) + sp.eye(n), ds["adj"] + sp.eye(n), dev)
    torch.manual_seed(seed)
    model = Model(ds["f"], h, "prelu", 1, "avg").to(dev)
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
    feats = torch.from_numpy(ds["features"])[None].to(dev)
    return full, model, opt, feats


def _time_call(fn, reps: int = 20, graph: bool = None) -> float:
    """Seconds per call on the device.  `graph` (default: GGAD_TIME_GRAPH != 0): the calls are captured into one hipGraph and the
    replay is timed, so the figure is the kernel's, not the Python launch path's (20-30 us per eager call -- more than the sparse
    products themselves take)."""
    import torch
    fn()
    torch.cuda.synchronize()
    if graph is None:
        graph = os.environ.get("GGAD_TIME_GRAPH", "1") != "0"
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(reps):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def bench_one(name: str, dev, epochs: int = 30, seed: int = 0, h: int = 300) -> Dict[str, object]:
    """Median epoch time of the training epoch (eager launches, then -- as `run.py` does -- one captured hipGraph when the eager
    epoch is launch-bound) and the times of the N x N x H product and the N x H x H projection of that config."""
    import torch
    from . import fullgraph as FG
    random.seed(seed)
    np.random.seed(seed)
    ds = make_dataset(name, seed)
    full, model, opt, feats = build_model(ds, dev, h, seed)
    args = types.SimpleNamespace(mean=ds["mean"], var=ds["var"])
    abn, nrm = ds["abn_idx"], ds["normal_idx"]
    ls = full.loss_structs(nrm, abn)

    one = torch.ones((), dtype=torch.float32, device=dev)

    def train_epoch():
        opt.zero_grad()
        emb, emb_combine, logits, emb_con, emb_abnormal = model(feats, full, abn, nrm, True, args)
        out = FG.ggad_loss(emb, logits, emb_con, emb_abnormal, full, ls, 0.7)
        out[0].backward(gradient=one)            # (d loss / d loss = 1 from a kept tensor: autograd would fill a new one every epoch)
        opt.step()
        return out

    model.train()
    # the only host-side tensor work of an epoch is the N(mean, var) noise draw: cap torch's intra-op threads like run.py does
    # (one thread per core turns the 844 x 300 randn into a 1-90 ms lottery on a loaded 128-core host)
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(min(8, prev_threads))
    try:
        eager = []
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = train_epoch()
            torch.cuda.synchronize()
            eager.append(time.perf_counter() - t)
        eager_med = float(np.median(eager[2:]))
        times, mode = [], "eager"
        graph = None
        if eager_med < 20e-3:
            noise_buf = torch.zeros(1, len(abn), h, device=dev)
            model.noise_override = noise_buf
            out = None
            opt.zero_grad()
            import gc
            gc.collect()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static = train_epoch()
            model.noise_override = None
            mode = "hipGraph"

        def draw():
            return torch.randn(1, len(abn), h) * ds["var"] + ds["mean"]              # same draw as Model.forward

        pending = None
        for _ in range(epochs):
            torch.cuda.synchronize()
            t = time.perf_counter()
            if graph is not None:
                noise_buf.copy_(pending if pending is not None else draw())
                graph.replay()
                pending = draw()               # the next epoch's draw while the GPU runs this one (run.py does the same)
            else:
                out = train_epoch()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t)
    finally:
        torch.set_num_threads(prev_threads)
    med = float(np.median(times))
    loss = float((static if graph is not None else out)[0].item())
    # the two kernels that bound the epoch, alone: A_hat (N x H)  and  (N x H)(H x H)^T
    n, nnz = ds["n"], int(full.A.nnz)
    x = torch.randn(n, h, device=dev)
    w = torch.randn(h, h, device=dev)
    xa = FG.padded_rows(n, h, dev, (full.A, None))                     # the rows the layer's projection writes (128-byte aligned
    xa.copy_(x)                                                        # on the sparse configs: the line-granular product)
    t_spmm = _time_call(lambda: FG.spmm(full.A, xa))
    plan0 = full.A.plan()
    pp0 = FG._use_panel(full.A, plan0, x)
    kernel = (("k_spmm_ring" if "wave_sb" in pp0 else "k_spmm_panel") if pp0 is not None
              else "k_spmm_sliced" if FG._use_sliced(full.A, plan0, x)
              else "k_spmm_rowline" if not xa.is_contiguous()
              else "k_spmm_rowslice" if FG._use_rowslice(full.A, plan0, x) else "k_spmm_seg")
    t_gemm = _time_call(lambda: FG.gemm(x, w, False, True))
    spmm_bytes = 8.0 * nnz + 4.0 * (n + 1) + 8.0 * n * h
    spmm_flops = 2.0 * nnz * h
    gemm_flops = 2.0 * n * h * h
    bound = "fp32-fma" if spmm_flops / F32_PEAK > spmm_bytes / HBM_PEAK else "hbm"
    lds = {}
    if pp0 is not None and "wave_sb" in pp0:
        # k_spmm_ring never reaches the FP32-FMA peak: every stored entry is one 4 H-byte row read out of LDS (ds_read_b128, 1 KB per wave
        # instruction) and one row addition.  Its real bound is the LDS read rate (MI355X_MICROARCH: 256 B/clk/CU, ~150 TB/s over the chip):
        # floor = 4 H bytes per walked entry / that rate; `lds_bytes` is what the schedule actually issues (padded steps, the zero columns of
        # the last 32-float slice) -- the gap between the two is the schedule's, the gap to `us` the kernel's.
        n_sl = (h + 31) // 32
        walked = nnz - (n if pp0.get("diag") is not None else 0)
        lds_alg = 4.0 * h * walked
        lds_issued = float(pp0["quads"]) * 4 * 1024 * n_sl
        bound = "lds"
        lds = {"lds_alg_bytes": lds_alg, "lds_bytes": lds_issued, "lds_peak_tbs": LDS_PEAK / 1e12, "lds_floor_us": lds_alg / LDS_PEAK * 1e6,
               "frac_of_lds_floor": lds_alg / LDS_PEAK / t_spmm, "lds_issued_floor_us": lds_issued / LDS_PEAK * 1e6,
               "frac_of_lds_issued_floor": lds_issued / LDS_PEAK / t_spmm, "schedule_fill": float(pp0["fill"])}
    return {"nodes": n, "stored_entries_incl_identity": nnz, "directed_entries": int(ds["adj"].nnz), "feat": ds["f"], "hidden": h,
            "epoch_ms": med * 1e3, "nodes_per_s": n / med, "mode": mode, "epochs_timed": epochs, "eager_epoch_ms": eager_med * 1e3,
            "loss_after": loss,
            "spmm_NxNxH": {"kernel": kernel, "us": t_spmm * 1e6, "tflops": spmm_flops / t_spmm / 1e12, "alg_gbs": spmm_bytes / t_spmm / 1e9,
                           "bound": bound, "frac_of_f32_fma_peak": spmm_flops / t_spmm / F32_PEAK,
                           "frac_of_hbm_peak": spmm_bytes / t_spmm / HBM_PEAK,
                           "floor_us": max(spmm_flops / F32_PEAK, spmm_bytes / HBM_PEAK) * 1e6,
                           "frac_of_bounding_roofline": (lds["frac_of_lds_floor"] if lds else
                                                         max(spmm_flops / F32_PEAK, spmm_bytes / HBM_PEAK) / t_spmm), **lds},
            "gemm_NxHxH": {"us": t_gemm * 1e6, "tflops": gemm_flops / t_gemm / 1e12, "frac_of_f32_mfma_peak": gemm_flops / t_gemm / F32_PEAK}}


def bench_fullgraph(dev, epochs: int = 30, names: Optional[list] = None) -> Dict[str, object]:
    """`run.py --synthetic` epochs of the four BASELINE full-graph configs (Reddit is the reference's CPU-runnable case)."""
    import torch
    out = {}
    state = random.getstate()
    try:
        for name in (names or ["reddit", "Amazon", "photo", "t_finance"]):
            with torch.cuda.device(dev):
                out[name] = bench_one(name, dev, epochs)
            torch.cuda.empty_cache()
    finally:
        random.setstate(state)
    return out
