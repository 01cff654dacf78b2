#!/usr/bin/env python3
"""Entry point of the DGraph mini-batch GGAD run:  cd src && python main.py [--config dgraph.yml] [--multi_run]

Mirrors the reference's `src/main.py` CLI (`--config`, `--multi_run`, YAML keys of dgraph.yml) on top of the
MI355X-native `ggad_amd.model_handler.ModelHandler`.  For N GPUs launch it with
`python -m torch.distributed.run --nproc-per-node N main.py ...`: batches are then sharded over the ranks."""
import argparse
import itertools
import os
import sys
import time

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ggad_amd.model_handler import ModelHandler  # noqa: E402


def set_random_seed(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


def print_config(config):
    bar = "**************** MODEL CONFIGURATION ****************"
    print(bar)
    for key in sorted(config):
        if key == "data":              # in-memory dataset handed over by --synthetic: not a setting
            continue
        print("{}{} -->   {}".format(key, " " * (24 - len(key)), config[key]))
    print(bar)


def expand_grid(config):
    """Every list-valued key is a swept hyper-parameter; yields one flat config per combination."""
    swept = [k for k, v in config.items() if isinstance(v, list)]
    fixed = {k: v for k, v in config.items() if k not in swept}
    for combo in itertools.product(*(config[k] for k in swept)):
        cfg = dict(fixed)
        cfg.update(dict(zip(swept, combo)))
        yield swept, cfg


HANDLER = ModelHandler


def use_handler(name):
    """The reference switches models by editing its import of `ModelHandler` (`src/main.py:11`); here it is a flag."""
    global HANDLER
    if name == "dominant":
        from ggad_amd.model_handler_dominate import ModelHandler as H
    elif name == "anomalydae":
        from ggad_amd.model_handler_anomalydae import ModelHandler as H
    elif name == "aegis":
        from ggad_amd.model_handler_aegis import ModelHandler as H
    else:
        H = ModelHandler
    HANDLER = H


def run_once(config):
    set_random_seed(config["seed"])
    return HANDLER(config).train()


def main(config):
    print_config(config)
    res = run_once(config)
    if res is None:            # the comparison models' handlers only print (src/model_handler_dominate.py:171)
        return
    f1_mac, f1_1, f1_0, auc, gmean = res
    print("F1-Macro: {}".format(f1_mac))
    print("AUC: {}".format(auc))
    print("G-Mean: {}".format(gmean))


def multi_run_main(config):
    print_config(config)
    results = []
    for i, (swept, cfg) in enumerate(expand_grid(config)):
        print("Running {}:\n".format(i))
        for k in swept:
            cfg["save_dir"] += "{}_{}_".format(k, cfg[k])
        print(cfg["save_dir"])
        st = time.time()
        results.append(run_once(cfg))
        print("Running {} done, elapsed time {}s".format(i, time.time() - st))
    arr = np.array(results, dtype=np.float64)
    names = ["F1-Macro", "F1-binary-1", "F1-binary-0", "AUC", "G-Mean"]
    for j in (0, 3, 4):
        print("{}: {}".format(names[j], arr[:, j].tolist()))
    for j, name in enumerate(names):
        std = arr[:, j].std(ddof=1) if len(arr) > 1 else float("nan")
        print("{}: {}+{}".format(name, arr[:, j].mean(), std))


def init_distributed():
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="dgraph.yml", help="")
    ap.add_argument("--multi_run", action="store_true", help="flag: multi run")
    ap.add_argument("--synthetic", action="store_true",
                    help="no ../data/dgraphfin.npz here: train on a synthetic graph of DGraph-Fin's size (power-law degrees, "
                         "U[0,1) features, 0.42 %% anomalies) built from the config's seed")
    ap.add_argument("--synthetic_entries", type=int, default=73105508, help="directed entries of the synthetic graph "
                    "(73.1 M = BASELINE's figure; 8600000 = the public dataset's average degree 2.3)")
    ap.add_argument("--num_epochs", type=int, default=None, help="override the config's num_epochs")
    ap.add_argument("--handler", choices=["ggad", "dominant", "anomalydae", "aegis"], default="ggad",
                    help="which ModelHandler drives the run: GGAD (default) or one of the mini-batch comparison models")
    a = ap.parse_args()
    with open(a.config, "r") as fh:
        cfg = yaml.load(fh, Loader=yaml.FullLoader)
    if a.num_epochs is not None:
        cfg["num_epochs"] = a.num_epochs
    use_handler(a.handler)
    init_distributed()
    torch.set_num_threads(min(8, os.cpu_count() or 1))        # host tensors are tiny here; 128 intra-op threads only add jitter
    if a.synthetic:
        from ggad_amd import synth
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        n = 3700550
        rowptr, col = synth.make_graph_torch(n, a.synthetic_entries, cfg["seed"], dev, kind="powerlaw", max_degree=2000)
        cfg["data"] = ((rowptr, col), synth.make_features(n, 17, cfg["seed"]),
                       synth.make_labels(n, 15509.0 / 3700550.0, cfg["seed"]).astype(np.int32))
    (multi_run_main if a.multi_run else main)(cfg)
