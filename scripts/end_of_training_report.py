"""End-of-training parity report (README.md quotes its output): python scripts/end_of_training_report.py
Runs the two long schedules of tests/parity_long.py on cuda:0 and prints the deltas against the reference's run as JSON + a table."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_long as P  # noqa: E402

full = P.full_graph_long()
print("FULL", json.dumps(full))
with tempfile.TemporaryDirectory() as d:
    mini = P.handler_long(d)
print("MINI", json.dumps(mini))
print("| schedule | steps | max |d loss| over the run | final AUROC (HIP / reference) | |d AUROC| | |d AP| | max |d score| |")
print("|---|---|---|---|---|---|---|")
print(f"| full-graph script, Photo's schedule (N = 4,200, H = 300) | {full['epochs']} epochs | {full['loss_delta_max']:.2e} | "
      f"{full['final_auc'][0]:.6f} / {full['final_auc'][1]:.6f} | {full['final_auc_delta']:.1e} | {full['final_ap_delta']:.1e} | {full['final_score_delta_max']:.1e} |")
print(f"| ModelHandler, 5 epochs x 150 batches, 3 validation sweeps (N = 90,000) | {mini['batches']} batches | {mini['loss_delta_max']:.2e} | "
      f"{mini['test_metrics'][0][3]:.6f} / {mini['test_metrics'][1][3]:.6f} | {mini['test_auc_delta']:.1e} | - | end weights {mini['end_weight_delta_max']:.1e} |")
