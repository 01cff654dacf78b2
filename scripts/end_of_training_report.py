"""End-of-training parity report (README.md quotes its output): python scripts/end_of_training_report.py
Runs the long schedules of tests/parity_long.py on cuda:0 -- the two round-5 fixtures (labels independent of the inputs) and the three
round-6 fixtures on PLANTED anomalies (the reference separates the classes) -- and prints the deltas against the reference's run as
JSON + a table."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GGAD_CAPTURE_BELOW_S", "10")
import parity_long as P  # noqa: E402
from conftest import load_golden  # noqa: E402

rows = []
for tag, fixture in (("full-graph script, Photo's schedule, independent labels", "fullgraph_long_photo_schedule.npz"),
                     ("full-graph script, planted anomalies, --num_epoch 50", "fullgraph_long_planted.npz"),
                     ("full-graph script, planted anomalies, 100 epochs (ill-conditioned past epoch 59)", "fullgraph_long_planted_100.npz")):
    r = P.full_graph_long(fixture=fixture)
    g = load_golden(fixture)
    print("FULL", fixture, json.dumps(r))
    sens = (f"; reference's own sensitivity {float(g['self_sens_auc']):.1e} / {float(g['self_sens_ap']):.1e}" if "self_sens_auc" in g else "")
    rows.append(f"| {tag} | {r['epochs']} epochs | {r['loss_delta_max']:.1e} | {r['final_auc'][0]:.6f} / {r['final_auc'][1]:.6f} | "
                f"{r['final_auc_delta']:.1e} | {r['final_ap'][0]:.6f} / {r['final_ap'][1]:.6f} | {r['final_ap_delta']:.1e} | max score delta {r['final_score_delta_max']:.1e}{sens} |")
for tag, fixture in (("ModelHandler, 5 epochs x 150 batches, independent labels", "handler_dgraph_like_5ep.npz"),
                     ("ModelHandler, 5 epochs x 150 batches, planted anomalies", "handler_dgraph_like_planted.npz")):
    with tempfile.TemporaryDirectory() as d:
        m = P.handler_long(d, fixture=fixture)
    print("MINI", fixture, json.dumps(m))
    ap = m.get("sweep_ap")
    rows.append(f"| {tag} | {m['batches']} batches | {m['loss_delta_max']:.1e} | {m['test_metrics'][0][3]:.6f} / {m['test_metrics'][1][3]:.6f} | "
                f"{m['test_auc_delta']:.1e} | " + (f"{ap[0][-1]:.6f} / {ap[1][-1]:.6f} | {m['sweep_ap_delta_max']:.1e}" if ap else "- | -") +
                f" | end weights {m['end_weight_delta_max']:.1e} |")
print("| schedule | steps | max |d loss| | final AUROC (HIP / reference) | |d AUROC| | final AP (HIP / reference) | |d AP| | |")
print("|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
