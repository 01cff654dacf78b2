cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, json, sys
def run(args):
    out = subprocess.run([sys.executable, "bench.py"] + args, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    try:
        d = json.loads(out)
        return round(d["value"]), round(d["ms_per_step"] * 1e3, 2), d["roofline"]["avg_launch_ms"]
    except Exception as e:
        return out[-300:]
for cus in "56 64 72 80 96 112".split():
    print("dense_cus", cus, run(["--steps", "4500", "--warmup", "150", "--no-extras", "--dense-cus", cus]), flush=True)
PY
