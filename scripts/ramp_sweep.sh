cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, json, sys
def run(args):
    out = subprocess.run([sys.executable, "bench.py"] + args, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    try:
        d = json.loads(out)
        return d["value"], d["ms_per_step"], d["config"]["chunks_of_timed_region"][:6], d["roofline"]["frac"]
    except Exception as e:
        return out[-300:]
# one process per variant is slow (setup 20 s each): do a few only
for ramp in ["7,7", "14", "10", "20", "12"]:
    print("ramp", ramp, run(["--steps", "20", "--warmup", "5", "--no-extras", "--ramp", ramp]), flush=True)
print("default600", run(["--steps", "600", "--warmup", "150", "--no-extras"]), flush=True)
print("default9000", run(["--no-extras"]), flush=True)
PY
