cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, json, sys
def run(args):
    out = subprocess.run([sys.executable, "bench.py"] + args, capture_output=True, text=True)
    l = out.stdout.strip().split("\n")[-1]
    try:
        d = json.loads(l); return round(d["value"]), round(d["ms_per_step"] * 1e3, 2)
    except Exception:
        return (l[-300:], out.stderr[-500:])
for extra in ([], ["--dp-path"], ["--dp-path", "--exchange", "rccl"], []):
    print(extra, run(["--steps", "4500", "--warmup", "150", "--no-extras"] + extra), flush=True)
PY
