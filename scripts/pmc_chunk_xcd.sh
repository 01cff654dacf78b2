#!/bin/bash
# Regenerates profiles/r0N_pmc_chunk_xcd.json from THIS build: hardware counters of k_train_chunk_xcd (the dense steps of a chunk as one
# launch resident on one XCD) in the K = 20 command the driver runs -- bench.py --steps 20 --warmup 5 --no-extras --, one rocprofv3 pass
# per counter group (--kernel-trace only).  bench.py reads its `issue_us_per_step` for the chunk kernel's `by_kernel` entry.
# Usage (GPU box): GGAD_COMMIT=$(git log -1 --format=%h -- ggad_amd/csrc/step_xcd.hip ggad_amd/csrc/step_common.h) bash scripts/pmc_chunk_xcd.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/cx
rm -rf $OUT; mkdir -p $OUT
i=0
rocprofv3 --kernel-trace -d $OUT/t -o t -- python $R/bench.py --steps 20 --warmup 5 --no-extras > $OUT/t.log 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p -- python $R/bench.py --steps 20 --warmup 5 --no-extras > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed"
done
python3 - <<PY
import glob, json, sqlite3
R, tag = "$R", "$TAG"
def dbs(pat):
    return sorted(glob.glob(pat, recursive=True))
# kernel durations of the trace pass: dispatches of k_train_chunk_xcd in launch order (warm-up 5 steps, timed 20, instrumented repeat 20)
db = sqlite3.connect(dbs("$OUT/t/**/*.db")[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
nc = "name" if "name" in cols else "kernel_name"
dur = [(e - s) / 1e3 for n, s, e in db.execute(f"select {nc}, start, end from kernels order by start") if "k_train_chunk_xcd" in n]
ctr = {}
for d in dbs("$OUT/p*/**/*.db"):
    c = sqlite3.connect(d)
    rows = c.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection").fetchall()
    by = {}
    for name, cn, disp, val in rows:
        if "k_train_chunk_xcd" in name:
            by.setdefault(cn, {}).setdefault(disp, 0.0)
            by[cn][disp] += float(val)
    for cn, dd in by.items():
        ctr[cn] = [dd[k] for k in sorted(dd)]
steps = 20
pick = lambda v: float(sum(v[-2:]) / len(v[-2:])) if v else None      # the two 20-step launches (timed window + its repeat)
out = {"note": "k_train_chunk_xcd in bench.py --steps 20 --warmup 5 --no-extras: rocprofv3 PMC, one counter group per run (--kernel-trace only); values = mean "
               "of the two 20-step launches, whole launch (all workgroups, including the 7 x 28 that leave at once).  SQ_* cycle counters are in "
               "quad-cycles (MI355X_MICROARCH.md), instruction counters in wave-instructions.  issue_us_per_step = SQ_ACTIVE_INST_ANY x 4 clocks / "
               "(28 CUs x 4 SIMDs) / 2.4 GHz / 20 steps: the time the launch's instructions need if every SIMD of the XCD issued one of them at a time "
               "without a gap",
       "commit": "${GGAD_COMMIT:-unknown}", "generated_by": "scripts/pmc_chunk_xcd.sh", "steps_per_launch": steps,
       "kernel_us": pick(dur), "kernel_us_per_step": (pick(dur) / steps) if dur else None, "counters": {k: pick(v) for k, v in sorted(ctr.items())}}
a = out["counters"].get("SQ_ACTIVE_INST_ANY")
out["issue_us_per_step"] = (a * 4.0 / (28 * 4) / 2.4e3 / steps) if a else None
insts = sum(out["counters"].get(k) or 0.0 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
out["wave_instructions_per_step"] = insts / steps if insts else None
out["wave_instructions_per_wave_per_step"] = insts / steps / (28 * 8) if insts else None
json.dump(out, open(f"{R}/gpurun_out/{tag}_pmc_chunk_xcd.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
