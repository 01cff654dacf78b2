#!/bin/bash
# Regenerates profiles/r0N_pmc_gather2_items.json from THIS build: HBM-side bytes per gathered neighbour of k_gather2_items at the
# two launch sizes bench.py uses (20 and 150 batches per launch).  Each counter group in its own rocprofv3 run (--kernel-trace
# only), as MI355X_MICROARCH.md's HBM section prescribes; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B).
# Usage (GPU box): GGAD_COMMIT=$(git log -1 --format=%h -- ggad_amd/csrc/hop2_ldsw.hip ggad_amd/csrc/plan_build.cpp ggad_amd/csrc/plan.hip) \
#                   bash scripts/pmc_gather2.sh [tag]      -> gpurun_out/<tag>_pmc_gather2_items.json   (tests/test_profiles.py checks the stamp)
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/g0 /tmp/g1 /tmp/g2
rocprofv3 --kernel-trace -d /tmp/g0 -o t -- python $R/scripts/plan_kernel_times.py 20,150 1 > /tmp/g0.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/g1 -o p -- python $R/scripts/plan_kernel_times.py 20,150 1 > /tmp/g1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d /tmp/g2 -o p -- python $R/scripts/plan_kernel_times.py 20,150 1 > /tmp/g2.log 2>&1
python $R/scripts/rocpd_by_size.py $(find /tmp/g0 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_plan_kernels_by_size.csv
python $R/scripts/rocpd_pmc_by_dispatch.py $(find /tmp/g1 -name "*.db" | head -1) > $R/gpurun_out/${TAG}_plan_pmc_by_dispatch.csv
python $R/scripts/rocpd_pmc_by_dispatch.py $(find /tmp/g2 -name "*.db" | head -1) >> $R/gpurun_out/${TAG}_plan_pmc_by_dispatch.csv
python - <<PY
import csv, json, re
R = "$R"; tag = "$TAG"
pairs = [int(m.group(1)) for m in re.finditer(r"pairs (\d+)", open("/tmp/g0.log").read())]
times = list(csv.DictReader(open(f"{R}/gpurun_out/{tag}_plan_kernels_by_size.csv")))
pmc = {}
for k, order, ctr, val in csv.reader(open(f"{R}/gpurun_out/{tag}_plan_pmc_by_dispatch.csv")):
    pmc[(k, int(order), ctr)] = float(val)
out = {"note": "HBM-side bytes per gathered neighbour (per (batch, owner) occurrence) of k_gather2_items, rocprofv3 PMC passes of "
               "scripts/plan_kernel_times.py 20,150 (each counter group its own run, --kernel-trace only): TCC_EA0_RDREQ_sum x 128 B "
               "(= FETCH_SIZE KB x 1024 x 2 on gfx950, MI355X_MICROARCH.md HBM section) / (owner, neighbour) pairs of the launch",
       "commit": "${GGAD_COMMIT:-unknown}", "generated_by": "scripts/pmc_gather2.sh", "by_batches_per_launch": {}}
for i, nb in enumerate((20, 150)):
    g = lambda c: pmc.get(("k_gather2_items", i, c))
    rd = g("TCC_EA0_RDREQ_sum")
    out["by_batches_per_launch"][str(nb)] = {
        "pairs": pairs[i] if i < len(pairs) else None, "tcc_ea0_rdreq": rd, "fetch_size_kb": g("FETCH_SIZE"),
        "hbm_bytes": rd * 128 if rd else None,
        "hbm_bytes_per_neighbour": (rd * 128 / pairs[i]) if (rd and i < len(pairs)) else None,
        "tcc_hit": g("TCC_HIT_sum"), "tcc_miss": g("TCC_MISS_sum"),
        "kernel_us_alone": float(times[i]["k_gather2_items"]) if i < len(times) else None,
        "k_tile_counts_us_alone": float(times[i]["k_tile_counts"]) if i < len(times) else None,
        "k_tile_counts_tcc_ea0_rdreq": pmc.get(("k_tile_counts", i, "TCC_EA0_RDREQ_sum")),
        "k_tile_counts_hbm_bytes_per_neighbour": (pmc.get(("k_tile_counts", i, "TCC_EA0_RDREQ_sum")) * 128 / pairs[i])
                                                 if (pmc.get(("k_tile_counts", i, "TCC_EA0_RDREQ_sum")) and i < len(pairs)) else None,
        "plan_span_us_alone": float(times[i]["span_us"]) if i < len(times) else None}
json.dump(out, open(f"{R}/gpurun_out/{tag}_pmc_gather2_items.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
