#!/bin/bash
# GEMM evidence for profiles/: kernel times + MFMA-busy counters of the full-graph projection shapes (scripts/gemm_pmc.py).
# PMC passes run with --kernel-trace only (never with hip/hsa/sys traces), one counter group per run.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gemm_prof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/scripts/gemm_pmc.py > $OUT/trace.log 2>&1
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$tag -o p -- python $R/scripts/gemm_pmc.py > $OUT/pmc_$tag.log 2>&1 || echo "pass $tag failed" >> $OUT/errors.log
done
ls -R $OUT | head -40
# reductions
python $R/scripts/rocpd_stats.py $(ls $OUT/trace/*/*.db | head -1) $OUT/gemm_kernel_stats.csv > /dev/null 2>&1
python - $(ls $OUT/trace/*/*.db | head -1) > $OUT/gemm_by_dispatch.csv <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print("dispatch,kernel,us")
for i, (n, s, e) in enumerate(db.execute("select name, start, end from kernels order by start")):
    if "gemm" in n or "splitk" in n:
        print(f"{i},{n.split('(')[0][:90]},{(e - s) / 1e3:.2f}")
PY
cat $OUT/gemm_by_dispatch.csv
: > $OUT/gemm_pmc_by_dispatch.csv
for d in $OUT/pmc_*/; do
  db=$(ls $d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/scripts/rocpd_pmc_by_dispatch.py $db >> $OUT/gemm_pmc_by_dispatch.csv
done
cat $OUT/gemm_pmc_by_dispatch.csv | grep gemm | head -80
cat $OUT/errors.log 2>/dev/null
