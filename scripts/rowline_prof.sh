#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace) of the sparse-neighbourhood product variants on Reddit and Photo (eager launches).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for DS in reddit photo; do
  rm -rf /tmp/rlp_$DS
  GGAD_TIME_GRAPH=0 rocprofv3 --kernel-trace --stats -d /tmp/rlp_$DS -o p -- python $R/scripts/spmm_sparse_variants.py $DS > /tmp/rlp_$DS.log 2>&1
  python $R/scripts/rocpd_stats.py $(find /tmp/rlp_$DS -name "*.db" | head -1) $R/gpurun_out/${ROUND:-r04}_spmm_${DS}_kernel_stats.csv > /dev/null 2>&1
  echo "== $DS"; grep "spmm\|Name" $R/gpurun_out/${ROUND:-r04}_spmm_${DS}_kernel_stats.csv | head
done
