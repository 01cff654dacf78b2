#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace) of the sparse-neighbourhood product variants; GGAD_ROWLINE_DBG variants of k_spmm_rowline.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for D in ${@:-0}; do
  rm -rf /tmp/rlp$D
  GGAD_ROWLINE_DBG=$D GGAD_TIME_GRAPH=0 rocprofv3 --kernel-trace --stats -d /tmp/rlp$D -o p -- python $R/scripts/spmm_sparse_variants.py reddit photo > /tmp/rlp$D.log 2>&1
  echo "== DBG $D"
  python $R/scripts/rocpd_stats.py $(find /tmp/rlp$D -name "*.db" | head -1) /tmp/rlp$D.csv > /dev/null 2>&1
  grep "spmm\|Name" /tmp/rlp$D.csv | head
  tail -3 /tmp/rlp$D.log
done
