#!/bin/bash
# PMC passes over the N x N x 300 product on a SPARSE-neighbourhood config (Reddit / Photo): k_spmm_seg (+ k_spmm_combine) and the
# column-sliced k_spmm_rowslice.  Each counter group in its own run (--kernel-trace only).  Usage: bash scripts/pmc_spmm_sparse.sh [reddit]
DS=${1:-reddit}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${ROUND:-r04}_spmm_${DS}_pmc.csv
echo "kernel,counter,dispatches,sum,avg_per_dispatch" > $OUT
i=0
for G in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pms$i
  rocprofv3 --kernel-trace --pmc $G -d /tmp/pms$i -o p -- python $R/scripts/spmm_sparse_variants.py $DS > /tmp/pms$i.log 2>&1
  DB=$(find /tmp/pms$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_pmc.py $DB | grep -v "^kernel,counter" | grep "spmm_seg\|spmm_rowslice\|spmm_rowline\|spmm_combine" >> $OUT; else echo "pass $i failed: $G" >> $OUT; fi
done
cat $OUT
