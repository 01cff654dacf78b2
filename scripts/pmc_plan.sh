#!/bin/bash
# PMC passes over the plan kernels alone on the chip (each counter group in its own run, --kernel-trace only).
# Usage (GPU box): bash scripts/pmc_plan.sh <sizes> <tag>     -> gpurun_out/<tag>_pmc.csv
SIZES=${1:-150}
TAG=${2:-r02_plan}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_pmc.csv
: > $OUT
i=0
for G in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $G -d /tmp/pmc$i -o p -- python $R/scripts/plan_kernel_times.py $SIZES 2 > /tmp/pmc$i.log 2>&1
  DB=$(find /tmp/pmc$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_pmc.py $DB | grep -v "^kernel,counter" >> $OUT; else echo "pass $i failed: $G" >> $OUT; tail -3 /tmp/pmc$i.log >> $OUT; fi
done
