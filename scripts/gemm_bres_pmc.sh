#!/bin/bash
# Counters of k_gemm_bres on the tall products (scripts/gemm_bres_ab.py bres3): one counter group per run, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${ROUND:-r04}_gemm_bres_pmc.csv
echo "kernel,counter,dispatches,sum,avg_per_dispatch" > $OUT
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY"; do
  i=$((i+1)); rm -rf /tmp/gb$i
  rocprofv3 --kernel-trace --pmc $G -d /tmp/gb$i -o p -- python $R/scripts/gemm_bres_ab.py bres3 > /tmp/gb$i.log 2>&1
  DB=$(find /tmp/gb$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_pmc.py $DB | grep "gemm_bres" >> $OUT; else echo "pass $i failed: $G" >> $OUT; fi
done
cat $OUT
