#!/bin/bash
# PMC passes over the N x N x 300 product kernels (each counter group in its own run, --kernel-trace only).
# Usage (GPU box): bash scripts/pmc_spmm.sh <dataset> <tag>     -> gpurun_out/<tag>_pmc.csv
DS=${1:-t_finance}
TAG=${2:-r02_spmm}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_pmc.csv
: > $OUT
i=0
for G in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $G -d /tmp/pmc$i -o p -- python $R/scripts/spmm_time.py $DS > /tmp/pmc$i.log 2>&1
  DB=$(find /tmp/pmc$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_pmc.py $DB | grep -v "^kernel,counter" | grep "spmm" >> $OUT; else echo "pass $i failed: $G" >> $OUT; tail -3 /tmp/pmc$i.log >> $OUT; fi
done
cat $OUT
