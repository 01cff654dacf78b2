#!/bin/bash
# Hardware counters of k_gemm_slab on the T-Finance projection (scripts/gemm_slab_probe_base t = that shape alone): one rocprofv3 pass per
# counter group, --kernel-trace only.  Output: gpurun_out/slab_pmc/summary.csv (kernel,counter,dispatches,sum,avg_per_dispatch)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
BIN=${1:-$R/scripts/gemm_slab_probe_base}
OUT=$R/gpurun_out/slab_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p -- $BIN t > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed" >> $OUT/errors.log
done
: > $OUT/summary.csv
for d in $OUT/p*/; do
  db=$(ls $d/*.db $d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 $R/scripts/rocpd_pmc.py $db | grep -v "^kernel" >> $OUT/summary.csv
done
cat $OUT/summary.csv
cat $OUT/errors.log 2>/dev/null
rm -rf $OUT/p*/
