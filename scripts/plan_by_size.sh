#!/bin/bash
# Per-kernel durations of the plan alone on the chip at 20 and 150 batches per chunk (rocprofv3 kernel trace).  Usage: bash scripts/plan_by_size.sh [tag]
R=$GRAFT_REPO_ROOT
TAG=${1:-plan}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q0
rocprofv3 --kernel-trace -d /tmp/q0 -o t -- python $R/scripts/plan_kernel_times.py ${SIZES:-20,150} 2 > /tmp/q0.log 2>&1
python $R/scripts/rocpd_by_size.py $(find /tmp/q0 -name "*.db" | head -1) | tee $R/gpurun_out/${TAG}_by_size.csv
