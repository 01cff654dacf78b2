#!/bin/bash
# Kernel list of one traced eager training epoch of a full-graph config: scripts/epoch_kernels.sh <dataset> <tag>  -> gpurun_out/<tag>_<dataset>_last_epoch.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
ds=${1:-reddit}; T=${2:-r05}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe_$ds
(cd $R && rocprofv3 --kernel-trace -d /tmp/pe_$ds -o e -- python run.py --dataset $ds --synthetic --num_epoch 12 --no_graph > /tmp/pe_$ds.log 2>&1)
DB=$(find /tmp/pe_$ds -name "*.db" | head -1)
python $R/scripts/rocpd_last_epoch.py $DB 4 > $R/gpurun_out/${T}_${ds}_last_epoch.txt 2>&1
cat $R/gpurun_out/${T}_${ds}_last_epoch.txt
