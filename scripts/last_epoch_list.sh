#!/bin/bash
# kernel list of one traced eager epoch: scripts/last_epoch_list.sh <dataset> [tag]
R=$GRAFT_REPO_ROOT; ds=$1; T=${2:-r05}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pe_$ds
(cd $R && rocprofv3 --kernel-trace -d /tmp/pe_$ds -o e -- python run.py --dataset $ds --synthetic --num_epoch 12 --no_graph > $R/gpurun_out/${T}_${ds}_run.log 2>&1)
DB=$(find /tmp/pe_$ds -name "*.db" | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_last_epoch.py $DB 4 > $R/gpurun_out/${T}_${ds}_last_epoch.txt 2>&1
cat $R/gpurun_out/${T}_${ds}_last_epoch.txt
