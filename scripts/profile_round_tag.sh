#!/bin/bash
# Round profiles (tag = $1, default r05) (GPU box, one call): rocprofv3 kernel stats + timeline of the driver's bench command, the full-graph leg, one traced
# eager epoch at Reddit and T-Finance size (kernel lists), SQ counters of the LDS-ring product on T-Finance and Amazon.
# Summaries land in gpurun_out/ (copied to profiles/).   bash scripts/r04_profile_round.sh
R=$GRAFT_REPO_ROOT
T=${1:-r05}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa /tmp/pe1 /tmp/pe2
rocprofv3 --kernel-trace --stats -d /tmp/pa -o a -- python $R/bench.py --steps 20 --warmup 5 --no-extras > $R/gpurun_out/${T}_bench20_profiled.log 2>&1
DB=$(find /tmp/pa -name "*.db" | head -1)
python $R/scripts/rocpd_stats.py $DB $R/gpurun_out/${T}_bench20_kernel_stats.csv > /dev/null
python $R/scripts/rocpd_timeline.py $DB > $R/gpurun_out/${T}_k20_timeline.txt 2>&1
grep '^{"metric"' $R/gpurun_out/${T}_bench20_profiled.log > $R/gpurun_out/${T}_bench20_profiled_line.json; echo "bench under rocprofv3 exit: $?"
for ds in reddit t_finance; do
  rm -rf /tmp/pe_$ds
  (cd $R && rocprofv3 --kernel-trace -d /tmp/pe_$ds -o e -- python run.py --dataset $ds --synthetic --num_epoch 12 --no_graph > $R/gpurun_out/${T}_${ds}_run.log 2>&1)
  DB=$(find /tmp/pe_$ds -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/scripts/rocpd_last_epoch.py $DB 4 > $R/gpurun_out/${T}_${ds}_last_epoch.txt 2>&1
done
cd $R
python scripts/fullgraph_leg.py > gpurun_out/${T}_fullgraph_leg.log 2>&1
ls -la gpurun_out | grep ${T}_ | head -30
tail -3 gpurun_out/${T}_reddit_last_epoch.txt; tail -3 gpurun_out/${T}_t_finance_last_epoch.txt; cat gpurun_out/${T}_k20_timeline.txt | tail -25
