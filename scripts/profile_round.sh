#!/bin/bash
# Round profiles (GPU box): rocprofv3 kernel stats of the driver's bench line and of a steady-state run, PMC passes (memory-side
# bytes, L2 hits) of the plan kernels at the chunk sizes those runs use.  Summaries land in gpurun_out/ (copy to profiles/).
R=$GRAFT_REPO_ROOT
T=${1:-r02}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pa -o a -- python $R/bench.py --steps 20 --warmup 5 --no-extras > $R/gpurun_out/${T}_bench20_profiled.log 2>&1
python $R/scripts/rocpd_stats.py $(find /tmp/pa -name "*.db" | head -1) $R/gpurun_out/${T}_bench20_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace -d /tmp/pb -o b -- python $R/bench.py --steps 1500 --warmup 150 --no-extras > $R/gpurun_out/${T}_bench1500_profiled.log 2>&1
python $R/scripts/rocpd_stats.py $(find /tmp/pb -name "*.db" | head -1) $R/gpurun_out/${T}_bench1500_kernel_stats.csv > /dev/null
python $R/scripts/rocpd_streams.py $(find /tmp/pb -name "*.db" | head -1) > $R/gpurun_out/${T}_bench1500_streams.txt 2>&1
rocprofv3 --kernel-trace -d /tmp/pc -o c -- python $R/scripts/plan_kernel_times.py 20,150 3 > $R/gpurun_out/${T}_plan_sizes.log 2>&1
python $R/scripts/rocpd_by_size.py $(find /tmp/pc -name "*.db" | head -1) > $R/gpurun_out/${T}_plan_kernels_by_size.csv
i=0
: > $R/gpurun_out/${T}_plan_pmc.csv
for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  rocprofv3 --kernel-trace --pmc $G -d /tmp/pm$i -o p -- python $R/scripts/plan_kernel_times.py 20,150 1 > /tmp/pm$i.log 2>&1
  DB=$(find /tmp/pm$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_pmc_by_dispatch.py $DB >> $R/gpurun_out/${T}_plan_pmc.csv; else echo "pass $i failed" >> $R/gpurun_out/${T}_plan_pmc.csv; fi
done
cd $R && python scripts/fullgraph_leg.py > gpurun_out/${T}_fullgraph_leg.log 2>&1
