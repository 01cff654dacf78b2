#!/bin/bash
# Round 3: kernel stats / stream occupancy of a steady-state run with the XCD-resident chunk kernel, and a sweep of its CU count.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pb -o b -- python $R/bench.py --steps 1500 --warmup 150 --no-extras > $R/gpurun_out/r03_bench1500_profiled.log 2>&1
python $R/scripts/rocpd_stats.py $(find /tmp/pb -name "*.db" | head -1) $R/gpurun_out/r03_bench1500_kernel_stats.csv > /dev/null
python $R/scripts/rocpd_streams.py $(find /tmp/pb -name "*.db" | head -1) > $R/gpurun_out/r03_bench1500_streams.txt 2>&1
cd $R
for c in 24 26 28 30 31; do
  echo "dense_cus $c" >> gpurun_out/r03_xcd_cus_sweep.log
  python bench.py --steps 3000 --warmup 150 --no-extras --dense-cus $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r03_xcd_cus_sweep.log
done
echo "round-2 split (64 CUs over all XCDs, launch chain)" >> gpurun_out/r03_xcd_cus_sweep.log
GGAD_XCD=0 python bench.py --steps 3000 --warmup 150 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r03_xcd_cus_sweep.log
