#!/bin/bash
# Kernel timeline of the K = 20 headline window (bench.py --steps 20 --warmup 5 --no-extras) from a rocprofv3 kernel trace.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/k20t
rocprofv3 --kernel-trace --stats -d /tmp/k20t -o p -- python $R/bench.py --steps 20 --warmup 5 --no-extras > /tmp/k20t.log 2>&1
tail -1 /tmp/k20t.log | cut -c1-300
python $R/scripts/rocpd_timeline.py $(find /tmp/k20t -name "*.db" | head -1) > $R/gpurun_out/${ROUND:-r04}_k20_timeline.txt
cat $R/gpurun_out/${ROUND:-r04}_k20_timeline.txt
