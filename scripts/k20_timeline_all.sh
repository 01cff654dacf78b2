#!/bin/bash
# Kernel timeline (all kernels, overlapping streams) of the K = 20 headline window from a rocprofv3 kernel trace: scripts/k20_timeline_all.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/k20t
rocprofv3 --kernel-trace --stats -d /tmp/k20t -o p -- python $R/bench.py --steps 20 --warmup 5 --no-extras > /tmp/k20t.log 2>&1
tail -1 /tmp/k20t.log | cut -c1-200
python $R/scripts/rocpd_timeline_all.py $(find /tmp/k20t -name "*.db" | head -1) -2 > $R/gpurun_out/${1:-r05}_k20_timeline.txt
cat $R/gpurun_out/${1:-r05}_k20_timeline.txt
