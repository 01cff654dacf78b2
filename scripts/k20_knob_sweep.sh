#!/bin/bash
# K = 20 headline under the launch-geometry knobs of the gather and the resident chunk kernel (two runs each, same box).
cd ${GRAFT_REPO_ROOT:-.}
run() { for rep in 1 2; do env "$@" python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', 'value %.3f M' % (d['value']/1e6), 'gather %.3f ms' % d['roofline']['avg_launch_ms'])"; done; }
run X=0
for w in 3 5 6; do run GGAD_G2_WG_PER_CU=$w; done
for c in 8 16 24; do run GGAD_G2_TAKE_CAP=$c; done
for n in 24 32; do run GGAD_XCD_WGS=$n; done
run X=1
