#!/bin/bash
# K = 20 headline (bench.py --steps 20 --warmup 5) under different chunk ramps: one chunk against overlapped sub-chunks.
cd ${GRAFT_REPO_ROOT:-.}
for R in ${@:-20 10,10 8,12 6,14 5,5,10 6,7,7 5,5,5,5}; do
  for rep in 1 2; do
    GGAD_RAMP=$R timeout 600 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ramp $R', 'value %.3f M' % (d['value']/1e6), 'ms/step %.4f' % d['ms_per_step'], d['config'].get('chunks_of_timed_region'), d['config'].get('overlap'))"
  done
done
