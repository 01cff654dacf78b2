#!/bin/bash
# End-to-end leg (sampler inside the window) with the sampler pipeline on the launching thread's L3 (same) or on another one (other).
for rep in 1 2; do for m in same other; do
GGAD_SAMPLER_LLC=$m timeout 300 python bench.py --steps 20 --warmup 5 --fullgraph-epochs 0 --sparse-entries 0 --cpu-batches 0 --steady-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())['e2e_with_sampler']; print('$m', round(d['value']/1e6,2), [round(v/1e6,2) for v in d['values']], round(d['sampler_us_per_batch'],1))"
done; done
lscpu | grep -i "L3\|socket\|^CPU(s)\|Thread\|Model name"; uptime
