#!/bin/bash
# A/B of one environment switch on the four full-graph configs (median hipGraph epoch): scripts/fullgraph_env_ab.sh VAR off_value on_value
for s in $2 $3 $2 $3; do
  echo "$1=$s"
  env $1=$s timeout 600 python scripts/fullgraph_leg.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    k, _, v = l.partition(' ')
    try: d = json.loads(v)
    except Exception: continue
    print('  ', k, round(d['epoch_ms'], 4), 'loss', d.get('loss_after'))
"
done
