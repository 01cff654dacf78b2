cd ${GRAFT_REPO_ROOT:-.}
run() { (cd $2 && python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'value %.3f M' % (d['value']/1e6), 'ms/step %.4f' % d['ms_per_step'])"); }
for rep in 1 2 3 4; do
  run r03 _r03
  run head .
done
