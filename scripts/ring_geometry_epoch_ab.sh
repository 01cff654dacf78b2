for so in default gpurun_variants/libggad_S2_RS624.so default gpurun_variants/libggad_S2_RS624.so; do
  echo "== $so"
  if [ $so = default ]; then unset GGAD_LIB_PATH; else export GGAD_LIB_PATH=$PWD/$so; fi
  timeout 600 python scripts/fullgraph_leg.py Amazon t_finance 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    k, _, v = l.partition(' ')
    try: d = json.loads(v)
    except Exception: continue
    print('  ', k, round(d['epoch_ms'], 4), 'full product', round(d['spmm_NxNxH']['us'], 1))
"
done
