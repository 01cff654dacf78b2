# Ring geometry (slots x rows per slot) A/B on the epochs: variants built with -DGGAD_RING_S / -DGGAD_RING_RS into gpurun_variants/ 
for so in default $(ls gpurun_variants/libggad_S*.so 2>/dev/null) default; do
  echo "== $so"
  if [ $so = default ]; then unset GGAD_LIB_PATH; else export GGAD_LIB_PATH=$PWD/$so; fi
  timeout 600 python scripts/fullgraph_leg.py Amazon t_finance 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    k, _, v = l.partition(' ')
    try: d = json.loads(v)
    except Exception: continue
    print('  ', k, round(d['epoch_ms'], 4), 'full product', round(d['spmm_NxNxH']['us'], 1), 'fill', round(d['spmm_NxNxH'].get('schedule_fill', 0), 3))
"
done
