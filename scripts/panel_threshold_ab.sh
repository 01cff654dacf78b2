for cfg in "1048576 64" "262144 32" "262144 16" "131072 32" "1048576 64" "262144 32"; do set -- $cfg; echo "MIN_NNZ=$1 MIN_ROW=$2"
GGAD_SPMM_PANEL_MIN_NNZ=$1 GGAD_SPMM_PANEL_MIN_ROW=$2 timeout 600 python scripts/fullgraph_leg.py Amazon t_finance 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    k, _, v = l.partition(' ')
    try: d = json.loads(v)
    except Exception: continue
    print('  ', k, round(d['epoch_ms'], 4))
"; done
