# phase clocks of the resident step kernel as seen by wave W of workgroup R:  GGAD_XCD_DEBUG = 4 + 16 W + 256 R
for r in ${RANKS:-0 15 31}; do for w in ${WAVES:-0 3 6 7}; do echo "rank $r wave $w"; GGAD_XCD_DEBUG=$((4+16*w+256*r)) python scripts/xcd_step_time.py 150 2>&1 | grep -E "xcd " | sed 's/.*by phase://'; done; done
