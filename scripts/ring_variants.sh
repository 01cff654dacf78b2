#!/bin/bash
# Timing-only variants of k_spmm_ring (wrong results): which part of the kernel the time goes to.  Builds here (hipcc, no GPU needed):
#   bash scripts/ring_variants.sh build      -> gpurun_variants/libggad_<variant>.so
# and on the GPU box:  bash scripts/ring_variants.sh run [t_finance]
cd "$(dirname "$0")/.."
V=${RING_VARIANTS:-"NO_BARRIER NO_ADD NO_IDX NO_BARRIER_NO_ADD"}
if [ "$1" = build ]; then
  mkdir -p gpurun_variants
  python -m ggad_amd.build > /dev/null
  for v in $V; do
    F=""; for f in $(echo $v | sed 's/_NO_/ NO_/g'); do F="$F -DRING_$f"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -fno-fast-math $F -x hip -c ggad_amd/csrc/fullgraph.hip -o /tmp/fullgraph_$v.o || exit 1
    OBJS=$(ls ggad_amd/build/*.o | grep -v fullgraph.o | grep -v hop2_tiled)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/libggad_$v.so $OBJS /tmp/fullgraph_$v.o || exit 1
    echo built $v: $F
  done
else
  DS=${2:-t_finance}
  python scripts/ring_time.py $DS
  for v in $V; do GGAD_LIB_PATH=$PWD/gpurun_variants/libggad_$v.so python scripts/ring_time.py $DS $v; done
fi
