#!/bin/bash
# Ring geometry (slots x rows per slot) of k_spmm_ring: builds in gpurun_variants/ (scripts: see the history of this file), timing here.
cd "$(dirname "$0")/.."
for ds in Amazon t_finance; do
  python scripts/ring_time.py $ds "S5xRS248(default)"
  for so in gpurun_variants/libggad_S*_RS*.so; do GGAD_LIB_PATH=$PWD/$so python scripts/ring_time.py $ds $(basename $so .so); done
done
