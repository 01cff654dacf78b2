#!/bin/bash
# Grid geometry of k_spmm_rowline (workgroups per XCD) on the N x N x 300 product of Reddit / Photo.
# Usage (GPU box): bash scripts/rowline_sweep.sh
cd ${GRAFT_REPO_ROOT:-.}
for BPX in 96 128 160 192 256; do
  echo "== workgroups/XCD $BPX"
  GGAD_ROWLINE_BPX=$BPX timeout 300 python scripts/spmm_sparse_variants.py reddit photo 2>&1 | grep "rowline"
done
