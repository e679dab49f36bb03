#!/bin/bash
# Timing ablations of the f16mx kernels (DBG instantiations; results are garbage, only the durations mean anything):
# 1 no in-loop DMA, 2 no in-loop fragment reads, 4 no wait + barrier, 8 no epilogue.  Second argument of bench_kernels.py:
# batch (4 = 32 workgroups: an eighth of the chip, no power limit).
for B in ${BATCHES:-64}; do
for d in ${DBGS:-0 1 2 4 8 3 7 15}; do
  echo "== ALVQ_FX_DBG=$d B=$B"
  ALVQ_FX_DBG=$d python3 tools/bench_kernels.py f16mx $B
done
done
