#!/bin/bash
# Timing ablations of the f16mx kernels (DBG instantiations; results are garbage, only the durations mean anything):
# 1 no in-loop DMA, 2 no in-loop fragment reads, 4 no wait + barrier, 8 no epilogue.  Second argument of bench_kernels.py:
# batch (4 = 32 workgroups: an eighth of the chip, no power limit).
# The DBG instantiations live only in the debug library: python3 acoustic_locating_vq-vae_amd/build.py --debug-kernels
export ALVQ_LIB=${ALVQ_LIB:-$PWD/acoustic_locating_vq-vae_amd/lib/libalvq_dbg.so}
[ -f "$ALVQ_LIB" ] || { echo "build the debug library first: python3 acoustic_locating_vq-vae_amd/build.py --debug-kernels"; exit 1; }
for B in ${BATCHES:-64}; do
for d in ${DBGS:-0 1 2 4 8 3 7 15}; do
  echo "== ALVQ_FX_DBG=$d B=$B"
  ALVQ_FX_DBG=$d python3 tools/bench_kernels.py f16mx $B
done
done
