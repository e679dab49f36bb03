#!/bin/bash
# A/B of two builds of libalvq.so on ONE box (boxes differ by several per cent): tools/ab.sh <a.so> <b.so> [bench flags]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    ALVQ_LIB=$PWD/$L python3 bench.py --no-secondary --no-parity --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', d['config']['workload'].split()[0], d['dtype'], round(d['value'],1), 'spectrograms/s', round(d['ms_per_step'],4), 'ms/step', 'loss', d['final_loss'])"
  done
done
