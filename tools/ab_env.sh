#!/bin/bash
# A/B of an environment switch on ONE box: tools/ab_env.sh VAR a b [bench flags]   (three interleaved repetitions)
V=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do
  for x in $A $B; do
    env $V=$x python3 bench.py --no-secondary --no-parity --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V=$x', d['config']['workload'].split()[0], d['dtype'], round(d['value'],1), 'spectrograms/s', round(d['ms_per_step'],4), 'ms/step')"
  done
done
