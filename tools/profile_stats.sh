#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py configuration (GPU box, repo root): tools/profile_stats.sh <tag> [bench flags]
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity --no-kernel-timer "$@" > $OUT/bench.log 2>&1
cd $R
find $OUT -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/*kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms over the profiled steps" % (tot / 1e6))
for r in rows[:28]:
    print("%6.2f%%  calls %5s  avg %8.1f us  %s" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
