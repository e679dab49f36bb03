#!/bin/bash
# Idle time between consecutive kernels of the replayed step (GPU box, repo root): rocprofv3 kernel trace of a headline-only
# bench run, then per step: sum of kernel durations, sum of the gaps between a kernel's end and the next one's start.
set -e
R=$PWD
OUT=$R/gpurun_out/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o gaps -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-parity --no-kernel-timer "$@" > $OUT/bench.log 2>&1
cd $R
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region = the last 20 steps: find the step boundaries by the first kernel of a step (standardise)
starts = [i for i, r in enumerate(rows) if "standardise" in r["Kernel_Name"]]
steps = []
for a, b in zip(starts[-21:-1], starts[-20:]):
    seg = rows[a:b]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    span = int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
    gaps.append(int(rows[b]["Start_Timestamp"]) - int(seg[-1]["End_Timestamp"]))
    steps.append((len(seg), busy / 1e6, span / 1e6, sum(g for g in gaps if g > 0) / 1e6, max(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
import statistics as st
print("kernels per step %d; per step (median of %d): kernel time %.3f ms, step span %.3f ms, idle between kernels %.3f ms (%.1f %%), largest gap %.1f us, median gap %.2f us"
      % (steps[0][0], len(steps), st.median(s[1] for s in steps), st.median(s[2] for s in steps), st.median(s[3] for s in steps),
         100 * st.median(s[3] for s in steps) / st.median(s[2] for s in steps), st.median(s[4] for s in steps), st.median(s[5] for s in steps)))
# where the big gaps are: after which kernel
seg = rows[starts[-2]:starts[-1]]
big = sorted(((int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3, seg[i]["Kernel_Name"][:60], seg[i + 1]["Kernel_Name"][:60]) for i in range(len(seg) - 1))[-8:]
for g, a, b in reversed(big):
    print("  %.1f us between %s -> %s" % (g, a, b))
PY
rm -f $OUT/*kernel_trace.csv
