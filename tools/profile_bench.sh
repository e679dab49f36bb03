#!/bin/bash
# Collect the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root):
#   tools/profile_bench.sh <dtype> [extra bench.py flags]
# Three separate passes, as MI355X_MICROARCH.md prescribes: kernel trace + stats, then one --pmc pass per counter.
set -e
DT=${1:-x3mx_hb}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$DT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--dtype $DT --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r04 -- python3 $R/bench.py $FLAGS > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r04 -- python3 $R/bench.py $FLAGS --no-graph --no-kernel-timer > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r04 -- python3 $R/bench.py $FLAGS --no-graph --no-kernel-timer > $OUT/bench_write.log 2>&1
# matrix-pipe and LDS counters of the same run (their own pass; the guide's prescription: counters never share a pass
# with a trace domain)
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o r04 -- python3 $R/bench.py $FLAGS --no-graph --no-kernel-timer > $OUT/bench_sq.log 2>&1 || true
cd $R
grep '^{' $OUT/bench_stats.log | tail -1 > $OUT/bench_under_rocprof.json
# keep only what summarize.py reads (the traces themselves are large)
find $OUT -name '*kernel_trace.csv' -delete
ls -la $OUT/*
