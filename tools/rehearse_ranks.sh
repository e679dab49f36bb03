#!/bin/bash
# Rehearse the N>1 control flow of bench.py on ONE GPU: two ranks share the card and exchange gradients over gloo
# (RCCL refuses two ranks on one device).  Throughput is meaningless here; the point is that both bucket modes run,
# stay finite and agree.   tools/rehearse_ranks.sh [bucket modes, default "2 1"]
for b in ${@:-2 1}; do
  ALVQ_GRAD_BUCKETS=$b ALVQ_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --no-f32-line --no-kernel-timer \
    2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('buckets=$b', d['n_gpus'], 'ranks', round(d['value'],1), 'spec/s', d['final_loss'])"
done
