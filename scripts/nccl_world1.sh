#!/bin/bash
# The RCCL branch of bench.py and of FitnessEvaluator on the one-GPU box: a job of ONE rank started the way the driver
# starts N (torch.distributed.run), with the collectives forced on (MONSOON_BENCH_FORCE_DIST=1).  What it can show:
# init_process_group("nccl"), barrier, all_reduce SUM / MAX on device tensors and FitnessEvaluator's all-reduce of the
# counts run on this image.  What it cannot: more than one rank (8-GPU runs are the driver's).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export MONSOON_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for wl in c2 c3 c4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu --workload $wl 2> gpurun_out/nccl_world1_$wl.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', 'value %.1f M %s' % (d['value']/1e6, d['unit']), 'ranks', d['ranks'])"
done
