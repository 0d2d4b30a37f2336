#!/bin/bash
# Profile recipe (run on the GPU box via gpurun from the repo root): $1 = tag, $2.. = bench.py arguments.
# Kernel trace + stats and the PMC counters in SEPARATE passes (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots,
# WRITE_SIZE 2; gpurun refuses --pmc combined with trace domains other than kernel-trace).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
ARGS="${@:---rounds 8 --steps 12 --warmup 3} --no-cpu"
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
echo "$ARGS" > $OUT/bench_args.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err || true
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err || true
find $OUT -name "*.csv" | head -20
