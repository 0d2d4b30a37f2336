#!/bin/bash
# Extra PMC passes for k_decide (run on the GPU box via gpurun from the repo root): $1 = tag, rest = counter sets
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
ARGS="--steps 20 --warmup 20 --no-cpu"
i=0
shift
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python3 $ROOT/bench.py $ARGS > $OUT/bench_$i.json 2> $OUT/err_$i.txt || echo "set $i failed: $set"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/set*")):
    for f in glob.glob(d+"/**/*_counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            if "k_decide" in r["Kernel_Name"]:
                a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
        for k,(n,v) in acc.items(): print(k, n, v/n)
PY
