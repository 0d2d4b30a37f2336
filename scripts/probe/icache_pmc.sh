ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|SQC_" | head -40 > $OUT/counters.txt
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --no-cpu > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "k_play" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        print(k, n, v / n)
PY
tail -n 3 $OUT/err.txt; head -20 $OUT/counters.txt
