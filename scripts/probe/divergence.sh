#!/bin/bash
# gpurun -- bash scripts/probe/divergence.sh   (PMC pass + kernel trace of scripts/probe/divergence.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/divergence
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ROOT/scripts/probe/divergence.py > $OUT/run.txt 2> $OUT/run.err
cat $OUT/run.txt
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Kernel_Name"] or "k_legal" in r["Kernel_Name"]:
            rows.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for (d, k), c in sorted(rows.items())[-12:]:
        w = max(c.get("SQ_WAVES", 1), 1)
        print(d, k, "waves %d VALU/wave %.0f SALU/wave %.0f LDS/wave %.0f wave-cycles/wave %.0f" % (w, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w, c.get("SQ_WAVE_CYCLES", 0) / w))
PY
