#!/bin/bash
# Study (GPU box): what parts of a look-ahead step cost in instructions.  Counter runs of the product build and of study
# builds that execute every look-ahead step (or a prefix of it: -DMSB_STUDY_CUT_AT) one more time, scripts/ab_build.sh
# <name> "-DMSB_STUDY_REPEAT=1 ..."; differences per launch are printed by scripts/step_cost.py.
#   bash scripts/step_cost.sh base=monsoon_amd/libmonsoon_hip.so rep=monsoon_amd/libmonsoon_hip_rep.so ...
# Study builds: -DMSB_STUDY_REPEAT=1 [-DMSB_STUDY_CUT_AT=n] (look-ahead step), -DMSB_STUDY_FEATURES=1, -DMSB_STUDY_LEGAL=1.
# tag=library[@tree]: `tree` = another source tree with its own bench.py and the same study block in its kernels.h (round 3
# compared itself with round 2's recursive core this way: `git archive 18c1ffb bench.py monsoon_amd include | tar -x -C study_r2`).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/step_cost
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%=*}; rest=${spec#*=}; lib=${rest%%@*}; tree=$ROOT
  [ "$rest" != "$lib" ] && tree=$ROOT/${rest#*@}
  export MONSOON_LIB=$ROOT/$lib
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $OUT/$tag -- python3 $tree/bench.py --rounds 8 --steps 6 --warmup 2 --no-cpu > $OUT/$tag.json 2> $OUT/$tag.err || exit 1
  timeout -k 10 200 python3 $tree/bench.py --rounds 8 --steps 12 --warmup 3 --no-cpu > $OUT/$tag.time.json 2>> $OUT/$tag.err || exit 1
done
python3 $ROOT/scripts/step_cost.py $OUT
