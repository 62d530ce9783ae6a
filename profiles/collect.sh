#!/bin/bash
# profiles/collect.sh <tag> -- the rocprofv3 runs behind profiles/<tag>_*: run on the GPU box from the repo root, e.g.
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r01_d'
# then copy the summaries from gpurun_out/<tag>/ into profiles/ (see profiles/README.md).  Counters are collected in their
# own passes with --kernel-trace only (no --stats / other trace domains in the same run), FETCH_SIZE and WRITE_SIZE apart.
set -e -o pipefail
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
STEPS="--steps 300 --warmup 30"
python3 bench.py $STEPS > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py $STEPS --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o k -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --frames-in-flight 1 > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o k -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --frames-in-flight 1 > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -o k -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --frames-in-flight 1 > $OUT/sq.json 2> $OUT/sq.err
cd $R
ls $OUT $OUT/stats | head -40
