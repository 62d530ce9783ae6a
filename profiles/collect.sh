#!/bin/bash
# profiles/collect.sh <tag> <workload> [bench.py arguments] -- the rocprofv3 runs behind profiles/<tag>_*: run on the GPU box from the
# repo root, e.g.
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh r02_c2 demo-1920x1080-1spp-8b'
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh r02_c5 procedural-1920x1080-1spp-8b --scene procedural'
# then copy the summaries from gpurun_out/<tag>/ into profiles/ (profiles/README.md).  Counters are collected in their own passes
# with --kernel-trace only (no --stats / other trace domains in the same run), FETCH_SIZE and WRITE_SIZE apart, one frame in flight.
set -e -o pipefail
TAG=$1; WORKLOAD=$2; shift 2
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
LONG="--steps ${PT_PROFILE_STEPS:-300} --warmup ${PT_PROFILE_WARMUP:-30}"
SHORT="--steps ${PT_PROFILE_PMC_STEPS:-20} --warmup 2 --prewarm 0 --no-cpu-baseline --no-roofline --frames-in-flight 1"
python3 bench.py $LONG "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_20_steps.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
# Under the profiler frames run one at a time, and a 1-spp frame submitted to an idle context would take the one-launch latency form:
# pin the throughput run's own choice (a separate looping pass from 400 k slots) for the traced and counted passes.
if [ "${PT_PROFILE_PIN_SCHEDULE:-1}" = 1 ]; then export PT_FUSE_LOOP=${PT_FUSE_LOOP:-0} PT_PROFILE_PINNED=PT_FUSE_LOOP; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py $LONG --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "[collect] kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o k -- python3 $R/bench.py $SHORT "$@" > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o k -- python3 $R/bench.py $SHORT "$@" > $OUT/write.json 2> $OUT/write.err
echo "[collect] HBM counters done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -o k -- python3 $R/bench.py $SHORT "$@" > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq2 -o k -- python3 $R/bench.py $SHORT "$@" > $OUT/sq2.json 2> $OUT/sq2.err
echo "[collect] SQ counters done"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcp -o k -- python3 $R/bench.py $SHORT "$@" > $OUT/tcp.json 2> $OUT/tcp.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc -o k -- python3 $R/bench.py $SHORT "$@" > $OUT/tcc.json 2> $OUT/tcc.err
echo "[collect] cache counters done"
cd $R
python3 profiles/make_counters.py $OUT $WORKLOAD $(( ${PT_PROFILE_PMC_STEPS:-20} + 2 )) > $OUT/counters.log 2>&1
# with the counters of THIS code in place, the bench line carries traffic / valu figures that match its kernels
cp $OUT/counters.json profiles/counters_$WORKLOAD.json
python3 bench.py $LONG "$@" > $OUT/bench_with_counters.json 2>> $OUT/bench.err
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
ls $OUT
