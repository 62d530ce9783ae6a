#!/bin/bash
# tools/experiments/instmix.sh -- VALU instructions per wave of the C2 kernels with and without the in-register second bounce
# (PT_INLINE2_MIN_SLOTS huge = one bounce per primary pass): separates ray generation + primary trace + first shade from the
# incoherent bounce trace + shade.   gpurun -- 'bash tools/experiments/instmix.sh'
set -e -o pipefail
R=$(pwd); OUT=$R/gpurun_out/instmix; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in inline2 single bounces0; do
  extra=""; [ $mode = bounces0 ] && extra="--bounces 0"
  [ $mode = single ] && export PT_INLINE2_MIN_SLOTS=4000000000 || unset PT_INLINE2_MIN_SLOTS
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/$mode -o k -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --frames-in-flight 1 $extra > $OUT/$mode.json 2> $OUT/$mode.err
done
cd $R
python3 - <<'PY'
import csv, collections, glob
for mode in ("inline2", "single", "bounces0"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for path in glob.glob(f"gpurun_out/instmix/{mode}/**/k_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if "bounce_kernel" not in n: continue
            key = n[n.index("bounce_kernel<"):].split(">")[0]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": cnt[key] += 1
    for k, v in agg.items():
        w = v["SQ_WAVES"]
        print(mode, k, "launches", cnt[k], "waves/launch %.0f" % (w / cnt[k]), " per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f" % (v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w, v["SQ_INSTS_VMEM_RD"] / w))
PY
