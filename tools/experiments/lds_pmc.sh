# LDS side of the BVH walk in LDS: bank-conflict cycles against all LDS-array cycles, per kernel (C2, one frame in flight)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lds_pmc
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/lds -o k -- python3 $R/bench.py --steps 20 --warmup 5 --prewarm 0 --frames-in-flight 1 --no-cpu-baseline --no-roofline "$@" > $OUT/lds.json 2> $OUT/lds.err
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/lds_pmc/lds/**/*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:110]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, v in acc.items():
    print(k, n[k], {c: round(x / max(n[k], 1)) for c, x in v.items()})
PY
