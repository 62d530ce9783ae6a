#!/bin/bash
# first GPU call of round 2: VALU issue-rate microbenchmark + baseline bench lines (driver's short run, long run, split schedule)
set -e -o pipefail
O=gpurun_out/r02_a; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/valu_rate tools/experiments/valu_rate.hip
/tmp/valu_rate > $O/valu_rate.txt 2>&1
cat $O/valu_rate.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b_20.json 2> $O/b_20.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b_20b.json 2>> $O/b_20.err
python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/b_300.json 2> $O/b_300.err
PT_SPLIT=1 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/b_split.json 2> $O/b_split.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_a/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'], (d.get('roofline') or {}).get('avg_launch_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
