#!/bin/bash
# why is a 20-step run slower per step than a 300-step run?  (clock ramp vs pipeline fill/drain)
set -e -o pipefail
O=gpurun_out/r02_c; mkdir -p $O
export PT_BEAMS=0
for cfg in "20 5" "20 300" "20 2000" "60 5" "300 30" ; do
  set -- $cfg
  python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-roofline > $O/b_$1_$2.json 2>> $O/err.log
  python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-roofline --frames-in-flight 6 > $O/b6_$1_$2.json 2>> $O/err.log
done
PT_FUSE_LOOP=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bfuse_20_5.json 2>> $O/err.log
PT_FUSE_LOOP=1 python3 bench.py --steps 20 --warmup 300 --no-cpu-baseline --no-roofline > $O/bfuse_20_300.json 2>> $O/err.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_c/b*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],5))
PY
