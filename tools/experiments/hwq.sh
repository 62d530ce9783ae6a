set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  for f in 3 4 5 6; do python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "hwq${q}_1080_lanes$f"; done
  for f in 3 4 6; do python bench.py --width 640 --height 384 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "hwq${q}_640_lanes$f"; done
  for f in 3 4 6; do python bench.py --width 640 --height 384 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --force-tiles --frames-in-flight $f 2>/dev/null | python -c "$P" "hwq${q}_tiles640_lanes$f"; done
done
