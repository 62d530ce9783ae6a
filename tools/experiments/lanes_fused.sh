set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
for f in 2 3 4 6 8; do python bench.py --scene small --width 256 --height 256 --bounces 4 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "C1_lanes$f"; done
for f in 2 3 4 6 8; do python bench.py --width 640 --height 384 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "640_lanes$f"; done
for f in 2 3 4 6 8; do python bench.py --width 640 --height 384 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --force-tiles --frames-in-flight $f 2>/dev/null | python -c "$P" "tiles640_lanes$f"; done
