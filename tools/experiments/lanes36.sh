set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
for rep in 1 2; do
for f in 3 6; do
  python bench.py --scene small --width 256 --height 256 --bounces 4 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "C1_lanes$f"
  python bench.py --width 640 --height 384 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "640_lanes$f"
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "C2_lanes$f"
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --animate --frames-in-flight $f 2>/dev/null | python -c "$P" "C2anim_lanes$f"
  python bench.py --width 3840 --height 2160 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "4K_lanes$f"
  python bench.py --width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "C3_lanes$f"
done; done
