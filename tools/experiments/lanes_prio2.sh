set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
B="--no-cpu-baseline --no-roofline"
for prio in 0 1; do
  export PT_LANE_PRIORITY=$prio
  for lanes in 2 3 4 6 8; do
    python bench.py --scene small --width 256 --height 256 --bounces 4 --frames-in-flight $lanes --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" "prio $prio C1 256x256, $lanes lanes"
    python bench.py --width 640 --height 384 --frames-in-flight $lanes --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" "prio $prio 640x384, $lanes lanes"
    python bench.py --width 960 --height 540 --frames-in-flight $lanes --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" "prio $prio 960x540, $lanes lanes"
    python bench.py --force-tiles --frames-in-flight $lanes --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" "prio $prio C2 tiled (1 rank rehearsal), $lanes lanes"
    python bench.py --width 3840 --height 2160 --spp 16 --frames-in-flight $lanes --steps 20 --warmup 3 $B 2>/dev/null | python -c "$P" "prio $prio C3, $lanes lanes"
  done
done
