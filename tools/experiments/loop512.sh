set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
for t in 256 512; do
  for b in 2 8; do PT_LOOP_THREADS=$t PT_TAIL_BLOCKS_PER_CU=$b python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "C2_t${t}_b$b"; done
  PT_LOOP_THREADS=$t python bench.py --scene small --width 256 --height 256 --bounces 4 --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "C1_t$t"
  PT_LOOP_THREADS=$t python bench.py --width 640 --height 384 --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "640x384_t$t"
  PT_LOOP_THREADS=$t python bench.py --width 960 --height 540 --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "960x540_t$t"
  PT_LOOP_THREADS=$t python bench.py --width 3840 --height 2160 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "4K1spp_t$t"
  PT_LOOP_THREADS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --animate 2>/dev/null | python -c "$P" "animate_t$t"
  PT_LOOP_THREADS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --di 2>/dev/null | python -c "$P" "di_t$t"
  PT_LOOP_THREADS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --force-tiles 2>/dev/null | python -c "$P" "tiles_t$t"
done
