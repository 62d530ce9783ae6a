set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
for f in 2 3 4 5 6; do python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "lanes$f"; done
for b in 1 2 3 4; do PT_TRAVERSE_BLOCKS_PER_CU=$b python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "trav_blocks$b"; done
for b in 2 4 8; do PT_TAIL_BLOCKS_PER_CU=$b python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "tail_blocks$b"; done
for t in 128 256 512; do PT_LOOP_THREADS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "loop_threads$t"; done
C3="--width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline"
for t in 256 512; do for b in 2 4; do PT_LOOP_THREADS=$t PT_TAIL_BLOCKS_PER_CU=$b python bench.py $C3 2>/dev/null | python -c "$P" "C3_loop_threads${t}_blocks$b"; done; done
for b in 2 4 8; do PT_TRAVERSE_BLOCKS_PER_CU=$b python bench.py $C3 2>/dev/null | python -c "$P" "C3_trav_blocks$b"; done
for f in 2 3 4; do python bench.py $C3 --frames-in-flight $f 2>/dev/null | python -c "$P" "C3_lanes$f"; done
