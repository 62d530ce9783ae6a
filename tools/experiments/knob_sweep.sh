set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],5))'
for f in 2 3 4 5 6; do python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --frames-in-flight $f 2>/dev/null | python -c "$P" "lanes$f"; done
for b in 1 2 3 4; do PT_TRAVERSE_BLOCKS_PER_CU=$b python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "trav_blocks$b"; done
for b in 1 2 4; do PT_TAIL_BLOCKS_PER_CU=$b python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "tail_blocks$b"; done
for t in 0 1 2; do PT_TAIL_AFTER=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "tail_after$t"; done
