# C5 (2^20 spheres) against its knobs after the round's changes.  usage: bash tools/experiments/c5_knobs.sh
set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4))'
B="--scene procedural --steps 40 --warmup 5 --no-cpu-baseline --no-roofline"
python bench.py $B 2>/dev/null | python -c "$P" "defaults"
for d in 3 4 6 12 16; do PT_DESCENT=$d python bench.py $B 2>/dev/null | python -c "$P" "PT_DESCENT=$d"; done
for b in 3 4 5 8; do PT_DYN_BLOCKS_PER_CU=$b python bench.py $B 2>/dev/null | python -c "$P" "PT_DYN_BLOCKS_PER_CU=$b"; done
for t in 0 1 2 4; do PT_TAIL_AFTER=$t python bench.py $B 2>/dev/null | python -c "$P" "PT_TAIL_AFTER=$t"; done
for l in 1 2 4; do python bench.py $B --frames-in-flight $l 2>/dev/null | python -c "$P" "lanes $l"; done
for b in 4 16; do PT_TRAVERSE_BLOCKS_PER_CU=$b python bench.py $B 2>/dev/null | python -c "$P" "PT_TRAVERSE_BLOCKS_PER_CU=$b"; done
python bench.py $B 2>/dev/null | python -c "$P" "defaults"
