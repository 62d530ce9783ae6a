set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4))'
B="--scene procedural --steps 40 --warmup 5 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
python bench.py $B 2>/dev/null | python -c "$P" "defaults"
PT_DYN_BLOCKS_PER_CU=4 python bench.py $B 2>/dev/null | python -c "$P" "dyn 4"
PT_DYN_BLOCKS_PER_CU=4 PT_TAIL_AFTER=4 python bench.py $B 2>/dev/null | python -c "$P" "dyn 4 tail_after 4"
PT_DYN_BLOCKS_PER_CU=4 PT_TAIL_AFTER=5 python bench.py $B 2>/dev/null | python -c "$P" "dyn 4 tail_after 5"
PT_DYN_BLOCKS_PER_CU=4 PT_TAIL_AFTER=4 PT_DESCENT=6 python bench.py $B 2>/dev/null | python -c "$P" "dyn 4 tail_after 4 descent 6"
PT_DYN_BLOCKS_PER_CU=4 PT_TAIL_AFTER=6 python bench.py $B 2>/dev/null | python -c "$P" "dyn 4 tail_after 6"
done
