set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4))'
for n in 5000 20000 100000 1048576 4000000; do
B="--scene procedural --spheres $n --steps 40 --warmup 5 --no-cpu-baseline --no-roofline"
python bench.py $B 2>/dev/null | python -c "$P" "$n spheres: defaults"
PT_DYN_BLOCKS_PER_CU=4 PT_TAIL_AFTER=4 PT_DESCENT=6 python bench.py $B 2>/dev/null | python -c "$P" "$n spheres: dyn 4 tail_after 4 descent 6"
PT_DYN_BLOCKS_PER_CU=4 PT_TAIL_AFTER=4 python bench.py $B 2>/dev/null | python -c "$P" "$n spheres: dyn 4 tail_after 4"
PT_TAIL_AFTER=4 PT_DESCENT=6 python bench.py $B 2>/dev/null | python -c "$P" "$n spheres: tail_after 4 descent 6"
done
B="--scene procedural --width 3840 --height 2160 --spp 4 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline"
python bench.py $B 2>/dev/null | python -c "$P" "2^20 spheres 4K 4spp: defaults"
PT_DYN_BLOCKS_PER_CU=4 PT_TAIL_AFTER=4 PT_DESCENT=6 python bench.py $B 2>/dev/null | python -c "$P" "2^20 spheres 4K 4spp: dyn 4 tail_after 4 descent 6"
