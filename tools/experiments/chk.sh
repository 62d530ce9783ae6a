set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" C2
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" C2
python bench.py --scene small --width 256 --height 256 --bounces 4 --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" C1
python bench.py --width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" C3
python bench.py --width 3840 --height 2160 --spp 64 --bounces 16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" C4
python bench.py --width 1280 --height 720 --spp 4 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" 720p4spp
PT_TRAVERSE_BLOCKS_PER_CU=8 python bench.py --width 3840 --height 2160 --spp 64 --bounces 16 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" C4_trav8
