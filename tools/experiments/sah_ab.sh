set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["config"].get("lbvh"))'
for sw in 256 32 4; do PT_SAH_SWEEP=$sw timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "$P" "sweep$sw"; done
PT_SAH=0 timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "$P" lbvh
timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --animate 2>/dev/null | python -c "$P" animate_sah
timeout -k 10 200 python bench.py --scene small --width 256 --height 256 --bounces 4 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "$P" C1_sah
timeout -k 10 300 python bench.py --width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P" C3_sah
