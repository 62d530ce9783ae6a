# A/B/n on one box: every library named on the command line, interleaved, twice.
# usage: bash tools/experiments/abn.sh tools/experiments/libpt_base.so tools/experiments/libpt_A.so ...
set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
B="--no-cpu-baseline --no-roofline"
for rep in 1 2 3; do
for lib in "$@"; do
  export PT_HIP_LIB=$PWD/$lib
  echo "== $lib"
  python bench.py --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" C2_300
  python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "$P" C2_20
  python bench.py --width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 $B 2>/dev/null | python -c "$P" C3
done
done
