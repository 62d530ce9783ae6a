# A/B on one box: tools/experiments/libpt_prev.so (a build of an earlier commit) against the in-tree library.
# usage: bash tools/experiments/ab2.sh [c2|c3|all]
set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
B="--no-cpu-baseline --no-roofline"
W=${1:-all}
for rep in 1 2; do
for lib in tools/experiments/libpt_prev.so directx-raytracing-spheres-demo_amd/libpt_hip.so; do
  export PT_HIP_LIB=$PWD/$lib
  echo "== $lib"
  if [ "$W" != c3 ]; then
  python bench.py --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" C2_300
  python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "$P" C2_20
  python bench.py --steps 300 --warmup 30 $B --moving-camera 2>/dev/null | python -c "$P" C2_moving || true
  fi
  if [ "$W" != c2 ]; then
  python bench.py --width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 $B 2>/dev/null | python -c "$P" C3
  python bench.py --width 3840 --height 2160 --spp 64 --bounces 16 --steps 6 --warmup 2 $B 2>/dev/null | python -c "$P" C4
  python bench.py --width 1280 --height 720 --spp 4 --steps 100 --warmup 10 $B 2>/dev/null | python -c "$P" 720p4spp
  fi
done
done
