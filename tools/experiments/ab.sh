# A/B of builds of libpt_hip.so on one box: every tools/experiments/libpt_*.so (variants built by hand) and the in-tree library.
# usage: bash tools/experiments/ab.sh [quick]
set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do
for lib in tools/experiments/libpt_*.so directx-raytracing-spheres-demo_amd/libpt_hip.so; do
  export PT_HIP_LIB=$PWD/$lib
  echo "== $lib"
  python bench.py --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" C2_300
  python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "$P" C2_20
  python bench.py --width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 $B 2>/dev/null | python -c "$P" C3
  python bench.py --width 3840 --height 2160 --spp 64 --bounces 16 --steps 6 --warmup 2 $B 2>/dev/null | python -c "$P" C4
  python bench.py --width 1280 --height 720 --spp 4 --steps 100 --warmup 10 $B 2>/dev/null | python -c "$P" 720p4spp
  if [ "$1" != quick ]; then
  python bench.py --animate --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" C2_animated
  python bench.py --width 960 --height 540 --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" 960x540
  fi
done
done
