# Lane streams in their own hardware-queue pool (highest priority) vs the default pool: texdbg.py (same frame after different
# numbers of earlier streams) and the frames-in-flight sweep.  usage: bash tools/experiments/lanes_prio.sh
set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
B="--no-cpu-baseline --no-roofline"
for prio in 1 0; do
  export PT_LANE_PRIORITY=$prio
  echo "== PT_LANE_PRIORITY=$prio"
  python tools/experiments/texdbg.py 2>&1 | grep -v amdgpu.ids
  for lanes in 2 3 4 5 6; do
    python bench.py --frames-in-flight $lanes --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" "C2, $lanes lanes, 300 steps"
    python bench.py --frames-in-flight $lanes --steps 20 --warmup 5 $B 2>/dev/null | python -c "$P" "C2, $lanes lanes, 20 steps"
  done
done
