# the looping pass on a second stream per lane (PT_SPLIT_STREAMS, 6 lanes on 3 + 3 streams) against the default 3 lanes
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do
  python bench.py --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" "3_lanes"
  python bench.py --steps 300 --warmup 30 $B --frames-in-flight 6 2>/dev/null | python -c "$P" "6_lanes_plain"
  PT_SPLIT_STREAMS=1 python bench.py --steps 300 --warmup 30 $B --frames-in-flight 6 2>/dev/null | python -c "$P" "6_lanes_split_default_prio"
  PT_SPLIT_STREAMS=2 python bench.py --steps 300 --warmup 30 $B --frames-in-flight 6 2>/dev/null | python -c "$P" "6_lanes_split_high_prio"
  PT_SPLIT_STREAMS=1 python bench.py --steps 20 --warmup 5 $B --frames-in-flight 6 2>/dev/null | python -c "$P" "6_lanes_split_20steps"
  python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "$P" "3_lanes_20steps"
done
