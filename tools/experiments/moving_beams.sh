# primary-beam lists for a moving camera: off (PT_BEAM_REACH=0) against lists that reach 8 .. 48 frames of travel (PT_BEAM_REACH), built behind the frame on its lane
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]), d["config"].get("primary_beams"))'
B="--no-cpu-baseline --no-roofline --steps 300 --warmup 30 --moving-camera"
for rep in 1 2; do
  for r in 0 6 8 12 16 24 48; do
    PT_BEAM_REACH=$r python bench.py $B 2>/dev/null | python -c "$P" "moving_reach_$r"
  done
  python bench.py --no-cpu-baseline --no-roofline --steps 300 --warmup 30 2>/dev/null | python -c "$P" "resting"
  PT_BEAM_REACH=0 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --moving-camera 2>/dev/null | python -c "$P" "moving_20_steps_off"
  python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --moving-camera 2>/dev/null | python -c "$P" "moving_20_steps"
done
