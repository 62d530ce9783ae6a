# primary-beam lists for a moving camera, built in shares inside the primary passes: off (PT_BEAM_REACH=0) against lists that reach 16 .. 48 frames of travel
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]), d["config"].get("primary_beams"))'
B="--no-cpu-baseline --no-roofline --steps 300 --warmup 30 --moving-camera"
for rep in 1 2; do
  for r in 0 16 24 32 48; do
    PT_BEAM_REACH=$r python bench.py $B 2>/dev/null | python -c "$P" "moving_reach_$r"
  done
  python bench.py --no-cpu-baseline --no-roofline --steps 300 --warmup 30 2>/dev/null | python -c "$P" "resting"
done
