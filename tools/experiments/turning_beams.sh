# primary-beam lists for a turning camera (bench.py --turning-camera, with and without --moving-camera): none (PT_BEAM_MAX_MARGIN=0: the pixel margin that
# lets lists survive a turn is off) against margins of up to 2 .. 8 pixels
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]), d["config"].get("primary_beams"))'
B="--no-cpu-baseline --no-roofline --steps 300 --warmup 30"
for rep in 1 2; do
  for m in 0 2 4 6 8; do
    PT_BEAM_MAX_MARGIN=$m python bench.py $B --turning-camera 2>/dev/null | python -c "$P" "turning_margin_$m"
  done
  PT_BEAM_MAX_MARGIN=0 python bench.py $B --turning-camera --moving-camera 2>/dev/null | python -c "$P" "turning_moving_margin_0"
  python bench.py $B --turning-camera --moving-camera 2>/dev/null | python -c "$P" "turning_moving"
  python bench.py $B --moving-camera 2>/dev/null | python -c "$P" "moving"
  python bench.py $B 2>/dev/null | python -c "$P" "resting"
done
