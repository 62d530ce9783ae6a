# how much LDS the looping and the primary workgroups of C2 can grow before they stop fitting beside each other (temporary PT_*_EXTRA_LDS hooks in launch_bounce_for)
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
for x in 0 4096 8192 10240 12288; do
  PT_LOOP_EXTRA_LDS=$x python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "loop_extra_$x"
done
for x in 2048 4096 5120 6144 8192; do
  PT_PRIMARY_EXTRA_LDS=$x python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "primary_extra_$x"
done
