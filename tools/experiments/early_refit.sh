# animated C2 (bench.py --animate): the small-scene refit at the head of the frame's chain on the lane's stream (PT_EARLY_REFIT=0) against the refit queued at
# pt_update_spheres time into the lane's second scene copy, on a refit stream of the lanes' priority (PT_REFIT_PRIORITY=1) or of the default priority (0)
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
B="--no-cpu-baseline --no-roofline --steps 300 --warmup 30 --animate"
for rep in 1 2 3; do
  PT_EARLY_REFIT=0 python bench.py $B 2>/dev/null | python -c "$P" "refit_on_lane"
  PT_REFIT_PRIORITY=1 python bench.py $B 2>/dev/null | python -c "$P" "early_refit_high_priority"
  PT_REFIT_PRIORITY=0 python bench.py $B 2>/dev/null | python -c "$P" "early_refit_default_priority"
done
