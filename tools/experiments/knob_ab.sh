# A/B of one PT_* knob (0 against 1) on one box:  bash tools/experiments/knob_ab.sh PT_LDS_WIDE [bench.py arguments]
K=$1; shift
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]), round(d.get("latency_ms_one_frame") or 0, 4))'
for rep in 1 2 3; do for v in 0 1; do
  env $K=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "$P" "$K=$v"
done; done
