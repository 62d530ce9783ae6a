P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]), round(d.get("latency_ms_one_frame") or 0,4))'
for rep in 1 2; do for x in 0 16384 28672 49152; do
  PT_LOOP_EXTRA_LDS=$x python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "extra_lds_$x"
done; done
