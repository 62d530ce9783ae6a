# the driver's run is 20 timed frames after 5 warm-up frames (~2 ms of GPU work in all): does it see the steady state?
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
B="--no-cpu-baseline --no-roofline"
for rep in 1 2 3 4; do
  python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "$P" "w5_20"
  python bench.py --steps 20 --warmup 300 $B 2>/dev/null | python -c "$P" "w300_20"
  python bench.py --steps 20 --warmup 3000 $B 2>/dev/null | python -c "$P" "w3000_20"
  python bench.py --steps 40 --warmup 5 $B 2>/dev/null | python -c "$P" "w5_40"
  python bench.py --steps 300 --warmup 30 $B 2>/dev/null | python -c "$P" "w30_300"
done
