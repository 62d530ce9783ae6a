# frames in flight against the steady state and the driver's 20-step run (round 3 kernels)
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
B="--no-cpu-baseline --no-roofline"
for rep in 1 2; do for n in 2 3 4 5 6 8; do
  python bench.py --steps 300 --warmup 30 $B --frames-in-flight $n 2>/dev/null | python -c "$P" "lanes_${n}_300"
  python bench.py --steps 20 --warmup 5 $B --frames-in-flight $n 2>/dev/null | python -c "$P" "lanes_${n}_20"
done; done
