# frames in flight against the number of hardware queues HIP spreads its streams over (GPU_MAX_HW_QUEUES, default 4)
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5), round(d["value"]))'
B="--no-cpu-baseline --no-roofline"
for q in 4 8 2; do for n in 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 300 --warmup 30 $B --frames-in-flight $n 2>/dev/null | python -c "$P" "hwq_${q}_lanes_${n}_300"
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 $B --frames-in-flight $n 2>/dev/null | python -c "$P" "hwq_${q}_lanes_${n}_20"
done; done
