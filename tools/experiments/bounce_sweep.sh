# C2 frame time against the bounce limit: the rays beyond bounce 2 are 4 % of the frame's rays but their dependent chain is the looping pass's latency
for b in 8 5 3 2; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --bounces $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bounces', d['config']['workload'].split(' bounces')[0][-2:], round(d['ms_per_step'],5), round(d['config']['rays_per_frame']), round(d['latency_ms_one_frame'],4))"
done
