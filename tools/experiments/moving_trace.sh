# kernel trace of the moving-camera run: how long the beam build takes beside the frames, and what it does to the frames' cadence
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PT_BEAM_REACH=${1:-32}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/moving_trace -o k -- python3 $R/bench.py --steps 200 --warmup 10 --prewarm 0 ${PT_TRACE_FLAGS---moving-camera} --no-cpu-baseline --no-roofline > $R/gpurun_out/moving_trace.json 2> $R/gpurun_out/moving_trace.err
python3 - <<'PY'
import csv, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
rows = list(csv.DictReader(open(R + "/gpurun_out/moving_trace/k_kernel_trace.csv")))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = "beam" if "beam_kernel" in n else "primary" if "bounce_kernel<true, unsigned short, true" in n else "loop" if "bounce_kernel" in n else None
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
ev.sort()
t0 = ev[0][0]
d = collections.defaultdict(list)
for s, e, k, q in ev: d[k].append(e - s)
for k, v in d.items(): print(k, len(v), "avg us", round(sum(v) / len(v) / 1e3, 1))
ends = [e for s, e, k, q in ev if k == "loop"]
beams = [(s, e) for s, e, k, q in ev if k == "beam"]
gaps = [(ends[i + 1] - ends[i]) / 1e3 for i in range(len(ends) - 1)]
print("frame completion cadence us: mean", round(sum(gaps[20:]) / len(gaps[20:]), 1), "sorted tail", [round(g) for g in sorted(gaps[20:])[-12:]])
for bs, be in beams[:8]:
    near = [round((ends[i + 1] - ends[i]) / 1e3) for i in range(len(ends) - 1) if bs - 400e3 < ends[i] < be + 600e3]
    print("build at", round((bs - t0) / 1e3), "us for", round((be - bs) / 1e3), "us; cadence around it:", near)
PY
