# kernel trace of the moving-camera run (how long the beam build takes beside the frames, what it delays)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PT_BEAM_REACH=${1:-12}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/moving_trace -o k -- python3 $R/bench.py --steps 120 --warmup 10 --prewarm 0 --moving-camera --no-cpu-baseline --no-roofline > $R/gpurun_out/moving_trace.json 2> $R/gpurun_out/moving_trace.err
python3 - <<'PY'
import csv, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
rows = list(csv.DictReader(open(R + "/gpurun_out/moving_trace/k_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    k = "beam" if "beam_kernel" in n else "primary" if "bounce_kernel<true, unsigned short, true" in n else "loop" if "bounce_kernel" in n else n[:40]
    d[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in d.items():
    print(k, len(v), "avg us", round(sum(e - s for s, e in v) / len(v) / 1e3, 1))
PY
