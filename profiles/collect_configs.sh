#!/bin/bash
# profiles/collect_configs.sh <tag> -- one bench.py line per BASELINE config that fits one GPU (C1..C5) plus the animated and
# textured variants of C2:  gpurun --timeout 1200 -- 'bash profiles/collect_configs.sh r01_d'
# -> gpurun_out/<tag>_configs.jsonl (copy to profiles/).  C1-C3 carry the CPU baseline; C4 (64 spp, 16 bounces at 4K) is
# BASELINE's 8-GPU configuration run here on one GPU; C5's brute-force oracle is not runnable (cpu_baseline null).
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/${TAG}_configs.jsonl
: > $OUT
run() { echo "== $*" >&2; python3 bench.py "$@" 2>/dev/null | tail -1 >> $OUT; }
run --scene small --width 256 --height 256 --bounces 4 --steps 300 --warmup 30                      # C1
run --steps 300 --warmup 30                                                                           # C2 (headline)
run --steps 300 --warmup 30 --moving-camera --no-cpu-baseline                                         # C2 under a camera that moves every frame (App.cpp:531-553)
run --steps 300 --warmup 30 --turning-camera --no-cpu-baseline                                        # C2 under a camera that turns every frame (mouse look)
run --steps 300 --warmup 30 --animate --no-cpu-baseline                                               # C2 animated (N2)
run --steps 300 --warmup 30 --textures --no-cpu-baseline                                              # C2 textured (N1)
run --steps 300 --warmup 30 --env-map --no-cpu-baseline                                               # C2 lit by the lat-long environment map (a18)
run --steps 300 --warmup 30 --di --no-cpu-baseline                                                    # C2 with sphere-light direct illumination (N4)
run --width 3840 --height 2160 --spp 16 --steps 20 --warmup 3 --cpu-row-step 16                       # C3
run --width 3840 --height 2160 --spp 64 --bounces 16 --steps 6 --warmup 2 --no-cpu-baseline           # C4 on one GPU
run --scene procedural --steps 40 --warmup 5                                                          # C5
run --steps 300 --warmup 30 --force-tiles --no-cpu-baseline --no-roofline                            # C2 through the tile path on one rank (render tiles, gather, un-swizzle)
TAG=$TAG python3 - <<'PY'
import json, os
for l in open("gpurun_out/%s_configs.jsonl" % os.environ["TAG"]):
    d = json.loads(l)
    print("%-112s %9.3f ms/frame %9.0f Mrays/s  cpu %s" % (d["config"]["workload"][:112], d["ms_per_step"], d["value"], (d.get("cpu_baseline") or {}).get("value")))
PY
