#!/bin/bash
# profiles/publish.sh <round tag> -- copy the summaries of a collection (gpurun_out/<tag>_c2 ... _c5, collect_round.sh) into profiles/, where they are tracked:
# the bench lines, the --stats kernel summary, the stamped counters (also as counters_<workload>.json, which bench.py merges into its roofline object).
set -e
R=${1:-r03}
declare -A W=( [c2]=demo-1920x1080-1spp-8b [c3]=demo-3840x2160-16spp-8b [c4]=demo-3840x2160-64spp-16b [c5]=procedural-1920x1080-1spp-8b )
for k in c2 c3 c4 c5; do
  d=gpurun_out/${R}_$k
  for f in bench bench_20_steps bench_under_rocprof; do cp $d/$f.json profiles/${R}_${k}_$f.json; done
  cp $d/stats/k_kernel_stats.csv profiles/${R}_${k}_kernel_stats.csv
  cp $d/counters.json profiles/${R}_${k}_counters.json
  cp $d/counters.json profiles/counters_${W[$k]}.json
  [ -f gpurun_out/${R}_${k}_bwc.json ] && cp gpurun_out/${R}_${k}_bwc.json profiles/${R}_${k}_bench_with_counters.json
done
cp gpurun_out/${R}_configs.jsonl profiles/${R}_configs.jsonl
