#!/bin/bash
# profiles/collect_round.sh <round tag, e.g. r03> -- every rocprofv3 collection of a round in one call: C2 (headline), C3, C4 (one GPU), C5, and the
# one-line-per-config table.  gpurun --timeout 1100 -- 'bash profiles/collect_round.sh r03'; then copy gpurun_out/<tag>_*/ summaries into profiles/ (README.md).
set -e -o pipefail
R=${1:-r03}
bash profiles/collect.sh ${R}_c2 demo-1920x1080-1spp-8b > gpurun_out/${R}_c2.log 2>&1
echo "[round] C2 done"
PT_PROFILE_STEPS=60 PT_PROFILE_WARMUP=5 PT_PROFILE_PMC_STEPS=6 bash profiles/collect.sh ${R}_c3 demo-3840x2160-16spp-8b --width 3840 --height 2160 --spp 16 > gpurun_out/${R}_c3.log 2>&1
echo "[round] C3 done"
PT_PROFILE_STEPS=16 PT_PROFILE_WARMUP=2 PT_PROFILE_PMC_STEPS=3 bash profiles/collect.sh ${R}_c4 demo-3840x2160-64spp-16b --width 3840 --height 2160 --spp 64 --bounces 16 > gpurun_out/${R}_c4.log 2>&1
echo "[round] C4 done"
PT_PROFILE_STEPS=60 PT_PROFILE_WARMUP=5 PT_PROFILE_PMC_STEPS=8 bash profiles/collect.sh ${R}_c5 procedural-1920x1080-1spp-8b --scene procedural > gpurun_out/${R}_c5.log 2>&1
echo "[round] C5 done"
bash profiles/collect_configs.sh ${R} > gpurun_out/${R}_configs.log 2>&1
echo "[round] configs done"
