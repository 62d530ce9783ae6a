# Where the animated C2 frame loses against the static one: beams (a moving scene has no cached lists) vs the per-frame upload + refit.
set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
B="--no-cpu-baseline --no-roofline --steps 300 --warmup 30"
for rep in 1 2; do
python bench.py $B 2>/dev/null | python -c "$P" "static, beams"
PT_BEAMS=0 python bench.py $B 2>/dev/null | python -c "$P" "static, no beams"
python bench.py $B --animate 2>/dev/null | python -c "$P" "animated (no beams by construction)"
done
