set -e -o pipefail
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],5))'
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "C2_default"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "C2_default"
for f in 0 1; do
  export PT_FUSE_LOOP=$f
  python bench.py --scene small --width 256 --height 256 --bounces 4 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "C1_fuse$f"
  python bench.py --width 960 --height 540 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "960_fuse$f"
  python bench.py --width 1280 --height 720 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "720p_fuse$f"
  python bench.py --width 1280 --height 720 --spp 4 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "720p4spp_fuse$f"
  python bench.py --width 640 --height 384 --spp 4 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$P" "640_4spp_fuse$f"
  python bench.py --width 640 --height 384 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --force-tiles 2>/dev/null | python -c "$P" "tiles640_fuse$f"
  python bench.py --width 640 --height 384 --steps 400 --warmup 40 --no-cpu-baseline --no-roofline --textures --di 2>/dev/null | python -c "$P" "640_tex_di_fuse$f"
done
