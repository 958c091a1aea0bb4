cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { printf "%-40s" "$1"; shift; env "$@" timeout 600 python bench.py --mode sequential --timesteps 200 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
run "stream off" IMAGEN_CONV_STREAM=0
run "stream on" IMAGEN_CONV_STREAM=1
run "stream on, min tiles 256" IMAGEN_CONV_STREAM=1 IMAGEN_STREAM_MIN_TILES=256
run "stream on + concat pro" IMAGEN_CONV_STREAM=1 IMAGEN_STREAM_CONCAT_PRO=1
