cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { printf "%-40s" "$1"; shift; env "$@" timeout 600 python bench.py --mode sequential --timesteps 200 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
run "defaults (16-byte stores everywhere)"
run "defaults again"
