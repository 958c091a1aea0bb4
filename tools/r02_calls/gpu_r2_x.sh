cd $GRAFT_REPO_ROOT
run() { python bench.py --timesteps 250 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), '->', r['ms_per_step'], 'ms/pass', r['value'], 'img/s', 'seq', r.get('sequential',{}).get('ms_per_step'))" "$@"; }
run --mode sequential --steps 3 --warmup 1
run --mode pipeline --steps 6 --warmup 2
run --mode lanes --lanes 2 --steps 6 --warmup 2
run --mode lanes --lanes 3 --steps 9 --warmup 3
run --mode lanes --lanes 4 --steps 12 --warmup 4
run --mode lanes --lanes 5 --steps 15 --warmup 5
run --mode lanes --lanes 6 --steps 18 --warmup 6
