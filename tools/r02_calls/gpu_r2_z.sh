cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l3 -- python $R/bench.py --mode lanes --lanes 3 --timesteps 60 --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_l3.log 2>&1
tail -2 /tmp/prof_l3.log | cut -c1-300
f=$(find /tmp/prof_l3 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r02_kernel_stats_lanes3.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $R/bench.py --mode sequential --timesteps 60 --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_s.log 2>&1
tail -2 /tmp/prof_s.log | cut -c1-300
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r02_kernel_stats_seq.csv
