#!/bin/bash
# Round 6, call I: the per-request time table filled in chunks of steps (host-side change only): the sampler / time-table tests on hardware (the table
# must stay bit-identical to the per-step chain), the step (unchanged by construction) and the C2 leg (8 chunks).
#   gpurun --timeout 1200 -- 'bash tools/gpu/r06_i.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "time_table or sample or elucidated or lanes or pipelined or conditioning" > $OUT/pytest_sampler.txt 2>&1; tail -n 12 $OUT/pytest_sampler.txt | cut -c1-220
timeout 400 python tools/step_time.py --steps 60 --reps 3 --tag chunked_time_table 2>/dev/null | tail -n 1 | tee $OUT/step.jsonl
timeout 400 python bench.py --config c2 --steps 2 --config-steps 50 2>/dev/null | tail -n 1 | cut -c1-400 | tee $OUT/bench_c2.json
