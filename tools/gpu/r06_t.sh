#!/bin/bash
# Call T: package power and rocm-smi clocks while one kernel shape loops (tools/probe/power_under_kernel.py).
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r06_t
mkdir -p $OUT
timeout 600 python tools/probe/power_under_kernel.py > $OUT/power_under_kernel.jsonl 2> $OUT/err.txt; tail -n 3 $OUT/err.txt; cat $OUT/power_under_kernel.jsonl | cut -c1-700
