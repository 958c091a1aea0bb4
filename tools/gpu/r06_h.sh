#!/bin/bash
# Round 6, call H: HEAD after the last (no-op in the product build) kernel-source change: kernel / fusion tests, smoke(), bench.py with its DEFAULT flags
# (what the driver runs).
#   gpurun --timeout 1200 -- 'bash tools/gpu/r06_h.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_h
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fusion_gpu.py tests/test_rowchain_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_kernels.txt 2>&1; tail -n 2 $OUT/pytest_kernels.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench.py (default flags) wall seconds: $(( $(date +%s) - t0 ))"
tail -n 3 $OUT/bench_default.err
python - <<PY
import json
r = json.load(open("$OUT/bench_default.json"))
print({k: r[k] for k in ("value", "ms_per_step", "steps", "warmup")}, r["sequential"]["value"], r["sequential"].get("box"))
print(r["roofline"]["frac"], r["roofline"]["traffic"], r["cpu_baseline"]["value"])
PY
