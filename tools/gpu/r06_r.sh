#!/bin/bash
# Call R: requests merged into one batch (Imagen.sample_requests; ABI 11: per-row Philox keys in DDPM_UPDATE): sampler / model tests on hardware, then the
# bench line with the informational merged-requests leg.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_r
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "ddpm or merged or philox or lane or sample_vs_reference or sample_options or graph_capture" > $OUT/pytest.txt 2>&1; tail -n 8 $OUT/pytest.txt | cut -c1-220
timeout 1200 python bench.py --steps 6 --warmup 6 --no-pmc --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -n 4 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "sequential", d["sequential"]["value"], "merged", d.get("merged_requests"), d.get("merged_requests_error"))
PY
