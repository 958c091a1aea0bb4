#!/bin/bash
# round 2, call P: 16-byte operand loads / stores in the generic igemm epilogue — parity, A/B on the res_conv shapes, sequential bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/p_pytest.txt
for d in 0 256; do
  echo "== IMAGEN_IGEMM_DBG=$d" >> gpurun_out/p_latency.txt
  IMAGEN_IGEMM_DBG=$d timeout 200 python tools/latency_probe.py res_conv 2>&1 | grep -v MIX >> gpurun_out/p_latency.txt
done
timeout 300 python bench.py --steps 6 --warmup 3 --no-pmc > gpurun_out/p_bench.json 2> gpurun_out/p_bench.err
IMAGEN_IGEMM_DBG=256 timeout 300 python bench.py --steps 6 --warmup 3 --no-pmc --mode sequential > gpurun_out/p_bench_old.json 2>> gpurun_out/p_bench.err
tail -3 gpurun_out/p_pytest.txt; cat gpurun_out/p_latency.txt; cut -c1-400 gpurun_out/p_bench.json; cut -c1-300 gpurun_out/p_bench_old.json
