#!/bin/bash
# Round-3 GPU call C: (1) -m gpu suite on the current tree (GCA_TAIL grid cap, ticket path removed, probes), (2) sequential step time,
# (3) lanes throughput vs the number of lanes and vs persistent grids capped at one workgroup per CU (co-residency of other lanes' kernels).
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_c
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 6 $OUT/pytest.log
cp $R/gpurun_out/parity_measured.json $OUT/ 2>/dev/null
T="timeout 200 python tools/step_time.py"
$T --tag "default, lanes 3" --lanes 3 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "lanes 4" --lanes 4 --reps 1 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "lanes 6" --lanes 6 --reps 1 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_PERSIST_WG_PER_CU=1 $T --tag "persistent grids: 1 WG/CU, lanes 3" --lanes 3 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_PERSIST_WG_PER_CU=1 $T --tag "persistent grids: 1 WG/CU, lanes 5" --lanes 5 --reps 1 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
GPU_MAX_HW_QUEUES=8 $T --tag "GPU_MAX_HW_QUEUES=8, lanes 6" --lanes 6 --reps 1 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
GPU_MAX_HW_QUEUES=8 IMAGEN_PERSIST_WG_PER_CU=1 $T --tag "GPU_MAX_HW_QUEUES=8, 1 WG/CU, lanes 6" --lanes 6 --reps 1 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
python - <<'PY'
import ctypes, torch, sys
sys.path.insert(0, '.')
import bench
print("calibration:", bench.calibration_leg(torch.device("cuda:0")))
PY
