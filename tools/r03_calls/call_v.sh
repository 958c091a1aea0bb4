#!/bin/bash
# Round-3 GPU call V: the real benchmark loop (6 lanes, 1000 steps per stage) with / without the per-request time table; cost of the table pass.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_v
mkdir -p $OUT
python - <<'PY' 2>/dev/null | tee $OUT/table_pass.txt
import time, torch, bench
dev = torch.device("cuda:0")
imagen = bench.build_imagen(1000, dev)
te = torch.randn(8, 256, 768, generator=torch.Generator().manual_seed(1234)).to(dev)
imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1, max_steps=2)
torch.cuda.synchronize()
for key, st in imagen._stages.items():
    eng = st['eng']
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); eng._tt_plan.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"stage {key[0]}: time-table pass {dt*1e3:.2f} ms for {st['coef'].shape[0]} steps x {eng.R} rows ({len(eng._tt_plan)} launches)")
PY
timeout 600 python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-roofline > $OUT/bench_tt.json 2>$OUT/bench_tt.err
python -c "import json;d=json.load(open('$OUT/bench_tt.json'));print('table:',d['value'],d['ms_per_step'],d['sequential']['value'],d['sequential']['ms_per_ddpm_step_pair'])"
IMAGEN_TIME_TABLE=0 timeout 600 python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-roofline > $OUT/bench_chain.json 2>$OUT/bench_chain.err
python -c "import json;d=json.load(open('$OUT/bench_chain.json'));print('chain:',d['value'],d['ms_per_step'],d['sequential']['value'],d['sequential']['ms_per_ddpm_step_pair'])"
