cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fusion_gpu.py -m gpu -q --tb=short 2>&1 | tail -4
run() { printf "%-60s" "$1"; shift; env "$@" timeout 600 python bench.py --timesteps 100 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run "r01 path + new res_conv epilogue (DMA=0 GCA=0)" IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
run "DMA=1 prep=0 GCA=1 (tiles<=1024)" IMAGEN_CONV_DMA=1 IMAGEN_ACT_PREP_MIN_COUT=0 IMAGEN_GCA_IN_EPILOGUE=1
run "DMA=1 prep=0 GCA=1 (tiles<=512)" IMAGEN_CONV_DMA=1 IMAGEN_ACT_PREP_MIN_COUT=0 IMAGEN_GCA_IN_EPILOGUE=1 IMAGEN_GCA_EPILOGUE_MAX_TILES=512
run "DMA=1 prep=0 GCA=1 (tiles<=4096)" IMAGEN_CONV_DMA=1 IMAGEN_ACT_PREP_MIN_COUT=0 IMAGEN_GCA_IN_EPILOGUE=1 IMAGEN_GCA_EPILOGUE_MAX_TILES=4096
run "DMA=1 prep=0 GCA=0" IMAGEN_CONV_DMA=1 IMAGEN_ACT_PREP_MIN_COUT=0 IMAGEN_GCA_IN_EPILOGUE=0
run "DMA=1 prep=128 GCA=1 (tiles<=1024)" IMAGEN_CONV_DMA=1 IMAGEN_ACT_PREP_MIN_COUT=128 IMAGEN_GCA_IN_EPILOGUE=1
run "r01 path again" IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
