set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
run() { echo "== LDS=$1 DMA=$2 PREP=$3 GCA=$4"; IMAGEN_CONV_LDS=$1 IMAGEN_CONV_DMA=$2 IMAGEN_ACT_PREP_MIN_COUT=$3 IMAGEN_GCA_IN_EPILOGUE=$4 timeout 600 python bench.py --timesteps 100 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_h.err | cut -c95-200; }
run 0 0 0 0
run 1 0 0 1
run 1 1 0 1
run 0 1 0 1
run 0 1 128 1
run 0 0 0 0
