# True in-graph cost of op classes: the benchmark (100 DDPM steps per stage) with classes of launches removed from the plans
# (IMAGEN_SKIP, see ops.py) — the wall-clock difference to the full run is what the class costs inside the hipGraph, launch
# boundaries and cache effects included.  Results of ablated runs are garbage by construction.
cd $GRAFT_REPO_ROOT
export IMAGEN_CONV_DMA=${IMAGEN_CONV_DMA:-0} IMAGEN_GCA_IN_EPILOGUE=${IMAGEN_GCA_IN_EPILOGUE:-0}
run() { printf "%-46s" "$1"; IMAGEN_SKIP="$2" IMAGEN_SKIP_NOOP="$3" timeout 600 python bench.py --timesteps 100 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run "full" "" 0
run "no gca_partial/gca_final" "^gca_" 0
run "no gca_final" "^gca_final" 0
run "no rowstat" "^rowstat" 0
run "no gate_residual" "^gate_residual" 0
run "no ln_residual" "^ln_residual" 0
run "no attention+kv_prep" "^(attention|kv_prep)" 0
run "no quantile/cfg/ddpm (sampler)" "^(quantile|cfg_x0|ddpm_update)" 0
run "no igemm res_conv" "^igemm:.*res_conv" 0
run "no igemm block1/block2 (3x3)" "^igemm:.*block[12]$" 0
run "no igemm 1x1 attn/ff (qkv,to_q,to_out,ff)" "^igemm:.*(qkv|to_q|to_out|ff\.lin|ctx\.)" 0
run "no igemm init/final conv" "^igemm:(init_conv|final_conv)" 0
run "no igemm at all" "^igemm" 0
run "everything a no-op launch" "." 1
run "full (again)" "" 0
