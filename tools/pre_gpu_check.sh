#!/bin/bash
# What to run in the build container BEFORE spending GPU minutes on a kernel change (none of it needs a GPU):
#   1. the product library and the emulated twins build;
#   2. the changed kernels are functionally right: the 23 structural cases on the CPU emulation (product + every variant named on the
#      command line), bit-compared with the product build;
#   3. the spill placement did not get worse than the last version that met hardware (profiles/r02_static_spills_igemm_default.txt);
#   4. the host logic is green.
#     bash tools/pre_gpu_check.sh [variant-name -Dflag ...]        e.g.  bash tools/pre_gpu_check.sh deepring -DMY_AB_FLAG=1
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
P=imagen-pytorch_amd
python -c "import __graft_entry__ as g; g.build()" | tail -n 3
TMP=$(mktemp -d /tmp/pre_gpu.XXXX)
IMAGEN_LIB_PATH=$PWD/$P/libimagen_emul.so python tools/emul/run_cases.py --out $TMP/product.pt > $TMP/product.log
tail -n 3 $TMP/product.log
LIBS="product"
if [ -n "${1:-}" ]; then
  NAME=$1
  bash tools/emul/build_emul_lib.sh "$@" | tail -n 1
  IMAGEN_LIB_PATH=$PWD/$P/libimagen_emul_$NAME.so python tools/emul/run_cases.py --out $TMP/$NAME.pt > $TMP/$NAME.log
  LIBS="product $NAME"
fi
python - "$TMP" $LIBS <<'PY'
import sys, torch
tmp, libs = sys.argv[1], sys.argv[2:]
res = {l: torch.load(f"{tmp}/{l}.pt", weights_only=False) for l in libs}
bad = [(l, k, r["err"]) for l, d in res.items() for k, r in d.items() if r["err"] >= 1e-3]
print("worst normwise error vs the fp32 contract:", max(r["err"] for d in res.values() for r in d.values()), "| failures:", bad)
for l in libs[1:]:
    same = all(torch.equal(res["product"][k]["y"], res[l][k]["y"]) for k in res["product"])
    print(f"{l}: bit-identical to the product build on all {len(res['product'])} cases: {same}")
assert not bad
PY
shift $(( $# > 0 ? 1 : 0 )) || true
python tools/scratch_report.py $P/csrc/igemm.hip "$@" | tail -n 1
echo "(last version that met hardware: $(tail -n 1 profiles/r02_static_spills_igemm_default.txt))"
echo "device code of the PRODUCT sources vs the versions that met hardware (profiles/r02_device_code_hashes.txt):"
for f in igemm conv_dma conv_stream; do
  h=$(python tools/scratch_report.py $P/csrc/$f.hip | tail -n 1 | awk '{print $4}')
  grep -q "$f.hip $h" profiles/r02_device_code_hashes.txt && echo "  $f.hip unchanged" || echo "  $f.hip CHANGED ($h): its kernels have to meet hardware again before their numbers are quoted"
done
python -m pytest tests/test_host_logic.py tests/test_plan_interp.py -q -x 2>&1 | tail -n 1
