#!/bin/bash
# Round-5 call S (session 2): tools/op_audit.py over the HEADLINE plans on hardware — README unet1 @64^2 and unet2 @256^2 (null rows): every
# launch on the GPU and in the plan interpreter from identical inputs.
#   gpurun --timeout 200 -- 'bash tools/gpu/r05_s.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_s
mkdir -p $OUT
echo "=== op audit, README unet1 @64, null row"
timeout 60 python tools/op_audit.py --config u1 --size 64 --null --top 12 --json $OUT/op_audit_u1_null.json > $OUT/op_audit_u1_null.txt 2>&1; head -n 18 $OUT/op_audit_u1_null.txt | cut -c1-190
echo "=== op audit, README unet2 @256, null row"
timeout 120 python tools/op_audit.py --config u2 --size 256 --null --top 12 --json $OUT/op_audit_u2_null.json > $OUT/op_audit_u2_null.txt 2>&1; head -n 18 $OUT/op_audit_u2_null.txt | cut -c1-190
