#!/bin/bash
# Round-5 call T (session 2): the per-launch audits again with the final tool (bit-pattern change detection, the C5 test's own weights) —
# C5 null / cond row, README unet1 @64^2 and unet2 @256^2 null rows.
#   gpurun --timeout 150 -- 'bash tools/gpu/r05_t.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_t
mkdir -p $OUT
for c in "c5 --null" "c5" "u1 --size 64 --null" "u2 --size 256 --null"; do
  tag=$(echo $c | tr -d '-' | tr ' ' '_')
  timeout 60 python tools/op_audit.py --config $c --top 14 --json $OUT/op_audit_$tag.json > $OUT/op_audit_$tag.txt 2>&1
  grep -E "^#|nan" $OUT/op_audit_$tag.txt | head -n 4 | cut -c1-200
  sed -n '/largest/,/most coherent/p' $OUT/op_audit_$tag.txt | head -n 5 | cut -c1-190
done
