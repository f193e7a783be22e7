#!/bin/bash
# Round 5, GPU call k: per-chunk phase times of the one-launch clean pass, variants of the instrumented build (tools/ab/libmaskfusion_amd_prof_<v>.so)
TAG=${1:-r05k}
mkdir -p gpurun_out
cp maskfusion_amd/libmaskfusion_amd.so /tmp/lib_product.so
for v in $(ls tools/ab | sed -n 's/libmaskfusion_amd_prof_\(.*\)\.so/\1/p'); do
  cp tools/ab/libmaskfusion_amd_prof_$v.so maskfusion_amd/libmaskfusion_amd.so
  echo "=== variant $v"
  timeout 300 python tools/clean_prof.py ${TAG}_$v 2>&1 | tee gpurun_out/${TAG}_${v}_clean_prof.txt
done
cp /tmp/lib_product.so maskfusion_amd/libmaskfusion_amd.so
