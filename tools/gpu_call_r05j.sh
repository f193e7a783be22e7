#!/bin/bash
# Round 5, GPU call j: per-chunk phase times of the one-launch clean pass on the full background map (tools/clean_prof.py, instrumented build)
TAG=${1:-r05j}
mkdir -p gpurun_out
cp maskfusion_amd/libmaskfusion_amd.so /tmp/lib_product.so
cp tools/ab/libmaskfusion_amd_prof.so maskfusion_amd/libmaskfusion_amd.so
timeout 300 python tools/clean_prof.py $TAG 2>&1 | tee gpurun_out/${TAG}_clean_prof.txt
cp /tmp/lib_product.so maskfusion_amd/libmaskfusion_amd.so
