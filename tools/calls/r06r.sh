#!/bin/bash
# Round 6, GPU call r: the tile pass of the VGA prediction -- list-length histogram and in-kernel stamps against the list length
TAG=${1:-r06r}
mkdir -p gpurun_out
PYTHONPATH=. timeout 400 python tools/tile_hist.py 600 > gpurun_out/${TAG}_tile_hist.txt 2>&1
PYTHONPATH=. timeout 400 python tools/splat_prof.py 600 > gpurun_out/${TAG}_splat_prof.txt 2>&1
cat gpurun_out/${TAG}_tile_hist.txt | cut -c1-400; cat gpurun_out/${TAG}_splat_prof.txt
