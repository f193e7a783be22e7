#!/bin/bash
# Round 6, GPU call i: same-box A/B of S2 on one GPU (--config 2s) and of the default line: the tree at the start of the round (tools/ab/tree_old) against HEAD
TAG=${1:-r06i}
REPO=$(pwd)
CACHE=/tmp/mf_frames
mkdir -p gpurun_out
for rep in 1 2; do
  for t in new old; do
    D=$REPO; [ $t = old ] && D=$REPO/tools/ab/tree_old
    (cd $D && timeout 300 python bench.py --config 2s --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline > $REPO/gpurun_out/${TAG}_2s_${t}_$rep.json 2> $REPO/gpurun_out/${TAG}_2s_${t}_$rep.err)
    python -c "
import json; d = json.load(open('$REPO/gpurun_out/${TAG}_2s_${t}_$rep.json')); print('2s $t $rep', round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,3) for k,v in (d['roofline'].get('stage_ms') or {}).items() if v} if d.get('roofline') else None)"
  done
done
for t in new old; do
  D=$REPO; [ $t = old ] && D=$REPO/tools/ab/tree_old
  (cd $D && timeout 300 python bench.py --frame-cache $CACHE --no-variants --no-host-input --no-cpu-baseline > $REPO/gpurun_out/${TAG}_1_${t}.json 2> $REPO/gpurun_out/${TAG}_1_${t}.err)
  python -c "
import json; d = json.load(open('$REPO/gpurun_out/${TAG}_1_${t}.json')); print('config 1 $t', round(d['value'],1), round(d['ms_per_step'],5), {k: round(v,4) for k,v in (d['roofline'].get('stage_ms') or {}).items() if v})"
done
