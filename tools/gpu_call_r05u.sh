#!/bin/bash
# Round 5, GPU call u: what the driver does at round end, on the final tree: the -m gpu suite, __graft_entry__.smoke(), bench.py with its defaults
TAG=${1:-r05u}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest -m gpu rc=$? wall $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.log
t1=$(date +%s)
timeout 600 python bench.py > gpurun_out/${TAG}_bench_defaults.json 2> gpurun_out/${TAG}_bench_defaults.err
echo "bench.py (no flags) rc=$? wall $(( $(date +%s) - t1 )) s"
cut -c1-330 gpurun_out/${TAG}_bench_defaults.json
