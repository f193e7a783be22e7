timeout 400 python -m pytest tests -m gpu -q > gpurun_out/pytest_r02b.log 2>&1; tail -5 gpurun_out/pytest_r02b.log
timeout 120 python bench.py --no-cpu-baseline > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; cut -c1-300 gpurun_out/r02b_bench.json
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
B="--frames 40 --gen-workers 1 --min-seconds 0 --no-cpu-baseline --no-host-input --no-roofline"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $REPO/bench.py --steps 12 --warmup 45 $B > /tmp/prof_f.log 2>&1; echo "fetch rc $?"
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python $REPO/bench.py --steps 12 --warmup 45 $B > /tmp/prof_w.log 2>&1; echo "write rc $?"
timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_cf -o f -- $REPO/tools/micro/fetch_calib > /tmp/prof_cf.log 2>&1; echo "cf rc $?"
timeout 60 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_cw -o w -- $REPO/tools/micro/fetch_calib > /tmp/prof_cw.log 2>&1; echo "cw rc $?"
cd $REPO
for t in f w cf cw; do F=$(find /tmp/prof_$t -name "*counter_collection.csv" | head -1); echo "$t -> $F"; done
python tools/pmc_summary.py $(find /tmp/prof_f -name "*counter_collection.csv" | head -1) 300 > gpurun_out/r02_pmc_fetch.csv
python tools/pmc_summary.py $(find /tmp/prof_w -name "*counter_collection.csv" | head -1) 300 > gpurun_out/r02_pmc_write.csv
python tools/pmc_summary.py $(find /tmp/prof_cf -name "*counter_collection.csv" | head -1) > gpurun_out/r02_calib_fetch.csv
python tools/pmc_summary.py $(find /tmp/prof_cw -name "*counter_collection.csv" | head -1) > gpurun_out/r02_calib_write.csv
python tools/make_pmc_json.py r02 gpurun_out/r02_pmc_fetch.csv gpurun_out/r02_pmc_write.csv gpurun_out/r02_calib_fetch.csv gpurun_out/r02_calib_write.csv "$(cat .tree_id)" && cp profiles/r02_pmc.json gpurun_out/
tail -2 /tmp/prof_f.log | cut -c1-200; cat gpurun_out/r02_calib_fetch.csv | head -5
