#!/bin/bash
# Round 6, GPU call w: test_static_switches_vs_oracle failed once (object count 2475 against 2476 on the first tracked frame): flaky, or a switch?
T="tests/test_gpu_switches.py::test_static_switches_vs_oracle"
for i in 1 2 3; do timeout 300 python -m pytest $T -q -m gpu -n 0 -x 2>&1 | tail -1; done
echo "--- objectStream=0"
for i in 1 2; do MF_TEST_PARAMS=objectStream=0 timeout 300 python -m pytest $T -q -m gpu -n 0 -x 2>&1 | tail -1; done
echo "--- fusedPreprocessLaunch=0"
for i in 1 2; do MF_TEST_PARAMS=fusedPreprocessLaunch=0 timeout 300 python -m pytest $T -q -m gpu -n 0 -x 2>&1 | tail -1; done
echo "--- both off"
for i in 1 2; do MF_TEST_PARAMS=fusedPreprocessLaunch=0,objectStream=0 timeout 300 python -m pytest $T -q -m gpu -n 0 -x 2>&1 | tail -1; done
echo "--- new tests"
timeout 600 python -m pytest tests/test_gpu_switches.py -q -m gpu -k "fused_preprocess or frame_pyramid_launch" 2>&1 | tail -3
