#!/bin/bash
# Everything that can be checked WITHOUT a GPU, in rising order of cost (run from the repo root).  The parity claims rest on the MI355X
# runs (python -m pytest tests -m gpu through gpurun); this is the development gate in a container that has no GPU.
#   quick (default, ~4 min): build, CPU suite (oracle vs reference-compiled code, host logic, ABI, gloo, emulated smoke tests)
#   full  (~50 min)        : + the whole -m gpu suite against the kernels executed on the CPU (logic)
#   deep  (~3 h)           : + the same suite under AddressSanitizer and UBSan, the reversed schedule, the fuzzers
set -e
MODE=${1:-quick}
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -q -m "not gpu"
[ "$MODE" = quick ] && exit 0
MF_EMU=1 python -m pytest tests -m gpu -q -p no:cacheprovider
[ "$MODE" = full ] && exit 0
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 MF_EMU=1 MF_EMU_ASAN=1 python -m pytest tests -m gpu -q -p no:cacheprovider
UBSAN_OPTIONS=print_stacktrace=1 MF_EMU=1 MF_EMU_UBSAN=1 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee /tmp/ubsan.log; ! grep -q "runtime error" /tmp/ubsan.log
python tools/emu_schedule_check.py; MF_BIG_FORMS=1 python tools/emu_schedule_check.py
python tools/emu_fuzz.py 100 1000; FUZZ_ODD_SIZES=1 python tools/emu_fuzz.py 60 11000; python tools/emu_fuzz.py 30 7000 mm
python tools/emu_fuzz_labels.py 300 100
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 MF_EMU_ASAN=1 python tools/emu_fuzz_api.py 30 700
# (gpurun refuses a snapshot that holds -fsanitize objects, whatever .gpurunignore says: the sanitizer builds do not outlive this script)
rm -f tests/_build/emua_* tests/_build/emuca_* tests/_build/*asan* tests/_build/*ubsan*
