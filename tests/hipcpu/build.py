"""Builds tests/_build/libmaskfusion_emu.so: the product's own sources (maskfusion_amd/csrc/*.hip) compiled with g++ against
tests/hipcpu/hipcpu.h and run on the CPU by tests/hipcpu/hipcpu_runtime.cpp.  TEST TOOLING: lets the kernel-logic tests run where there
is no GPU.  The product never loads this library, and nothing here is a fallback: maskfusion_amd.lib.load() knows only the HIP build.

The sources are compiled as they are, except for three mechanical edits applied in memory (the files on disk are untouched):
  * `asm volatile("s_waitcnt ...")` statements (profiling stamps) are dropped -- AMD mnemonics do not assemble on the host;
  * `extern __shared__ T name[];` becomes `T* name = (T*)hipcpu::dyn_shared();`;
  * `#pragma unroll` / `#pragma clang ...` are dropped (g++ warns about them; they do not change meaning).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "maskfusion_amd", "csrc")
OUT = os.path.join(os.path.dirname(HERE), "_build")
# the plain build lives in tests/_emu/ -- git-ignored like tests/_build/, but NOT gpurun-ignored: it travels to the GPU box, where
# tests/test_gpu_emu_agrees.py compares it with the hardware (building it there cost that test 200 s of the driver's GPU-suite limit)
SHIP = os.path.join(os.path.dirname(HERE), "_emu")
LIB = os.path.join(SHIP, "libmaskfusion_emu.so")
LIB_COOP = os.path.join(OUT, "libmaskfusion_emu_coop.so")     # -DHIPCPU_COOP: cooperative launches possible, slower (see hipcpu.h)
LIB_ASAN = os.path.join(OUT, "libmaskfusion_emu_asan.so")     # -fsanitize=address: out-of-bounds / use-after-free accesses of "device" memory
                                                              # (every hipMalloc is a malloc) abort with a report -- run under
                                                              # LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
sys.path.insert(0, ROOT)
from maskfusion_amd.build import SOURCES, HEADERS  # noqa: E402

CXX = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-pthread", "-w", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include"),
       "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-I", HERE, "-include", os.path.join(HERE, "hipcpu.h")]


def rewrite(text: str) -> str:
    text = re.sub(r"asm\s+volatile\s*\([^;]*\);", "", text)
    text = re.sub(r"extern\s+__shared__\s+([\w:]+)\s+(\w+)\s*\[\s*\]\s*;", r"\1* \2 = (\1*)hipcpu::dyn_shared();", text)
    text = re.sub(r"^\s*#pragma\s+(unroll|clang)[^\n]*$", "", text, flags=re.M)
    return text


def _stale(lib) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(HERE, f) for f in ("hipcpu.h", "hipcpu_runtime.cpp", "build.py")]   # (not the directory listing: __pycache__ changes whenever Python imports from here)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


LIB_UBSAN = os.path.join(OUT, "libmaskfusion_emu_ubsan.so")   # -fsanitize=undefined: LDS / local array indices out of bounds, misaligned vector accesses,
                                                              # shifts, signed overflow ... reported on stderr (UBSAN_OPTIONS=print_stacktrace=1)


def build(force: bool = False, only=None, coop: bool = False, asan: bool = False, ubsan: bool = False) -> str:
    lib = LIB_UBSAN if ubsan else LIB_ASAN if asan else (LIB_COOP if coop else LIB)
    cxx = CXX + (["-DHIPCPU_COOP"] if coop else []) + (["-fsanitize=address", "-fno-omit-frame-pointer", "-g1"] if asan else [])
    if ubsan:
        cxx = CXX + ["-fsanitize=undefined", "-fno-sanitize=float-cast-overflow", "-fno-omit-frame-pointer", "-g1"]
    tag = "emuu_" if ubsan else "emua_" if asan else "emuc_" if coop else "emu_"
    if coop and asan:                       # all workgroups resident AND AddressSanitizer: for kernels with device-wide barriers
        lib, tag = os.path.join(OUT, "libmaskfusion_emu_coop_asan.so"), "emuca_"
    if not (force or _stale(lib)) and only is None:
        return lib
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(OUT, tag + src.replace(".hip", ".o"))
        objs.append(obj)
        if only is not None and src not in only:
            continue
        text = rewrite(open(os.path.join(CSRC, src)).read())
        r = subprocess.run([*cxx, "-x", "c++", "-c", "-", "-o", obj], input=text.encode(), capture_output=True)
        if r.returncode != 0:
            sys.stderr.write(f"--- {src}\n" + r.stderr.decode()[:6000])
            raise SystemExit(1)
    if only is not None:
        return ""
    rt = os.path.join(OUT, tag + "runtime.o")
    subprocess.check_call([*cxx, "-c", os.path.join(HERE, "hipcpu_runtime.cpp"), "-o", rt])
    subprocess.check_call(["g++", "-shared", "-pthread", *(["-fsanitize=address"] if asan else ["-fsanitize=undefined"] if ubsan else []), "-o", lib, *objs, rt])
    return lib


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a.endswith(".hip")] or None
    print(build(force=True, only=only, coop="--coop" in sys.argv, asan="--asan" in sys.argv, ubsan="--ubsan" in sys.argv))
