"""Builds libmaskfusion_amd.so (HIP, gfx950 only) in-tree with hipcc.  No CPU fallback exists."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmaskfusion_amd.so")
SOURCES = ["mf_preproc.hip", "mf_odometry.hip", "mf_rgbd.hip", "mf_surfel.hip", "mf_splat.hip", "mf_segment.hip", "mf_labels.hip", "mf_labels_gpu.hip", "mf_context.hip"]
HEADERS = ["mf_internal.h", "mf_device.h", "mf_labels.h", os.path.join("..", "..", "include", "maskfusion_amd.h"), "mf_rgbd_device.h", "mf_walk.h",
           "mf_frame.inl", "mf_model_api.inl", "mf_ktest.inl"]   # (the .inl files are parts of mf_context.hip)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# per-file additions.  mf_odometry: the SLP vectoriser pairs the 28 upper-triangle products of the ICP row into v_pk_fma_f32 and then
# spends more v_mov_b32 on assembling the operand pairs than it saves (k_icp_iter<512>: 2643 -> 2432 instructions, 589 -> 286 moves)
FILE_FLAGS = {"mf_odometry.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not (force or _stale()):
        return LIB
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
