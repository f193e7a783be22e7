"""Builds oracle/_ref/libmf_glsl.so: the reference's own GLSL shaders (Core/Shaders/*.vert / *.frag / *.glsl) compiled as C++ and run
on the CPU -- what pins the oracle's restatement of the OpenGL half of the hot path to
reference-compiled code.

TEST INFRASTRUCTURE ONLY.  The shader sources are read where they lie under /root/reference and are never written into the
repository; the translation unit is assembled in memory and piped to g++ on stdin.  The only edits, all mechanical and listed here:
  * `#version ...` lines dropped; `#include "x.glsl"` replaced by the text of x.glsl (what pangolin's shader loader does);
  * `layout(...)` removed, the storage qualifiers `in` / `out` / `uniform` / `flat` at the start of a declaration removed (the
    variables become namespace-scope globals that the harness sets and reads);
  * every floating-point literal without a suffix gets an `f` (GLSL 3.30 has no double: `1.0 / cols` is a float division there);
  * each shader is wrapped in `namespace mfgl { namespace sh_<name> { ... } }`.
oracle/glsl_shim/mfgl.h supplies vec2/3/4 (with the .xy / .zw / .xyz swizzles the shaders use), mat3/4, samplers and the built-ins;
oracle/glsl_shim/mfgl_api.cpp runs the passes (vertex loop, point / sprite raster rule, depth test) around the shaders' main().
The geometry shaders (vertex_feedback.geom, data.geom, copy_unstable.geom) are one-line emit conditions and are restated in the
harness next to the call of the vertex shader they follow.

The GPU box has no /root/reference: there the prebuilt .so (it travels with the snapshot) is used as is.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "glsl_shim")
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libmf_glsl.so")
REF_SHADERS = os.environ.get("MF_REFERENCE_SHADER_DIR", "/root/reference/Core/Shaders")
SHADERS = ["depth_bilateral_metric.frag", "vertex_feedback.vert", "init_unstable.vert", "index_map.vert", "index_map.frag", "data.vert",
           "data.frag", "update.vert", "copy_unstable.vert", "splat.vert", "combo_splat.frag", "splat_models.vert",
           "combo_splat_models.frag", "fill_vertex.frag", "fill_normal.frag", "fill_rgb.frag"]
CXXFLAGS = ["-std=c++14", "-O2", "-ffp-contract=off", "-fPIC", "-w", "-I", SHIM]

_FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")
_QUALIFIER = re.compile(r"^(\s*)(?:flat\s+)?(?:in|out|uniform)\s+", re.M)
_LAYOUT = re.compile(r"layout\s*\([^)]*\)\s*")
_INCLUDE = re.compile(r'^\s*#include\s+"([^"]+)"\s*$', re.M)


def reference_available() -> bool:
    return all(os.path.exists(os.path.join(REF_SHADERS, s)) for s in SHADERS)


def _read(name: str) -> str:
    with open(os.path.join(REF_SHADERS, name), "r", encoding="utf-8", errors="replace") as f:
        return f.read()


def shader_to_cxx(name: str) -> str:
    """the mechanical edits listed in the module docstring"""
    text = _read(name)
    text = _INCLUDE.sub(lambda m: _read(m.group(1)), text)
    text = re.sub(r"^\s*#version.*$", "", text, flags=re.M)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                      # licence header (holds words like "in" at line starts)
    text = _LAYOUT.sub("", text)
    text = _QUALIFIER.sub(lambda m: m.group(1), text)
    text = _FLOAT_LIT.sub(lambda m: m.group(1) + "f", text)
    ns = "sh_" + name.replace(".", "_")
    return f"namespace mfgl {{ namespace {ns} {{\n{text}\n}} }}\n"


def translation_unit() -> str:
    parts = ['#include "mfgl.h"\n']
    parts += [shader_to_cxx(s) for s in SHADERS]
    with open(os.path.join(SHIM, "mfgl_api.cpp")) as f:
        parts.append(f.read())
    return "".join(parts)


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SHIM, f) for f in os.listdir(SHIM)] + [os.path.abspath(__file__), os.path.join(HERE, "mf_oracle.c")]
    deps += [os.path.join(REF_SHADERS, s) for s in SHADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False) -> str | None:
    """Returns the library path, or None when neither the reference shaders nor a prebuilt library are present."""
    if not reference_available():
        return LIB if os.path.exists(LIB) else None
    if not (force or _stale()):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "mf_glsl.o")
    subprocess.run(["g++", *CXXFLAGS, "-x", "c++", "-c", "-", "-o", obj], input=translation_unit().encode(), check=True)
    # exp() / acos() of the shaders are the oracle's shared polynomials (mfgl.h): link the oracle's object code in
    orc = os.path.join(OUT, "mf_oracle_for_glsl.o")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-std=gnu11", "-w", "-c", os.path.join(HERE, "mf_oracle.c"), "-o", orc])
    subprocess.check_call(["g++", "-shared", "-fopenmp", "-o", LIB, obj, orc, "-lm"])
    return LIB


if __name__ == "__main__":
    if "--print" in sys.argv:
        print(translation_unit())
    else:
        print(build(force="--force" in sys.argv))
