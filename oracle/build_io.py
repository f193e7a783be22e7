"""Builds oracle/_ref/libmf_io.so: the reference's own log-reader text -- GUI/Tools/KlgLogReader.cpp (constructor, getNext, readFrame, ...),
the LogReader / KlgLogReader class declarations, Core/FrameData.h, Core/Utils/Resolution.h, Core/Utils/Macros.h and
ImageLogReader::loadMaskIDs (GUI/Tools/ImageLogReader.cpp) -- compiled over oracle/io_shim/mfio_cv.h (a cv::Mat stand-in) and zlib.
It pins the byte formats of SURVEY.md row 8f-1 (`.klg`, `Mask####.txt`) that maskfusion_amd/io/readers.py reads: tests/test_io_pin.py.

TEST INFRASTRUCTURE ONLY.  Nothing is copied into the repository: the pieces are cut out of the reference files in memory, put where
oracle/io_shim/mfio_api.cpp carries a marker, and piped to g++ on stdin.  Output: oracle/_ref/libmf_io.so (git-ignored).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "io_shim")
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libmf_io.so")
REF = os.environ.get("MF_REFERENCE_DIR", "/root/reference")
FILES = {k: os.path.join(REF, *v) for k, v in dict(
    klg_cpp=("GUI", "Tools", "KlgLogReader.cpp"), klg_h=("GUI", "Tools", "KlgLogReader.h"), log_h=("GUI", "Tools", "LogReader.h"),
    img_cpp=("GUI", "Tools", "ImageLogReader.cpp"), frame_h=("Core", "FrameData.h"), res_h=("Core", "Utils", "Resolution.h"),
    macros_h=("Core", "Utils", "Macros.h"), jpeg_h=("GUI", "Tools", "JPEGLoader.h")).items()}
JPEG_SO = "libjpeg.so.8"     # libjpeg-turbo's IJG-v8 ABI: what the image ships (no headers: oracle/io_shim/jpeglib.h + jpeg_probe.c)


def reference_available() -> bool:
    return all(os.path.exists(p) for p in FILES.values())


def _read(key: str) -> str:
    return open(FILES[key], encoding="utf-8", errors="replace").read()


def _strip_includes(text: str) -> str:
    return "\n".join(l for l in text.split("\n") if not l.lstrip().startswith("#include") and not l.lstrip().startswith("#pragma once"))


def _class(text: str, head: str) -> str:
    """`class X ... {` up to the matching `};`"""
    a = text.index(head)
    depth, i = 0, text.index("{", a)
    while True:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
        if depth == 0:
            return text[a:text.index(";", i) + 1]


def _function(text: str, head: str) -> str:
    a = text.index(head)
    depth, i = 0, text.index("{", a)
    while True:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
        if depth == 0:
            return text[a:i]


def translation_unit() -> str:
    api = open(os.path.join(SHIM, "mfio_api.cpp")).read()
    klg = _read("klg_cpp")
    parts = {
        "MFIO_RESOLUTION_H": _strip_includes(_read("res_h")),
        "MFIO_MACROS_H": _strip_includes(_read("macros_h")),
        "MFIO_FRAMEDATA_H": _strip_includes(_read("frame_h")),
        "MFIO_LOGREADER_CLASS": _class(_read("log_h"), "class LogReader {"),
        "MFIO_KLGLOGREADER_CLASS": _class(_read("klg_h"), "class KlgLogReader : public LogReader {"),
        "MFIO_KLGLOGREADER_CPP": klg[klg.index("KlgLogReader::KlgLogReader("):],
        "MFIO_LOADMASKIDS": _function(_read("img_cpp"), "void ImageLogReader::loadMaskIDs("),
        "MFIO_JPEGLOADER_H": _strip_includes(_read("jpeg_h")),
    }
    for k, v in parts.items():
        assert api.count("\n" + k + "\n") == 1, k
        api = api.replace("\n" + k + "\n", "\n" + v + "\n")
    return api


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SHIM, f) for f in os.listdir(SHIM)] + [os.path.abspath(__file__)] + list(FILES.values())
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False) -> str | None:
    """Returns the library path, or None when neither the reference nor a prebuilt library are present."""
    if not reference_available():
        return LIB if os.path.exists(LIB) else None
    if not (force or _stale()):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    version, size = jpeg_abi()
    # -DNDEBUG: Macros.h's RELEASE flavour of CHECK_THROW (`if (!x) throw`), the way upstream ships
    subprocess.run(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-w", "-DNDEBUG", f"-DMFIO_JPEG_LIB_VERSION={version}",
                    f"-DMFIO_JPEG_DECOMPRESS_SIZE={size}", "-I", SHIM, "-x", "c++", "-", "-o", LIB, "-lz", "-l:" + JPEG_SO],
                   input=translation_unit().encode(), check=True)
    return LIB


def jpeg_abi():
    """(JPEG_LIB_VERSION, sizeof(struct jpeg_decompress_struct)) as the installed libjpeg reports them (oracle/io_shim/jpeg_probe.c)"""
    exe = os.path.join(OUT, "jpeg_probe")
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["gcc", "-O1", "-o", exe, os.path.join(SHIM, "jpeg_probe.c"), "-l:" + JPEG_SO], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    return int(out[0]), int(out[1])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
