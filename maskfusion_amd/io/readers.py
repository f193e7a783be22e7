"""Log readers with the reference's formats and conventions.

Replaces (reference, relative to /root/reference):
  KlgLogReader    GUI/Tools/KlgLogReader.cpp:22-89      `.klg`: int32 numFrames, then per frame int64 timestamp, int32 depthSize,
                                                        int32 imageSize, depth (uint16 mm, raw or zlib), image (raw RGB or JPEG)
  ImageLogReader  GUI/Tools/ImageLogReader.cpp:62-320   directory of Color####.{jpg,png,ppm}, Depth####.{png,exr}, Mask####.{png,pgm}
                                                        (+ Mask####.txt: class ids, optional boxes), optional calibration.txt
  loadCalibration GUI/MainController.cpp:346-383        "fx fy cx cy [w h]" text file

Frames come out as the library's inputs: rgb uint8 [H,W,3] (R,G,B), depth float32 metres [H,W], mask uint8 [H,W] or None,
class ids (index 0 = background).  Decoding uses zlib and Pillow on the host; none of this is on the GPU path.
"""
from __future__ import annotations

import io
import os
import struct
import zlib
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class FrameData:  # Core/FrameData.h:25-48
    timestamp: int
    index: int
    rgb: np.ndarray
    depth: np.ndarray
    mask: Optional[np.ndarray] = None
    classIDs: List[int] = field(default_factory=list)
    rois: List[tuple] = field(default_factory=list)

    def flipColors(self):
        self.rgb = np.ascontiguousarray(self.rgb[..., ::-1])


def _pil():
    try:
        from PIL import Image
        return Image
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("compressed images need Pillow on the host") from e


def load_calibration(path: str):
    """(fx, fy, cx, cy, width or None, height or None) -- MainController::loadCalibration."""
    vals = open(path).read().split()
    if len(vals) < 4:
        raise ValueError(f"calibration file {path}: expected 'fx fy cx cy [w h]'")
    fx, fy, cx, cy = (float(v) for v in vals[:4])
    w, h = (int(float(vals[4])), int(float(vals[5]))) if len(vals) >= 6 else (None, None)
    return fx, fy, cx, cy, w, h


class KlgLogReader:
    def __init__(self, file: str, width: int = 640, height: int = 480, flipColors: bool = False):
        if not os.path.exists(file):
            raise FileNotFoundError(file)
        self.file, self.W, self.H, self.flip = file, width, height, flipColors
        self.fp = open(file, "rb")
        head = self.fp.read(4)
        if len(head) != 4:
            raise ValueError("Could not open log-file: " + file)   # KlgLogReader.cpp:29
        self.numFrames = struct.unpack("<i", head)[0]
        self.currentFrame = 0

    def getNumFrames(self) -> int:
        return self.numFrames

    def hasMore(self) -> bool:  # KlgLogReader.cpp:118 (sic: the last frame is never delivered)
        return self.currentFrame + 1 < self.numFrames

    def getNext(self) -> FrameData:
        hdr = self.fp.read(16)
        if len(hdr) != 16:
            raise EOFError("truncated .klg frame header")
        ts, dsz, isz = struct.unpack("<qii", hdr)
        P = self.W * self.H
        dbuf = self.fp.read(dsz)
        ibuf = self.fp.read(isz) if isz > 0 else b""
        if len(dbuf) != dsz or len(ibuf) != max(isz, 0):
            raise EOFError("truncated .klg frame")
        if dsz != P * 2:
            dbuf = zlib.decompress(dbuf)
        depth = np.frombuffer(dbuf, np.uint16, P).reshape(self.H, self.W).astype(np.float32) * np.float32(0.001)
        if isz <= 0:
            rgb = np.zeros((self.H, self.W, 3), np.uint8)
        elif isz == P * 3:
            rgb = np.frombuffer(ibuf, np.uint8, P * 3).reshape(self.H, self.W, 3).copy()
        else:
            # JPEG: GUI/Tools/JPEGLoader.h decodes with libjpeg's defaults (ISLOW DCT, fancy upsampling, JCS_RGB) and stores every pixel with
            # its first and third channel SWAPPED (`rgb[2] = t0; rgb[1] = t1; rgb[0] = t2`, :73-79: the loggers of this family compress
            # OpenCV's BGR images, the swap gives RGB back).  Pillow is the same libjpeg-turbo decoder; the swap is reproduced here.  Pinned
            # byte for byte against the reference's loader compiled from its text (tests/test_io_pin.py, round 4 -- round 3 returned the
            # decoder's channel order, i.e. the reverse of what upstream hands to processFrame).
            rgb = np.asarray(_pil().open(io.BytesIO(ibuf)).convert("RGB"), np.uint8)[..., ::-1]
            if rgb.shape[:2] != (self.H, self.W):
                raise ValueError("JPEG frame size does not match the configured resolution")
        f = FrameData(timestamp=ts, index=self.currentFrame, rgb=np.ascontiguousarray(rgb), depth=depth)
        if self.flip:
            f.flipColors()
        self.currentFrame += 1
        return f

    def close(self):
        self.fp.close()

    def __iter__(self):
        while self.hasMore():   # MainController::run: "if (logReader->hasMore()) getNext()" -- upstream never delivers the last frame
            yield self.getNext()


def _count_files(path, prefix, exts):
    n, ext = 0, ""
    for name in sorted(os.listdir(path)):
        stem, e = os.path.splitext(name)
        e = e.lower()
        if stem.startswith(prefix) and e in exts and os.path.isfile(os.path.join(path, name)):
            if ext == "":
                ext = e
            elif ext != e:
                raise ValueError(f"Error: Files in the dataset ( {path}, {prefix}) are required to have the same extension.")
            n += 1
    return n, ext


class ImageLogReader:
    rateHz = 24.0  # ImageLogReader.h:96

    def __init__(self, colorDirectory: str, depthDirectory: str = "", maskDirectory: str = "", indexWidth: int = 4,
                 colorPrefix: str = "", depthPrefix: str = "", maskPrefix: str = "", flipColors: bool = False, maxMasks: int = -1):
        cd = colorDirectory if colorDirectory.endswith(os.sep) else colorDirectory + os.sep
        dd = (depthDirectory or cd)
        md = (maskDirectory or cd)
        dd = dd if dd.endswith(os.sep) else dd + os.sep
        md = md if md.endswith(os.sep) else md + os.sep
        if (dd == cd or md == cd or md == dd) and (depthPrefix == colorPrefix == maskPrefix):
            colorPrefix, depthPrefix, maskPrefix = "Color", "Depth", "Mask"      # ImageLogReader.cpp:78-83
        self.cd, self.dd, self.md = cd, dd, md
        self.cp, self.dp, self.mp, self.indexW, self.flip = colorPrefix, depthPrefix, maskPrefix, indexWidth, flipColors
        nc, self.cext = _count_files(cd, colorPrefix, (".jpg", ".png", ".ppm"))
        nd, self.dext = _count_files(dd, depthPrefix, (".exr", ".png"))
        nm, self.mext = _count_files(md, maskPrefix, (".png", ".pgm"))
        self.hasMasksGT = nm > 0
        self.maxMasks = nm if maxMasks < 0 else maxMasks
        if nc != nd:
            raise ValueError("Error: Number of RGB-frames != Depth-frames!")
        if self.hasMasksGT and nc != nm:
            raise ValueError("Error: Number of RGB-frames != Mask-frames!")
        self.numFrames = nc
        for idx in range(2):
            if os.path.exists(f"{cd}{colorPrefix}{idx:0{indexWidth}d}{self.cext}"):
                self.startIndex = idx
                break
        else:
            raise ValueError("Error: Could not find start index.")
        cal = os.path.join(cd, "calibration.txt")
        self.calibrationFile = cal if os.path.exists(cal) else None
        self.currentFrame = -1

    def getNumFrames(self) -> int:
        return self.numFrames

    def hasMore(self) -> bool:
        return self.currentFrame + 1 < self.numFrames

    @staticmethod
    def loadMaskIDs(path):  # ImageLogReader.cpp:290-308
        lines = open(path).read().split("\n")
        ids = [0] + [int(t) for t in lines[0].split(" ") if t]
        rois = []
        for line in lines[1:]:
            if not line.strip():
                continue
            a, b, c, d = (int(t) for t in line.split()[:4])
            rois.append((b, a, d - b, c - a))   # cv::Rect(b, a, d - b, c - a)
        if rois and len(rois) != len(ids) - 1:
            raise ValueError("Bounding-boxes provided, but number does not match class ids.")
        return ids, rois

    def load(self, index: int) -> FrameData:
        Image = _pil()
        s = f"{index + self.startIndex:0{self.indexW}d}"
        dpath, cpath = f"{self.dd}{self.dp}{s}{self.dext}", f"{self.cd}{self.cp}{s}{self.cext}"
        for p, what in ((dpath, "depth"), (cpath, "rgb")):
            if not os.path.exists(p):
                raise FileNotFoundError(f"Could not find {what}-image file: {p}")
        rgb = np.asarray(Image.open(cpath).convert("RGB"), np.uint8)   # imread (BGR) + flipColors() == RGB
        d = dimg = None
        if self.dext == ".exr":
            from .exr import read_exr_depth
            depth = read_exr_depth(dpath)        # CV_32FC1, or channel 0 (blue) of CV_32FC3: ImageLogReader.cpp:250-260
        else:
            dimg = Image.open(dpath)
            d = np.asarray(dimg)
        if d is None:
            pass
        elif d.dtype == np.uint16 or d.dtype == np.int32 or dimg.mode in ("I;16", "I"):
            depth = d.astype(np.float32) * np.float32(0.001)            # CV_16UC1 branch, ImageLogReader.cpp:262-268
        elif d.dtype == np.float32:
            depth = d if d.ndim == 2 else d[..., 0]
        else:
            raise ValueError("Unsupported depth-files: " + str(d.dtype))
        f = FrameData(timestamp=int(index * 1000.0 / self.rateHz), index=index, rgb=np.ascontiguousarray(rgb),
                      depth=np.ascontiguousarray(depth, np.float32))
        if self.hasMasksGT:
            mpath = f"{self.md}{self.mp}{s}"
            if not os.path.exists(mpath + self.mext):
                raise FileNotFoundError("Could not find mask-image file: " + mpath + self.mext)
            if os.path.exists(mpath + ".txt"):
                f.classIDs, f.rois = self.loadMaskIDs(mpath + ".txt")
            if index < self.maxMasks:
                m = np.asarray(Image.open(mpath + self.mext).convert("L"), np.uint8)
                if m.shape != rgb.shape[:2]:
                    raise ValueError("Could not read mask-image file.")
                f.mask = np.ascontiguousarray(m)
        if self.flip:
            f.flipColors()
        return f

    def getNext(self) -> FrameData:
        self.currentFrame += 1
        return self.load(self.currentFrame)

    def __iter__(self):
        while self.hasMore():
            yield self.getNext()


def open_log(path: str, width=640, height=480, **kw):
    """`-l file.klg` or `-dir directory` (MainController.cpp:137-180)."""
    if os.path.isdir(path):
        return ImageLogReader(path, **kw)
    return KlgLogReader(path, width, height, flipColors=kw.get("flipColors", False))
