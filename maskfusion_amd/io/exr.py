"""Minimal OpenEXR scan-line reader / writer (single part, NONE / ZIPS / ZIP compression, HALF / FLOAT channels): enough for
the `Depth####.exr` files of the reference's image-directory datasets (GUI/Tools/ImageLogReader.cpp:243-261 reads them with
cv::imread(IMREAD_UNCHANGED): a one-channel file is the depth in metres, a three-channel file carries it in every channel
and OpenCV's channel 0 -- blue -- is taken).  Host-side Python; no GPU involved."""
from __future__ import annotations

import struct
import zlib

import numpy as np

MAGIC = 20000630
_LINES = {0: 1, 2: 1, 3: 16}     # scan lines per chunk: NO_COMPRESSION, ZIPS, ZIP
_PT = {1: np.float16, 2: np.float32}


def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _unzip(data, expected):
    if len(data) == expected:            # stored raw when deflate did not help
        return data
    t = np.frombuffer(zlib.decompress(data), np.uint8).astype(np.int32)
    if t.size != expected:
        raise ValueError("EXR: chunk inflates to an unexpected size")
    # predictor: t[i] = t[i-1] + t[i] - 128 (mod 256) == 128-offset prefix sum
    t[1:] -= 128
    t = (np.cumsum(t) & 255).astype(np.uint8)
    out = np.empty(expected, np.uint8)
    half = (expected + 1) // 2
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _zip(raw):
    b = np.frombuffer(raw, np.uint8)
    t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
    comp = zlib.compress(d.astype(np.uint8).tobytes())
    return comp if len(comp) < len(raw) else raw


def read_exr(path):
    """-> dict channel name -> float32 array [H, W]."""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != MAGIC:
        raise ValueError("not an OpenEXR file: " + path)
    if version & 0x1E00:
        raise ValueError("EXR: tiled / deep / multi-part files are not supported")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        size, = struct.unpack_from("<i", buf, pos)
        pos += 4
        attrs[name] = (typ, buf[pos:pos + size])
        pos += size
    pos += 1
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise ValueError(f"EXR: compression {comp} is not supported (NONE, ZIPS, ZIP only)")
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = xmax - xmin + 1, ymax - ymin + 1
    chans, cb, cp = [], attrs["channels"][1], 0
    while cb[cp] != 0:
        name, cp = _cstr(cb, cp)
        ptype, _plin, xs, ys = struct.unpack_from("<iB3xii", cb, cp)
        cp += 16
        if ptype not in _PT or xs != 1 or ys != 1:
            raise ValueError("EXR: only HALF / FLOAT channels without sub-sampling are supported")
        chans.append((name, _PT[ptype]))
    lines = _LINES[comp]
    nblocks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f"<{nblocks}Q", buf, pos)
    out = {n: np.zeros((H, W), np.float32) for n, _ in chans}
    row_bytes = sum(W * np.dtype(t).itemsize for _, t in chans)
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        nl = min(lines, ymin + H - y)
        raw = buf[off + 8:off + 8 + size]
        data = raw if comp == 0 else _unzip(raw, nl * row_bytes)
        p = 0
        for l in range(nl):
            for name, t in chans:
                n = W * np.dtype(t).itemsize
                out[name][y - ymin + l] = np.frombuffer(data, t, W, p).astype(np.float32)
                p += n
    return out


def read_exr_depth(path):
    """The channel cv::imread + ImageLogReader end up using: the only channel, or blue of a colour file."""
    ch = read_exr(path)
    if len(ch) == 1:
        return next(iter(ch.values()))
    for name in ("B", "Z", "Y", "R"):
        if name in ch:
            return ch[name]
    return ch[sorted(ch)[0]]


def write_exr(path, channels, compression=3, half=False):
    """channels: dict name -> [H, W] array.  Scan-line file, increasing y."""
    names = sorted(channels)
    H, W = channels[names[0]].shape
    t = np.float16 if half else np.float32
    ptype = 1 if half else 2
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val

    head = struct.pack("<ii", MAGIC, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression])) + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = _LINES[compression]
    chunks = []
    for y0 in range(0, H, lines):
        raw = b"".join(np.ascontiguousarray(channels[n][y], t).tobytes() for y in range(y0, min(H, y0 + lines)) for n in names)
        data = raw if compression == 0 else _zip(raw)
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    table_pos = len(head)
    offs, cur = [], table_pos + 8 * len(chunks)
    for c in chunks:
        offs.append(cur)
        cur += len(c)
    with open(path, "wb") as f:
        f.write(head + struct.pack(f"<{len(offs)}Q", *offs) + b"".join(chunks))
