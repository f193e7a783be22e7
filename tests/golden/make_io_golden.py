"""Generates tests/golden/io/: a tiny .klg log and Mask####.txt descriptors together with what the REFERENCE's own code makes of them.

  tiny.klg           written here from the format KlgLogReader.cpp:22-89 parses (there is no .klg writer in the reference repository): four
                     16 x 12 frames -- raw depth + raw colour, zlib depth + raw colour, raw depth + no colour (imageSize 0), and a last
                     frame that upstream's `hasMore()` loop never delivers
  Mask0000.txt ...   written by the reference's PYTHON writer: save_id_image() of Core/Segmentation/MaskRCNN/helpers.py:101-113, cut out of
                     that file and executed here (the module itself imports Mask R-CNN)
  io_vectors.npz     the frames as the reference's compiled KlgLogReader (oracle/_ref/libmf_io.so, oracle/build_io.py) returns them, plain and
                     with flipColors, and the class ids / boxes its ImageLogReader::loadMaskIDs parses out of the descriptors

Needs /root/reference (the build container).  tests/test_io_pin.py holds maskfusion_amd/io/readers.py to these files on any machine."""
import ast
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "io")
W, H = 16, 12


def reference_save_id_image():
    """save_id_image as the reference defines it, with PIL's Image replaced by a no-op (only the .txt half is wanted)"""
    src = open("/root/reference/Core/Segmentation/MaskRCNN/helpers.py").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "save_id_image")
    text = "\n".join(src.split("\n")[fn.lineno - 1:fn.end_lineno])

    class _Img:
        @staticmethod
        def fromarray(a):
            class _S:
                def save(self, p):
                    pass
            return _S()
    ns = {"os": os, "Image": _Img}
    exec(compile(text, "helpers.py:save_id_image", "exec"), ns)
    return ns["save_id_image"]


def main():
    from oracle import mfio
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    frames = []
    for k in range(4):
        depth = rng.integers(0, 6000, size=(H, W), dtype=np.uint16)
        depth[rng.random((H, W)) < 0.1] = 0
        if k == 1:   # the frame that is stored deflated must deflate: the reference reads the compressed bytes into a W*H*2-byte buffer
            depth = (1000 + 3 * np.add.outer(np.arange(H), np.arange(W))).astype(np.uint16)
        rgb = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        frames.append((1000000 * (k + 1) + 37 * k, depth, rgb))
    with open(os.path.join(OUT, "tiny.klg"), "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for k, (ts, depth, rgb) in enumerate(frames):
            d = depth.tobytes()
            if k == 1:
                d = zlib.compress(d)                   # depthSize != W*H*2 -> the reader inflates it
                assert len(d) < W * H * 2
            c = b"" if k == 2 else rgb.tobytes()       # imageSize 0 -> the reader delivers a black image
            f.write(struct.pack("<qii", ts, len(d), len(c)))
            f.write(d)
            f.write(c)
    vec = {}
    for flip in (0, 1):
        n, out = mfio.read_klg(os.path.join(OUT, "tiny.klg"), W, H, bool(flip))
        vec[f"klg_num_frames_flip{flip}"] = np.int32(n)
        vec[f"klg_delivered_flip{flip}"] = np.int32(len(out))
        for i, (ts, depth, rgb) in enumerate(out):
            vec[f"klg_ts_{i}_flip{flip}"] = np.int64(ts)
            vec[f"klg_depth_{i}_flip{flip}"] = depth
            vec[f"klg_rgb_{i}_flip{flip}"] = rgb
    # jpeg.klg: colour stored JPEG-compressed (imageSize != W*H*3 -> JPEGLoader::readData, KlgLogReader.cpp:74-77), what every real .klg of the
    # ElasticFusion family carries.  Baseline JFIF files without COM / EXIF segments (the reference's source manager leaves skip_input_data
    # unset), written with Pillow: 4:2:0 at quality 90, 4:4:4 at 75, 4:2:2 at 95, and a last frame that is never delivered.
    import io as _io
    from PIL import Image
    JW, JH = 32, 24
    yy, xx = np.mgrid[0:JH, 0:JW]
    base = np.stack([(xx * 8) % 256, (yy * 10) % 256, ((xx + yy) * 5) % 256], -1).astype(np.uint8)
    base[5:12, 8:20] = [200, 40, 90]
    jframes = []
    for k, (q, ss) in enumerate([(90, 2), (75, 0), (95, 1), (50, 2)]):
        b = _io.BytesIO()
        Image.fromarray(base if k % 2 == 0 else np.ascontiguousarray(base[::-1])).save(b, format="JPEG", quality=q, subsampling=ss)
        jframes.append((2000000 + 41 * k, (900 + 7 * k + np.add.outer(np.arange(JH), np.arange(JW))).astype(np.uint16), b.getvalue()))
    with open(os.path.join(OUT, "jpeg.klg"), "wb") as f:
        f.write(struct.pack("<i", len(jframes)))
        for ts, depth, c in jframes:
            assert len(c) != JW * JH * 3
            f.write(struct.pack("<qii", ts, depth.nbytes, len(c)))
            f.write(depth.tobytes())
            f.write(c)
    for flip in (0, 1):
        n, out = mfio.read_klg(os.path.join(OUT, "jpeg.klg"), JW, JH, bool(flip))
        vec[f"jpeg_num_frames_flip{flip}"] = np.int32(n)
        vec[f"jpeg_delivered_flip{flip}"] = np.int32(len(out))
        for i, (ts, depth, rgb) in enumerate(out):
            vec[f"jpeg_ts_{i}_flip{flip}"] = np.int64(ts)
            vec[f"jpeg_depth_{i}_flip{flip}"] = depth
            vec[f"jpeg_rgb_{i}_flip{flip}"] = rgb
    save = reference_save_id_image()
    cases = [([41, 57, 1], [[10, 20, 110, 220], [5, 6, 7, 9], [0, 0, 480, 640]]),   # class ids + one box each (y1 x1 y2 x2, Mask R-CNN order)
             ([3], []),                                                             # ids only
             ([], [])]                                                              # no detections: an empty first line
    for i, (ids, rois) in enumerate(cases):
        base = f"Mask{i:04d}"
        save(np.zeros((H, W), np.uint8), OUT, base, exported_class_ids=ids, export_classes=True, exported_rois=rois)
        got_ids, got_rois = mfio.load_mask_ids(os.path.join(OUT, base + ".txt"))
        vec[f"mask_ids_{i}"] = np.array(got_ids, np.int32)
        vec[f"mask_rois_{i}"] = np.array(got_rois, np.int32).reshape(-1, 4)
    np.savez_compressed(os.path.join(OUT, "io_vectors.npz"), **vec)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
