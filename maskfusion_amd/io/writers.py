"""Writers of the reference's INPUT formats (test fixtures, dataset conversion); the library's OUTPUT formats
(`poses-<id>.txt`, `cloud-<id>.ply`) are written by the C ABI (`mf_export_poses`, `mf_save_ply`)."""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np


def write_klg(path, frames, compress_depth=True):
    """frames: iterable of (timestamp_us, rgb uint8 [H,W,3], depth float32 metres [H,W]); layout of KlgLogReader.cpp:55-89."""
    frames = list(frames)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for ts, rgb, depth in frames:
            d16 = np.clip(np.rint(np.asarray(depth, np.float64) * 1000.0), 0, 65535).astype(np.uint16).tobytes()
            if compress_depth:
                d16 = zlib.compress(d16)
            img = np.ascontiguousarray(rgb, np.uint8).tobytes()
            f.write(struct.pack("<qii", int(ts), len(d16), len(img)))
            f.write(d16)
            f.write(img)


def write_image_dir(path, frames, masks=None, class_ids=None, calibration=None, index_width=4, start_index=0):
    """Color####.png / Depth####.png (16-bit millimetres) / Mask####.png + Mask####.txt, as ImageLogReader expects."""
    from PIL import Image
    os.makedirs(path, exist_ok=True)
    for i, (rgb, depth) in enumerate(frames):
        s = f"{i + start_index:0{index_width}d}"
        Image.fromarray(np.ascontiguousarray(rgb, np.uint8), "RGB").save(os.path.join(path, f"Color{s}.png"))
        d16 = np.clip(np.rint(np.asarray(depth, np.float64) * 1000.0), 0, 65535).astype(np.uint16)
        Image.fromarray(d16).save(os.path.join(path, f"Depth{s}.png"))
        if masks is not None:
            Image.fromarray(np.ascontiguousarray(masks[i], np.uint8), "L").save(os.path.join(path, f"Mask{s}.png"))
            if class_ids is not None:
                with open(os.path.join(path, f"Mask{s}.txt"), "w") as f:
                    f.write(" ".join(str(c) for c in class_ids[i][1:]) + "\n")   # index 0 (background) is implicit
    if calibration is not None:
        with open(os.path.join(path, "calibration.txt"), "w") as f:
            f.write(" ".join(repr(float(v)) for v in calibration[:4]))
            if len(calibration) >= 6:
                f.write(f" {int(calibration[4])} {int(calibration[5])}")
            f.write("\n")
