"""Readers of the reference's input formats and the flag handling of the headless driver.  CPU only
(the GPU run of the driver is tests/test_gpu_api.py::test_cli_runs_image_directory)."""
import os

import numpy as np
import pytest

from maskfusion_amd import cli
from maskfusion_amd.io import ImageLogReader, KlgLogReader, load_calibration, open_log, write_image_dir, write_klg

W, H = 64, 48


def _frames(n=4, seed=0):
    rng = np.random.default_rng(seed)
    return [(33333 * k, rng.integers(0, 255, (H, W, 3), dtype=np.uint8),
             np.round(rng.uniform(0.5, 4, (H, W)), 3).astype(np.float32)) for k in range(n)]


@pytest.mark.parametrize("compress", [True, False])
def test_klg_round_trip(tmp_path, compress):
    fr = _frames()
    p = str(tmp_path / "a.klg")
    write_klg(p, fr, compress_depth=compress)
    r = KlgLogReader(p, W, H)
    assert r.getNumFrames() == 4
    out = list(r)
    assert len(out) == 3          # hasMore() is "currentFrame + 1 < numFrames": upstream never delivers the last frame
    for f, (ts, rgb, depth) in zip(out, fr):
        assert f.timestamp == ts and np.array_equal(f.rgb, rgb)
        assert np.array_equal(f.depth, (np.rint(depth.astype(np.float64) * 1000).astype(np.uint16)).astype(np.float32) * np.float32(0.001))
    r2 = KlgLogReader(p, W, H, flipColors=True)
    assert np.array_equal(r2.getNext().rgb, fr[0][1][..., ::-1])
    with pytest.raises(FileNotFoundError):
        KlgLogReader(str(tmp_path / "missing.klg"))


def test_klg_jpeg_frames(tmp_path):
    import io
    import struct
    import zlib
    from PIL import Image
    rgb = np.zeros((H, W, 3), np.uint8); rgb[:, :, 0] = 200; rgb[10:20, 10:30, 1] = 180
    buf = io.BytesIO(); Image.fromarray(rgb).save(buf, "JPEG", quality=95)
    d16 = zlib.compress(np.full((H, W), 1500, np.uint16).tobytes())
    p = str(tmp_path / "j.klg")
    with open(p, "wb") as f:
        f.write(struct.pack("<i", 2))
        for k in range(2):
            f.write(struct.pack("<qii", k, len(d16), len(buf.getvalue()))); f.write(d16); f.write(buf.getvalue())
    fr = KlgLogReader(p, W, H).getNext()
    # GUI/Tools/JPEGLoader.h swaps the first and third channel of what libjpeg decodes (the loggers compress BGR images): without -f the frame
    # comes back channel-reversed, with it as encoded (pinned against the compiled loader in tests/test_io_pin.py)
    assert np.allclose(fr.depth, 1.5) and np.abs(fr.rgb.astype(int) - rgb[..., ::-1].astype(int)).mean() < 4
    fr = KlgLogReader(p, W, H, flipColors=True).getNext()
    assert np.abs(fr.rgb.astype(int) - rgb.astype(int)).mean() < 4


def test_image_directory(tmp_path):
    fr = _frames()
    masks = [np.zeros((H, W), np.uint8) for _ in fr]
    masks[1][5:20, 8:30] = 1
    d = str(tmp_path / "seq") + os.sep
    write_image_dir(d, [(a, b) for _, a, b in fr], masks=masks, class_ids=[[0, 41]] * 4, calibration=(50.5, 51.5, 32, 24, W, H))
    r = ImageLogReader(d)
    assert r.getNumFrames() == 4 and r.startIndex == 0 and r.hasMasksGT and r.calibrationFile.endswith("calibration.txt")
    out = list(r)
    assert len(out) == 4
    assert np.array_equal(out[2].rgb, fr[2][1]) and np.abs(out[2].depth - fr[2][2]).max() < 6e-4
    assert out[1].classIDs == [0, 41] and np.array_equal(out[1].mask, masks[1])
    assert out[3].timestamp == int(3 * 1000.0 / 24.0)        # ImageLogReader.cpp:280, rateHz = 24
    assert load_calibration(r.calibrationFile) == (50.5, 51.5, 32.0, 24.0, W, H)
    assert isinstance(open_log(d), ImageLogReader)
    # start index 1 and a missing depth image
    d2 = str(tmp_path / "seq1") + os.sep
    write_image_dir(d2, [(a, b) for _, a, b in fr], start_index=1)
    r2 = ImageLogReader(d2)
    assert r2.startIndex == 1 and not r2.hasMasksGT and np.array_equal(r2.getNext().rgb, fr[0][1])
    os.remove(d2 + "Depth0002.png")
    with pytest.raises(ValueError):
        ImageLogReader(d2)


def test_mask_description_file(tmp_path):
    p = tmp_path / "Mask0000.txt"
    p.write_text("41 64\n10 20 110 220\n5 6 50 60\n")
    ids, rois = ImageLogReader.loadMaskIDs(str(p))
    assert ids == [0, 41, 64] and rois == [(20, 10, 200, 100), (6, 5, 54, 45)]   # cv::Rect(b, a, d - b, c - a)
    p.write_text("41 64\n10 20 110 220\n")
    with pytest.raises(ValueError):
        ImageLogReader.loadMaskIDs(str(p))


def test_cli_flags_and_defaults():
    f = cli.parse(["-dir", "seq/", "-tum3", "-run", "-q", "-ep", "-em", "-static", "-i", "100", "-nso", "-exportdir", "out"])
    st = cli.settings(f)
    assert (st["W"], st["H"], st["fx"], st["fy"], st["cx"], st["cy"]) == (640, 480, 535.4, 539.2, 320.1, 247.6)
    assert st["icpWeight"] == 100.0 and not st["so3"] and not st["multi"] and st["depthCutoff"] == 4.0
    d = cli.settings(cli.parse(["-l", "x.klg"]))
    # the values the GUI pushes into the core every frame (GUI.h:188-196,347; MainController.cpp:215-228)
    assert (d["depthCutoff"], d["icpWeight"], d["outlierCoefficient"], d["confGlobal"], d["confObject"], d["modelSpawnOffset"]) == \
        (4.0, 20.0, 0.1, 10.0, 0.01, 22)
    assert d["so3"] and d["multi"] and d["timeDelta"] == (2 ** 31 - 1) // 2 and (d["fx"], d["cx"]) == (528.0, 320.0)
    # SURVEY.md 2.4, "Effective (GUI/CLI) default" column, row by row (GUI/Tools/GUI.h:342-347,367-374; MainController.cpp:215-246)
    assert d["trackAllModels"] is False
    assert d["mf"] == dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphEdgeRadius=1,
                           mfMorphMaskIterations=0, mfMorphMaskRadius=2, newModelMinRelativeSize=0.015, newModelMaxRelativeSize=0.4)
    assert (d["frameQueueRequested"], d["frameQueue"], d["preallocate"], d["start"], d["end"]) == (30, 0, 0, 1, 65535)
    assert not d["fastOdom"] and not d["rgbOnly"] and not d["exportSegmentation"]
    e = cli.settings(cli.parse(["-l", "x.klg", "-segMinNew", "0.02", "-segMaxNew", "0.5", "-offset", "5", "-a", "3", "-frameQ", "10", "-es",
                                "-thNew", "5", "-method", "maskfusion", "-fo", "-keep"]))
    assert (e["mf"]["newModelMinRelativeSize"], e["mf"]["newModelMaxRelativeSize"], e["modelSpawnOffset"], e["preallocate"]) == (0.02, 0.5, 5, 3)
    assert e["frameQueueRequested"] == 10 and e["frameQueue"] == 0 and e["exportSegmentation"] and e["fastOdom"]
    with pytest.raises(SystemExit):
        cli.settings(cli.parse(["-l", "x.klg", "-method", "cofusion"]))
    with pytest.raises(SystemExit):
        cli.parse(["-bogus"])
    with pytest.raises(SystemExit):
        cli.parse(["-l"])


@pytest.mark.parametrize("compression", [0, 2, 3])
@pytest.mark.parametrize("half", [False, True])
def test_exr_round_trip(tmp_path, compression, half):
    from maskfusion_amd.io import read_exr, read_exr_depth, write_exr
    rng = np.random.default_rng(3)
    d = rng.uniform(0.3, 5, (37, 53)).astype(np.float32)      # odd sizes: the last ZIP chunk is short
    p = str(tmp_path / "d.exr")
    write_exr(p, {"Z": d}, compression, half)
    tol = 4e-3 if half else 0.0
    assert np.abs(read_exr_depth(p) - d).max() <= tol
    write_exr(p, {"R": d, "G": 2 * d, "B": 3 * d}, compression, half)
    ch = read_exr(p)
    assert sorted(ch) == ["B", "G", "R"] and np.abs(ch["G"] - 2 * d).max() <= 2 * tol
    assert np.abs(read_exr_depth(p) - 3 * d).max() <= 3 * tol     # blue = OpenCV channel 0 (ImageLogReader.cpp:252-256)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"not an exr file")
        read_exr(p)


def test_image_directory_with_exr_depth(tmp_path):
    from PIL import Image
    from maskfusion_amd.io import write_exr
    fr = _frames(3)
    d = str(tmp_path / "seq") + os.sep
    os.makedirs(d)
    for i, (_, rgb, depth) in enumerate(fr):
        Image.fromarray(rgb, "RGB").save(f"{d}Color{i:04d}.png")
        write_exr(f"{d}Depth{i:04d}.exr", {"R": depth, "G": depth, "B": depth})
    r = ImageLogReader(d)
    assert r.dext == ".exr"
    out = list(r)
    assert len(out) == 3 and np.array_equal(out[1].depth, fr[1][2]) and np.array_equal(out[1].rgb, fr[1][1])


def test_trackable_classes_from_config_toml(tmp_path):
    """MainController.cpp:273-287: trackable ids = positions of [MaskRCNN].trackable_classes in class_names"""
    cfg = tmp_path / "config.toml"
    cfg.write_text('[MaskRCNN]\nclass_names = ["BG", "person", "bicycle", "teddy bear", "bottle"]\ntrackable_classes = ["teddy bear", "bottle"]\n')
    assert cli.trackable_class_ids(str(cfg)) == [3, 4]
    assert cli.trackable_class_ids(str(tmp_path / "missing.toml")) is None


@pytest.mark.parametrize("shape", [(480, 640), (7, 5), (300, 333)])
def test_png_writer_round_trip(tmp_path, shape):
    """mf_write_png_gray8 (what exportSegmentation writes, Core/MaskFusion.cpp:299-303): any PNG reader gets the same pixels back; rows
    longer than one stored-deflate block (65 535 B) and odd sizes included.  Host code: runs without a GPU."""
    from PIL import Image
    from maskfusion_amd.lib import load
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    path = str(tmp_path / "seg.png")
    assert load().mf_write_png_gray8(path.encode(), img.ctypes.data, shape[1], shape[0]) == 0
    back = np.asarray(Image.open(path))
    assert back.dtype == np.uint8 and np.array_equal(back, img)
    assert load().mf_write_png_gray8(str(tmp_path / "no" / "dir.png").encode(), img.ctypes.data, shape[1], shape[0]) != 0
