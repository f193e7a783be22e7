"""SURVEY.md row 8f-1 (input formats) pinned to the reference's own text.

tests/golden/io/ holds a tiny `.klg` log and three `Mask####.txt` descriptors (the latter written by the reference's Python writer,
Core/Segmentation/MaskRCNN/helpers.py:101-113) together with what the reference's compiled parse code returns for them
(GUI/Tools/KlgLogReader.cpp:22-89 and ImageLogReader::loadMaskIDs, compiled from their text by oracle/build_io.py;
tests/golden/make_io_golden.py generated the files).  maskfusion_amd/io/readers.py must return the same frames BYTE FOR BYTE: time stamps,
depth (float32 metres, raw and zlib-deflated), colour (raw, absent, flipped), which frames are delivered (upstream's `hasMore()` loop never
delivers the last one), class ids and boxes.  Where /root/reference is present (the build container) the compiled reference reader is also
run live against the same files."""
import os

import numpy as np
import pytest

from maskfusion_amd.io import readers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io")
W, H = 16, 12


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLD, "io_vectors.npz"))


@pytest.mark.parametrize("flip", [0, 1])
def test_klg_frames_are_the_reference_readers(vec, flip):
    r = readers.KlgLogReader(os.path.join(GOLD, "tiny.klg"), W, H, flipColors=bool(flip))
    assert r.getNumFrames() == int(vec[f"klg_num_frames_flip{flip}"]) == 4
    frames = list(r)                                    # MainController::run's loop: `if (hasMore()) getNext()`
    assert len(frames) == int(vec[f"klg_delivered_flip{flip}"]) == 3
    for i, f in enumerate(frames):
        assert f.timestamp == int(vec[f"klg_ts_{i}_flip{flip}"])
        assert f.depth.dtype == np.float32 and f.depth.tobytes() == vec[f"klg_depth_{i}_flip{flip}"].tobytes(), i   # bit-exact metres
        assert f.rgb.dtype == np.uint8 and np.array_equal(f.rgb, vec[f"klg_rgb_{i}_flip{flip}"]), i
    assert not frames[2].rgb.any()                       # imageSize 0: a black image
    assert np.array_equal(frames[0].rgb[..., ::-1], list(readers.KlgLogReader(os.path.join(GOLD, "tiny.klg"), W, H, flipColors=not flip))[0].rgb)
    r.close()


@pytest.mark.parametrize("flip", [0, 1])
def test_jpeg_klg_frames_are_the_reference_readers(vec, flip):
    """colour stored JPEG-compressed (KlgLogReader.cpp:74-77 -> GUI/Tools/JPEGLoader.h): the golden frames come from the reference's loader,
    compiled from its text against the installed libjpeg (oracle/build_io.py, oracle/io_shim/jpeglib.h); readers.py decodes with Pillow and
    applies the loader's channel swap -- byte for byte, 4:2:0 / 4:4:4 / 4:2:2 sampling"""
    JW, JH = 32, 24
    r = readers.KlgLogReader(os.path.join(GOLD, "jpeg.klg"), JW, JH, flipColors=bool(flip))
    assert r.getNumFrames() == int(vec[f"jpeg_num_frames_flip{flip}"]) == 4
    frames = list(r)
    assert len(frames) == int(vec[f"jpeg_delivered_flip{flip}"]) == 3
    for i, f in enumerate(frames):
        assert f.timestamp == int(vec[f"jpeg_ts_{i}_flip{flip}"])
        assert f.depth.tobytes() == vec[f"jpeg_depth_{i}_flip{flip}"].tobytes(), i
        assert f.rgb.dtype == np.uint8 and f.rgb.shape == (JH, JW, 3)
        assert np.array_equal(f.rgb, vec[f"jpeg_rgb_{i}_flip{flip}"]), (i, int(np.abs(f.rgb.astype(int) - vec[f"jpeg_rgb_{i}_flip{flip}"].astype(int)).max()))
    # the decoded frame is a picture of what was encoded, in the channel order upstream hands on: the 200 / 40 / 90 patch of the source image
    # comes back swapped (90 / 40 / 200) without -f and as encoded with it
    patch = frames[0].rgb[6:11, 10:18].reshape(-1, 3).mean(0)
    want = np.array([200, 40, 90] if flip else [90, 40, 200], float)
    assert np.abs(patch - want).max() < 12
    r.close()


@pytest.mark.parametrize("i", [0, 1, 2])
def test_mask_descriptor_is_parsed_like_the_reference(vec, i):
    ids, rois = readers.ImageLogReader.loadMaskIDs(os.path.join(GOLD, f"Mask{i:04d}.txt"))
    assert ids == vec[f"mask_ids_{i}"].tolist()
    assert [tuple(r) for r in rois] == [tuple(int(v) for v in r) for r in vec[f"mask_rois_{i}"]]
    assert ids[0] == 0                                   # mask value 0 is always background


def test_descriptor_with_mismatched_boxes_is_rejected(tmp_path):
    p = tmp_path / "Mask0000.txt"
    p.write_text("41 57\n1 2 3 4")
    with pytest.raises(ValueError):
        readers.ImageLogReader.loadMaskIDs(str(p))
    from oracle import mfio
    if mfio.available():
        with pytest.raises(ValueError):
            mfio.load_mask_ids(str(p))                   # the reference throws std::invalid_argument too


def test_live_reference_reader_agrees_with_the_golden_file(vec):
    from oracle import mfio
    if not mfio.available():
        pytest.skip("oracle/_ref/libmf_io.so absent and no /root/reference to build it")
    for flip in (0, 1):
        n, out = mfio.read_klg(os.path.join(GOLD, "tiny.klg"), W, H, bool(flip))
        assert n == 4 and len(out) == 3
        for i, (ts, depth, rgb) in enumerate(out):
            assert ts == int(vec[f"klg_ts_{i}_flip{flip}"])
            assert depth.tobytes() == vec[f"klg_depth_{i}_flip{flip}"].tobytes() and np.array_equal(rgb, vec[f"klg_rgb_{i}_flip{flip}"])
    for i in range(3):
        ids, rois = mfio.load_mask_ids(os.path.join(GOLD, f"Mask{i:04d}.txt"))
        assert ids == vec[f"mask_ids_{i}"].tolist() and rois == [tuple(int(v) for v in r) for r in vec[f"mask_rois_{i}"]]
