"""Input formats of the reference's log readers: `.klg` logs and image directories."""
from .readers import FrameData, ImageLogReader, KlgLogReader, load_calibration, open_log  # noqa: F401
from .writers import write_image_dir, write_klg  # noqa: F401
from .exr import read_exr, read_exr_depth, write_exr  # noqa: F401
