"""maskfusion_amd -- MI355X-native hot path of MaskFusion::processFrame behind a C ABI.

The product is `libmaskfusion_amd.so` (hand-written HIP for gfx950, see csrc/ and include/maskfusion_amd.h).
This package is the thin host-side mirror of the reference's MaskFusion / Model interface over that C ABI
(ctypes), used by the tests and by bench.py.  There is no CPU fallback: importing `lib` without the built
extension, or creating a context without a GPU, fails loudly.
"""
from .lib import load, MFError, Config  # noqa: F401
from .api import MaskFusion, Model  # noqa: F401
