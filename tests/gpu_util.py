"""Helpers shared by the -m gpu parity tests: torch is used only to hold device memory."""
import os

import numpy as np
import torch

# MF_EMU=1: the tests drive tests/_emu/libmaskfusion_emu.so (the product's kernels executed on the CPU, tests/hipcpu) instead of the GPU
# library -- a logic check for machines without a GPU, selected explicitly and never by default.  "Device" memory is host memory then.
EMU = os.environ.get("MF_EMU") == "1"
DEVICE = "cpu" if EMU else "cuda"


def dev(a: np.ndarray) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.clone() if EMU else t.cuda()


def empty(shape, dtype=torch.float32) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=DEVICE)


def host(t: torch.Tensor) -> np.ndarray:
    if not EMU:
        torch.cuda.synchronize()
    return t.cpu().numpy()


def nan_equal_close(a, b, rtol, atol):
    """max abs error over entries finite in both; NaN patterns must agree."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    assert (na == nb).all(), f"NaN pattern differs at {int((na != nb).sum())} entries"
    ok = ~na
    err = np.abs(a[ok] - b[ok])
    tol = atol + rtol * np.abs(b[ok])
    bad = err > tol
    return float(err.max()) if err.size else 0.0, int(bad.sum())


def scene_frames(n, W=640, H=480, noise=False, n_objects=0):
    from maskfusion_amd import synth
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=noise, n_objects=n_objects)
    return st, [st.frame(k) for k in range(n)]
