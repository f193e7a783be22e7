"""The oracle against the frozen vectors of tests/golden/ (CPU).  The vectors are oracle outputs, not reference outputs (the
reference ships none and cannot run here -- see tests/golden/make_golden.py): this test pins the restatement against
accidental change; integer results must reproduce exactly, floats to rounding of the host libm."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
EXACT = {"gray1", "gray1_l1", "gray0", "dIdx", "dIdy", "rgb_count_sigma", "rgb_corr_valid", "rgb_corr_u0", "rgb_corr_v0",
         "rgb_corr_diff", "pipeline_counts", "seg_binary", "seg_full", "seg_new"}


def test_oracle_reproduces_golden_vectors():
    now = make_golden.build()
    assert set(now) == set(GOLD.files)
    for k in GOLD.files:
        a, b = np.asarray(now[k]), GOLD[k]
        assert a.shape == b.shape, k
        if k in EXACT:
            if k in ("rgb_corr_u0", "rgb_corr_v0", "rgb_corr_diff"):
                v = GOLD["rgb_corr_valid"]
                assert np.array_equal(a[v], b[v]), k
            else:
                assert np.array_equal(a, b), k
        else:
            both_nan = np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64))
            assert np.array_equal(np.isnan(a.astype(np.float64)), np.isnan(b.astype(np.float64))), k
            assert np.allclose(np.where(both_nan, 0, a), np.where(both_nan, 0, b), rtol=2e-6, atol=1e-7), k
