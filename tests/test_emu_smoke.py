"""Kernel LOGIC on every CPU run: a few frames through the product's own kernels executed on the CPU (tests/hipcpu: the .hip sources
compiled with g++ against a HIP stand-in, fibers for threads, 64-wide wavefronts) next to the oracle.  This is test tooling, not a
product path -- it runs in a subprocess so that nothing of it leaks into this process; the product library is the HIP build only, and
the parity claims rest on the -m gpu runs on MI355X (which MF_EMU=1 can rehearse here in full, see tests/gpu_util.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_kernels_track_and_fuse_like_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipcpu", "smoke.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for k, fr in enumerate(res["single"]):                       # geometric tracking: one launch per Gauss-Newton iteration
        assert fr["count"] == fr["ocount"], (k, fr)              # fused surfel count exact, every frame
        assert fr["pose_diff"] < 2e-5, (k, fr)
        assert k == 0 or fr["inliers"] > 10000
    for k, fr in enumerate(res["rgbd"]):                         # ICP + photometric term + SO(3) pre-alignment
        assert abs(fr["count"] - fr["ocount"]) <= max(5, fr["ocount"] // 500), (k, fr)
        assert fr["pose_diff"] < 1e-4, (k, fr)
        assert k == 0 or (fr["rgb_count"] > 0 and fr["so3_iterations"] >= 1)
    for k, fr in enumerate(res["bad_depth"]):                    # NaN / +inf / negative patches in the sensor's depth image
        assert fr["nan_pattern_equal"] and fr["nan_pixels"] > 400, (k, fr)       # the filter poisons exactly the pixels the shader's would
        assert fr["filtered_diff"] < 2e-6 and fr["count"] == fr["ocount"], (k, fr)
        assert fr["pose_finite"] and fr["pose_diff"] < 2e-5, (k, fr)
    for k, fr in enumerate(res["mm_bad_depth"]):                 # multi-model frame on sensor garbage, filtered depth shared
        assert fr["ids"] == fr["oids"] and fr["counts"] == fr["ocounts"] and fr["label_diff"] == 0, (k, fr)
        assert fr["pose_diff"] < 2e-6, (k, fr)
    assert len(res["mm_bad_depth"][-1]["ids"]) == 3              # both objects spawned
    for wm, rows in res["weight"].items():                       # processFrame's weightMultiplier, poses given, filtered depth shared
        for k, fr in enumerate(rows):
            assert fr["count"] == fr["ocount"] and fr["stamps_equal"] and 0.0 <= fr["max_diff"] <= 1e-6, (wm, k, fr)
    assert res["weight"]["3.0"][-1]["conf_mean"] > 1.5 * res["weight"]["0.3"][-1]["conf_mean"]      # ... and it does change the map
    for k, fr in enumerate(res["dev_masks"]):                    # device-resident frames + masks + class table == host-pointer frames + classIDs
        assert fr["ids"] == fr["ids_dev"] and fr["classes"] == fr["classes_dev"] and fr["counts_equal"] and fr["poses_equal"] and fr["label_diff"] == 0, (k, fr)
    assert max(len(fr["ids"]) for fr in res["dev_masks"]) >= 2
    sw = res["static"]                                            # makeNonStatic / makeStatic / updateStaticPose
    assert sw["flag_after_nonstatic"] is True and sw["flag_after_static"] is False and sw["tracked_or_replaced"] is True, sw
    import numpy as np
    by_id = {}
    for k, mid, rel in sw["relative"]:
        by_id.setdefault(mid, []).append((k, np.array(rel)))
    for mid, rows in by_id.items():                               # while an object is static, pose_obj * pose_bg^-1 does not move
        static_rows = [r for k, r in rows if k != 5]
        for r in static_rows[1:]:
            assert np.abs(r - static_rows[0]).max() < 2e-6, (mid, rows)
    for name, rows in res["schedule"].items():                   # fastOdom / pyramid off: the loop's schedule follows the oracle's
        for k, fr in enumerate(rows):
            assert abs(fr["count"] - fr["ocount"]) <= (2 if "rgbd" in name else 0), (name, k, fr)
            assert fr["pose_diff"] < 2e-5, (name, k, fr)
    hp = res["host_paths"]                                        # mf_process_frame's host-side variants == the blocking form, bit for bit
    for name, v in hp.items():
        assert v["poses"] == hp["blocking"]["poses"] and v["count"] == hp["blocking"]["count"] and v["cloud_sha1"] == hp["blocking"]["cloud_sha1"], name
    mfm = res["map_forms"]                                        # the size-dependent forms of the fuse / clean passes: one result
    for name, v in mfm.items():
        ref = mfm["copy_two_launch"]
        assert v["poses"] == ref["poses"] and v["count"] == ref["count"] and v["cloud_sha1"] == ref["cloud_sha1"], name
    assert mfm["in_place_runs"]["runs"] > 0 and mfm["in_place_runs"]["visible_runs"] > 0 and mfm["in_place_runs"]["clean_runs"] > 0   # ... and the culled passes really ran
    for size, v in res["rgb_pyramid"].items():                    # one-launch intensity pyramid + derivative / gate images == the four single kernels
        assert v["images_equal"] and v["pose_equal"] and v["cloud_equal"], (size, v)
        assert v["gate_pixels"] > 100 and v["zero_texels"] > 500, (size, v)      # ... on images that exercise the gate and the zero-skipping rule


def test_emulated_kernels_reproduce_the_reference_generated_vectors():
    """The kernel-level -m gpu tests that hold the HIP kernels to vectors generated by the REFERENCE's own CUDA code compiled for the CPU
    (tests/golden/ref_vectors.npz: pyramids, vertex / normal maps, ICP / RGB / SO(3) steps, residuals, edge maps) and to the frozen oracle
    vectors, run here against the CPU-executed kernels (MF_EMU=1, tests/gpu_util.py): the same assertions the MI355X run makes, minus
    the device's own rounding.  In a subprocess: the emulated library never enters this process."""
    env = dict(os.environ, MF_EMU="1")
    files = [os.path.join(ROOT, "tests", f) for f in ("test_gpu_ref_golden.py", "test_gpu_golden.py", "test_gpu_kernels.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-q", "-x", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
