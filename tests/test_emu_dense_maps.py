"""configs[4]'s scenario on every CPU run: the parity test of the dense-map scenario (tests/test_gpu_parity_long.py::test_config4_dense_maps) at a
quarter of the resolution and 1 / 128 of the surfel budgets, through the product's kernels EXECUTED ON THE CPU (tests/hipcpu) with the passes of
FULL maps forced (run table + culled projection passes, in-place update, in-place clean on the runs its rules can touch): lead-in with a dense map
uploaded behind every spawn, the dense room map, then three frames against OracleMM -- model list, every model's surfel count, label image exact,
every surfel of every model in its slot.  A subprocess, as tests/test_emu_smoke.py: nothing of the emulator leaks into this process; the parity
claim itself rests on the -m gpu run of the same test at the budgets the reference is compiled with."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dense_map_scenario_on_the_cpu_executed_kernels():
    env = dict(os.environ, MF_EMU="1", MF_PARITY_C4_SCALE="4", MF_PARITY_C4_OBJECTS="3", MF_PARITY_C4_GSURFELS=str(1 << 18), MF_PARITY_C4_OSURFELS=str(1 << 14),
               MF_PARITY_C4_FORMS="big", MF_PARITY_C4_FRAMES="3", MF_NO_PREBUILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity_long.py") + "::test_config4_dense_maps", "-q", "-m", "gpu",
                        "-n", "0", "-x", "-s", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "1 passed" in r.stdout and "every one in its slot" in r.stdout, tail


def _emu_gpu_tests(tests, extra_env=None, timeout=1500):
    env = dict(os.environ, MF_EMU="1", MF_NO_PREBUILD="1", **(extra_env or {}))
    r = subprocess.run([sys.executable, "-m", "pytest", *[os.path.join(ROOT, "tests", t) for t in tests], "-q", "-m", "gpu", "-n", "0", "-x", "-s", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return r, (r.stdout + r.stderr)[-3000:]


def test_dense_map_scenario_with_tracked_objects_on_the_cpu_executed_kernels():
    """The same scenario with the object models TRACKED after the lead-in (S3 as SURVEY.md 8d defines it; tests/test_gpu_parity_long.py::
    test_config4_dense_maps_tracked): every Gauss-Newton system and update of every tracked model against the oracle on equal input, through the
    batched loop with its slab culling."""
    r, tail = _emu_gpu_tests(["test_gpu_parity_long.py::test_config4_dense_maps_tracked"],
                             dict(MF_PARITY_C4_SCALE="4", MF_PARITY_C4_OBJECTS="3", MF_PARITY_C4_GSURFELS=str(1 << 18), MF_PARITY_C4_OSURFELS=str(1 << 14),
                                  MF_PARITY_C4_FORMS="big", MF_PARITY_C4_FRAMES="3"))
    assert r.returncode == 0, tail
    assert "1 passed" in r.stdout and "Gauss-Newton systems compared iteration by iteration" in r.stdout, tail


def test_in_place_clean_rules_on_the_cpu_executed_kernels():
    """Model::clean in place (round 6) where it differs most from a dense buffer: the first live surfel as vertex 0 after the head of the buffer has
    died (tests/test_gpu_switches.py::test_in_place_clean_first_surfel_rule), run culling on and off (::test_run_culling_changes_nothing) -- the
    assertions of the -m gpu tests, on the CPU-executed kernels."""
    r, tail = _emu_gpu_tests(["test_gpu_switches.py::test_in_place_clean_first_surfel_rule", "test_gpu_switches.py::test_run_culling_changes_nothing"])
    assert r.returncode == 0, tail
    assert "2 passed" in r.stdout, tail
