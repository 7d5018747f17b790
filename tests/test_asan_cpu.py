"""The host-side code under AddressSanitizer + UBSan (`make asan`, tools/asan_driver.cpp): URDF+ reader, plan compiler and
oracle over every robot URDF of the reference and the serialised TelloWithArms model, plus truncated and randomly
corrupted model descriptions (which must be rejected, not read out of bounds -- the round-3 run of this driver found and
fixed such a read in the loop-constraint payload of the plan compiler).  GPU sanitizers are not available on the pool;
the kernels are covered by the parity tests."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    r = subprocess.run(["make", "-C", ROOT, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    from generalized_rbda_amd.robots import jvrc1_humanoid, tello_with_arms

    blobs = []
    for name, model in (("tello", tello_with_arms()), ("jvrc1_hand_built", jvrc1_humanoid())):
        p = tmp_path / f"{name}.grbd"
        p.write_bytes(model.serialize())
        blobs.append(str(p))
    urdfs = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "robot-models", "*.urdf")))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "build", "asan", "asan_driver"), *urdfs, *blobs], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("OK"), (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
