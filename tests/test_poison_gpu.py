"""The device path must not rely on hipMalloc handing out zeroed memory: with EXP_AMD_POISON=1 every
fresh device allocation is filled with 0xff (NaN doubles, 0xffffffff indices) and the parity tests
must still pass.  Runs them in a child process (the flag is read once per process).  GPU only."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("files", [("tests/test_sph_gpu.py", "tests/test_basis_gpu.py"),
                                   ("tests/test_cyl_gpu.py", "tests/test_multistep_gpu.py",
                                    "tests/test_orient_gpu.py")])
def test_parity_suite_with_poisoned_allocations(files):
    env = dict(os.environ, EXP_AMD_POISON="1")
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-x", "-q", "-p",
                        "no:cacheprovider", "-k", "not full_size and not at_scale"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
