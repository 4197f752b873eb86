"""GPU: the C-ABI without Python -- examples/lasso_c_abi.c is compiled with gcc against include/lasso_hip.h and the
in-tree shared library, proves 2^12 and 2^18 range-check lookups, verifies them with the host verifier and rejects a
tampered proof."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_example_builds_and_runs(tmp_path):
    exe = str(tmp_path / "lasso_c_abi")
    lib_dir = os.path.join(ROOT, "halo2-lasso_amd")
    subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "lasso_c_abi.c"), "-L" + lib_dir, "-llasso_hip",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True, capture_output=True, text=True)
    for n in ("12", "18"):
        r = subprocess.run([exe, n], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "verified on the host, 0 bytes left unread" in r.stdout
        assert "tampered proof -> status -" in r.stdout
