"""CPU checks of the arithmetic experiments of round 4: the FP64-FMA Montgomery product and the
9 x 29-bit lazy-carry form are the same sources the GPU micro-benchmarks time; here they are compiled for the host and
compared with big-integer arithmetic / the CIOS product."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp64_fma_montgomery_product_matches_big_integers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ubench", "mul_fp64_host_check.py")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Fr cases 20005 bad 0" in r.stdout and "Fq cases 20005 bad 0" in r.stdout, r.stdout


def test_29_bit_limb_product_matches_cios(tmp_path):
    exe = str(tmp_path / "mul29_host_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "halo2-lasso_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tools", "ubench", "mul29_host_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("bad 0 "), r.stdout + r.stderr
