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
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "halo2-lasso_amd", "csrc"), "-I",
                           os.path.join(ROOT, "tools", "ubench"), "-o", exe,
                           os.path.join(ROOT, "tools", "ubench", "mul29_host_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("bad 0 "), r.stdout + r.stderr


def test_generated_column_blocks_are_current_and_schedule_every_product():
    """csrc/ff_cols.inc is what tools/gen_ff_cols.py prints, and its blocks hold the product-scanning schedule of
    ff.cuh mul_scan: column k multiplies a_i b_(k-i) for every valid i and m_i p_(k-i) for every earlier quotient digit -
    64 operand products and 56 reduction products over the 15 columns (the 8 products m_k p_0 are ff.cuh's LH_MACS);
    accumulator and carry word are early-clobber operands of every block."""
    import re
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_ff_cols.py")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    text = open(os.path.join(ROOT, "halo2-lasso_amd", "csrc", "ff_cols.inc")).read()
    assert r.stdout == text, "halo2-lasso_amd/csrc/ff_cols.inc is stale: python3 tools/gen_ff_cols.py > halo2-lasso_amd/csrc/ff_cols.inc"
    blocks = dict(re.findall(r"#define (LH_COL_MUL_\d+)\(A, B, M, PP\) \\\n((?:.*\\\n)*.*)\n", text))
    assert len(blocks) == 15
    vv_total = vs_total = 0
    for k in range(15):
        body = blocks["LH_COL_MUL_%d" % k]
        vv = re.findall(r'"v"\(\(A\)\.l\[(\d)\]\), "v"\(\(B\)\.l\[(\d)\]\)', body)
        vs = re.findall(r'"v"\(\(M\)\[(\d)\]\), "s"\(PP::mod\((\d)\)\)', body)
        assert sorted(int(i) for i, _ in vv) == [i for i in range(8) if 0 <= k - i < 8]
        assert all(int(i) + int(j) == k for i, j in vv + vs) and all(int(j) >= 1 for _, j in vs)
        assert sorted(int(i) for i, _ in vs) == [i for i in range(8) if 1 <= k - i < 8]
        assert body.count("v_mad_u64_u32") == len(vv) + len(vs)
        vv_total, vs_total = vv_total + len(vv), vs_total + len(vs)
    assert (vv_total, vs_total) == (64, 56)
    outs = re.findall(r':\s*"([=+]&?)v"\(acc\), "([=+]&?)v"\(top\)', text)
    assert outs and all(a.endswith("&") and b.endswith("&") for a, b in outs)
