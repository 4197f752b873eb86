"""Host check of tools/ubench/mul_fp64_core.h (the FP64-FMA Montgomery product measured by mul_fp64.hip): the same source
compiled for the CPU (g++ -O2 -mfma -frounding-math, fesetround(FE_TOWARDZERO)) against Python big-integer arithmetic on
20 000 random and the extreme operands, both BN254 fields.  usage: python tools/ubench/mul_fp64_host_check.py"""
import os
import random
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
Q_MOD = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def limbs(x):
    return [(x >> (52 * i)) & ((1 << 52) - 1) for i in range(5)]


def main():
    exe = os.path.join(tempfile.mkdtemp(), "mul_fp64_host_check")
    subprocess.check_call(["g++", "-O2", "-mfma", "-frounding-math", "-Wno-unused-result", "-I", HERE, "-o", exe,
                           os.path.join(HERE, "mul_fp64_host_check.cpp")])
    ok = True
    for name, n in (("Fr", R_MOD), ("Fq", Q_MOD)):
        np_ = (-pow(n, -1, 1 << 52)) % (1 << 52)
        rng = random.Random(5)
        cases = [(rng.randrange(n), rng.randrange(n)) for _ in range(20000)]
        cases += [(n - 1, n - 1), (0, 5), (1, 1), (n - 1, 1), ((1 << 254) % n, n - 2)]
        inp = " ".join(map(str, limbs(n))) + " %d\n%d\n" % (np_, len(cases))
        inp += "\n".join(" ".join(map(str, limbs(a) + limbs(b))) for a, b in cases) + "\n"
        out = subprocess.run([exe], input=inp, capture_output=True, text=True).stdout.strip().splitlines()
        rinv = pow(1 << 260, -1, n)
        bad = sum(1 for (a, b), line in zip(cases, out)
                  if sum(int(v) << (52 * i) for i, v in enumerate(line.split())) != a * b * rinv % n)
        print(name, "cases", len(cases), "bad", bad)
        ok = ok and bad == 0 and len(out) == len(cases)
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
