#!/usr/bin/env python3
"""Instruction audit of the device kernels: compiles a .hip file of csrc/ to gfx950 assembly and prints,
per kernel whose demangled name matches the pattern, the instruction count by class (multiply-adds, carry additions, moves,
selects, hazard nops, wait counts, memory), registers, occupancy and scratch - the numbers the instruction-count model of
DESIGN.md section 3 is checked against.  No GPU needed.
usage: python tools/isa_audit.py kernels_sumcheck.hip 'sc_round_pp_kernel|sc_round_open_kernel' [extra hipcc flags...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_mad_u64"):
        return "v_mad_u64_u32"
    if op.startswith(("v_addc", "v_add_co", "v_subb", "v_sub_co", "v_subbrev")):
        return "carry add/sub"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "v_mov/accvgpr"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "memory"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "other valu"
    if op.startswith("s_"):
        return "scalar"
    return "other"


def main():
    src, pat = sys.argv[1], re.compile(sys.argv[2])
    extra = sys.argv[3:]
    path = src if os.path.exists(src) else os.path.join(ROOT, "halo2-lasso_amd", "csrc", src)
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950",
                               "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-o", asm, path] + extra,
                              stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    kernels, cur = [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z[\w]+):\s*(;.*)?$", l)
        if m:
            cur = (m.group(1), i)
        if l.startswith("; Kernel info:") and cur:
            kernels.append((cur[0], cur[1], i))
            cur = None
    names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in kernels), capture_output=True, text=True).stdout.split("\n")
    for (mangled, a, b), name in zip(kernels, names):
        if not pat.search(name):
            continue
        cnt = collections.Counter()
        for l in lines[a:b]:
            t = l.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            cnt[t.split()[0]] += 1
        info = {}
        for x in lines[b:b + 20]:
            if x.startswith(";") and ":" in x:
                k, v = x.strip("; ").split(":", 1)
                info[k.strip()] = v.strip()
        groups = collections.Counter()
        for op, c in cnt.items():
            groups[classify(op)] += c
        print(re.sub(r"\(.*", "", name))
        print("   instructions %d  vgprs %s  sgprs %s  occupancy %s  scratch %s B  code %s B" % (
            sum(cnt.values()), info.get("NumVgprs"), info.get("NumSgprs"), info.get("Occupancy"), info.get("ScratchSize"),
            info.get("codeLenInByte")))
        print("   " + "  ".join("%s %d" % kv for kv in groups.most_common()))
        other = [(o, c) for o, c in cnt.most_common() if classify(o) == "other valu"][:10]
        print("   other valu: " + "  ".join("%s %d" % kv for kv in other))


if __name__ == "__main__":
    main()
