import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lasso_amd as hl
from halo2_lasso_amd import hyperplonk as hp, synthetic
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = hl.Context(0)
circ = synthetic.vanilla_plonk_with_lookup(ctx, k)
rng = np.random.default_rng(k)
ss = [int(v) for v in rng.integers(1, 1 << 62, size=k)]
pcs_pp, pcs_vp = hl.MultilinearKzg.setup(ctx, ss), hl.MultilinearKzgVerifierParams.setup(ss)
pp, vp = synthetic.prover_param(pcs_pp, circ, pcs_vp)
for _ in range(2):
    hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, hl.Keccak256Transcript())
hl.profile_enable(ctx, True)
hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, hl.Keccak256Transcript())
recs = hl.profile_read(ctx)
hl.profile_enable(ctx, False)
for r in recs:
    if r["name"].startswith("sc_round_jit") or r["name"].startswith("sc_round_prog") or r["name"].startswith("fix_var"):
        print("%-22s items %9d  %8.3f ms" % (r["name"], r["items"], r["ms"]))
