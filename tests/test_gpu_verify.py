"""GPU: prove on the device, verify with the product's own host verifier (PlonkishBackend::{prove, verify} round
trip, backend.rs:213-239), at sizes beyond what the Python oracle verifies quickly."""
import random

import numpy as np
import pytest

from oracle.pyref import hyperplonk as o_hp
from oracle.pyref.field import R_MOD as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,n", [("range", 10), ("range", 18), ("and", 16), ("xor", 12), ("range", 20), ("and", 20)])
def test_lasso_prove_then_verify(hl, ctx, kind, n):
    rng = np.random.default_rng(n)
    table = hl.LassoTable.range(2, 16) if kind == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, 4, 16)
    nv = max(n, table.l)
    ss = [int(v) for v in rng.integers(1, 1 << 62, size=nv)]
    pp, vp = hl.MultilinearKzg.setup(ctx, ss), hl.MultilinearKzgVerifierParams.setup(ss)
    dims = [ctx.upload(rng.integers(0, 1 << table.l, size=1 << n, dtype=np.uint32).tobytes()) for _ in range(table.c)]
    t = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, dims, t)
    proof = t.into_proof()
    hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(proof))
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 4
    with pytest.raises(hl.Error):
        hl.lasso_verify(vp, table, n, hl.Keccak256Transcript.from_proof(bytes(bad)))
    other_vp = hl.MultilinearKzgVerifierParams.setup([s + 1 for s in ss])
    with pytest.raises(hl.InvalidPcsOpen):
        hl.lasso_verify(other_vp, table, n, hl.Keccak256Transcript.from_proof(proof))


@pytest.mark.parametrize("num_vars,with_lookup", [(4, True), (9, False), (12, True)])
def test_hyperplonk_prove_then_verify(hl, ctx, num_vars, with_lookup):
    from halo2_lasso_amd import hyperplonk as g_hp
    rng = random.Random(num_vars)
    ss = [rng.randrange(1, P) for _ in range(num_vars)]
    pcs_pp, pcs_vp = hl.MultilinearKzg.setup(ctx, ss), hl.MultilinearKzgVerifierParams.setup(ss)
    gen = o_hp.rand_vanilla_plonk_with_lookup_circuit if with_lookup else o_hp.rand_vanilla_plonk_circuit
    o_info, instances, witness = gen(num_vars, rng)
    mk = g_hp.vanilla_plonk_with_lookup_circuit_info if with_lookup else g_hp.vanilla_plonk_circuit_info
    info = mk(num_vars, len(instances[0]), o_info.preprocess_polys, o_info.permutations)
    pp, vp = g_hp.HyperPlonk.preprocess(pcs_pp, info, pcs_vp)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness], t)
    proof = t.into_proof()
    r = hl.Keccak256Transcript.from_proof(proof)
    g_hp.HyperPlonk.verify(vp, instances, r)
    assert r.remaining() == 0
    bad = [list(instances[0])]
    bad[0][0] = (bad[0][0] + 1) % P
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(vp, bad, hl.Keccak256Transcript.from_proof(proof))
