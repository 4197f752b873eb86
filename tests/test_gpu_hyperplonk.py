"""GPU: HyperPlonk::prove with LogUp lookups and the permutation argument (backend/hyperplonk.rs:164-291)
on the reference's two sample circuits (backend/hyperplonk.rs:387-408 run vanilla_plonk and
vanilla_plonk_with_lookup for num_vars 2..16): proof bytes equal to the oracle's, then the oracle's
verifier accepts the GPU proof."""
import random

import pytest

from oracle.pyref import hyperplonk as o_hp, kzg as o_kzg
from oracle.pyref.field import R_MOD as P
from oracle.pyref.transcript import Keccak256Transcript as OT

pytestmark = pytest.mark.gpu


def _setup(hl, ctx, num_vars, seed):
    rng = random.Random(seed)
    ss = [rng.randrange(1, P) for _ in range(num_vars)]
    return o_kzg.setup(ss), hl.MultilinearKzg.setup(ctx, ss)


def _circuit(hl, num_vars, with_lookup, seed):
    from halo2_lasso_amd import hyperplonk as g_hp
    rng = random.Random(seed)
    gen = o_hp.rand_vanilla_plonk_with_lookup_circuit if with_lookup else o_hp.rand_vanilla_plonk_circuit
    o_info, instances, witness = gen(num_vars, rng)
    mk = g_hp.vanilla_plonk_with_lookup_circuit_info if with_lookup else g_hp.vanilla_plonk_circuit_info
    g_info = mk(num_vars, len(instances[0]), o_info.preprocess_polys, o_info.permutations)
    return o_info, g_info, instances, witness


def _prove_both(hl, ctx, num_vars, with_lookup, seed):
    from halo2_lasso_amd import hyperplonk as g_hp
    o_pcs, g_pcs = _setup(hl, ctx, num_vars, seed)
    o_info, g_info, instances, witness = _circuit(hl, num_vars, with_lookup, seed + 1)
    o_pp = o_hp.preprocess(o_pcs, o_info)
    g_pp = g_hp.HyperPlonk.preprocess(g_pcs, g_info)
    assert g_pp.preprocess_comms == o_pp.preprocess_comms
    assert g_pp.permutation_comms == o_pp.permutation_comms
    ot = OT()
    o_hp.prove(o_pp, instances, lambda rnd, ch: witness, ot)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness], t)
    return o_pp, instances, ot.into_proof(), t.into_proof()


@pytest.mark.parametrize("with_lookup", [False, True])
@pytest.mark.parametrize("num_vars", [2, 3, 4, 6])
def test_hyperplonk_prove_matches_oracle(hl, ctx, num_vars, with_lookup):
    o_pp, instances, o_proof, g_proof = _prove_both(hl, ctx, num_vars, with_lookup, 40 + 2 * num_vars + with_lookup)
    assert g_proof == o_proof
    o_hp.verify(o_pp, instances, OT(g_proof))


def test_hyperplonk_invalid_lookup_input(hl, ctx):
    """prover.rs:176-178: an input row that is not in the table -> InvalidSnark("Invalid lookup input")"""
    from halo2_lasso_amd import hyperplonk as g_hp
    num_vars = 4
    _, g_pcs = _setup(hl, ctx, num_vars, 5)
    o_info, g_info, instances, witness = _circuit(hl, num_vars, True, 6)
    q_lookup = o_info.preprocess_polys[5]
    rows = [b for b in range(1 << num_vars) if q_lookup[b] == 1]
    assert rows
    witness = [list(w) for w in witness]
    witness[0][rows[0]] = (witness[0][rows[0]] + 1) % P
    g_pp = g_hp.HyperPlonk.preprocess(g_pcs, g_info)
    with pytest.raises(hl.InvalidSnark, match="Invalid lookup input"):
        g_hp.HyperPlonk.prove(g_pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness],
                              hl.Keccak256Transcript())


def test_hyperplonk_wrong_witness_is_rejected(hl, ctx):
    """a gate-violating witness still proves (the prover does not check), and the oracle verifier rejects"""
    from halo2_lasso_amd import hyperplonk as g_hp
    num_vars = 3
    o_pcs, g_pcs = _setup(hl, ctx, num_vars, 9)
    o_info, g_info, instances, witness = _circuit(hl, num_vars, False, 10)
    witness = [list(w) for w in witness]
    witness[2][5] = (witness[2][5] + 1) % P
    o_pp = o_hp.preprocess(o_pcs, o_info)
    g_pp = g_hp.HyperPlonk.preprocess(g_pcs, g_info)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness], t)
    with pytest.raises(Exception):
        o_hp.verify(o_pp, instances, OT(t.into_proof()))


@pytest.mark.parametrize("num_vars,with_lookup", [(10, True), (13, False), (14, True)])
def test_hyperplonk_large_proof_verifies(hl, ctx, num_vars, with_lookup):
    """sizes past what the pure-Python prover restatement finishes in seconds: the GPU proof goes through the
    oracle's verifier (trapdoor-form KZG check), with the verifier key taken from the GPU preprocess"""
    from halo2_lasso_amd import hyperplonk as g_hp
    rng = random.Random(num_vars)
    ss = [rng.randrange(1, P) for _ in range(num_vars)]
    g_pcs = hl.MultilinearKzg.setup(ctx, ss)
    o_info, g_info, instances, witness = _circuit(hl, num_vars, with_lookup, 77 + num_vars)
    g_pp = g_hp.HyperPlonk.preprocess(g_pcs, g_info)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness], t)
    proof = t.into_proof()
    vp = o_hp.Param()
    vp.pcs, vp.num_vars = o_kzg.Params(ss, None), num_vars
    vp.num_instances, vp.num_witness_polys, vp.num_challenges = o_info.num_instances, [3], [0]
    vp.lookups = o_info.lookups
    vp.preprocess_comms, vp.permutation_comms = g_pp.preprocess_comms, g_pp.permutation_comms
    vp.num_permutation_z_polys, vp.expression = o_hp.compose(o_info)
    o_hp.verify(vp, instances, OT(proof))
    # tampering with any evaluation breaks it
    bad = bytearray(proof)
    bad[-64 * num_vars - 5] ^= 1
    with pytest.raises(Exception):
        o_hp.verify(vp, instances, OT(bytes(bad)))


@pytest.mark.parametrize("num_vars,with_lookup", [(3, False), (4, True), (6, True)])
def test_hyperplonk_over_zeromorph_matches_oracle(hl, ctx, num_vars, with_lookup):
    """HyperPlonk<Zeromorph<UnivariateKzg<Bn256>>> (backend/hyperplonk.rs:426): same schedule, other PCS"""
    from halo2_lasso_amd import hyperplonk as g_hp
    from oracle.pyref import zeromorph as o_zm
    rng = random.Random(300 + num_vars)
    s = rng.randrange(1, P)
    o_info, g_info, instances, witness = _circuit(hl, num_vars, with_lookup, 900 + num_vars)
    o_pp = o_hp.preprocess(o_zm.trim(o_zm.setup(s, 1 << num_vars), 1 << num_vars), o_info, o_zm)
    pcs_pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, s, 1 << num_vars), 1 << num_vars)
    pcs_vp = hl.ZeromorphVerifierParam.setup(s, 1 << num_vars, 1 << num_vars)
    g_pp, g_vp = g_hp.HyperPlonk.preprocess(pcs_pp, g_info, pcs_vp)
    assert g_pp.preprocess_comms == o_pp.preprocess_comms and g_pp.permutation_comms == o_pp.permutation_comms
    ot = OT()
    o_hp.prove(o_pp, instances, lambda rnd, ch: witness, ot)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness], t)
    proof = t.into_proof()
    assert proof == ot.into_proof()
    o_hp.verify(o_pp, instances, OT(proof))
    r = hl.Keccak256Transcript.from_proof(proof)
    g_hp.HyperPlonk.verify(g_vp, instances, r)
    assert r.remaining() == 0
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 2
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(g_vp, instances, hl.Keccak256Transcript.from_proof(bytes(bad)))


@pytest.mark.parametrize("pcs,k", [("mkzg", 16), ("mkzg", 20), ("zeromorph", 18)])
def test_hyperplonk_synthetic_circuit_verifies(hl, ctx, pcs, k):
    """the numpy/device-built circuit of bench.py --workload hyperplonk at its measured sizes: the proof must verify,
    and a proof for a witness with one wrong cell must not"""
    import numpy as np
    from halo2_lasso_amd import hyperplonk as g_hp, synthetic
    circ = synthetic.vanilla_plonk_with_lookup(ctx, k)
    ss = [int(v) for v in np.random.default_rng(k).integers(1, 1 << 62, size=k)]
    if pcs == "mkzg":
        pcs_pp, pcs_vp = hl.MultilinearKzg.setup(ctx, ss), hl.MultilinearKzgVerifierParams.setup(ss)
    else:
        pcs_pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, ss[0], 1 << k), 1 << k)
        pcs_vp = hl.ZeromorphVerifierParam.setup(ss[0], 1 << k, 1 << k)
    pp, vp = synthetic.prover_param(pcs_pp, circ, pcs_vp)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, t)
    proof = t.into_proof()
    assert len(proof) == 1024 + 352 * k + (128 if pcs == "zeromorph" else 0)
    g_hp.HyperPlonk.verify(vp, circ.instances, hl.Keccak256Transcript.from_proof(proof))
    # break one gate: w_o of row 0 (an add gate)
    bad = circ.h_witness[2].copy()
    bad[0, 0] ^= np.uint64(1)
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness[:2] + [hl.MultilinearPolynomial(ctx, ctx.upload(bad.tobytes()), k)], t)
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(vp, circ.instances, hl.Keccak256Transcript.from_proof(t.into_proof()))


# ------------------------------------------------------------------ multi-phase circuits (hyperplonk.rs:185-205)
def _two_phase_circuit(mod_ex, mk_info, k, rng, theta_of):
    """pi | q, q_inst | phase 0: w0, w1 -> challenge theta | phase 1: w2 = w0 + theta * w1 -> challenge (used as a
    constraint separator).  Constraints: q (w2 - w0 - theta w1), ch1 (w2 - w0 - theta w1) w1, q_inst (w0 - pi); one copy
    constraint between (w0, row a) and (w2, row b)."""
    from oracle.pyref import hyperplonk as o_hp
    n = 1 << k
    rows = o_hp.row_mapping(k)
    num_inst = 3
    instances = [[rng.randrange(P) for _ in range(num_inst)]]
    q = [rng.randrange(2) for _ in range(n)]
    q_inst = [0] * n
    w0 = [rng.randrange(P) for _ in range(n)]
    w1 = [rng.randrange(P) for _ in range(n)]
    for i in range(num_inst):
        q_inst[rows[i]] = 1
        w0[rows[i]] = instances[0][i]
    a, b = rows[num_inst], rows[num_inst + 1]   # rows outside the instance rows, never row 0
    w1[b], w0[b] = 0, w0[a]                      # then w2[b] = w0[b] = w0[a]: the copy holds
    E = mod_ex
    pi, pq, pqi, pw0, pw1, pw2 = (E.Polynomial(i) if hasattr(E, "Polynomial") else E.Poly(i) for i in range(6))
    theta, sep = E.Challenge(0), E.Challenge(1)
    gate = pw2 - pw0 - theta * pw1
    constraints = [pq * gate, sep * gate * pw1, pqi * (pw0 - pi)]
    info = mk_info(k, [num_inst], [q, q_inst], [2, 1], [1, 1], constraints, [], [[(3, a), (5, b)]], None)

    def synthesize(rnd, challenges):
        if rnd == 0:
            assert challenges == []
            return [w0, w1]
        assert len(challenges) == 1
        return [[(x + challenges[0] * y) % P for x, y in zip(w0, w1)]]
    return info, instances, synthesize


@pytest.mark.parametrize("num_vars", [3, 5, 9])
def test_hyperplonk_two_phase_circuit(hl, ctx, num_vars):
    """the phase loop: synthesize(round, challenges) is called per phase through the C-ABI callback, the second phase's
    witness depends on the first phase's challenge; bytes equal the oracle's, both verifiers accept, a witness that
    ignores the challenge is rejected"""
    from halo2_lasso_amd import hyperplonk as g_hp, expression as g_ex
    from oracle.pyref import expression as o_ex
    o_pcs, g_pcs = _setup(hl, ctx, num_vars, 900 + num_vars)
    o_info, instances, o_synth = _two_phase_circuit(o_ex, o_hp.CircuitInfo, num_vars, random.Random(num_vars), None)
    g_info, instances2, _ = _two_phase_circuit(g_ex, g_hp.PlonkishCircuitInfo, num_vars, random.Random(num_vars), None)
    assert instances == instances2
    o_pp = o_hp.preprocess(o_pcs, o_info)
    ot = OT()
    o_hp.prove(o_pp, instances, o_synth, ot)
    rng = random.Random(900 + num_vars)
    ss = [rng.randrange(1, P) for _ in range(num_vars)]
    g_pp, g_vp = g_hp.HyperPlonk.preprocess(g_pcs, g_info, hl.MultilinearKzgVerifierParams.setup(ss))
    calls = []

    def synth(rnd, challenges):
        calls.append((rnd, list(challenges)))
        return [hl.MultilinearPolynomial.new(ctx, w) for w in o_synth(rnd, challenges)]
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, synth, t)
    proof = t.into_proof()
    assert [c[0] for c in calls] == [0, 1] and calls[0][1] == [] and len(calls[1][1]) == 1
    assert proof == ot.into_proof()
    o_hp.verify(o_pp, instances, OT(proof))
    g_hp.HyperPlonk.verify(g_vp, instances, hl.Keccak256Transcript.from_proof(proof))

    # a second-phase witness computed with the wrong challenge does not satisfy the circuit
    def cheat(rnd, challenges):
        return [hl.MultilinearPolynomial.new(ctx, w) for w in o_synth(rnd, [c + 1 for c in challenges])]
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, cheat, t)
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(g_vp, instances, hl.Keccak256Transcript.from_proof(t.into_proof()))
    # a synthesize that returns the wrong number of polys is the reference's assert_eq (hyperplonk.rs:198)
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.prove(g_pp, instances, lambda r, ch: [], hl.Keccak256Transcript())


@pytest.mark.parametrize("num_vars", [3, 6])
def test_hyperplonk_two_phase_circuit_over_zeromorph(hl, ctx, num_vars):
    """the phase loop is generic over the PCS (hyperplonk.rs:185-205 under HyperPlonk<Zeromorph<..>>, hyperplonk.rs:426):
    lh_hyperplonk_prove_phases_zeromorph / lh_hyperplonk_verify_phases_zeromorph against the specification's bytes"""
    from halo2_lasso_amd import hyperplonk as g_hp, expression as g_ex
    from oracle.pyref import expression as o_ex, zeromorph as o_zm
    s = random.Random(940 + num_vars).randrange(1, P)
    o_info, instances, o_synth = _two_phase_circuit(o_ex, o_hp.CircuitInfo, num_vars, random.Random(num_vars), None)
    g_info, _, _ = _two_phase_circuit(g_ex, g_hp.PlonkishCircuitInfo, num_vars, random.Random(num_vars), None)
    o_pp = o_hp.preprocess(o_zm.trim(o_zm.setup(s, 1 << num_vars), 1 << num_vars), o_info, o_zm)
    ot = OT()
    o_hp.prove(o_pp, instances, o_synth, ot)
    pcs_pp = hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, s, 1 << num_vars), 1 << num_vars)
    g_pp, g_vp = g_hp.HyperPlonk.preprocess(pcs_pp, g_info, hl.ZeromorphVerifierParam.setup(s, 1 << num_vars, 1 << num_vars))
    calls = []

    def synth(rnd, challenges):
        calls.append(rnd)
        return [hl.MultilinearPolynomial.new(ctx, w) for w in o_synth(rnd, challenges)]
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, synth, t)
    proof = t.into_proof()
    assert calls == [0, 1] and proof == ot.into_proof()
    o_hp.verify(o_pp, instances, OT(proof))
    r = hl.Keccak256Transcript.from_proof(proof)
    g_hp.HyperPlonk.verify(g_vp, instances, r)
    assert r.remaining() == 0
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(g_vp, instances, hl.Keccak256Transcript.from_proof(bytes(bad)))


def test_jit_code_objects_are_kept_on_disk(tmp_path):
    """csrc/jit.cpp: the runtime-compiled round kernels of a HyperPlonk proof are written to LH_JIT_CACHE_DIR and the next
    process loads them instead of compiling (LH_HP_DEBUG says which); a torn file is ignored and rewritten; same bytes."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, random, hashlib
        sys.path.insert(0, %r)
        import halo2_lasso_amd as hl
        from halo2_lasso_amd import synthetic
        ctx = hl.Context(0)
        k = 10
        rng = random.Random(5)
        pcs = hl.MultilinearKzg.setup(ctx, [rng.randrange(1, hl.R_MOD) for _ in range(k)])
        circ = synthetic.vanilla_plonk_with_lookup(ctx, k, seed=3)
        pp = synthetic.prover_param(pcs, circ)
        t = hl.Keccak256Transcript()
        from halo2_lasso_amd import hyperplonk as g_hp
        g_hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, t)
        print("PROOF", hashlib.sha256(t.into_proof()).hexdigest())
    """) % root
    cache = tmp_path / "jit"
    env = dict(os.environ, LH_JIT_CACHE_DIR=str(cache), LH_EXPR_JIT_MIN_VARS="4", LH_HP_DEBUG="1")

    def run():
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        proof = [ln for ln in r.stdout.splitlines() if ln.startswith("PROOF")][0]
        return proof, r.stderr.count("[expr] compiled"), r.stderr.count("[expr] loaded from the disk cache")

    p1, compiled1, loaded1 = run()
    files = sorted(cache.glob("gfx950-*.co"))
    assert compiled1 >= 1 and loaded1 == 0 and len(files) == compiled1, (compiled1, loaded1, files)
    p2, compiled2, loaded2 = run()
    assert p2 == p1 and compiled2 == 0 and loaded2 == compiled1
    files[0].write_bytes(files[0].read_bytes()[:100])   # a torn file: compiled again, rewritten
    p3, compiled3, loaded3 = run()
    assert p3 == p1 and compiled3 == 1 and loaded3 == compiled1 - 1
    assert files[0].stat().st_size > 100
    env["LH_JIT_CACHE"] = "0"                           # switched off: compiles, reads and writes nothing
    p4, compiled4, loaded4 = run()
    assert p4 == p1 and compiled4 == compiled1 and loaded4 == 0


# ------------------------------------------------------------------ Lasso as HyperPlonk's lookup argument
@pytest.mark.parametrize("kind,c,l,num_vars", [("range", 2, 2, 4), ("and", 2, 4, 5), ("xor", 2, 4, 4), ("range", 2, 4, 4),
                                               ("and", 4, 4, 7)])
def test_hyperplonk_with_lasso_lookup(hl, ctx, kind, c, l, num_vars):
    """north_star's "hyperplonk::prover Lasso/Surge memory-check" (BASELINE.json configs[4] stand-in: vanilla gates plus
    one lookup into a decomposable table proven by Lasso inside HyperPlonk::prove; specification
    oracle/pyref/hyperplonk.py LassoLookup): proof bytes equal the oracle's, both verifiers accept, a chunk value outside
    the subtable or a wrong output is the reference's "Invalid lookup input"."""
    from halo2_lasso_amd import hyperplonk as g_hp
    from test_verifier import _lasso_circuit
    o_pcs, g_pcs = _setup(hl, ctx, num_vars, 300 + num_vars)
    o_info, g_info, instances, witness = _lasso_circuit(hl, kind, c, l, num_vars, 31 + num_vars)
    o_pp = o_hp.preprocess(o_pcs, o_info)
    ot = OT()
    o_hp.prove(o_pp, instances, lambda r, ch: witness, ot)
    rng = random.Random(300 + num_vars)
    ss = [rng.randrange(1, P) for _ in range(num_vars)]
    g_pp, g_vp = g_hp.HyperPlonk.preprocess(g_pcs, g_info, hl.MultilinearKzgVerifierParams.setup(ss))
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness], t)
    proof = t.into_proof()
    assert proof == ot.into_proof()
    o_hp.verify(o_pp, instances, OT(proof))
    g_hp.HyperPlonk.verify(g_vp, instances, hl.Keccak256Transcript.from_proof(proof))
    # a chunk index outside the subtable / an output that is not the table's value
    row = [i for i in range(1 << num_vars) if o_info.preprocess_polys[5][i] == 1][0]
    for col, val in ((3, 1 << l), (len(witness) - 1, (witness[-1][row] + 1) % P)):
        bad = [list(w) for w in witness]
        bad[col][row] = val
        with pytest.raises(hl.InvalidSnark, match="Invalid lookup input"):
            g_hp.HyperPlonk.prove(g_pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in bad],
                                  hl.Keccak256Transcript())


def test_hyperplonk_lasso_zero_columns_and_two_lookups(hl, ctx):
    """two Lasso lookups in one circuit (a range and an XOR table over different columns), one of them with pairwise
    distinct chunk indices (read_ts identically zero: an identity commitment inside the Lasso group, framed by the
    mask): bytes against the specification, both verifiers"""
    from halo2_lasso_amd import hyperplonk as g_hp, expression as g_ex
    from oracle.pyref import expression as o_ex, lasso as o_lasso
    k = 4
    n = 1 << k
    rng = random.Random(77)
    specs = [o_lasso.range_table(2, 2), o_lasso.bitwise_table(o_lasso.SUBTABLE_XOR, 1, 4)]
    tables = [hl.LassoTable.range(2, 2), hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 1, 4)]
    # witness columns: lookup 0: d0, d1, a0 (range over 2 x 2-bit limbs); lookup 1: e0 (a permutation of 0..15), a1
    d0, d1 = [rng.randrange(4) for _ in range(n)], [rng.randrange(4) for _ in range(n)]
    a0 = [x + 4 * y for x, y in zip(d0, d1)]
    e0 = list(range(n))
    rng.shuffle(e0)
    a1 = [(v >> 2) ^ (v & 3) for v in e0]
    w = [rng.randrange(P) for _ in range(n)]
    witness = [w, d0, d1, a0, e0, a1]          # polys 2..7 (pi = 0, q = 1)
    q = [rng.randrange(2) for _ in range(n)]

    def info(E, Info, Lk, tabs):
        mk = E.Polynomial if hasattr(E, "Polynomial") else E.Poly
        pq, pd0, pd1, pa0 = mk(1), mk(3), mk(4), mk(5)
        i = Info(k, [0], [q], [6], [0], [pq * (pa0 - pd0 - pd1 * 4)], [], [], None)  # a0 recomposes its limbs
        i.lasso_lookups = [Lk(tabs[0], 5, [3, 4]), Lk(tabs[1], 7, [6])]
        return i
    o_info = info(o_ex, o_hp.CircuitInfo, o_hp.LassoLookup, specs)
    g_info = info(g_ex, g_hp.PlonkishCircuitInfo, g_hp.LassoLookup, tables)
    o_pcs, g_pcs = _setup(hl, ctx, k, 771)
    o_pp = o_hp.preprocess(o_pcs, o_info)
    ot = OT()
    o_hp.prove(o_pp, [[]], lambda r, ch: witness, ot)
    prng = random.Random(771)
    ss = [prng.randrange(1, P) for _ in range(k)]
    g_pp, g_vp = g_hp.HyperPlonk.preprocess(g_pcs, g_info, hl.MultilinearKzgVerifierParams.setup(ss))
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(g_pp, [[]], [hl.MultilinearPolynomial.new(ctx, x) for x in witness], t)
    proof = t.into_proof()
    assert proof == ot.into_proof()
    o_hp.verify(o_pp, [[]], OT(proof))
    g_hp.HyperPlonk.verify(g_vp, [[]], hl.Keccak256Transcript.from_proof(proof))
