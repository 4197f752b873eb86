"""CPU: the product's verifiers (host code in liblasso_hip.so, no GPU) against the oracle and the golden proofs.

* BN254 pairing: the C++ tower implementation vs the oracle's flat-extension implementation
  (oracle/pyref/pairing.py) through `pairings_product_is_identity` (util/arithmetic.rs:24-33), bilinearity.
* TranscriptRead: reference error behaviour (transcript.rs:138-154,185-210).
* MultilinearKzg::{verify, batch_verify}, Lasso verify, HyperPlonk::verify: accept the committed golden proofs
  (tests/golden/vectors.json), reject tampered ones with the reference's error kinds.
"""
import json
import os
import random

import pytest

from oracle.pyref import curve, kzg as o_kzg, pairing as o_pair
from oracle.pyref.field import R_MOD as P, Q_MOD
from oracle.pyref.poly import evaluate
from oracle.pyref.transcript import Keccak256Transcript as OT

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
I = lambda xs: [int(x, 16) for x in xs]
SS = I(GOLDEN["srs"]["ss"])


@pytest.fixture(scope="module")
def vp5(hl):
    return hl.MultilinearKzgVerifierParams.setup(SS)


# ------------------------------------------------------------------ pairing
def test_verifier_params_match_oracle_g2(hl, vp5):
    g1, g2, ss = vp5.export()
    assert g1 == curve.G1_GEN and g2 == o_pair.G2_GEN
    assert ss[0] == o_pair.g2_mul(o_pair.G2_GEN, SS[0]) and ss[4] == o_pair.g2_mul(o_pair.G2_GEN, SS[4])
    again = hl.MultilinearKzgVerifierParams.new(g1, g2, ss)
    assert again.export() == (g1, g2, ss) and again.num_vars == 5
    with pytest.raises(hl.Error):
        hl.MultilinearKzgVerifierParams.new(g1, ((1, 2), (3, 4)), ss)  # not on the twist


def test_pairing_product_bilinear(hl):
    rng = random.Random(5)
    a, b = rng.randrange(P), rng.randrange(P)
    g1, g2 = curve.G1_GEN, o_pair.G2_GEN
    ag1, bg2 = curve.mul(g1, a), o_pair.g2_mul(g2, b)
    abg1 = curve.mul(g1, a * b % P)
    cases = [([(ag1, bg2), (curve.neg(abg1), g2)], True),
             ([(ag1, bg2), (curve.neg(ag1), bg2)], True),
             ([(ag1, bg2), (curve.neg(abg1), bg2)], False),
             ([(g1, g2)], False),
             ([(None, g2), (g1, None)], True),
             ([], True)]
    for pairs, want in cases:
        assert hl.pairings_product_is_identity(pairs) is want
    # the oracle's independent implementation agrees on the non-trivial ones
    assert o_pair.pairings_product_is_identity(cases[0][0]) and not o_pair.pairings_product_is_identity(cases[2][0])


def test_pairing_public_precompile_vectors(hl):
    """EIP-196 / EIP-197 known answers (tests/golden/alt_bn128_vectors.py) against BOTH pairing implementations - the
    product's 2-3-2 tower (csrc/pairing.hpp, through lh_pairing_check) and the oracle's flat extension: a positive pairing
    check over a G2 point that is not the generator, its negative (one G1 point negated; one G2 point replaced), and the
    ECADD / ECMUL vectors on the oracle's G1 (the device's G1 takes them in test_gpu_parity.py)."""
    from tests.golden import alt_bn128_vectors as v
    assert curve.is_on_curve(v.PAIRING_P1) and curve.is_on_curve(v.PAIRING_P2)
    assert o_pair.g2_is_on_curve(v.PAIRING_Q1) and v.PAIRING_Q2 == o_pair.G2_GEN and v.PAIRING_Q1 != o_pair.G2_GEN
    good = [(v.PAIRING_P1, v.PAIRING_Q1), (v.PAIRING_P2, v.PAIRING_Q2)]
    bad_g1 = [(v.PAIRING_P1, v.PAIRING_Q1), (curve.neg(v.PAIRING_P2), v.PAIRING_Q2)]
    bad_g2 = [(v.PAIRING_P1, v.PAIRING_Q2), (v.PAIRING_P2, v.PAIRING_Q2)]
    for impl in (hl.pairings_product_is_identity, o_pair.pairings_product_is_identity):
        assert impl(good) is True
        assert impl(bad_g1) is False and impl(bad_g2) is False
    assert curve.add(v.ECADD_A, v.ECADD_B) == v.ECADD_C
    assert curve.mul(v.ECMUL_P, v.ECMUL_K) == v.ECMUL_Q
    assert curve.msm([1, 1], [v.ECADD_A, v.ECADD_B]) == v.ECADD_C


# ------------------------------------------------------------------ TranscriptRead
def test_transcript_read_side(hl):
    w = hl.Keccak256Transcript()
    w.write_field_element(5)
    w.write_commitment((1, 2))
    c_w = w.squeeze_challenge()
    proof = w.into_proof()
    r = hl.Keccak256Transcript.from_proof(proof)
    assert r.remaining() == 96
    assert r.read_field_element() == 5 and r.read_commitment() == (1, 2)
    assert r.squeeze_challenge() == c_w and r.remaining() == 0
    with pytest.raises(hl.TranscriptError):  # read_exact past the end
        r.read_field_element()
    # non-canonical field element / point off the curve (transcript.rs:146-151,199-206)
    bad_fe = (P).to_bytes(32, "big")
    with pytest.raises(hl.TranscriptError, match="Invalid field element encoding in proof"):
        hl.Keccak256Transcript.from_proof(bad_fe).read_field_element()
    bad_pt = (1).to_bytes(32, "big") + (3).to_bytes(32, "big")
    with pytest.raises(hl.TranscriptError, match="Invalid elliptic curve point encoding in proof"):
        hl.Keccak256Transcript.from_proof(bad_pt).read_commitment()
    with pytest.raises(hl.TranscriptError, match="Invalid elliptic curve point encoding in proof"):
        hl.Keccak256Transcript.from_proof(bytes(64)).read_commitment()  # (0, 0)
    with pytest.raises(hl.TranscriptError, match="Invalid elliptic curve point encoding in proof"):
        hl.Keccak256Transcript.from_proof((Q_MOD).to_bytes(32, "big") + (2).to_bytes(32, "big")).read_commitment()


# ------------------------------------------------------------------ sum-check verify
@pytest.mark.parametrize("name,kind,degree", [("sum_check_eval", "eval", 3), ("sum_check_coeff", "coeff", 2)])
def test_sum_check_verify_golden(hl, name, kind, degree):
    from oracle.pyref import sum_check as o_sc
    g = GOLDEN[name]
    nv, claim, proof = g["num_vars"], int(g["claim"], 16), bytes.fromhex(g["proof"])
    o_cls, g_kind = (o_sc.Evaluations, hl.LH_SC_EVALUATIONS) if kind == "eval" else (o_sc.Coefficients, hl.LH_SC_COEFFICIENTS)
    want = o_sc.verify(o_cls, nv, degree, claim, OT(proof))
    got = hl.sum_check_verify(g_kind, nv, degree, claim, hl.Keccak256Transcript.from_proof(proof))
    assert got == want and got[1] == I(g["x"])
    with pytest.raises(hl.InvalidSumcheck):
        hl.sum_check_verify(g_kind, nv, degree, (claim + 1) % P, hl.Keccak256Transcript.from_proof(proof))
    bad = bytearray(proof)
    bad[32 * (degree + 1) + 7] ^= 1  # second round message
    with pytest.raises(hl.InvalidSumcheck, match="Consistency failure at round 1"):
        hl.sum_check_verify(g_kind, nv, degree, claim, hl.Keccak256Transcript.from_proof(bytes(bad)))


# ------------------------------------------------------------------ MultilinearKzg verify
def test_mkzg_open_verify_round_trip(hl, vp5):
    """commit/open by the oracle (CPU), verify by the product's pairing check (pcs/multilinear.rs:321-331 shape)"""
    rng = random.Random(11)
    pp = o_kzg.setup(SS)
    for nv in (1, 3, 5):
        poly = [rng.randrange(P) for _ in range(1 << nv)]
        point = [rng.randrange(P) for _ in range(nv)]
        comm = o_kzg.commit(pp.trim(nv), poly)
        t = OT()
        o_kzg.open_(pp.trim(nv), poly, point, t)
        proof, ev = t.into_proof(), evaluate(poly, point)
        hl.mkzg_verify(vp5, comm, point, ev, hl.Keccak256Transcript.from_proof(proof))
        with pytest.raises(hl.InvalidPcsOpen, match="Invalid multilinear KZG open"):
            hl.mkzg_verify(vp5, comm, point, (ev + 1) % P, hl.Keccak256Transcript.from_proof(proof))
        with pytest.raises(hl.InvalidPcsOpen):
            hl.mkzg_verify(vp5, curve.add(comm, curve.G1_GEN), point, ev, hl.Keccak256Transcript.from_proof(proof))
    with pytest.raises(hl.InvalidPcsParam, match="Too many variates"):
        hl.mkzg_verify(vp5, comm, [1] * 6, 0, hl.Keccak256Transcript.from_proof(bytes(64 * 6)))


def test_mkzg_batch_verify_golden(hl, vp5):
    g = GOLDEN["kzg_batch"]
    nv = g["num_vars"]
    t = hl.Keccak256Transcript.from_proof(bytes.fromhex(g["proof"]))
    comms = t.read_commitments(len(g["polys"]))
    pts = [t.squeeze_challenges(nv) for _ in range(2)]
    vals = t.read_field_elements(len(g["pairs"]))
    evals = [hl.Evaluation(p, q, v) for (p, q), v in zip(g["pairs"], vals)]
    hl.MultilinearKzg.batch_verify(vp5, nv, comms, pts, evals, t)
    assert t.remaining() == 0
    # a wrong claimed evaluation fails in the sum-check or in the final pairing
    t = hl.Keccak256Transcript.from_proof(bytes.fromhex(g["proof"]))
    comms = t.read_commitments(len(g["polys"]))
    pts = [t.squeeze_challenges(nv) for _ in range(2)]
    vals = t.read_field_elements(len(g["pairs"]))
    evals = [hl.Evaluation(p, q, v) for (p, q), v in zip(g["pairs"], vals)]
    evals[1].value = (evals[1].value + 1) % P
    with pytest.raises((hl.InvalidSumcheck, hl.InvalidPcsOpen)):
        hl.MultilinearKzg.batch_verify(vp5, nv, comms, pts, evals, t)


# ------------------------------------------------------------------ Lasso verify
def _table(hl, g):
    return hl.LassoTable.range(g["c"], g["l"]) if g["kind"] == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if g["kind"] == "and" else hl.SUBTABLE_XOR, g["c"], g["l"])


@pytest.mark.parametrize("idx", range(5))
def test_lasso_verify_golden(hl, vp5, idx):
    g = GOLDEN["lasso"][idx]
    proof = bytes.fromhex(g["proof"])
    hl.lasso_verify(vp5, _table(hl, g), g["n"], hl.Keccak256Transcript.from_proof(proof))
    rng = random.Random(idx)
    for _ in range(12):
        bad = bytearray(proof)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        with pytest.raises(hl.Error):
            hl.lasso_verify(vp5, _table(hl, g), g["n"], hl.Keccak256Transcript.from_proof(bytes(bad)))
    with pytest.raises(hl.Error):  # truncated
        hl.lasso_verify(vp5, _table(hl, g), g["n"], hl.Keccak256Transcript.from_proof(proof[:-32]))
    with pytest.raises(hl.InvalidSnark, match="trailing bytes"):
        hl.lasso_verify(vp5, _table(hl, g), g["n"], hl.Keccak256Transcript.from_proof(proof + bytes(32)))
    with pytest.raises(hl.Error):  # a different table
        other = hl.LassoTable.range(g["c"], g["l"]) if g["kind"] != "range" else hl.LassoTable.bitwise(
            hl.SUBTABLE_XOR, g["c"], g["l"])
        hl.lasso_verify(vp5, other, g["n"], hl.Keccak256Transcript.from_proof(proof))


# ------------------------------------------------------------------ HyperPlonk verify
class _FakePcs:
    """stands in for the GPU prover param: HyperPlonk.verify needs only commitments, which the oracle computes"""


@pytest.mark.parametrize("idx", range(3))
def test_hyperplonk_verify_golden(hl, vp5, idx):
    from halo2_lasso_amd import hyperplonk as g_hp
    from oracle.pyref import hyperplonk as o_hp
    g = GOLDEN["hyperplonk"][idx]
    nv = g["num_vars"]
    perms = [[tuple(c) for c in cyc] for cyc in g["permutations"]]
    pre = [I(a) for a in g["preprocess_polys"]]
    mk = g_hp.vanilla_plonk_with_lookup_circuit_info if g["with_lookup"] else g_hp.vanilla_plonk_circuit_info
    info = mk(nv, len(g["instances"][0]), pre, perms)
    o_mk = o_hp.vanilla_plonk_with_lookup_circuit_info if g["with_lookup"] else o_hp.vanilla_plonk_circuit_info
    o_pp = o_hp.preprocess(o_kzg.setup(SS), o_mk(nv, len(g["instances"][0]), pre, perms))
    vp = g_hp.HyperPlonkVerifierParam()
    vp.pcs, vp.num_vars, vp.info = vp5, nv, info
    vp.num_permutation_z_polys, vp.expression = g_hp.compose(info)
    vp.preprocess_comms, vp.permutation_comms = o_pp.preprocess_comms, o_pp.permutation_comms
    instances, proof = [I(a) for a in g["instances"]], bytes.fromhex(g["proof"])
    t = hl.Keccak256Transcript.from_proof(proof)
    g_hp.HyperPlonk.verify(vp, instances, t)
    assert t.remaining() == 0
    bad_inst = [list(instances[0])]
    bad_inst[0][1] = (bad_inst[0][1] + 1) % P
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(vp, bad_inst, hl.Keccak256Transcript.from_proof(proof))
    rng = random.Random(100 + idx)
    for _ in range(12):
        bad = bytearray(proof)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        with pytest.raises(hl.Error):
            g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(bytes(bad)))


# ------------------------------------------------------------------ Zeromorph (oracle proofs, product pairing verifier)
def test_zeromorph_oracle_round_trip_and_golden():
    from oracle.pyref import zeromorph as zm
    g = GOLDEN["zeromorph"]
    nv, s = g["num_vars"], int(g["s"], 16)
    pp, vp = zm.trim(zm.setup(s, g["param_size"]), 1 << nv)
    tabs, point = [I(a) for a in g["polys"]], I(g["point"])
    assert [hex(c) for c in zm.commit(pp, tabs[0])] == g["commitment"]
    t = OT()
    zm.open_(pp, tabs[0], point, evaluate(tabs[0], point), t)
    assert t.into_proof().hex() == g["open_proof"]
    zm.verify(vp, zm.commit(pp, tabs[0]), point, evaluate(tabs[0], point), OT(bytes.fromhex(g["open_proof"])))
    with pytest.raises(o_kzg.PcsError):
        zm.verify(vp, zm.commit(pp, tabs[1]), point, evaluate(tabs[0], point), OT(bytes.fromhex(g["open_proof"])))


def test_zeromorph_product_verifier_on_golden(hl):
    from oracle.pyref import zeromorph as zm
    g = GOLDEN["zeromorph"]
    nv, s = g["num_vars"], int(g["s"], 16)
    vp = hl.ZeromorphVerifierParam.setup(s, g["param_size"], 1 << nv)
    g1, g2, s_g2, s_off = vp.export()
    offset = g["param_size"] - (1 << nv)
    assert (g1, g2) == (curve.G1_GEN, o_pair.G2_GEN)
    assert s_g2 == o_pair.g2_mul(o_pair.G2_GEN, s) and s_off == o_pair.g2_mul(o_pair.G2_GEN, pow(s, offset, P))
    assert hl.ZeromorphVerifierParam.new(g1, g2, s_g2, s_off).export() == (g1, g2, s_g2, s_off)
    tabs, point = [I(a) for a in g["polys"]], I(g["point"])
    comm = tuple(int(c, 16) for c in g["commitment"])
    ev = evaluate(tabs[0], point)
    proof = bytes.fromhex(g["open_proof"])
    hl.Zeromorph.verify(vp, comm, point, ev, hl.Keccak256Transcript.from_proof(proof))
    with pytest.raises(hl.InvalidPcsOpen, match="Invalid Zeromorph KZG open"):
        hl.Zeromorph.verify(vp, comm, point, (ev + 1) % P, hl.Keccak256Transcript.from_proof(proof))
    wrong_offset = hl.ZeromorphVerifierParam.setup(s, g["param_size"] + 1, 1 << nv)
    with pytest.raises(hl.InvalidPcsOpen):
        hl.Zeromorph.verify(wrong_offset, comm, point, ev, hl.Keccak256Transcript.from_proof(proof))
    # batch
    r = hl.Keccak256Transcript.from_proof(bytes.fromhex(g["batch_proof"]))
    comms = r.read_commitments(3)
    pts = [r.squeeze_challenges(nv) for _ in range(2)]
    vals = r.read_field_elements(len(g["pairs"]))
    assert vals == [evaluate(tabs[p], pts[q]) for p, q in g["pairs"]]
    hl.Zeromorph.batch_verify(vp, nv, comms, pts, [hl.Evaluation(p, q, v) for (p, q), v in zip(g["pairs"], vals)], r)
    assert r.remaining() == 0


@pytest.mark.parametrize("num_vars", [3, 5])
def test_hyperplonk_verify_two_phase_circuit(hl, num_vars):
    """multi-phase circuits (hyperplonk.rs:309-316): the oracle proves a two-phase circuit whose second-phase witness
    depends on the first phase's challenge; lh_hyperplonk_verify_phases accepts it and rejects a tampered proof"""
    from halo2_lasso_amd import hyperplonk as g_hp, expression as g_ex
    from oracle.pyref import hyperplonk as o_hp, expression as o_ex
    from test_gpu_hyperplonk import _two_phase_circuit
    rng = random.Random(900 + num_vars)
    ss = [rng.randrange(1, P) for _ in range(num_vars)]
    o_info, instances, synth = _two_phase_circuit(o_ex, o_hp.CircuitInfo, num_vars, random.Random(num_vars), None)
    o_pp = o_hp.preprocess(o_kzg.setup(ss), o_info)
    t = OT()
    o_hp.prove(o_pp, instances, synth, t)
    proof = t.into_proof()
    g_info, _, _ = _two_phase_circuit(g_ex, g_hp.PlonkishCircuitInfo, num_vars, random.Random(num_vars), None)
    vp = g_hp.HyperPlonkVerifierParam()
    vp.pcs, vp.num_vars, vp.info = hl.MultilinearKzgVerifierParams.setup(ss), num_vars, g_info
    vp.num_permutation_z_polys, vp.expression = g_hp.compose(g_info)
    vp.preprocess_comms, vp.permutation_comms = o_pp.preprocess_comms, o_pp.permutation_comms
    g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(proof))
    bad = bytearray(proof)
    bad[3 * 64 + 40] ^= 1  # inside the m / h / z commitments or the first round message
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(bytes(bad)))
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(vp, [[v + 1 for v in instances[0]]], hl.Keccak256Transcript.from_proof(proof))


@pytest.mark.parametrize("num_vars", [3, 4])
def test_hyperplonk_verify_two_phase_circuit_over_zeromorph(hl, num_vars):
    """the same over Zeromorph (lh_hyperplonk_verify_phases_zeromorph): a proof of the specification is accepted, a
    tampered one and wrong instances are rejected"""
    from halo2_lasso_amd import hyperplonk as g_hp, expression as g_ex
    from oracle.pyref import hyperplonk as o_hp, expression as o_ex, zeromorph as o_zm
    from test_gpu_hyperplonk import _two_phase_circuit
    s = random.Random(940 + num_vars).randrange(1, P)
    o_info, instances, synth = _two_phase_circuit(o_ex, o_hp.CircuitInfo, num_vars, random.Random(num_vars), None)
    o_pp = o_hp.preprocess(o_zm.trim(o_zm.setup(s, 1 << num_vars), 1 << num_vars), o_info, o_zm)
    t = OT()
    o_hp.prove(o_pp, instances, synth, t)
    proof = t.into_proof()
    g_info, _, _ = _two_phase_circuit(g_ex, g_hp.PlonkishCircuitInfo, num_vars, random.Random(num_vars), None)
    vp = g_hp.HyperPlonkVerifierParam()
    vp.pcs, vp.num_vars, vp.info = hl.ZeromorphVerifierParam.setup(s, 1 << num_vars, 1 << num_vars), num_vars, g_info
    vp.num_permutation_z_polys, vp.expression = g_hp.compose(g_info)
    vp.preprocess_comms, vp.permutation_comms = o_pp.preprocess_comms, o_pp.permutation_comms
    r = hl.Keccak256Transcript.from_proof(proof)
    g_hp.HyperPlonk.verify(vp, instances, r)
    assert r.remaining() == 0
    bad = bytearray(proof)
    bad[3 * 64 + 40] ^= 1
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(bytes(bad)))
    with pytest.raises(hl.Error):
        g_hp.HyperPlonk.verify(vp, [[v + 1 for v in instances[0]]], hl.Keccak256Transcript.from_proof(proof))


LASSO_CIRCUITS = [("range", 2, 2, 4), ("and", 2, 4, 5), ("xor", 2, 4, 4)]


def _lasso_circuit(hl, kind, c, l, num_vars, seed):
    """the configs[4] stand-in circuit (vanilla gates + one Lasso lookup) in the oracle's and the product's description"""
    from halo2_lasso_amd import hyperplonk as g_hp
    from oracle.pyref import hyperplonk as o_hp, lasso as o_lasso
    spec = o_lasso.range_table(c, l) if kind == "range" else o_lasso.bitwise_table(
        o_lasso.SUBTABLE_AND if kind == "and" else o_lasso.SUBTABLE_XOR, c, l)
    table = hl.LassoTable.range(c, l) if kind == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, c, l)
    o_info, instances, witness = o_hp.rand_vanilla_plonk_with_lasso_circuit(num_vars, random.Random(seed), spec)
    g_info = g_hp.vanilla_plonk_with_lasso_circuit_info(num_vars, len(instances[0]), o_info.preprocess_polys,
                                                        o_info.permutations, table)
    return o_info, g_info, instances, witness


@pytest.mark.parametrize("kind,c,l,num_vars", LASSO_CIRCUITS)
def test_hyperplonk_verify_lasso_lookup(hl, kind, c, l, num_vars):
    """Lasso as HyperPlonk's lookup argument (specification: oracle/pyref/hyperplonk.py LassoLookup): a proof made by the
    oracle is accepted by lh_hyperplonk_verify, tampered proofs and wrong instances are rejected"""
    from halo2_lasso_amd import hyperplonk as g_hp
    from oracle.pyref import hyperplonk as o_hp
    rng = random.Random(70 + num_vars)
    ss = [rng.randrange(1, P) for _ in range(num_vars)]
    o_info, g_info, instances, witness = _lasso_circuit(hl, kind, c, l, num_vars, 7 + num_vars)
    o_pp = o_hp.preprocess(o_kzg.setup(ss), o_info)
    t = OT()
    o_hp.prove(o_pp, instances, lambda r, ch: witness, t)
    proof = t.into_proof()
    o_hp.verify(o_pp, instances, OT(proof))
    vp = g_hp.HyperPlonkVerifierParam()
    vp.pcs, vp.num_vars, vp.info = hl.MultilinearKzgVerifierParams.setup(ss), num_vars, g_info
    vp.num_permutation_z_polys, vp.expression = g_hp.compose(g_info)
    vp.preprocess_comms, vp.permutation_comms = o_pp.preprocess_comms, o_pp.permutation_comms
    g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(proof))
    for pos in (len(proof) // 3, len(proof) // 2, len(proof) - 40):
        bad = bytearray(proof)
        bad[pos] ^= 1
        with pytest.raises(hl.Error):
            g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(bytes(bad)))


@pytest.mark.parametrize("idx", range(2))
def test_hyperplonk_lasso_verify_golden(hl, vp5, idx):
    """the committed HyperPlonk + Lasso proofs through the host verifier (needs the preprocess / permutation commitments:
    recomputed with the oracle's commit)"""
    from halo2_lasso_amd import hyperplonk as g_hp
    from oracle.pyref import hyperplonk as o_hp, lasso as o_lasso
    g = GOLDEN["hyperplonk_lasso"][idx]
    nv = g["num_vars"]
    I = lambda xs: [int(x, 16) for x in xs]
    spec = o_lasso.range_table(g["c"], g["l"]) if g["kind"] == "range" else o_lasso.bitwise_table(
        o_lasso.SUBTABLE_AND, g["c"], g["l"])
    table = hl.LassoTable.range(g["c"], g["l"]) if g["kind"] == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND, g["c"], g["l"])
    perms = [[tuple(c) for c in cyc] for cyc in g["permutations"]]
    pre = [I(a) for a in g["preprocess_polys"]]
    instances = [I(a) for a in g["instances"]]
    o_pp = o_hp.preprocess(o_kzg.setup(SS), o_hp.vanilla_plonk_with_lasso_circuit_info(nv, len(instances[0]), pre, perms, spec))
    g_info = g_hp.vanilla_plonk_with_lasso_circuit_info(nv, len(instances[0]), pre, perms, table)
    vp = g_hp.HyperPlonkVerifierParam()
    vp.pcs, vp.num_vars, vp.info = vp5, nv, g_info
    vp.num_permutation_z_polys, vp.expression = g_hp.compose(g_info)
    vp.preprocess_comms, vp.permutation_comms = o_pp.preprocess_comms, o_pp.permutation_comms
    g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(bytes.fromhex(g["proof"])))
