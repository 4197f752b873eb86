"""CPU tests of the oracle (oracle/pyref): public KATs, the reference's round-trip properties restated
against the oracle's own verifier, and the committed golden vectors (regression pin).

Parity with the reference is UNPINNED at byte level: the reference has no golden vectors and cannot be
built here.  What pins the oracle: (1) public constants and KATs, (2) the
reference's test properties (prove -> verify -> re-evaluate), (3) agreement of independent
implementations (this oracle, the C++ oracle, the HIP path) on tests/golden/vectors.json.
"""
import json
import os
import random

import pytest

from oracle.pyref import curve, field, gkr, kzg, lasso, sum_check as sc, expression as ex
from oracle.pyref.field import R_MOD as P, Q_MOD
from oracle.pyref.keccak import keccak256
from oracle.pyref.poly import evaluate, eq_xy, eq_xy_eval, fix_var
from oracle.pyref.transcript import Keccak256Transcript as T, TranscriptError

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
I = lambda xs: [int(x, 16) for x in xs]


# ------------------------------------------------------------------ KATs
def test_field_constants():
    assert P == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert Q_MOD == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert (P - 1).bit_length() == 254  # reference util/arithmetic.rs:202-205 `field_size::<Fr>() == 254`
    # Montgomery constants quoted in SURVEY.md §8 a1/a2 (halo2curves' [u64;4] representation)
    assert field.MONT_R % P == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert (-pow(P, -1, 1 << 64)) % (1 << 64) == 0xc2e1f593efffffff
    assert (-pow(Q_MOD, -1, 1 << 64)) % (1 << 64) == 0x87d20782e4866389
    assert field.from_mont_bytes(field.to_mont_bytes(12345)) == 12345


def test_keccak256_kats():
    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # Ethereum's well-known selector hash: multi-word input through the sponge
    assert keccak256(b"transfer(address,uint256)").hex()[:8] == "a9059cbb"


def test_curve_generator_and_group_law():
    G = curve.G1_GEN
    assert curve.is_on_curve(G) and G == (1, 2)
    assert curve.mul(G, P) is None                      # group order is r
    assert curve.mul(G, P - 1) == curve.neg(G)
    assert curve.add(curve.mul(G, 5), curve.mul(G, 7)) == curve.mul(G, 12)
    assert curve.add(G, G) == curve.mul(G, 2)           # doubling through add
    assert curve.add(G, curve.neg(G)) is None
    # 2G on BN254 (public value)
    assert curve.mul(G, 2) == (
        0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
        0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)


def test_msm_matches_naive():
    rng = random.Random(3)
    fb = curve.FixedBase()
    bases = [fb.mul(rng.randrange(1, P)) for _ in range(40)] + [None]
    scalars = [rng.randrange(P) for _ in range(41)]
    acc = None
    for s, b in zip(scalars, bases):
        acc = curve.add(acc, curve.mul(b, s))
    assert curve.msm(scalars, bases) == acc


def test_transcript_encoding():
    """write = hash LE repr, stream BE repr; squeeze chains on the digest (transcript.rs:126-165)."""
    t = T()
    t.write_field_element(1)
    assert t.into_proof() == (1).to_bytes(32, "big")
    c1 = t.squeeze_challenge()
    h = keccak256((1).to_bytes(32, "little"))
    assert c1 == int.from_bytes(h, "little") % P
    assert t.squeeze_challenge() == int.from_bytes(keccak256(h), "little") % P
    t.write_commitment(curve.G1_GEN)
    assert t.into_proof()[32:] == (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    with pytest.raises(TranscriptError):
        T().write_commitment(None)  # identity (transcript.rs:172-179)
    r = T(t.into_proof())
    assert r.read_field_element() == 1 and r.squeeze_challenge() == c1
    with pytest.raises(TranscriptError):
        T((P).to_bytes(32, "big")).read_field_element()  # non-canonical


# ------------------------------------------------------------------ reference round-trip properties
def test_fix_var_vs_evaluate():
    """poly/multilinear.rs:663-687"""
    rng = random.Random(4)
    for nv in range(1, 8):
        evals = [rng.randrange(P) for _ in range(1 << nv)]
        x = [rng.randrange(P) for _ in range(nv)]
        cur = evals
        for x_i in x:
            cur = fix_var(cur, x_i)
        assert cur == [evaluate(evals, x)]
        eq = eq_xy(x)
        assert sum(a * b for a, b in zip(eq, evals)) % P == evaluate(evals, x)
        y = [rng.randrange(P) for _ in range(nv)]
        assert evaluate(eq, y) == eq_xy_eval(x, y)


@pytest.mark.parametrize("prover,msg", [(sc.EvaluationsProver, sc.Evaluations), (sc.CoefficientsProver, sc.Coefficients)])
def test_sum_check_round_trip(prover, msg):
    """run_sum_check (piop/sum_check.rs:140-177): prove -> verify -> expression at the final point."""
    rng = random.Random(5)
    for nv in range(1, 7):
        tabs = [[rng.randrange(P) for _ in range(1 << nv)] for _ in range(2)]
        ys = [[rng.randrange(P) for _ in range(nv)] for _ in range(2)]
        expr = ex.sum_exprs(ex.EqXY(j) * ex.Poly(j) * (j + 3) for j in range(2))
        claim = sum((j + 3) * evaluate(tabs[j], ys[j]) for j in range(2)) % P
        t = T()
        x, evals = sc.prove(prover, nv, sc.VirtualPolynomial(expr, tabs, [], ys), claim, t)
        final, vx = sc.verify(msg, nv, 2, claim, T(t.into_proof()))
        assert vx == x and evals == [evaluate(tab, x) for tab in tabs]
        assert final == ex.evaluate_fe(expr, [eq_xy_eval(x, y) for y in ys], evals, [])
        bad = bytearray(t.into_proof())
        bad[5] ^= 1
        with pytest.raises(sc.SumCheckError):
            sc.verify(msg, nv, 2, claim, T(bytes(bad)))


def test_fractional_sum_check_round_trip():
    """fractional_sum_check.rs:327-370"""
    rng = random.Random(6)
    B = 3
    for nv in range(1, 7):
        tabs = [[rng.randrange(P) for _ in range(1 << nv)] for _ in range(2 * B)]
        t = T()
        gkr.prove_fractional_sum_check([None] * B, [None] * B, tabs[:B], tabs[B:], t)
        p_xs, q_xs, x = gkr.verify_fractional_sum_check(nv, [None] * B, [None] * B, T(t.into_proof()))
        for tab, e in zip(tabs, p_xs + q_xs):
            assert evaluate(tab, x) == e
    # the roots are the fraction sums: sum_i p_i/q_i == P0/Q0
    p, q = tabs[0], tabs[B]
    t = T()
    gkr.prove_fractional_sum_check([None], [None], [p], [q], t)
    rd = T(t.into_proof())
    p0, q0 = rd.read_field_element(), rd.read_field_element()
    assert sum(a * pow(b, -1, P) for a, b in zip(p, q)) % P == p0 * pow(q0, -1, P) % P


def test_grand_product_round_trip():
    rng = random.Random(7)
    for sizes in ([2], [4, 2], [8, 32, 8], [16, 16, 4, 4]):
        vs = [[rng.randrange(1, P) for _ in range(s)] for s in sizes]
        t = T()
        roots, claims = gkr.prove_grand_product(vs, t)
        vroots, vclaims = gkr.verify_grand_product([s.bit_length() - 1 for s in sizes], T(t.into_proof()))
        assert (roots, claims) == (vroots, vclaims)
        for v, root, (cl, pt) in zip(vs, roots, claims):
            prod = 1
            for e in v:
                prod = prod * e % P
            assert prod == root and evaluate(v, pt) == cl


@pytest.fixture(scope="module")
def pp5():
    return kzg.setup(I(GOLDEN["srs"]["ss"]))


def test_kzg_setup_structure(pp5):
    """eqs[k][b] = eq_k(b; s) * G with bit i <-> s_i (kzg.rs:174-194)."""
    ss = pp5.ss
    assert [[hex(c) for c in pt] for pt in pp5.eqs[3]] == GOLDEN["srs"]["eqs_level3"]
    for k in (0, 1, 3):
        scal = eq_xy(ss[:k]) if k else [1]
        assert pp5.eqs[k] == [curve.mul(curve.G1_GEN, e) for e in scal]


def test_kzg_commit_open_verify(pp5):
    """run_commit_open_verify (pcs/multilinear.rs:293-332)"""
    rng = random.Random(8)
    for nv in range(1, 6):
        poly = [rng.randrange(P) for _ in range(1 << nv)]
        ppn = pp5.trim(nv)
        t = T()
        t.write_commitment(kzg.commit(ppn, poly))
        pt = t.squeeze_challenges(nv)
        e = evaluate(poly, pt)
        t.write_field_element(e)
        assert kzg.open_(ppn, poly, pt, t) == e
        v = T(t.into_proof())
        kzg.verify(ppn, v.read_commitment(), v.squeeze_challenges(nv), v.read_field_element(), v)
        v = T(t.into_proof())
        with pytest.raises(kzg.PcsError):
            kzg.verify(ppn, v.read_commitment(), v.squeeze_challenges(nv), (v.read_field_element() + 1) % P, v)
    with pytest.raises(kzg.PcsError):
        kzg.commit(pp5, [1] * 64)  # too many variates


def test_lasso_round_trip_and_tamper(pp5):
    rng = random.Random(9)
    spec = lasso.range_table(2, 3)
    dims = [[rng.randrange(8) for _ in range(16)] for _ in range(2)]
    t = T()
    lasso.prove(pp5, spec, dims, t)
    proof = t.into_proof()
    lasso.verify(pp5, spec, 4, T(proof))
    for pos in range(0, len(proof), 97):
        bad = bytearray(proof)
        bad[pos] ^= 0x01
        with pytest.raises(Exception):
            lasso.verify(pp5, spec, 4, T(bytes(bad)))
    # a wrong witness (E not the table value) must not verify: forge the output column
    w = lasso.witness(spec, dims)
    assert w["a"][0] == dims[0][0] + 8 * dims[1][0]
    assert all(sum(c) == 16 for c in w["final_cts"])
    assert lasso.subtable_mle_eval(lasso.SUBTABLE_AND, [1, 0, 1, 1]) == 1  # x=0b11, y=0b01 -> 1


# ------------------------------------------------------------------ golden vectors (regression pin)
def test_golden_sum_check_eval():
    g = GOLDEN["sum_check_eval"]
    c = I(g["coeffs"])
    expr = ex.EqXY(0) * (ex.Poly(0) * ex.Poly(1) * c[0] + ex.Poly(2) * c[1])
    t = T()
    x, ev = sc.prove(sc.EvaluationsProver, g["num_vars"],
                     sc.VirtualPolynomial(expr, [I(a) for a in g["tables"]], [], [I(g["y"])]), int(g["claim"], 16), t)
    assert (x, ev, t.into_proof().hex()) == (I(g["x"]), I(g["evals"]), g["proof"])


def test_golden_sum_check_coeff():
    g = GOLDEN["sum_check_coeff"]
    s = I(g["scalars"])
    expr = ex.sum_exprs(ex.EqXY(j) * ex.Poly(j) * s[j] for j in range(2))
    t = T()
    x, ev = sc.prove(sc.CoefficientsProver, g["num_vars"],
                     sc.VirtualPolynomial(expr, [I(a) for a in g["tables"]], [], [I(a) for a in g["ys"]]),
                     int(g["claim"], 16), t)
    assert (x, ev, t.into_proof().hex()) == (I(g["x"]), I(g["evals"]), g["proof"])


def test_golden_gkr():
    g = GOLDEN["frac_gkr"]
    t = T()
    out = gkr.prove_fractional_sum_check([None] * 2, [None] * 2, [I(a) for a in g["ps"]], [I(a) for a in g["qs"]], t)
    assert out == (I(g["p_xs"]), I(g["q_xs"]), I(g["x"])) and t.into_proof().hex() == g["proof"]
    g = GOLDEN["grand_product"]
    t = T()
    roots, claims = gkr.prove_grand_product([I(v) for v in g["leaves"]], t)
    assert roots == I(g["roots"]) and t.into_proof().hex() == g["proof"]
    assert [[hex(c), [hex(v) for v in p]] for c, p in claims] == g["claims"]


def test_golden_kzg_batch(pp5):
    g = GOLDEN["kzg_batch"]
    nv, polys = g["num_vars"], [I(a) for a in g["polys"]]
    t = T()
    kzg.batch_commit_and_write(pp5, polys, t)
    pts = [t.squeeze_challenges(nv) for _ in range(2)]
    vals = [evaluate(polys[p], pts[q]) for p, q in g["pairs"]]
    t.write_field_elements(vals)
    kzg.batch_open(pp5.trim(nv), nv, polys, pts, [kzg.Evaluation(p, q, v) for (p, q), v in zip(g["pairs"], vals)], t)
    assert t.into_proof().hex() == g["proof"]


@pytest.mark.parametrize("idx", range(5))  # 3, 4: identically zero columns (identity commitments)
def test_golden_lasso(pp5, idx):
    g = GOLDEN["lasso"][idx]
    spec = lasso.range_table(g["c"], g["l"]) if g["kind"] == "range" else lasso.bitwise_table(
        lasso.SUBTABLE_AND if g["kind"] == "and" else lasso.SUBTABLE_XOR, g["c"], g["l"])
    t = T()
    lasso.prove(pp5, spec, g["dims"], t)
    assert t.into_proof().hex() == g["proof"]
    lasso.verify(pp5, spec, g["n"], T(bytes.fromhex(g["proof"])))


# ------------------------------------------------------------------ public vectors that pin the primitives
def test_keccak_sponge_against_hashlib_sha3():
    """The oracle's sponge with the FIPS-202 domain byte IS SHA3-256: hashlib pins the permutation, the rate and the
    multi-block absorb/pad logic on every message length around the 136-byte block boundaries and on a long message;
    the legacy-Keccak domain byte (0x01) the reference uses is pinned by the published Keccak-256 vectors."""
    import hashlib
    from oracle.pyref.keccak import Keccak256, keccak256
    rng = random.Random(5)
    for n in list(range(0, 300)) + [407, 408, 409, 1000, 10_000]:
        msg = bytes(rng.randrange(256) for _ in range(n))
        h = Keccak256(pad=0x06)
        h.update(msg[:n // 3])
        h.update(msg[n // 3:])
        assert h.finalize_reset() == hashlib.sha3_256(msg).digest(), n
    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert keccak256(b"The quick brown fox jumps over the lazy dog").hex() == \
        "4d741b6f1eb29cb2a9b9911c82f56fa8d73b04959d3d9d222895df6c0b28aa15"


def test_bn254_public_points():
    """alt_bn128 (EIP-196 / EIP-197) public values: 2G, 3G, the order of G1, the G2 generator on its twist, and the
    precompile's canonical pairing checks e(P, Q) e(-P, Q) = 1 and e(aP, bQ) = e(abP, Q)"""
    G = curve.G1_GEN
    two_g = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
             9918110051302171585080402603319702774565515993150576347155970296011118125764)
    three_g = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
               19321533766552368860946552437480515441416830039777911637913418824951667761761)
    assert curve.add(G, G) == curve.mul(G, 2) == two_g
    assert curve.add(two_g, G) == curve.mul(G, 3) == three_g
    assert curve.mul(G, field.R_MOD) is None and curve.mul(G, field.R_MOD - 1) == curve.neg(G)
    assert curve.msm([5, field.R_MOD - 2], [G, two_g]) == G  # 5 G - 2 (2 G)
