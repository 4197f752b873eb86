"""GPU parity at the sizes where the STREAMING kernels run (round-1 parity tests stop below the thresholds at which
`sc_round_kernel<D,BIND>` replaces the LDS round kernel, the radix sort takes several passes, MSM windows reach
13-17 bits with continuation levels, `eq_outer_kernel` runs and the expression rounds are compiled at run time).

The checker is the multithreaded C++ oracle (oracle/cpu: the reference's algorithms, SOS Montgomery on 4 x u64,
Jacobian points, evaluate-then-bind, Pippenger per thread chunk), which finishes 2^16..2^20 in about a second on the
GPU box's host cores; it is itself pinned to the Python oracle and the golden vectors by tests/test_oracle_cpu.py.
Every comparison is on bytes: proof streams, challenges and evaluations as Montgomery limbs, affine points.
Inputs are uniformly random Montgomery limb patterns below the modulus (a uniform field element), handed unchanged to
both sides.  The last test proves BASELINE.json configs[2] (2^24 AND lookups) and puts the proof through the verifier.
"""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import cpu_oracle as co
from oracle.pyref.field import R_MOD as P

pytestmark = pytest.mark.gpu

TOP_LIMB = 0x30644E72E131A029  # top 64 bits of r: limb patterns with a smaller top limb are < r


def rand_mont(rng, n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] = rng.integers(0, TOP_LIMB, size=n, dtype=np.uint64)
    return a.tobytes()


def mle(hl, ctx, raw):
    n = len(raw) // 32
    return hl.MultilinearPolynomial(ctx, ctx.upload(raw), n.bit_length() - 1)


def trapdoor(nv, seed):
    rng = random.Random(seed)
    return [rng.randrange(1, P) for _ in range(nv)]


@pytest.fixture(scope="module")
def srs17(hl, ctx):
    """(ss, GPU params, the C++ oracle's flat SRS) for 17 variables; the two setups must agree on all 2^18 - 1 points"""
    ss = trapdoor(17, 1700)
    pp = hl.MultilinearKzg.setup(ctx, ss)
    flat = co.setup(ss)
    assert pp.eqs_bytes() == flat, "SRS of the GPU setup differs from the oracle's"
    return ss, pp, flat


# ------------------------------------------------------------------ a': Lasso
def _dims(rng, table, n, skew):
    cols = []
    for j in range(table.c):
        d = rng.integers(0, 1 << table.l, size=1 << n, dtype=np.uint32)
        if skew and j == 0:  # a third of the lookups hit one address, another third a handful: long runs for the
            hot = rng.random(1 << n)  # counters' sort, the MSM's continuation levels and one very hot bucket
            d[hot < 0.33] = 7
            d[(hot >= 0.33) & (hot < 0.66)] = rng.integers(0, 5, size=int(((hot >= 0.33) & (hot < 0.66)).sum()), dtype=np.uint32)
        cols.append(d)
    return cols


@pytest.mark.parametrize("kind,n,skew", [("range", 14, False), ("and", 14, False), ("xor", 14, True),
                                         ("range", 17, True), ("and", 17, False), ("xor", 17, False)])
def test_lasso_matches_cpp_oracle(hl, ctx, srs17, kind, n, skew):
    ss, pp, flat = srs17
    table = hl.LassoTable.range(2, 16) if kind == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, 4, 16)
    rng = np.random.default_rng(1000 + n + len(kind))
    dims = _dims(rng, table, n, skew)
    ot = co.Transcript()
    co.lasso_prove(ot, flat, 17, table.to_c(), n, [d.tobytes() for d in dims])
    t = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, [ctx.upload(d.tobytes()) for d in dims], t)
    proof = t.into_proof()
    assert proof == ot.into_proof()
    hl.lasso_verify(hl.MultilinearKzgVerifierParams.setup(ss), table, n, hl.Keccak256Transcript.from_proof(proof))


@pytest.mark.parametrize("n", [14, 17])
def test_lasso_nonlinear_g_matches_cpp_oracle(hl, ctx, srs17, n):
    """a table whose g has product terms (test_gpu_parity._nonlinear_tables) at streaming sizes: the output column's
    full-width MSM, the degree-4 Surge rounds over six read columns, the opening over field-element views"""
    from test_gpu_parity import _nonlinear_tables
    ss, pp, flat = srs17
    spec, table = _nonlinear_tables(hl, 4, 16)
    rng = np.random.default_rng(1700 + n)
    dims = _dims(rng, table, n, n == 17)
    ot = co.Transcript()
    co.lasso_prove(ot, flat, 17, table.to_c(), n, [d.tobytes() for d in dims])
    t = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, [ctx.upload(d.tobytes()) for d in dims], t)
    proof = t.into_proof()
    assert proof == ot.into_proof()
    hl.lasso_verify(hl.MultilinearKzgVerifierParams.setup(ss), table, n, hl.Keccak256Transcript.from_proof(proof))


def test_route_options_change_the_route_not_the_bytes(hl, srs17):
    """lh_ctx_set_option / lh_lasso_last_route (include/lasso_hip.h): every route switch yields the same proof bytes and
    the route report shows that the other code ran; unknown names are LH_ERR_ARG"""
    ss, pp0, flat = srs17
    n = 17
    table = hl.LassoTable.range(2, 16)
    rng = np.random.default_rng(1717)
    dims = _dims(rng, table, n, False)
    ot = co.Transcript()
    co.lasso_prove(ot, flat, 17, table.to_c(), n, [d.tobytes() for d in dims])
    want = ot.into_proof()
    ctx = hl.Context(0)                       # options are per ctx: a fresh one, sharing the session SRS (device memory)
    pp = hl.MultilinearKzgParams(ctx, pp0.h)
    try:
        bufs = [ctx.upload(d.tobytes()) for d in dims]

        def prove():
            t = hl.Keccak256Transcript()
            hl.lasso_prove(pp, table, n, bufs, t)
            return t.into_proof(), hl.lasso_last_route(ctx)
        proof, route = prove()
        assert proof == want and route["open_small_depth"] >= 1 and route["eq_factored_rounds"] > 0 and route["resident_tails"] > 0
        assert route["open_precommit"] == 1      # the column-wise commitments came from the helper ctx, beside the sum-checks
        assert hl.get_option(ctx, "open_small_min_vars") == 21 and hl.get_option(ctx, "sc_eq_factoring") == 1
        for smallest in (0, 18):                  # never / only proofs of >= 2^18 lookups: this one commits in the opening
            hl.set_option(ctx, "open_precommit", smallest)
            proof, route = prove()
            assert proof == want and route["open_precommit"] == 0 and route["open_small_depth"] >= 1
        hl.set_option(ctx, "open_precommit", 1)
        hl.set_option(ctx, "open_small_min_vars", 64)
        proof, route = prove()
        assert proof == want and route["open_small_depth"] == 0 and route["open_small_passes"] == 0
        hl.set_option(ctx, "sc_eq_factoring", 0)
        proof, route = prove()
        assert proof == want and route["eq_factored_rounds"] == 0 and route["rw_leaf_rounds"] == 0 and route["standard_rounds"] > 0
        hl.set_option(ctx, "sc_tail", 0)
        hl.set_option(ctx, "lasso_pack_ts", 0)
        proof, route = prove()
        assert proof == want and route["resident_tails"] == 0 and route["packed_ts_pairs"] == 0
        assert route["window_table_jobs"] == 0
        hl.set_option(ctx, "msm_window_tables", 17)   # window tables of the SRS levels: one bucket set per quotient job
        proof, route = prove()
        assert proof == want and route["window_table_jobs"] >= 10, route
        with pytest.raises(hl.Error):
            hl.set_option(ctx, "no_such_option", 1)
    finally:
        pp.h = None                            # the SRS belongs to the session fixture


# ------------------------------------------------------------------ a5-a8: both sum-check provers at 2^18
def _sop_struct(hl, terms, global_eq):
    return hl.SumOfProducts(terms, global_eq=global_eq).to_c()


def test_sum_check_evaluations_2p18(hl, ctx):
    """eq * (c0 p0 p1 + c1 p2 p3 + c2 p0 p3 p4), degree 4: sc_round_kernel<4,*> from 2^17 pairs down through the
    LDS kernel and the resident tail; challenges, evaluations and proof bytes against the C++ oracle"""
    nv = 18
    rng = np.random.default_rng(18)
    prng = random.Random(18)
    tables = [rand_mont(rng, 1 << nv) for _ in range(5)]
    y = [prng.randrange(P) for _ in range(nv)]
    c = [prng.randrange(P) for _ in range(3)]
    # a claim that is NOT the true sum: the reference still sends the true p(1..D) with p(0) = claim - p(1); the device
    # path's eq factoring derives q(0) from the claim, so it must notice in round 0 and take the standard path
    claim = prng.randrange(P)
    terms = [(c[0], [0, 1]), (c[1], [2, 3]), (c[2], [0, 3, 4])]
    ot = co.Transcript()
    ox, oev = co.sumcheck_prove(ot, 0, nv, _sop_struct(hl, terms, 0), tables, [y], claim)
    t = hl.Keccak256Transcript()
    x, ev = hl.ClassicSumCheck.prove(ctx, hl.EvaluationsProver, nv, hl.SumOfProducts(terms, global_eq=0),
                                     [mle(hl, ctx, r) for r in tables], [y], claim, t)
    assert (x, ev) == (ox, oev) and t.into_proof() == ot.into_proof()


def test_sum_check_coefficients_2p18(hl, ctx):
    """sum_j s_j eq_j poly_j (batch_open's expression) with the true claim; the verifier accepts"""
    nv = 18
    rng = np.random.default_rng(19)
    prng = random.Random(19)
    tables = [rand_mont(rng, 1 << nv) for _ in range(4)]
    ys = [[prng.randrange(P) for _ in range(nv)] for _ in range(4)]
    s = [prng.randrange(P) for _ in range(4)]
    polys = [mle(hl, ctx, r) for r in tables]
    claim = sum(sj * p.evaluate(yj) for sj, p, yj in zip(s, polys, ys)) % P
    terms = [(s[j], [4 + j, j]) for j in range(4)]
    ot = co.Transcript()
    ox, oev = co.sumcheck_prove(ot, 1, nv, _sop_struct(hl, terms, -1), tables, ys, claim)
    t = hl.Keccak256Transcript()
    x, ev = hl.ClassicSumCheck.prove(ctx, hl.CoefficientsProver, nv, hl.SumOfProducts(terms), polys, ys, claim, t)
    assert (x, ev) == (ox, oev) and t.into_proof() == ot.into_proof()
    from oracle.pyref import kzg as o_kzg
    final, vx = hl.sum_check_verify(hl.CoefficientsProver, nv, 2, claim, hl.Keccak256Transcript.from_proof(t.into_proof()))
    assert vx == x and final == sum(s[j] * o_kzg.eq_xy_eval(x, ys[j]) * ev[j] for j in range(4)) % P


# ------------------------------------------------------------------ a9: GKR
def test_fractional_sum_check_x3_2p16(hl, ctx):
    """the reference's GKR test shape (fractional_sum_check.rs:327-370: three batched fractions) at 16 variables"""
    nv = 16
    rng = np.random.default_rng(16)
    ps = [rand_mont(rng, 1 << nv) for _ in range(3)]
    qs = [rand_mont(rng, 1 << nv) for _ in range(3)]
    ot = co.Transcript()
    o = co.frac_gkr_prove(ot, ps, qs)
    t = hl.Keccak256Transcript()
    g = hl.prove_fractional_sum_check(ctx, [None] * 3, [None] * 3, [mle(hl, ctx, r) for r in ps],
                                      [mle(hl, ctx, r) for r in qs], t)
    assert tuple(g) == tuple(o) and t.into_proof() == ot.into_proof()


def test_grand_product_mixed_depth_large(hl, ctx):
    """the Lasso memory-check shape: trees of 2^18, 2^18, 2^12, 2^16 leaves in one batch"""
    rng = np.random.default_rng(17)
    leaves = [rand_mont(rng, 1 << nv) for nv in (18, 18, 12, 16)]
    ot = co.Transcript()
    o_roots, o_claims = co.grand_product_prove(ot, leaves)
    t = hl.Keccak256Transcript()
    roots, claims = hl.prove_grand_product(ctx, [mle(hl, ctx, r) for r in leaves], t)
    assert roots == o_roots and claims == o_claims and t.into_proof() == ot.into_proof()


# ------------------------------------------------------------------ a10: variable_base_msm
@pytest.fixture(scope="module")
def bases20(hl, ctx):
    pp = hl.MultilinearKzg.setup(ctx, trapdoor(20, 2000))
    flat = pp.eqs_bytes()
    off = 64 * ((1 << 20) - 1)  # level 20 of the flat layout
    raw = flat[off:off + 64 * (1 << 20)]
    return ctx.upload(raw), raw


@pytest.mark.parametrize("n", [1 << 16, (1 << 20) - 3])
def test_msm_fr_large(hl, ctx, bases20, n):
    d_bases, raw = bases20
    rng = np.random.default_rng(n)
    scalars = rand_mont(rng, n)
    got = hl.variable_base_msm(ctx, ctx.upload(scalars), d_bases, n)
    assert got == co.msm(scalars, raw[:64 * n])


def _fr_of_u32(hl, ctx, vals):
    src, out = ctx.upload(vals.tobytes()), ctx.alloc(32 * len(vals))
    hl._check(ctx.lib.lh_fr_from_u32(ctx.h, src.ptr, len(vals), out.ptr))
    ctx.sync()
    return src, out.read()


@pytest.mark.parametrize("shape", ["uniform16", "uniform32", "skewed"])
def test_msm_u32_large(hl, ctx, bases20, shape):
    """32-bit columns (Lasso's dim / read_ts / final_cts / E): one or two windows; `skewed`: 60 % of the scalars are
    the same value and another 30 % come from eight values (hot buckets cut into continuation chunks), zeros present"""
    d_bases, raw = bases20
    n = 1 << 20
    rng = np.random.default_rng(len(shape))
    if shape == "uniform16":
        v = rng.integers(0, 1 << 16, size=n, dtype=np.uint32)
    elif shape == "uniform32":
        v = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    else:
        v = rng.integers(0, 1 << 20, size=n, dtype=np.uint32)
        u = rng.random(n)
        v[u < 0.6] = 12345
        v[(u >= 0.6) & (u < 0.9)] = rng.integers(1, 9, size=int(((u >= 0.6) & (u < 0.9)).sum()), dtype=np.uint32)
        v[u > 0.99] = 0
    src, as_fr = _fr_of_u32(hl, ctx, v)
    got = hl.variable_base_msm_u32(ctx, src, d_bases, n)
    assert got == co.msm(as_fr, raw)
    # the same column as full field elements goes through the signed-digit path
    assert hl.variable_base_msm(ctx, ctx.upload(as_fr), d_bases, n) == got


def test_msm_fr_skewed_large(hl, ctx, bases20):
    """full-width scalars with heavy repetition: every window has one very hot bucket"""
    d_bases, raw = bases20
    n = 1 << 18
    rng = np.random.default_rng(5)
    a = np.frombuffer(rand_mont(rng, n), dtype=np.uint64).reshape(n, 4).copy()
    a[rng.random(n) < 0.5] = a[0]
    scalars = a.tobytes()
    assert hl.variable_base_msm(ctx, ctx.upload(scalars), d_bases, n) == co.msm(scalars, raw[:64 * n])


# ------------------------------------------------------------------ a16: HyperPlonk at 2^16 (runtime-compiled rounds)
@pytest.mark.parametrize("with_lookup", [False, True])
def test_hyperplonk_2p16_matches_cpp_oracle(hl, ctx, with_lookup):
    """the reference's sample circuits (backend/hyperplonk/util.rs) at 16 variables: the zero-check runs the
    expression compiled at run time (csrc/jit.cpp, from 2^16 rows), LogUp's sort-merge join and the permutation
    prefix product run at scale; proof bytes against the C++ oracle, then the host verifier"""
    from halo2_lasso_amd import hyperplonk as g_hp
    from oracle.pyref import hyperplonk as o_hp
    k = 16
    ss = trapdoor(k, 1600 + with_lookup)
    gen = o_hp.rand_vanilla_plonk_with_lookup_circuit if with_lookup else o_hp.rand_vanilla_plonk_circuit
    o_info, instances, witness = gen(k, random.Random(160 + with_lookup))
    mk = g_hp.vanilla_plonk_with_lookup_circuit_info if with_lookup else g_hp.vanilla_plonk_circuit_info
    g_info = mk(k, len(instances[0]), o_info.preprocess_polys, o_info.permutations)
    pcs = hl.MultilinearKzg.setup(ctx, ss)
    pp, vp = g_hp.HyperPlonk.preprocess(pcs, g_info, hl.MultilinearKzgVerifierParams.setup(ss))
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, instances, [hl.MultilinearPolynomial.new(ctx, w) for w in witness], t)
    proof = t.into_proof()
    # C++ oracle on the same circuit
    num_z, expression = o_hp.compose(o_info)
    perm_idx = o_info.permutation_polys()
    perm = o_hp.permutation_polys(o_info.k, perm_idx, o_info.permutations)
    lookups = [[(co.flatten_expression(i), co.flatten_expression(tb)) for i, tb in lk] for lk in o_info.lookups]
    ot = co.Transcript()
    co.hyperplonk_prove(ot, pcs.eqs_bytes(), k, k, o_info.num_instances, o_info.preprocess_polys,
                        o_info.num_witness_polys[0], o_info.num_challenges[0], lookups, perm_idx, perm, num_z,
                        co.flatten_expression(expression), instances, witness)
    assert proof == ot.into_proof()
    g_hp.HyperPlonk.verify(vp, instances, hl.Keccak256Transcript.from_proof(proof))


# ------------------------------------------------------------------ f3: Zeromorph at 2^16
def test_zeromorph_commit_open_2p16(hl, ctx):
    nv = 16
    prng = random.Random(316)
    s = prng.randrange(1, P)
    size = (1 << nv) + 5
    powers = co.usetup(s, size)
    params = hl.Zeromorph.setup(ctx, s, size)
    buf = C.create_string_buffer(64 * size)
    hl._check(ctx.lib.lh_usrs_download(ctx.h, params.h, buf))
    assert buf.raw == powers, "powers of s differ from the oracle's"
    pp = hl.Zeromorph.trim(params, 1 << nv)
    raw = rand_mont(np.random.default_rng(316), 1 << nv)
    poly = mle(hl, ctx, raw)
    assert hl.Zeromorph.commit(pp, poly) == co.zm_commit(powers, 1 << nv, raw)
    point = [prng.randrange(P) for _ in range(nv)]
    ot = co.Transcript()
    co.zm_open(ot, powers, 1 << nv, raw, point)
    t = hl.Keccak256Transcript()
    hl.Zeromorph.open(pp, poly, point, t)
    assert t.into_proof() == ot.into_proof()
    vp = hl.ZeromorphVerifierParam.setup(s, size, 1 << nv)
    hl.Zeromorph.verify(vp, hl.Zeromorph.commit(pp, poly), point, poly.evaluate(point),
                        hl.Keccak256Transcript.from_proof(t.into_proof()))


# ------------------------------------------------------------------ the headline config's REAL route, byte for byte
@pytest.fixture(scope="module")
def srs22(hl, ctx):
    """GPU params for 22 variables and their flat SRS for the C++ oracle (srs17 above ties the GPU setup to the oracle's)"""
    ss = trapdoor(22, 2200)
    pp = hl.MultilinearKzg.setup(ctx, ss)
    return ss, pp, pp.eqs_bytes()


@pytest.mark.heavy(est=10)
@pytest.mark.parametrize("kind,n", [("and", 21), ("xor", 21), ("range", 22), ("and", 22)])
def test_lasso_default_route_at_2p21_matches_cpp_oracle(hl, ctx, srs22, kind, n):
    """The route the 2^24 headline proof takes switches on at 2^21 lookups: the largest quotient(s) of the opening are
    committed column by column (mkzg_open's column route: packed pairs with 2^18-2^20 buckets, 32-share window sums, the
    lazy first fold; TWO column-wise levels for the range check), E columns come from their dim column's buckets, read_ts
    columns go in packed pairs.  A passing pairing check does not pin the n quotient commitments one by one; the C++
    oracle's bytes do.  Default options, nothing forced."""
    ss, pp, flat = srs22
    table = hl.LassoTable.range(2, 16) if kind == "range" else hl.LassoTable.bitwise(
        hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, 4, 16)
    rng = np.random.default_rng(2100 + n + len(kind))
    dims = _dims(rng, table, n, kind == "xor")
    ot = co.Transcript()
    co.lasso_prove(ot, flat, 22, table.to_c(), n, [d.tobytes() for d in dims])
    t = hl.Keccak256Transcript()
    hl.profile_enable(ctx, True)
    try:
        hl.lasso_prove(pp, table, n, [ctx.upload(d.tobytes()) for d in dims], t)
        names = {r["name"] for r in hl.profile_read(ctx)}
    finally:
        hl.profile_enable(ctx, False)
    assert t.into_proof() == ot.into_proof()
    # linear g: the Surge sum-check runs over the 32-bit output column, and while round 2 still is a streaming round (from 2^20
    # lookups on) its first three rounds never see a field-element view of it (csrc/sumcheck.cpp: two rounds of sums made
    # with the claim, one double-bind kernel)
    assert "sc_round_u32<bind2>" in names and "inner_products<quads>" in names, sorted(names)
    # the batch opening: no merged tables - its sum-check's first three rounds and its first fold from the columns (from 2^22)
    assert "lincomb<fold,u32>" in names and ("lincomb<bind2,u32>" in names) == (n >= 22), sorted(names)
    route = hl.lasso_last_route(ctx)
    assert route["open_small_depth"] == (2 if kind == "range" else 1) and route["open_small_passes"] >= 3, route
    assert route["eq_factored_rounds"] > 0 and route["rw_leaf_rounds"] > 0, route
    assert route["open_precommit"] == 1, route  # (those column-wise commitments ran on the helper ctx, beside the sum-checks)
    if kind != "range":
        assert route["derived_commitments"] == 4 and route["packed_ts_pairs"] >= 1, route  # (a skewed column has wide counts)


# ------------------------------------------------------------------ BASELINE.json configs[2]
@pytest.mark.parametrize("kind", ["and", pytest.param("xor", marks=pytest.mark.heavy(est=8))])
def test_lasso_2p24_and_prove_verify(hl, ctx, kind):
    """2^24 AND / XOR lookups (32-bit operands, 4 chunks of 8+8 bits) on one GPU: the proof verifies, is the same
    with its two large MSM batches pipelined in halves (msm_half_batches, the default) and undivided, and a flipped lookup
    index changes it.  (Bytes against the oracle at this size are checked by bench.py's cpu_baseline
    on the largest sample that fits its time bound; the XOR columns are skewed - a quarter of the lookups hit 16 cells -
    so the access counts are wide and the packed read_ts pairs take their large-bucket shape.)"""
    n = 24
    table = hl.LassoTable.bitwise(hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR, 4, 16)
    ss = trapdoor(n, 2400)
    pp = hl.MultilinearKzg.setup(ctx, ss)
    rng = np.random.default_rng(24)
    dims = [rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32) for _ in range(4)]
    if kind == "xor":
        for d in dims:
            hot = rng.random(1 << n) < 0.25
            d[hot] = rng.integers(0, 16, size=int(hot.sum()), dtype=np.uint32) * 4099 % (1 << 16)
    bufs = [ctx.upload(d.tobytes()) for d in dims]
    proofs = []
    for half in (1, 0):  # (the commit and the opening's remainder as two pipelined halves on two streams, then as one batch each)
        hl.set_option(ctx, "msm_half_batches", half)
        try:
            t = hl.Keccak256Transcript()
            hl.lasso_prove(pp, table, n, bufs, t)
        finally:
            hl.set_option(ctx, "msm_half_batches", 1)
        proofs.append(t.into_proof())
        assert hl.lasso_last_route(ctx)["msm_half_batches"] == (2 if half else 0), hl.lasso_last_route(ctx)
    assert proofs[0] == proofs[1]
    hl.lasso_verify(hl.MultilinearKzgVerifierParams.setup(ss), table, n, hl.Keccak256Transcript.from_proof(proofs[0]))
    dims[2][12345] ^= 1
    bufs[2] = ctx.upload(dims[2].tobytes())
    t = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, bufs, t)
    assert t.into_proof() != proofs[0]


# ------------------------------------------------------------------ Lasso inside HyperPlonk at scale (configs[4] stand-in)
@pytest.mark.parametrize("kind", ["and", "range"])
def test_hyperplonk_lasso_2p17_matches_cpp_oracle(hl, ctx, kind):
    """vanilla gates + a 32-bit lookup (4 x (8+8)-bit AND chunks / 2 x 16-bit range limbs: the production subtable size)
    proven by Lasso inside HyperPlonk::prove, 2^17 rows: proof bytes against the C++ oracle's restatement of the
    specification (oracle/pyref/hyperplonk.py LassoLookup), then the host verifier"""
    from halo2_lasso_amd import hyperplonk as g_hp, synthetic
    from oracle.pyref import hyperplonk as o_hp, lasso as o_lasso
    k = 17
    ss = trapdoor(k, 1717)
    pcs = hl.MultilinearKzg.setup(ctx, ss)
    circ = synthetic.vanilla_plonk_with_lasso(ctx, k, kind=kind, seed=17)
    pp, vp = synthetic.prover_param(pcs, circ, hl.MultilinearKzgVerifierParams.setup(ss))
    t = hl.Keccak256Transcript()
    g_hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, t)
    proof = t.into_proof()
    spec = o_lasso.range_table(2, 16) if kind == "range" else o_lasso.bitwise_table(o_lasso.SUBTABLE_AND, 4, 16)
    o_info = o_hp.vanilla_plonk_with_lasso_circuit_info(k, 0, [[]] * 6, [[(7, 1)], [(8, 1)], [(9, 1)]], spec)
    num_z, expression = o_hp.compose(o_info)
    lk = circ.info.lasso_lookups[0]
    ot = co.Transcript()
    co.hyperplonk_prove(ot, pcs.eqs_bytes(), k, k, [0], [a.tobytes() for a in circ.h_preprocess], len(circ.h_witness), 0, [],
                        [7, 8, 9], [p.buf.read() for p in circ.d_permutation], num_z, co.flatten_expression(expression),
                        [[]], [a.tobytes() for a in circ.h_witness],
                        lasso_lookups=[(lk.table.to_c(), lk.output_poly, lk.chunk_polys)])
    assert proof == ot.into_proof()
    g_hp.HyperPlonk.verify(vp, circ.instances, hl.Keccak256Transcript.from_proof(proof))


def test_context_used_from_another_thread(hl, ctx):
    """the current HIP device is a per-thread setting: every entry point makes the ctx's device current, so a ctx created
    on one host thread proves from another (device guard)"""
    import threading
    n = 14
    table = hl.LassoTable.range(2, 16)
    pp = hl.MultilinearKzg.setup(ctx, trapdoor(16, 99))
    rng = np.random.default_rng(99)
    dims = [ctx.upload(rng.integers(0, 1 << 16, size=1 << n, dtype=np.uint32).tobytes()) for _ in range(2)]
    t0 = hl.Keccak256Transcript()
    hl.lasso_prove(pp, table, n, dims, t0)
    out = {}

    def worker():
        t = hl.Keccak256Transcript()
        hl.lasso_prove(pp, table, n, dims, t)
        out["proof"] = t.into_proof()
    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert out["proof"] == t0.into_proof()


@pytest.mark.gpu
@pytest.mark.heavy(est=7)
@pytest.mark.parametrize("env", [{"LH_SC_TAIL_G": "64"}, {"LH_SC_TAIL_G": "2", "LH_SC_TAIL_MAX_LEN": "16384"},
                                 {"LH_SC_TAIL": "0", "LH_LASSO_PACK_TS": "0", "LH_MSM_SLAB_LOG": "31"},
                                 {"LH_OPEN_SMALL_MIN_VARS": "2", "LH_MSM_SLAB_LOG": "4"},
                                 {"LH_MSM_QUAD_MAX": "0", "LH_MSM_SEG": "16", "LH_MSM_TREE_MAX": "0"},
                                 {"LH_OPEN_SMALL_MIN_VARS": "2", "LH_OPEN_SMALL_DEPTH": "2", "LH_OPEN_SMALL_CHECK": "1"},
                                 {"LH_MSM_WINDOW_TABLES": "24", "LH_MSM_SLAB_LOG": "6"},
                                 {"LH_MSM_HALF_MIN_LOG": "6", "LH_MSM_SLAB_LOG": "12"},
                                 {"LH_MSM_HALF_MIN_LOG": "6", "LH_MSM_HALF_COVER": "1"}])
def test_small_parity_suite_under_forced_shapes(env):
    """The byte-parity tests of test_gpu_parity.py / test_gpu_golden.py again in a child process with the shape
    knobs forced (they are read once per process): 64 workgroups with slices of two entries (hand-over right after
    the first resident round), two workgroups with the longest resident tables, no resident tail at all, and every
    MSM job >= 2^14 points on the slab path (per-slab sorts, the dim columns' entry streams taken from the access
    counters' sorts), every Lasso batch opening through the small-column route for its largest quotient (32-bit
    differences, packed column pairs, base-sum offsets: by default only from 2^21 lookups on), read_ts columns
    committed one by one instead of in packed pairs, and the two largest quotients column by column whatever the table
    (with the route's own comparison against the plain commitments switched on), every full-width MSM job over a
    window table of its SRS level (one bucket set for all windows), and the MSM tails in their throughput forms whatever
    the size: plain (not quad-cooperative) kernels, 16-bucket segments with the two-level group reduction wherever a window
    has 4096 buckets, linear continuation levels instead of trees; and every MSM batch with two jobs as two pipelined halves
    on two streams (msm_half_batches: by default only from 2^24 entries on), derived jobs next to their parents, with the
    default and with the smallest second half."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_golden.py", "-m", "gpu",
                          "-x", "-q", "-k", "sum_check or grand_product or fractional or lasso or batch_open or golden or msm"],
                         cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]


_ROUTE_HASH = """
import hashlib, sys
sys.path.insert(0, %r)
import bench
import halo2_lasso_amd as hl
n, kind = int(sys.argv[1]), sys.argv[2]
ctx = hl.Context(0)
table, _ = bench.make_table(hl, kind)
pp = hl.MultilinearKzg.setup(ctx, bench.trapdoor(max(n, table.l)))
dims = [ctx.upload(c.tobytes()) for c in bench.gen_dims(table, n, 0)]
t = hl.Keccak256Transcript()
hl.lasso_prove(pp, table, n, dims, t)
print("proof", hashlib.sha256(t.into_proof()).hexdigest())
"""


@pytest.mark.gpu
@pytest.mark.heavy(est=5)
def test_round6_route_switches_change_the_route_not_the_bytes():
    """Round 6's routes each have an environment switch for A/B timing: the fence-free hand-off of the round
    kernels, Surge's and the batch opening's first rounds from the 32-bit columns, the opening's fold from the columns, the
    sort's XCD tile order.  A 2^22 range-check proof made with all of them OFF (a child process: they are read once) is byte
    for byte the proof made with the defaults - which test_lasso_default_route_at_2p21_matches_cpp_oracle pins to the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    off = {"LH_FIN_LANES_MIN_BYTES": "-1", "LH_SC_U32": "0", "LH_OPEN_U32_ROUNDS": "0", "LH_OPEN_FOLD_COLS": "0",
           "LH_SORT_XCD_ORDER": "0"}
    got = []
    for env in ({}, off):
        r = subprocess.run([sys.executable, "-c", _ROUTE_HASH % root, "22", "range"], cwd=root, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        got.append([l for l in r.stdout.splitlines() if l.startswith("proof ")][-1])
    assert got[0] == got[1], got
