"""The resident grand-product layers (csrc/kernels_gkr.hip: the layers near the roots of the memory-checking trees in ONE
launch - layer loop, eq tables, eq-factored rounds with the batching coefficients folded into the left factors, final
evaluations) against the C++ oracle (oracle/cpu, reference algorithms) on bytes: tree counts that exercise every lane
grouping (1, 2, 4 groups of four trees, ragged counts), depths at which the active set shrinks inside the resident range,
single- and multi-workgroup layers up to the largest resident layer; the option switched off gives the same bytes; a
transcript that fails in the middle of a resident layer releases the kernel and leaves the context usable."""
import ctypes as C
import random
import time

import numpy as np
import pytest

from oracle import cpu_oracle as co
from oracle.pyref.field import R_MOD as P

pytestmark = pytest.mark.gpu

TOP_LIMB = 0x30644E72E131A029


def rand_mont(rng, n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] = rng.integers(0, TOP_LIMB, size=n, dtype=np.uint64)
    return a.tobytes()


def mle(hl, ctx, raw):
    n = len(raw) // 32
    return hl.MultilinearPolynomial(ctx, ctx.upload(raw), n.bit_length() - 1)


def prove(hl, ctx, leaves):
    t = hl.Keccak256Transcript()
    roots, claims = hl.prove_grand_product(ctx, [mle(hl, ctx, r) for r in leaves], t)
    return roots, claims, t.into_proof()


@pytest.mark.parametrize("depths", [
    [1], [2, 1], [3, 3, 3, 3, 3],                       # single-workgroup layers only
    [7] * 4, [8] * 9, [9] * 16,                          # 1, 4 (ragged) and 4 tree groups; the first multi-workgroup layers
    [11, 11, 6, 6, 6, 11, 3, 11],                        # the active set shrinks inside the resident range
    [15] * 16,                                           # every resident layer of a 16-tree batch, up to 2^14 entries
    [16] * 8 + [12] * 8,                                 # the memory-check shape: resident layers, then launched ones
    [17, 17, 15],
])
def test_resident_layers_match_cpp_oracle(hl, ctx, depths):
    rng = np.random.default_rng(1000 + sum(depths) + len(depths))
    leaves = [rand_mont(rng, 1 << nv) for nv in depths]
    ot = co.Transcript()
    o_roots, o_claims = co.grand_product_prove(ot, leaves)
    assert hl.get_option(ctx, "gkr_resident") == 1
    roots, claims, proof = prove(hl, ctx, leaves)
    assert roots == o_roots and claims == o_claims and proof == ot.into_proof()
    # the launched path (one sum-check per layer) writes the same bytes
    hl.set_option(ctx, "gkr_resident", 0)
    try:
        roots0, claims0, proof0 = prove(hl, ctx, leaves)
    finally:
        hl.set_option(ctx, "gkr_resident", 1)
    assert (roots0, claims0, proof0) == (roots, claims, proof)


def test_lasso_route_counts_resident_layers(hl, ctx):
    """a 2^14 AND proof: 16 trees (8 of depth 14, 8 of depth 16); layers 1 .. 12 are ordinary layers of <= 2^14 entries,
    layer 13 is the paired leaf layer of the lookup-sized trees, layers 14 / 15 belong to the subtable-sized trees"""
    n, nv = 14, 16
    rng = random.Random(5)
    ss = [rng.randrange(1, P) for _ in range(nv)]
    table = hl.LassoTable.bitwise(hl.SUBTABLE_AND, 4, 16)
    dims = [np.random.default_rng(50 + j).integers(0, 1 << 16, size=1 << n, dtype=np.uint32) for j in range(4)]
    pp = hl.MultilinearKzg.setup(ctx, ss)
    bufs = [ctx.upload(d.tobytes()) for d in dims]
    out = {}
    for on in (1, 0):
        hl.set_option(ctx, "gkr_resident", on)
        try:
            tr = hl.Keccak256Transcript()
            hl.lasso_prove(pp, table, n, bufs, tr)
            out[on] = (tr.into_proof(), hl.lasso_last_route(ctx))
        finally:
            hl.set_option(ctx, "gkr_resident", 1)
    assert out[1][0] == out[0][0]
    assert out[1][1]["resident_layers"] >= 12 and out[0][1]["resident_layers"] == 0, (out[1][1], out[0][1])
    ot = co.Transcript()
    co.lasso_prove(ot, pp.eqs_bytes(), nv, table.to_c(), n, [d.tobytes() for d in dims])
    assert out[1][0] == ot.into_proof()


def test_transcript_failure_inside_a_resident_layer(hl, ctx):
    """The transcript fails while the resident kernel waits for a challenge: the error surfaces at once (the kernel is told
    to leave, not left to time out), and the next proofs on the context - resident again - have the oracle's bytes."""
    from halo2_lasso_amd import _ffi
    rng = np.random.default_rng(77)
    leaves = [rand_mont(rng, 1 << 12) for _ in range(6)]
    ot = co.Transcript()
    o_roots, o_claims = co.grand_product_prove(ot, leaves)
    polys = [mle(hl, ctx, r) for r in leaves]
    for fail_at in (3, 20, 41):  # a single-workgroup layer, a round of a multi-workgroup layer, a layer boundary
        inner = hl.Keccak256Transcript()
        vt = inner.p.contents
        calls = {"n": 0}

        def squeeze(user, out, calls=calls, vt=vt, fail_at=fail_at):
            calls["n"] += 1
            if calls["n"] == fail_at:
                return -6  # LH_ERR_TRANSCRIPT
            return vt.squeeze_challenge(vt.user, out)

        failing = _ffi.lh_transcript()
        C.memmove(C.byref(failing), inner.p, C.sizeof(failing))
        cb = _ffi._FE_CB(squeeze)
        failing.squeeze_challenge = cb

        class Wrapped:
            p = C.pointer(failing)

        t0 = time.perf_counter()
        with pytest.raises(hl.Error):
            hl.prove_grand_product(ctx, polys, Wrapped)
        assert calls["n"] == fail_at and time.perf_counter() - t0 < 1.5  # aborted, not timed out
        t1 = time.perf_counter()
        t = hl.Keccak256Transcript()
        roots, claims = hl.prove_grand_product(ctx, polys, t)
        assert time.perf_counter() - t1 < 1.0
        assert roots == o_roots and claims == o_claims and t.into_proof() == ot.into_proof()


def test_a_launch_that_cannot_start_all_its_workgroups_falls_back(hl):
    """On a GPU shared with other processes' resident kernels (several ranks on one device) a launch may never get all of its
    workgroups dispatched; the kernel signs everybody in first and leaves - before the transcript has seen anything - when
    the roll is not complete in time (GKR_START_FAILED); the prover then takes the launched path for those layers.  With
    LH_GKR_START_TIMEOUT_MS=0 every multi-workgroup launch gives up: same proof bytes as the oracle, no resident layer."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, random
        sys.path.insert(0, %r)
        import numpy as np
        import halo2_lasso_amd as hl
        from oracle import cpu_oracle as co
        n, nv = 17, 17
        rng = random.Random(55)
        ss = [rng.randrange(1, hl.R_MOD) for _ in range(nv)]
        ctx = hl.Context(0)
        table = hl.LassoTable.bitwise(hl.SUBTABLE_AND, 4, 16)
        dims = [np.random.default_rng(550 + j).integers(0, 1 << 16, size=1 << n, dtype=np.uint32) for j in range(4)]
        pp = hl.MultilinearKzg.setup(ctx, ss)
        tr = hl.Keccak256Transcript()
        hl.lasso_prove(pp, table, n, [ctx.upload(d.tobytes()) for d in dims], tr)
        route = hl.lasso_last_route(ctx)
        ot = co.Transcript()
        co.lasso_prove(ot, pp.eqs_bytes(), nv, table.to_c(), n, [d.tobytes() for d in dims])
        assert tr.into_proof() == ot.into_proof(), "bytes differ"
        assert route["resident_layers"] == 0 and route["resident_tails"] >= 10, route
        print("FALLBACK-OK", route["resident_layers"], route["resident_tails"])
    """) % root
    env = dict(os.environ, LH_GKR_START_TIMEOUT_MS="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FALLBACK-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_folded_layers_give_the_same_bytes_on_every_path(hl, ctx):
    """Coefficient folding (option sc_pp_fold): the generic layers store c_m l_m at their first bind, the leaf layers of the
    lookup-sized trees store cs (l + k) and r + k and go on as product-pair rounds - with the resident kernel finishing
    the layers, with the launched small rounds and the generic tail finishing them (gkr_resident = 0: they evaluate the
    REWRITTEN general expression over the folded tables), not folding at all, and (2) the product-pair kernel binding round
    after round with coefficients that are not one and tables it must leave unscaled: one proof, the oracle's."""
    n, nv = 18, 18
    rng = random.Random(18)
    ss = [rng.randrange(1, P) for _ in range(nv)]
    table = hl.LassoTable.bitwise(hl.SUBTABLE_XOR, 4, 16)
    dims = [np.random.default_rng(180 + j).integers(0, 1 << 16, size=1 << n, dtype=np.uint32) for j in range(4)]
    pp = hl.MultilinearKzg.setup(ctx, ss)
    bufs = [ctx.upload(d.tobytes()) for d in dims]
    ot = co.Transcript()
    co.lasso_prove(ot, pp.eqs_bytes(), nv, table.to_c(), n, [d.tobytes() for d in dims])
    want = ot.into_proof()
    seen = {}
    for fold, resident in ((1, 1), (1, 0), (0, 1), (0, 0), (2, 1), (2, 0)):
        hl.set_option(ctx, "sc_pp_fold", fold)
        hl.set_option(ctx, "gkr_resident", resident)
        try:
            tr = hl.Keccak256Transcript()
            hl.lasso_prove(pp, table, n, bufs, tr)
            seen[(fold, resident)] = (tr.into_proof(), hl.lasso_last_route(ctx))
        finally:
            hl.set_option(ctx, "sc_pp_fold", 1)
            hl.set_option(ctx, "gkr_resident", 1)
    for key, (proof, route) in seen.items():
        assert proof == want, key
        assert (route["pp_folds"] > 0) == (key[0] == 1), (key, route)
        assert route["rw_leaf_rounds"] > 0, (key, route)
    # with folding the leaf layers run the leaf kernel twice (first round, folding round), without it in every streaming round
    assert seen[(1, 1)][1]["rw_leaf_rounds"] < seen[(0, 1)][1]["rw_leaf_rounds"]


def test_a_proof_repeated_two_hundred_times_is_the_same_proof(hl, ctx):
    """What one parity test cannot see (tools/soak.py in small): the same 2^13 AND proof 200 times on one ctx - identical
    bytes, the same route every time (no resident launch that sometimes falls back, no sequence number that runs into a
    stale one), and a workspace arena whose high-water mark stops growing after the first proofs."""
    n, nv = 13, 16
    rng = random.Random(9)
    ss = [rng.randrange(1, P) for _ in range(nv)]
    table = hl.LassoTable.bitwise(hl.SUBTABLE_AND, 4, 16)
    dims = [np.random.default_rng(90 + j).integers(0, 1 << 16, size=1 << n, dtype=np.uint32) for j in range(4)]
    pp = hl.MultilinearKzg.setup(ctx, ss)
    bufs = [ctx.upload(d.tobytes()) for d in dims]
    first = route = mark = None
    for i in range(200):
        tr = hl.Keccak256Transcript()
        hl.lasso_prove(pp, table, n, bufs, tr)
        proof, rt = tr.into_proof(), hl.lasso_last_route(ctx)
        if i == 0:
            first = proof
        assert proof == first, "proof %d differs" % i
        if i == 2:
            route, mark = rt, hl.memory_stats(ctx)["arena_high_water_bytes"]
        if i > 2:
            assert rt == route, (i, rt, route)
    assert route["resident_layers"] >= 12 and hl.memory_stats(ctx)["arena_high_water_bytes"] == mark
    ot = co.Transcript()
    co.lasso_prove(ot, pp.eqs_bytes(), nv, table.to_c(), n, [d.tobytes() for d in dims])
    assert first == ot.into_proof()
