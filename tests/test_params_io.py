"""CPU: bincode layout of the PCS parameter structs (halo2_lasso_amd.params_io; kzg.rs:25-102,
univariate/kzg.rs:33-111, zeromorph.rs:42-65).  The curve crate's serde form is an assumption documented in the
module; what is tested here is the framing (bincode 1.x), both coordinate encodings, and round trips."""
import random
import struct

import pytest

from oracle.pyref import curve, kzg as o_kzg, pairing as o_pair
from oracle.pyref.field import R_MOD as P, Q_MOD, MONT_R


def _mont(v):
    return (v * MONT_R % Q_MOD).to_bytes(32, "little")


def test_multilinear_kzg_params_layout_and_round_trip(hl):
    from halo2_lasso_amd import params_io as io
    rng = random.Random(4)
    ss = [rng.randrange(1, P) for _ in range(3)]
    pp = o_kzg.setup(ss)
    flat = b"".join(hl.g1_to_bytes(p) for lvl in pp.eqs for p in lvl)
    g2s = [o_pair.g2_mul(o_pair.G2_GEN, s) for s in ss]
    blob = io.write_multilinear_kzg_params(curve.G1_GEN, flat, 4, o_pair.G2_GEN, g2s)
    # g1 | u64 4 | (u64 2^k | points)* | g2 | u64 3 | ss
    assert len(blob) == 64 + 8 + sum(8 + 64 * (1 << k) for k in range(4)) + 128 + 8 + 3 * 128
    assert blob[:64] == _mont(1) + _mont(2)                       # generator (1, 2), Montgomery limbs
    assert struct.unpack_from("<Q", blob, 64)[0] == 4 and struct.unpack_from("<Q", blob, 72)[0] == 1
    assert blob[80:144] == hl.g1_to_bytes(pp.eqs[0][0])
    back = io.read_multilinear_kzg_params(blob)
    assert back["g1"] == curve.G1_GEN and back["eqs_flat"] == flat and back["num_levels"] == 4
    assert back["g2"] == o_pair.G2_GEN and back["ss"] == g2s
    # canonical encoding: same framing, coordinates as plain little-endian integers
    canon = io.write_multilinear_kzg_params(curve.G1_GEN, flat, 4, o_pair.G2_GEN, g2s, repr_="canonical")
    assert len(canon) == len(blob) and canon[:64] == (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
    assert io.read_multilinear_kzg_params(canon, repr_="canonical") == back
    # prover / verifier halves
    pblob = io.write_multilinear_kzg_prover_params(curve.G1_GEN, flat, 4)
    assert io.read_multilinear_kzg_prover_params(pblob)["eqs_flat"] == flat and pblob == blob[:len(pblob)]
    vblob = io.write_multilinear_kzg_verifier_params(curve.G1_GEN, o_pair.G2_GEN, g2s)
    assert io.read_multilinear_kzg_verifier_params(vblob) == dict(g1=curve.G1_GEN, g2=o_pair.G2_GEN, ss=g2s)
    with pytest.raises(ValueError):
        io.read_multilinear_kzg_params(blob[:-1])
    with pytest.raises(ValueError):
        io.read_multilinear_kzg_params(blob + b"\x00")
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 72, 2)   # eqs[0] claims two points
    with pytest.raises(ValueError):
        io.read_multilinear_kzg_params(bytes(bad))


def test_verifier_params_file_feeds_the_host_verifier(hl):
    """a verifier-param file written here drives lh_mkzg_vp_new and verifies a golden proof"""
    import json
    import os
    from halo2_lasso_amd import params_io as io
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
    ss = [int(x, 16) for x in golden["srs"]["ss"]]
    g1, g2, ss_g2 = hl.MultilinearKzgVerifierParams.setup(ss).export()
    blob = io.write_multilinear_kzg_verifier_params(g1, g2, ss_g2)
    d = io.read_multilinear_kzg_verifier_params(blob)
    vp = hl.MultilinearKzgVerifierParams.new(d["g1"], d["g2"], d["ss"])
    g = golden["lasso"][0]
    hl.lasso_verify(vp, hl.LassoTable.range(g["c"], g["l"]), g["n"], hl.Keccak256Transcript.from_proof(bytes.fromhex(g["proof"])))


def test_univariate_and_zeromorph_param_round_trip(hl):
    from halo2_lasso_amd import params_io as io
    from oracle.pyref import zeromorph as zm
    s = 0x1234567
    param = zm.setup(s, 5)
    g1b = b"".join(hl.g1_to_bytes(p) for p in param.powers_g1)
    g2p = [o_pair.g2_mul(o_pair.G2_GEN, pow(s, i, P)) for i in range(3)]
    blob = io.write_univariate_kzg_param(g1b, g2p)
    assert len(blob) == 8 + 5 * 64 + 8 + 3 * 128
    assert io.read_univariate_kzg_param(blob) == dict(powers_g1_bytes=g1b, powers_g2=g2p)
    assert io.read_univariate_kzg_param(io.write_univariate_kzg_param(g1b, g2p, "canonical"), "canonical")["powers_g1_bytes"] == g1b
    vp = hl.ZeromorphVerifierParam.setup(s, 5, 4)
    vblob = io.write_zeromorph_verifier_param(*vp.export())
    assert len(vblob) == 64 + 3 * 128
    d = io.read_zeromorph_verifier_param(vblob)
    assert hl.ZeromorphVerifierParam.new(d["g1"], d["g2"], d["s_g2"], d["s_offset_g2"]).export() == vp.export()
