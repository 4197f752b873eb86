"""Host mirror of plonkish_backend::backend::hyperplonk (setup / preprocess / prove surface).

`PlonkishCircuitInfo` (backend.rs:46-130), `compose` and the permutation polys of
backend/hyperplonk/preprocessor.rs:25-203, `HyperPlonk::{preprocess, prove}` (hyperplonk.rs:97-291).
Host code here is circuit bookkeeping only (expressions, copy cycles); every polynomial operation of
`prove` runs on the GPU behind `lh_hyperplonk_prove`.
"""
import ctypes as C

from . import _ffi
from . import expression as ex
from .expression import R_MOD


class PlonkishCircuitInfo:
    """backend.rs:46-73"""

    def __init__(self, k, num_instances, preprocess_polys, num_witness_polys, num_challenges, constraints, lookups,
                 permutations, max_degree=None):
        self.k, self.num_instances = k, list(num_instances)
        self.preprocess_polys = [list(p) for p in preprocess_polys]
        self.num_witness_polys, self.num_challenges = list(num_witness_polys), list(num_challenges)
        self.constraints, self.lookups = list(constraints), [list(l) for l in lookups]
        self.permutations, self.max_degree = [list(c) for c in permutations], max_degree
        self.lasso_lookups = []  # LassoLookup: lookups proven by the Lasso argument instead of LogUp

    def num_poly(self):
        return len(self.num_instances) + len(self.preprocess_polys) + sum(self.num_witness_polys)

    def permutation_polys(self):
        return sorted({poly for cycle in self.permutations for poly, _ in cycle})


class LassoLookup:
    """A lookup into a decomposable table (`LassoTable`) proven by the Lasso argument inside HyperPlonk::prove in place of
    LogUp (include/lasso_hip.h lh_hp_lasso_lookup; specification: oracle/pyref/hyperplonk.py).  On every row the circuit
    poly `output_poly` holds g(T[dim_0], ..) and `chunk_polys[j]` the chunk index dim_j < 2^l."""

    def __init__(self, table, output_poly, chunk_polys):
        if len(chunk_polys) != table.c:
            raise ValueError("expected %d chunk polys" % table.c)
        self.table, self.output_poly, self.chunk_polys = table, output_poly, list(chunk_polys)


def _lasso_lookups_c(info):
    arr = (_ffi.lh_hp_lasso_lookup * max(len(info.lasso_lookups), 1))()
    for k, lk in enumerate(info.lasso_lookups):
        arr[k].table = lk.table.to_c()
        arr[k].output_poly = lk.output_poly
        for j, p in enumerate(lk.chunk_polys):
            arr[k].chunk_polys[j] = p
    return arr


def lookup_constraints(info, beta, gamma):
    """preprocessor.rs:79-109"""
    m_offset = info.num_poly() + len(info.permutation_polys())
    h_offset = m_offset + len(info.lookups)
    constraints = []
    for k, lookup in enumerate(info.lookups):
        m, h = ex.Polynomial(m_offset + k), ex.Polynomial(h_offset + k)
        inp = ex.distribute_powers([i for i, _ in lookup], beta)
        tab = ex.distribute_powers([t for _, t in lookup], beta)
        constraints.append(h * (inp + gamma) * (tab + gamma) - (tab + gamma) + m * (inp + gamma))
    return constraints, [ex.Polynomial(h_offset + k) for k in range(len(info.lookups))]


def max_degree(info, lookup_cs=None):
    """preprocessor.rs:62-77"""
    if lookup_cs is None:
        lookup_cs = lookup_constraints(info, ex.Constant(0), ex.Constant(0))[0]
    degs = [c.degree() for c in info.constraints] + [c.degree() for c in lookup_cs]
    if info.max_degree is not None:
        degs.append(info.max_degree)
    return max(degs + [2])


def permutation_constraints(info, max_deg, beta, gamma, num_builtin_witness_polys):
    """preprocessor.rs:111-170"""
    perm_polys = info.permutation_polys()
    chunk = max_deg - 1
    num_chunks = -(-len(perm_polys) // chunk) if perm_polys else 0
    perm_offset = info.num_poly()
    z_offset = perm_offset + len(perm_polys) + num_builtin_witness_polys
    polys = [ex.Polynomial(i) for i in perm_polys]
    ids = [ex.Constant(i << info.k) + ex.Identity() for i in range(len(polys))]
    perms = [ex.Polynomial(perm_offset + i) for i in range(len(perm_polys))]
    zs = [ex.Polynomial(z_offset + i) for i in range(num_chunks)]
    z_0_next = ex.Polynomial(z_offset, 1)
    constraints = []
    if zs:
        constraints.append(ex.Lagrange(1) * (zs[0] - ex.Constant(1)))
    for c in range(num_chunks):
        sl = slice(c * chunk, (c + 1) * chunk)
        z_lhs, z_rhs = zs[c], (zs[c + 1] if c + 1 < num_chunks else z_0_next)
        lhs = z_lhs * ex.product_exprs([p + beta * i + gamma for p, i in zip(polys[sl], ids[sl])])
        rhs = z_rhs * ex.product_exprs([p + beta * s + gamma for p, s in zip(polys[sl], perms[sl])])
        constraints.append(lhs - rhs)
    return num_chunks, constraints


def compose(info):
    """preprocessor.rs:25-60 -> (num_permutation_z_polys, expression)"""
    off = sum(info.num_challenges)
    beta, gamma, alpha = (ex.Challenge(off + i) for i in range(3))
    lookup_cs, lookup_zero_checks = lookup_constraints(info, beta, gamma)
    md = max_degree(info, lookup_cs)
    num_z, perm_cs = permutation_constraints(info, md, beta, gamma, 2 * len(info.lookups))
    constraints = list(info.constraints) + lookup_cs + perm_cs
    zero_check_on_every_row = ex.distribute_powers(constraints, alpha) * ex.EqXY(0)
    return num_z, ex.distribute_powers(lookup_zero_checks + [zero_check_on_every_row], alpha)


def permutation_polys(num_vars, perm_polys, cycles):
    """preprocessor.rs:172-203"""
    poly_index = {poly: idx for idx, poly in enumerate(perm_polys)}
    perms = [[((idx << num_vars) + j) % R_MOD for j in range(1 << num_vars)] for idx in range(len(perm_polys))]
    for cycle in cycles:
        i0, j0 = cycle[0]
        last = perms[poly_index[i0]][j0]
        for (i, j) in (cycle[1:] + cycle[:1]):
            assert j != 0
            perms[poly_index[i]][j], last = last, perms[poly_index[i]][j]
    return perms


class HyperPlonkProverParam:
    """hyperplonk.rs:38-55"""


def _pcs_of(pcs_pp):
    """the PolynomialCommitmentScheme a param belongs to: MultilinearKzg or Zeromorph (backend/hyperplonk.rs:76-95)"""
    from . import MultilinearKzg, Zeromorph, ZeromorphProverParam
    return Zeromorph if isinstance(pcs_pp, ZeromorphProverParam) else MultilinearKzg


class HyperPlonkVerifierParam:
    """hyperplonk.rs:57-74"""


class HyperPlonk:
    @staticmethod
    def preprocess(pcs_pp, info, pcs_vp=None):
        """hyperplonk.rs:97-162: preprocess / permutation polys go to the GPU once.  Returns the prover param, or
        (pp, vp) when the PCS verifier param is given."""
        from . import MultilinearPolynomial
        ctx = pcs_pp.ctx
        pcs = _pcs_of(pcs_pp)
        pp = HyperPlonkProverParam()
        pp.pcs, pp.num_vars, pp.info = pcs_pp, info.k, info
        pp.preprocess_polys = [MultilinearPolynomial.new(ctx, p) for p in info.preprocess_polys]
        pp.preprocess_comms = pcs.batch_commit(pcs_pp, pp.preprocess_polys)
        perm = permutation_polys(info.k, info.permutation_polys(), info.permutations)
        pp.permutation_polys = [MultilinearPolynomial.new(ctx, p) for p in perm]
        pp.permutation_comms = pcs.batch_commit(pcs_pp, pp.permutation_polys)
        pp.num_permutation_z_polys, pp.expression = compose(info)
        if pcs_vp is None:
            return pp
        vp = HyperPlonkVerifierParam()
        vp.pcs, vp.num_vars, vp.info = pcs_vp, info.k, info
        vp.num_permutation_z_polys, vp.expression = pp.num_permutation_z_polys, pp.expression
        vp.preprocess_comms, vp.permutation_comms = list(pp.preprocess_comms), list(pp.permutation_comms)
        return pp, vp

    @staticmethod
    def shard_param(pp, rank, size, shard_bit):
        """This rank's prover param of a sharded prove: preprocess and permutation polys as shards (device extraction),
        everything else shared with `pp`."""
        from . import shard_poly
        sp = HyperPlonkProverParam()
        sp.__dict__.update(pp.__dict__)
        sp.preprocess_polys = [shard_poly(p, rank, size, shard_bit) for p in pp.preprocess_polys]
        sp.permutation_polys = [shard_poly(p, rank, size, shard_bit) for p in pp.permutation_polys]
        return sp

    @staticmethod
    def rebind_param(pp, ctx):
        """The same prover param for use through another ctx of the same device (a second proof in flight): the polys and
        the SRS are device memory and stay shared, only the ctx the library is called with changes."""
        sp = HyperPlonkProverParam()
        sp.__dict__.update(pp.__dict__)
        sp.pcs = pp.pcs.view(ctx)
        return sp

    @staticmethod
    def prove_sharded(pp_local, instances, witness_polys_local, transcript):
        """lh_hyperplonk_prove_sharded: ONE proof over the ranks of the ctx's communicator, same bytes as `prove`.
        `pp_local` from shard_param, `witness_polys_local`: this rank's shards of the witness polys."""
        return HyperPlonk.prove(pp_local, instances, witness_polys_local, transcript, sharded=True)

    @staticmethod
    def prove(pp, instances, witness_polys, transcript, sharded=False):
        """hyperplonk.rs:164-291.  `witness_polys`: the device tables of a single-phase circuit (`synthesize(0, [])`),
        or a callable synthesize(round, challenges) -> list of MultilinearPolynomial (PlonkishCircuit::synthesize,
        backend.rs:139) which the phase loop of hyperplonk.rs:185-205 calls once per phase."""
        from . import _check, _ptr_array, _fr_array, lh_fr, ArgumentError
        info, ctx = pp.info, pp.pcs.ctx
        multi = callable(witness_polys)
        if sharded and multi:
            raise NotImplementedError("the sharded prove takes a single-phase circuit's witness tables")
        if len(info.num_witness_polys) != 1 and not multi:
            raise ArgumentError("multi-phase circuits need a synthesize(round, challenges) callable")
        keep = []
        prm = _ffi.lh_hp_param()
        prm.num_vars = pp.num_vars
        prm.num_instance_polys = len(info.num_instances)
        ni = (C.c_size_t * max(len(info.num_instances), 1))(*info.num_instances)
        prm.num_instances = ni
        prm.num_preprocess_polys = len(pp.preprocess_polys)
        pre = _ptr_array(pp.preprocess_polys)
        prm.d_preprocess_polys = C.cast(pre, C.POINTER(C.c_void_p))
        prm.num_witness_polys = sum(info.num_witness_polys)
        prm.num_challenges = sum(info.num_challenges)
        lookups = (_ffi.lh_hp_lookup * max(len(info.lookups), 1))()
        for k, lookup in enumerate(info.lookups):
            ins = (_ffi.lh_expr * len(lookup))()
            tabs = (_ffi.lh_expr * len(lookup))()
            for w, (i_e, t_e) in enumerate(lookup):
                ce, ka = i_e.to_c()
                ins[w] = ce
                ct, kb = t_e.to_c()
                tabs[w] = ct
                keep += [ka, kb]
            lookups[k].inputs, lookups[k].tables, lookups[k].width = ins, tabs, len(lookup)
            keep += [ins, tabs]
        prm.num_lookups, prm.lookups = len(info.lookups), lookups
        pidx = info.permutation_polys()
        prm.num_permutation_polys = len(pidx)
        pi_arr = (C.c_size_t * max(len(pidx), 1))(*pidx)
        prm.permutation_poly_index = pi_arr
        perm = _ptr_array(pp.permutation_polys)
        prm.d_permutation_polys = C.cast(perm, C.POINTER(C.c_void_p))
        prm.num_permutation_z_polys = pp.num_permutation_z_polys
        ce, knodes = pp.expression.to_c()
        prm.expression = ce
        lasso_arr = _lasso_lookups_c(info)
        prm.num_lasso_lookups, prm.lasso_lookups = len(info.lasso_lookups), lasso_arr
        inst_arrays = [_fr_array(i) for i in instances]
        inst = (C.POINTER(lh_fr) * max(len(instances), 1))(*[C.cast(a, C.POINTER(lh_fr)) for a in inst_arrays])
        from . import ZeromorphProverParam
        if multi:
            from . import fr_from_bytes
            alive, failure = [], []

            def synth(_user, rnd, challenges, num_ch, d_out, num_out):
                try:
                    ch = [fr_from_bytes(bytes(challenges[i])) for i in range(num_ch)]
                    polys = list(witness_polys(rnd, ch))
                    if len(polys) != num_out:
                        return _ffi.LH_ERR_ARG
                    alive.append(polys)  # the tables must outlive the prove
                    for i, p in enumerate(polys):
                        d_out[i] = p.ptr
                    return 0
                except Exception as e:  # surfaces as the prove's error
                    failure.append(e)
                    return _ffi.LH_ERR_INVALID_SNARK
            circ = _ffi.lh_hp_circuit()
            circ.user, circ.synthesize = None, _ffi._SYNTH_CB(synth)
            nph = len(info.num_witness_polys)
            nw = (C.c_size_t * max(nph, 1))(*info.num_witness_polys)
            nc = (C.c_size_t * max(nph, 1))(*info.num_challenges)
            if isinstance(pp.pcs, ZeromorphProverParam):
                rc = ctx.lib.lh_hyperplonk_prove_phases_zeromorph(ctx.h, pp.pcs.params.h, pp.pcs.poly_size, C.byref(prm), nph,
                                                                  nw, nc, inst, C.byref(circ), transcript.p)
            else:
                rc = ctx.lib.lh_hyperplonk_prove_phases(ctx.h, pp.pcs.h, C.byref(prm), nph, nw, nc, inst, C.byref(circ),
                                                        transcript.p)
            if failure:
                raise failure[0]
            _check(rc)
            return
        wit = _ptr_array(witness_polys)
        if sharded:
            if isinstance(pp.pcs, ZeromorphProverParam):
                raise NotImplementedError("the sharded prove is wired for multilinear KZG")
            _check(ctx.lib.lh_hyperplonk_prove_sharded(ctx.h, pp.pcs.h, C.byref(prm), inst, wit, transcript.p))
            return
        if isinstance(pp.pcs, ZeromorphProverParam):
            _check(ctx.lib.lh_hyperplonk_prove_zeromorph(ctx.h, pp.pcs.params.h, pp.pcs.poly_size, C.byref(prm), inst, wit,
                                                         transcript.p))
        else:
            _check(ctx.lib.lh_hyperplonk_prove(ctx.h, pp.pcs.h, C.byref(prm), inst, wit, transcript.p))

    @staticmethod
    def verify(vp, instances, transcript):
        """hyperplonk.rs:293-362 (host only).  Raises InvalidSumcheck / InvalidSnark / InvalidPcsOpen."""
        from . import _check, _fr_array, _g1_array, lh_fr
        info = vp.info
        if [len(i) for i in instances] != list(info.num_instances):
            raise AssertionError("instances do not match num_instances")  # assert_eq! hyperplonk.rs:300
        prm = _ffi.lh_hp_vparam()
        prm.num_vars = vp.num_vars
        prm.num_instance_polys = len(info.num_instances)
        ni = (C.c_size_t * max(len(info.num_instances), 1))(*info.num_instances)
        prm.num_instances = ni
        prm.num_witness_polys, prm.num_challenges = sum(info.num_witness_polys), sum(info.num_challenges)
        prm.num_lookups, prm.num_permutation_z_polys = len(info.lookups), vp.num_permutation_z_polys
        ce, knodes = vp.expression.to_c()
        prm.expression = ce
        lasso_arr = _lasso_lookups_c(info)
        prm.num_lasso_lookups, prm.lasso_lookups = len(info.lasso_lookups), lasso_arr
        pre, perm = _g1_array(vp.preprocess_comms), _g1_array(vp.permutation_comms)
        prm.num_preprocess_polys, prm.preprocess_comms = len(vp.preprocess_comms), pre
        prm.num_permutation_polys, prm.permutation_comms = len(vp.permutation_comms), perm
        inst_arrays = [_fr_array(i) for i in instances]
        inst = (C.POINTER(lh_fr) * max(len(instances), 1))(*[C.cast(a, C.POINTER(lh_fr)) for a in inst_arrays])
        from . import ZeromorphVerifierParam
        if len(info.num_witness_polys) != 1:
            nph = len(info.num_witness_polys)
            nw = (C.c_size_t * max(nph, 1))(*info.num_witness_polys)
            nc = (C.c_size_t * max(nph, 1))(*info.num_challenges)
            fn = vp.pcs.lib.lh_hyperplonk_verify_phases_zeromorph if isinstance(vp.pcs, ZeromorphVerifierParam) \
                else vp.pcs.lib.lh_hyperplonk_verify_phases
            _check(fn(vp.pcs.h, C.byref(prm), nph, nw, nc, inst, transcript.p))
            return
        fn = vp.pcs.lib.lh_hyperplonk_verify_zeromorph if isinstance(vp.pcs, ZeromorphVerifierParam) \
            else vp.pcs.lib.lh_hyperplonk_verify
        _check(fn(vp.pcs.h, C.byref(prm), inst, transcript.p))


# ------------------------------------------------------------------ the reference's sample circuits
def _vanilla_gate(base):
    pi, q_l, q_r, q_m, q_o, q_c = (ex.Polynomial(i) for i in range(6))
    w_l, w_r, w_o = (ex.Polynomial(base + i) for i in range(3))
    return q_l * w_l + q_r * w_r + q_m * w_l * w_r + q_o * w_o + q_c + pi


def vanilla_plonk_circuit_info(num_vars, num_instances, preprocess_polys, permutations):
    """backend/hyperplonk/util.rs:30-50: polys pi | q_l q_r q_m q_o q_c | w_l w_r w_o"""
    return PlonkishCircuitInfo(num_vars, [num_instances], preprocess_polys, [3], [0], [_vanilla_gate(6)], [],
                               permutations, 4)


def vanilla_plonk_with_lookup_circuit_info(num_vars, num_instances, preprocess_polys, permutations):
    """backend/hyperplonk/util.rs:63-86: polys pi | q_l q_r q_m q_o q_c q_lookup t_l t_r t_o | w_l w_r w_o"""
    q_lookup, t_l, t_r, t_o = (ex.Polynomial(i) for i in range(6, 10))
    w_l, w_r, w_o = (ex.Polynomial(i) for i in range(10, 13))
    lookups = [[(q_lookup * w_l, t_l), (q_lookup * w_r, t_r), (q_lookup * w_o, t_o)]]
    return PlonkishCircuitInfo(num_vars, [num_instances], preprocess_polys, [3], [0], [_vanilla_gate(10)], lookups,
                               permutations, 4)


def vanilla_plonk_with_lasso_circuit_info(num_vars, num_instances, preprocess_polys, permutations, table):
    """The BASELINE.json configs[4] stand-in (the reference has no Keccak-f circuit and no Lasso): vanilla gates plus ONE
    lookup into a decomposable `LassoTable` proven by Lasso.  polys pi | q_l q_r q_m q_o q_c q_lookup | w_l w_r w_o
    d_0..d_{c-1} a; constraints: the vanilla gate and q_lookup * (w_o - a)."""
    c = table.c
    q_lookup = ex.Polynomial(6)
    w_o, a = ex.Polynomial(9), ex.Polynomial(10 + c)
    info = PlonkishCircuitInfo(num_vars, [num_instances], preprocess_polys, [4 + c], [0],
                               [_vanilla_gate(7), q_lookup * (w_o - a)], [], permutations, 4)
    info.lasso_lookups = [LassoLookup(table, 10 + c, [10 + j for j in range(c)])]
    return info


def keccak_circuit_info(num_vars, preprocess_polys, permutations, table_xor, table_and):
    """Keccak-f as a PLONKish circuit whose bitwise operations are Lasso lookups (BASELINE.json configs[4]); layout and
    row kinds: halo2-lasso_amd/keccak_circuit.py.  polys pi | q_xor q_and q_lin c_x s_x c_y s_y | x y o d_X a_X d_A a_A;
    two Lasso lookups (one chunk of 2 ub bits each): a_X = T_xor[d_X], a_A = T_and[d_A] on every row."""
    q_xor, q_and, q_lin, c_x, s_x, c_y, s_y = (ex.Polynomial(1 + i) for i in range(7))
    x, y, o, d_x, a_x, d_a, a_a = (ex.Polynomial(8 + i) for i in range(7))
    u, v = c_x + s_x * x, c_y + s_y * y
    unit = 1 << (table_xor.l // 2)
    constraints = [q_xor * (d_x - u * unit - v), q_xor * (o - a_x), q_and * (d_a - u * unit - v), q_and * (o - a_a),
                   q_lin * (o - u - v)]
    info = PlonkishCircuitInfo(num_vars, [0], preprocess_polys, [7], [0], constraints, [], permutations, 4)
    info.lasso_lookups = [LassoLookup(table_xor, 12, [11]), LassoLookup(table_and, 14, [13])]
    return info
