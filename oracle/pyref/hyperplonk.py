"""HyperPlonk (zero-check + LogUp lookups + permutation) prover / verifier.  TEST INFRASTRUCTURE ONLY.

Restates reference plonkish_backend/src/backend/hyperplonk.rs:85-369,
backend/hyperplonk/{preprocessor.rs:13-203, prover.rs:32-409, verifier.rs:19-182, util.rs:30-405} and the
rotation helpers of poly/multilinear.rs:191-264,433-570.  The transcript schedule is the reference's
(SURVEY.md §3.1); circuits are the reference's test circuits (`vanilla_plonk[_with_lookup]`), generated
with Python's PRNG (the reference's StdRng streams are not reproducible here - data, not schedule).
"""
import math

from .field import R_MOD as P, batch_invert
from . import expression as ex
from . import sum_check as sc
from . import kzg
from . import lasso
from .bh import BooleanHypercube
from .poly import evaluate as mle_evaluate


class InvalidSnark(Exception):
    pass


# ------------------------------------------------------------------ circuit description (backend.rs:46-130)
class CircuitInfo:
    def __init__(self, k, num_instances, preprocess_polys, num_witness_polys, num_challenges, constraints,
                 lookups, permutations, max_degree):
        self.k, self.num_instances = k, list(num_instances)
        self.preprocess_polys = [list(p) for p in preprocess_polys]
        self.num_witness_polys, self.num_challenges = list(num_witness_polys), list(num_challenges)
        self.constraints, self.lookups = list(constraints), [list(l) for l in lookups]
        self.permutations, self.max_degree = [list(c) for c in permutations], max_degree
        self.lasso_lookups = []  # LassoLookup: lookups proven by the Lasso argument instead of LogUp (see below)

    def num_poly(self):
        return len(self.num_instances) + len(self.preprocess_polys) + sum(self.num_witness_polys)

    def permutation_polys(self):
        return sorted({poly for cycle in self.permutations for poly, _ in cycle})


class LassoLookup:
    """A lookup into a decomposable table proven by the Lasso argument (oracle/pyref/lasso.py) inside HyperPlonk::prove,
    in place of the LogUp m / h polys and constraint of hyperplonk.rs:211-252, preprocessor.rs:79-109 (the reference
    snapshot has no Lasso code: this schedule is the build's own).  On EVERY row k the circuit's poly `output_poly`
    holds a[k] = g(T_1[dim_1[k]], ..), the polys `chunk_polys[j]` hold the chunk indices dim_j[k] < 2^l.  These are
    ordinary circuit polys (normally witness columns, tied to the rest of the circuit by its own gates); the argument
    adds committed polys read_ts_j, E_i (2^num_vars entries) and final_cts_j (2^l <= 2^num_vars entries, zero padded),
    numbered after the permutation z polys: per lookup read_ts | E | final_cts.

    Schedule inside prove (hyperplonk.py): the Lasso polys are committed in round n right after the LogUp m
    commitments, framed with lasso.write_commitments (an identically zero read_ts is a valid witness); after the
    zero-check and its evaluations each lookup absorbs (num_vars, l, c, alpha) and runs lasso.argue; ONE batch_open
    then serves the zero-check's queries and every Lasso claim."""

    def __init__(self, spec, output_poly, chunk_polys):
        assert len(chunk_polys) == spec.c
        self.spec, self.output_poly, self.chunk_polys = spec, output_poly, list(chunk_polys)


def lasso_witnesses(lookups, polys, num_vars):
    """chunk columns -> Lasso witnesses; a chunk value outside the subtable or an output that is not the table's value
    is the reference's Error::InvalidSnark("Invalid lookup input") (prover.rs:176-178)"""
    out = []
    for lk in lookups:
        if lk.spec.l > num_vars:
            raise InvalidSnark("Lasso subtable larger than the circuit")
        dims = [polys[p] for p in lk.chunk_polys]
        if any(d >= 1 << lk.spec.l for col in dims for d in col):
            raise InvalidSnark("Invalid lookup input")
        w = lasso.witness(lk.spec, dims)
        if w["a"] != [v % P for v in polys[lk.output_poly]]:
            raise InvalidSnark("Invalid lookup input")
        out.append(w)
    return out


def lasso_poly_list(w, num_vars):
    pad = lambda t: list(t) + [0] * ((1 << num_vars) - len(t))
    return [pad(t) for t in w["read_ts"] + w["E"] + w["final_cts"]]


def lasso_evaluations(lk, base, point_base, vals):
    """Evaluation claims of one Lasso lookup: polys numbered from `base` (read_ts | E | final_cts), points numbered from
    `point_base` (r, r_z, r_N, r_M)"""
    v, e_rz, dim_e, rts_e, e_e, fc_e = vals
    c, alpha = lk.spec.c, lk.spec.alpha
    out = [kzg.Evaluation(lk.output_poly, point_base, v)]
    out += [kzg.Evaluation(base + c + i, point_base + 1, e_rz[i]) for i in range(alpha)]
    out += [kzg.Evaluation(lk.chunk_polys[j], point_base + 2, dim_e[j]) for j in range(c)]
    out += [kzg.Evaluation(base + j, point_base + 2, rts_e[j]) for j in range(c)]
    out += [kzg.Evaluation(base + c + i, point_base + 2, e_e[i]) for i in range(alpha)]
    out += [kzg.Evaluation(base + c + alpha + j, point_base + 3, fc_e[j]) for j in range(c)]
    return out


def row_mapping(k):
    """hyperplonk.rs:365-369"""
    return BooleanHypercube(k).iter()[1:] + [0]


def instance_polys(num_vars, instances):
    """prover.rs:32-48"""
    rm = row_mapping(num_vars)
    out = []
    for inst in instances:
        poly = [0] * (1 << num_vars)
        for b, v in zip(rm, inst):
            poly[b] = v % P
        out.append(poly)
    return out


# ------------------------------------------------------------------ preprocessor.rs
def lookup_constraints(info, beta, gamma):
    """preprocessor.rs:79-109"""
    m_offset = info.num_poly() + len(info.permutation_polys())
    h_offset = m_offset + len(info.lookups)
    constraints = []
    for k, lookup in enumerate(info.lookups):
        m, h = ex.Poly(m_offset + k), ex.Poly(h_offset + k)
        inp = ex.distribute_powers([i for i, _ in lookup], beta)
        tab = ex.distribute_powers([t for _, t in lookup], beta)
        constraints.append(h * (inp + gamma) * (tab + gamma) - (tab + gamma) + m * (inp + gamma))
    sum_check = [ex.Poly(h_offset + k) for k in range(len(info.lookups))]
    return constraints, sum_check


def max_degree(info, lookup_cs=None):
    """preprocessor.rs:62-77"""
    if lookup_cs is None:
        lookup_cs = lookup_constraints(info, ex.Constant(0), ex.Constant(0))[0]
    degs = [ex.degree(c) for c in info.constraints] + [ex.degree(c) for c in lookup_cs]
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
    polys = [ex.Poly(i) for i in perm_polys]
    ids = [ex.Constant((i << info.k) % P) + ex.Identity() for i in range(len(polys))]
    perms = [ex.Poly(perm_offset + i) for i in range(len(perm_polys))]
    zs = [ex.Poly(z_offset + i) for i in range(num_chunks)]
    z_0_next = ex.Poly(z_offset, 1)
    constraints = []
    if zs:
        constraints.append(ex.Lagrange(1) * (zs[0] - ex.Constant(1)))

    def prod(es):
        acc = es[0]
        for e in es[1:]:
            acc = acc * e
        return acc
    for c in range(num_chunks):
        sl = slice(c * chunk, (c + 1) * chunk)
        z_lhs, z_rhs = zs[c], (zs[c + 1] if c + 1 < num_chunks else z_0_next)
        lhs = z_lhs * prod([p + beta * i + gamma for p, i in zip(polys[sl], ids[sl])])
        rhs = z_rhs * prod([p + beta * s + gamma for p, s in zip(polys[sl], perms[sl])])
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
    perms = [[((idx << num_vars) + j) % P for j in range(1 << num_vars)] for idx in range(len(perm_polys))]
    for cycle in cycles:
        i0, j0 = cycle[0]
        last = perms[poly_index[i0]][j0]
        for (i, j) in (cycle[1:] + cycle[:1]):
            assert j != 0
            perms[poly_index[i]][j], last = last, perms[poly_index[i]][j]
    return perms


# ------------------------------------------------------------------ prover.rs: lookup / permutation polys
def _eval_row(expr, polys, challenges, b, bh, lagrange_rows):
    return ex.evaluate(
        expr, lambda c: c,
        lambda cp: (b % P) if isinstance(cp, ex.Identity) else (1 if (cp.i, b) in lagrange_rows else 0),
        lambda q: polys[q][b] if isinstance(q, int) else polys[q.idx][bh.rotate(b, q.rotation)],
        lambda i: challenges[i], lambda a: (-a) % P, lambda a, c: (a + c) % P, lambda a, c: a * c % P,
        lambda a, s: a * s % P)


def lookup_compressed_polys(lookups, polys, challenges, betas):
    """prover.rs:50-137"""
    if not lookups:
        return []
    num_vars = len(polys[0]).bit_length() - 1
    bh = BooleanHypercube(num_vars)
    order = bh.iter()
    lag = {l.i for lookup in lookups for pair in lookup for e in pair for l in ex.leaves(e) if isinstance(l, ex.Lagrange)}
    lagrange_rows = {(i, order[i % (1 << num_vars)]) for i in lag}
    out = []
    for lookup in lookups:
        def compress(exprs):
            acc = [0] * (1 << num_vars)
            for beta_i, e in zip(betas, exprs):
                for b in range(1 << num_vars):
                    acc[b] = (acc[b] + beta_i * _eval_row(e, polys, challenges, b, bh, lagrange_rows)) % P
            return acc
        out.append([compress([i for i, _ in lookup]), compress([t for _, t in lookup])])
    return out


def lookup_m_poly(compressed):
    """prover.rs:145-192: duplicate table values keep the LAST index (HashMap collect, :151)."""
    inp, table = compressed
    index = {}
    for i, t in enumerate(table):
        index[t] = i
    m = [0] * len(inp)
    for v in inp:
        if v not in index:
            raise InvalidSnark("Invalid lookup input")
        m[index[v]] += 1
    return [c % P for c in m]


def lookup_h_poly(compressed, m_poly, gamma):
    """prover.rs:206-250: h = 1/(gamma + f) - m/(gamma + t)"""
    inp, table = compressed
    hi = batch_invert([(gamma + v) % P for v in inp])
    ht = batch_invert([(gamma + v) % P for v in table])
    h = [(a - b * m) % P for a, b, m in zip(hi, ht, m_poly)]
    assert sum(h) % P == 0  # the reference's `sanity-check` feature (prover.rs:245-247): the LogUp identity closes
    return h


def permutation_z_polys(num_chunks, perm_polys, polys, beta, gamma):
    """prover.rs:252-345.  perm_polys: list of (poly index, permutation table)."""
    if not perm_polys:
        return []
    chunk_size = -(-len(perm_polys) // num_chunks)
    num_vars = len(polys[0]).bit_length() - 1
    n = 1 << num_vars
    products = []
    for c in range(num_chunks):
        chunk = perm_polys[c * chunk_size:(c + 1) * chunk_size]
        prod = [1] * n
        for poly, perm in chunk:
            for b in range(n):
                prod[b] = prod[b] * ((beta * perm[b] + gamma + polys[poly][b]) % P) % P
        prod = batch_invert(prod)
        for k, (poly, _) in enumerate(chunk):
            id_offset = (c * chunk_size + k) << num_vars
            for b in range(n):
                prod[b] = prod[b] * (((id_offset + b) * beta + gamma + polys[poly][b]) % P) % P
        products.append(prod)
    order = BooleanHypercube(num_vars).iter()
    z = [0] * num_chunks + [1]
    state = 1
    for b in order[1:]:
        for c in range(num_chunks):
            state = state * products[c][b] % P
            z.append(state)
    z = z[:num_chunks << num_vars]
    # the reference's `sanity-check` feature (prover.rs:325-331): the running product closes to one on the last row
    assert z[-1] * products[-1][order[-1]] % P == 1
    nth = BooleanHypercube(num_vars).nth_map()
    return [[z[offset + num_chunks * nth[b]] for b in range(n)] for offset in range(num_chunks)]


# ------------------------------------------------------------------ rotation helpers (poly/multilinear.rs:433-570)
def _point_pattern(nxt, num_vars, distance):
    bh = BooleanHypercube(num_vars)
    rem = bh.primitive if nxt else bh.x_inv
    pat = [0] * (1 << distance)
    for depth in range(distance):
        step = 1 << (distance - depth)
        for e in range(0, len(pat), step):
            o = e + step // 2
            rot = pat[e] << 1 if nxt else pat[e] >> 1
            pat[o] = rot ^ rem
            pat[e] = rot
    return pat


def _coeff_pattern(nxt, num_vars, distance):
    bh = BooleanHypercube(num_vars)
    rem = bh.primitive - (1 << num_vars) if nxt else bh.x_inv << distance
    pat = [0] * (1 << (distance - 1))
    for depth in range(distance - 1):
        step = 1 << (distance - depth - 1)
        for e in range(0, len(pat), step):
            o = e + step // 2
            rot = pat[e] << 1 if nxt else pat[e] >> 1
            pat[o] = rot ^ rem
            pat[e] = rot
    return pat


def rotation_eval_points(x, rotation):
    """multilinear.rs:477-526"""
    if rotation == 0:
        return [list(x)]
    distance, n = abs(rotation), len(x)
    num_x = n - distance
    bit = lambda p, i: (p >> i) & 1
    if rotation < 0:
        pat = _point_pattern(False, n, distance)
        xs = x[distance:]
        return [[(1 - xs[i]) % P if bit(p, i) else xs[i] for i in range(num_x)] +
                [bit(p, i + num_x) for i in range(distance)] for p in pat]
    pat = _point_pattern(True, n, distance)
    xs = x[:num_x]
    return [[bit(p, i) for i in range(distance)] +
            [(1 - xs[i]) % P if bit(p, i + distance) else xs[i] for i in range(num_x)] for p in pat]


def rotation_eval(x, rotation, evals_for_rotation):
    """multilinear.rs:435-475"""
    if rotation == 0:
        assert len(evals_for_rotation) == 1
        return evals_for_rotation[0]
    n, distance = len(x), abs(rotation)
    assert len(evals_for_rotation) == 1 << distance and distance <= n
    if rotation < 0:
        pat = _coeff_pattern(False, n, distance)
        nths = list(range(distance, 0, -1))
        xs = list(reversed(x[:distance]))
    else:
        pat = _coeff_pattern(True, n, distance)
        nths = [n - 1 + i for i in range(distance)]
        xs = list(x[n - distance:])
    evals = list(evals_for_rotation)
    for idx, (x_i, nth) in enumerate(zip(xs, nths)):
        bits = [(p >> nth) & 1 for p in pat[::1 << idx]]
        evals = [((e0 - e1) * x_i + e1) % P if b else ((e1 - e0) * x_i + e0) % P
                 for b, (e0, e1) in zip(bits, zip(evals[0::2], evals[1::2]))]
    return evals[0]


def evaluate_for_rotation(poly, x, rotation):
    """multilinear.rs:191-264: the 2^distance evaluations the opening proves, i.e. poly at
    rotation_eval_points(x, rotation) (the reference computes the same values with a fused fold)."""
    return [mle_evaluate(poly, pt) for pt in rotation_eval_points(x, rotation)]


# ------------------------------------------------------------------ verifier.rs helpers
def pcs_query(expression, num_instance_poly):
    return [q for q in ex.used_query(expression) if q[0] >= num_instance_poly]


def point_offset(query):
    rots = sorted({r for _, r in query})
    off, out = 0, {}
    for r in rots:
        out[r] = off
        off += 1 << abs(r)
    return out


def points(query, x):
    out = []
    for r in sorted({r for _, r in query}):
        out += rotation_eval_points(x, r)
    return out


def lagrange_eval(x, b):
    acc = 1
    for i, x_i in enumerate(x):
        acc = acc * (x_i if (b >> i) & 1 else (1 - x_i)) % P
    return acc


def evaluate_expression(expression, num_vars, evals, challenges, ys, x):
    """piop/sum_check.rs:60-98"""
    order = BooleanHypercube(num_vars).iter()
    lag = {i: lagrange_eval(x, order[i % (1 << num_vars)]) for i in ex.used_lagrange(expression)}
    from .poly import eq_xy_eval, identity_eval
    return ex.evaluate_general(expression, [eq_xy_eval(x, y) for y in ys], evals, challenges, identity_eval(x), lag)


def instance_evals(num_vars, expression, instances, x):
    """verifier.rs:92-145"""
    iq = [q for q in ex.used_query(expression) if q[0] < len(instances)]
    lo, hi = 0, 0
    for poly, rot in iq:
        i = -rot
        lo, hi = min(lo, i), max(hi, i + len(instances[poly]))
    if lo < 0:
        lo -= 1
    if hi > 0:
        hi += 1
    order = BooleanHypercube(num_vars).iter()
    lag = {i: lagrange_eval(x, order[i % (1 << num_vars)]) for i in range(lo, hi) if i != 0}
    out = {}
    for poly, rot in iq:
        cnt = len(instances[poly])
        if rot > 0:
            idxs = (list(range(-rot, 0)) + list(range(1, cnt + 1)))[:cnt]
        else:
            idxs = list(range(1 - rot, 1 - rot + cnt))
        out[(poly, rot)] = sum(v * lag[i] for v, i in zip(instances[poly], idxs)) % P
    return out


# ------------------------------------------------------------------ HyperPlonk::{preprocess, prove, verify}
class Param:
    pass


def preprocess(pcs_pp, info, pcs_mod=None):
    """hyperplonk.rs:97-162.  pcs_mod: the PolynomialCommitmentScheme the backend is generic over -- `kzg`
    (multilinear KZG, default; pcs_pp = its Params) or `zeromorph` (pcs_pp = the trimmed (ProverParam, VerifierParam))."""
    pp = Param()
    pp.pcs_mod = pcs_mod or kzg
    if pp.pcs_mod is kzg:
        pp.pcs, pp.num_vars = pcs_pp.trim(info.k), info.k
        pp.pcs_vp = pp.pcs
    else:
        (pp.pcs, pp.pcs_vp), pp.num_vars = pcs_pp, info.k
    pp.num_instances, pp.num_witness_polys, pp.num_challenges = info.num_instances, info.num_witness_polys, info.num_challenges
    pp.lookups = info.lookups
    pp.lasso_lookups = getattr(info, "lasso_lookups", [])
    pp.preprocess_polys = info.preprocess_polys
    pp.preprocess_comms = [pp.pcs_mod.commit(pp.pcs, p) for p in info.preprocess_polys]
    perm = permutation_polys(info.k, info.permutation_polys(), info.permutations)
    pp.permutation_polys = list(zip(info.permutation_polys(), perm))
    pp.permutation_comms = [pp.pcs_mod.commit(pp.pcs, p) for p in perm]
    pp.num_permutation_z_polys, pp.expression = compose(info)
    return pp


def prove(pp, instances, witness_fn, transcript):
    """hyperplonk.rs:164-291.  witness_fn(round, challenges) -> list of witness tables."""
    for n_i, inst in zip(pp.num_instances, instances):
        assert len(inst) == n_i
        transcript.common_field_elements(inst)
    inst_polys = instance_polys(pp.num_vars, instances)
    witness_polys, challenges = [], []
    for rnd, (nw, nc) in enumerate(zip(pp.num_witness_polys, pp.num_challenges)):
        polys = witness_fn(rnd, challenges)
        assert len(polys) == nw
        pp.pcs_mod.batch_commit_and_write(pp.pcs, polys, transcript)
        witness_polys += polys
        challenges += transcript.squeeze_challenges(nc)
    polys = inst_polys + pp.preprocess_polys + witness_polys

    beta = transcript.squeeze_challenge()
    width = max([len(l) for l in pp.lookups] + [0])
    betas = [pow(beta, i, P) for i in range(width)]
    compressed = lookup_compressed_polys(pp.lookups, polys, challenges, betas)
    m_polys = [lookup_m_poly(c) for c in compressed]
    pp.pcs_mod.batch_commit_and_write(pp.pcs, m_polys, transcript)
    lasso_w = lasso_witnesses(pp.lasso_lookups, polys, pp.num_vars)
    lasso_polys = [p for w in lasso_w for p in lasso_poly_list(w, pp.num_vars)]
    if pp.lasso_lookups:
        lasso.write_commitments(transcript, [pp.pcs_mod.commit(pp.pcs, p) for p in lasso_polys])

    gamma = transcript.squeeze_challenge()
    h_polys = [lookup_h_poly(c, m, gamma) for c, m in zip(compressed, m_polys)]
    z_polys = permutation_z_polys(pp.num_permutation_z_polys, pp.permutation_polys, polys, beta, gamma)
    pp.pcs_mod.batch_commit_and_write(pp.pcs, h_polys + z_polys, transcript)

    alpha = transcript.squeeze_challenge()
    y = transcript.squeeze_challenges(pp.num_vars)
    polys = polys + [p for _, p in pp.permutation_polys] + m_polys + h_polys + z_polys
    challenges = challenges + [beta, gamma, alpha]
    pts, evals = prove_sum_check(len(pp.num_instances), pp.expression, 0, polys, challenges, y, transcript)
    base = len(polys)
    polys = polys + lasso_polys
    pad_pt = lambda pt: list(pt) + [0] * (pp.num_vars - len(pt))
    for lk, w in zip(pp.lasso_lookups, lasso_w):
        transcript.common_field_elements([pp.num_vars, lk.spec.l, lk.spec.c, lk.spec.alpha])
        lpts, vals = lasso.argue(lk.spec, w, transcript)
        evals += lasso_evaluations(lk, base, len(pts), vals)
        pts += [pad_pt(pt) for pt in lpts]
        base += 2 * lk.spec.c + lk.spec.alpha
    pp.pcs_mod.batch_open(pp.pcs, pp.num_vars, polys, pts, evals, transcript)


def prove_sum_check(num_instance_poly, expression, sum_, polys, challenges, y, transcript):
    """prover.rs:368-409"""
    num_vars = len(polys[0]).bit_length() - 1
    vp = sc.VirtualPolynomial(expression, polys, challenges, [y])
    x, evals = sc.prove(sc.EvaluationsProver, num_vars, vp, sum_, transcript)
    query = pcs_query(expression, num_instance_poly)
    off = point_offset(query)
    out = []
    for poly, rot in query:
        vals = [evals[poly]] if rot == 0 else evaluate_for_rotation(polys[poly], x, rot)
        out += [kzg.Evaluation(poly, off[rot] + k, v) for k, v in enumerate(vals)]
    transcript.write_field_elements([e.value for e in out])
    return points(query, x), out


def verify(vp, instances, transcript):
    """hyperplonk.rs:293-362 (vp: the Param of preprocess, used for its public parts only)."""
    for n_i, inst in zip(vp.num_instances, instances):
        assert len(inst) == n_i
        transcript.common_field_elements(inst)
    witness_comms, challenges = [], []
    for nw, nc in zip(vp.num_witness_polys, vp.num_challenges):
        witness_comms += transcript.read_commitments(nw)
        challenges += transcript.squeeze_challenges(nc)
    beta = transcript.squeeze_challenge()
    m_comms = transcript.read_commitments(len(vp.lookups))
    lasso_lookups = getattr(vp, "lasso_lookups", [])
    lasso_comms = lasso.read_commitments(transcript, sum(2 * lk.spec.c + lk.spec.alpha for lk in lasso_lookups)) \
        if lasso_lookups else []
    gamma = transcript.squeeze_challenge()
    hz_comms = transcript.read_commitments(len(vp.lookups) + vp.num_permutation_z_polys)
    alpha = transcript.squeeze_challenge()
    y = transcript.squeeze_challenges(vp.num_vars)
    challenges += [beta, gamma, alpha]

    # verify_sum_check (verifier.rs:39-90)
    x_eval, x = sc.verify(sc.Evaluations, vp.num_vars, ex.degree(vp.expression), 0, transcript)
    query = pcs_query(vp.expression, len(instances))
    evals_for_rotation, evals = [], dict(instance_evals(vp.num_vars, vp.expression, instances, x))
    for poly, rot in query:
        efr = transcript.read_field_elements(1 << abs(rot))
        evals_for_rotation.append(efr)
        evals[(poly, rot)] = rotation_eval(x, rot, efr)
    if evaluate_expression(vp.expression, vp.num_vars, evals, challenges, [y], x) != x_eval:
        raise InvalidSnark("Unmatched between sum_check output and query evaluation")
    off = point_offset(query)
    pcs_evals = []
    for (poly, rot), efr in zip(query, evals_for_rotation):
        pcs_evals += [kzg.Evaluation(poly, off[rot] + k, v) for k, v in enumerate(efr)]
    comms = [None] * len(vp.num_instances) + vp.preprocess_comms + witness_comms + vp.permutation_comms + m_comms + hz_comms
    pts = points(query, x)
    base = len(comms)
    comms = comms + lasso_comms
    pad_pt = lambda pt: list(pt) + [0] * (vp.num_vars - len(pt))
    for lk in lasso_lookups:
        if lk.spec.l > vp.num_vars:
            raise InvalidSnark("Lasso subtable larger than the circuit")
        if lk.output_poly < len(vp.num_instances) or any(p < len(vp.num_instances) for p in lk.chunk_polys):
            raise InvalidSnark("Lasso lookups over instance polys are not supported")  # they have no commitment
        transcript.common_field_elements([vp.num_vars, lk.spec.l, lk.spec.c, lk.spec.alpha])
        try:
            lpts, vals = lasso.check(lk.spec, vp.num_vars, transcript)
        except lasso.LassoError as e:
            raise InvalidSnark(str(e))
        pcs_evals += lasso_evaluations(lk, base, len(pts), vals)
        pts += [pad_pt(pt) for pt in lpts]
        base += 2 * lk.spec.c + lk.spec.alpha
    getattr(vp, 'pcs_mod', kzg).batch_verify(getattr(vp, 'pcs_vp', vp.pcs), vp.num_vars, comms, pts, pcs_evals,
                                             transcript)
    if transcript.pos != len(transcript.stream):
        raise InvalidSnark("trailing bytes in proof")


# ------------------------------------------------------------------ test circuits (util.rs:30-405)
def _vanilla_gate(base):
    pi, q_l, q_r, q_m, q_o, q_c = (ex.Poly(i) for i in range(6))
    w_l, w_r, w_o = (ex.Poly(base + i) for i in range(3))
    return q_l * w_l + q_r * w_r + q_m * w_l * w_r + q_o * w_o + q_c + pi


def vanilla_plonk_circuit_info(num_vars, num_instances, preprocess_polys, permutations):
    """util.rs:30-50"""
    return CircuitInfo(num_vars, [num_instances], preprocess_polys, [3], [0], [_vanilla_gate(6)], [], permutations, 4)


def vanilla_plonk_with_lookup_circuit_info(num_vars, num_instances, preprocess_polys, permutations):
    """util.rs:63-86"""
    q_lookup, t_l, t_r, t_o = (ex.Poly(i) for i in range(6, 10))
    w_l, w_r, w_o = (ex.Poly(i) for i in range(10, 13))
    lookups = [[(q_lookup * w_l, t_l), (q_lookup * w_r, t_r), (q_lookup * w_o, t_o)]]
    return CircuitInfo(num_vars, [num_instances], preprocess_polys, [3], [0], [_vanilla_gate(10)], lookups,
                       permutations, 4)


class _Permutation:
    """util.rs:376-405"""

    def __init__(self):
        self.cycles, self.idx = [], {}

    def copy(self, lhs, rhs):
        if lhs in self.idx:
            c = self.idx[lhs]
            self.cycles[c].add(rhs)
            self.idx[rhs] = c
        else:
            self.cycles.append({lhs, rhs})
            for cell in (lhs, rhs):
                self.idx[cell] = len(self.cycles) - 1

    def into_cycles(self):
        return [sorted(c) for c in self.cycles]


def rand_vanilla_plonk_with_lookup_circuit(num_vars, rng):
    """util.rs:216-316 -> (CircuitInfo, instances, witness tables [w_l, w_r, w_o])"""
    size = 1 << num_vars
    rf = lambda: rng.randrange(P)
    polys = [[0] * size for _ in range(13)]
    for t in (7, 8, 9):
        polys[t] = [0, 0] + [rf() for _ in range(size - 2)]
    instances = [rf() for _ in range(num_vars)]
    polys[0] = instance_polys(num_vars, [instances])[0]
    instance_rows = set(BooleanHypercube(num_vars).iter()[:num_vars + 1])
    perm = _Permutation()
    for poly in (10, 11, 12):
        perm.copy((poly, 1), (poly, 1))
    for idx in range(size - 1):
        use_copy = rng.getrandbits(1) == 0 and idx > 1
        if use_copy:
            l_copy = (rng.randrange(10, 13), rng.randrange(1, idx))
            r_copy = (rng.randrange(10, 13), rng.randrange(1, idx))
            perm.copy(l_copy, (10, idx))
            perm.copy(r_copy, (11, idx))
            w_l, w_r = polys[l_copy[0]][l_copy[1]], polys[r_copy[0]][r_copy[1]]
        else:
            w_l, w_r = rf(), rf()
        q_c = rf()
        gate = use_copy or idx in instance_rows
        add = rng.getrandbits(1) == 0
        if gate and add:
            vals = [(1, 1), (2, 1), (4, P - 1), (5, q_c), (10, w_l), (11, w_r), (12, (w_l + w_r + q_c + polys[0][idx]) % P)]
        elif gate:
            vals = [(3, 1), (4, P - 1), (5, q_c), (10, w_l), (11, w_r), (12, (w_l * w_r + q_c + polys[0][idx]) % P)]
        else:
            t = rng.randrange(1, size)
            vals = [(6, 1), (10, polys[7][t]), (11, polys[8][t]), (12, polys[9][t])]
        for poly, v in vals:
            polys[poly][idx] = v
    info = vanilla_plonk_with_lookup_circuit_info(num_vars, len(instances), polys[1:10], perm.into_cycles())
    return info, [instances], polys[10:13]


def rand_vanilla_plonk_circuit(num_vars, rng):
    """util.rs:100-170"""
    size = 1 << num_vars
    rf = lambda: rng.randrange(P)
    polys = [[0] * size for _ in range(9)]
    instances = [rf() for _ in range(num_vars)]
    polys[0] = instance_polys(num_vars, [instances])[0]
    perm = _Permutation()
    for poly in (6, 7, 8):
        perm.copy((poly, 1), (poly, 1))
    for idx in range(size - 1):
        if rng.getrandbits(1) == 0 and idx > 1:
            l_copy = (rng.randrange(6, 9), rng.randrange(1, idx))
            r_copy = (rng.randrange(6, 9), rng.randrange(1, idx))
            perm.copy(l_copy, (6, idx))
            perm.copy(r_copy, (7, idx))
            w_l, w_r = polys[l_copy[0]][l_copy[1]], polys[r_copy[0]][r_copy[1]]
        else:
            w_l, w_r = rf(), rf()
        q_c = rf()
        if rng.getrandbits(1) == 0:
            vals = [(1, 1), (2, 1), (4, P - 1), (5, q_c), (6, w_l), (7, w_r), (8, (w_l + w_r + q_c + polys[0][idx]) % P)]
        else:
            vals = [(3, 1), (4, P - 1), (5, q_c), (6, w_l), (7, w_r), (8, (w_l * w_r + q_c + polys[0][idx]) % P)]
        for poly, v in vals:
            polys[poly][idx] = v
    info = vanilla_plonk_circuit_info(num_vars, len(instances), polys[1:6], perm.into_cycles())
    return info, [instances], polys[6:9]


def vanilla_plonk_with_lasso_circuit_info(num_vars, num_instances, preprocess_polys, permutations, spec):
    """The configs[4] stand-in: vanilla gates plus ONE lookup into a decomposable table proven by Lasso.
    polys pi | q_l q_r q_m q_o q_c q_lookup | w_l w_r w_o d_0..d_{c-1} a; constraints: the vanilla gate and
    q_lookup * (w_o - a); the Lasso lookup ties a to the chunk columns d_j on every row."""
    c = spec.c
    q_lookup = ex.Poly(6)
    w_o, a = ex.Poly(9), ex.Poly(10 + c)
    info = CircuitInfo(num_vars, [num_instances], preprocess_polys, [4 + c], [0],
                       [_vanilla_gate(7), q_lookup * (w_o - a)], [], permutations, 4)
    info.lasso_lookups = [LassoLookup(spec, 10 + c, [10 + j for j in range(c)])]
    return info

def keccak_circuit_info(num_vars, preprocess_polys, permutations, table_xor, table_and):
    """Keccak-f as a PLONKish circuit whose bitwise operations are Lasso lookups (BASELINE.json configs[4]); layout and
    row kinds: halo2-lasso_amd/keccak_circuit.py.  polys pi | q_xor q_and q_lin c_x s_x c_y s_y | x y o d_X a_X d_A a_A;
    two Lasso lookups (one chunk of 2 ub bits each): a_X = T_xor[d_X], a_A = T_and[d_A] on every row."""
    q_xor, q_and, q_lin, c_x, s_x, c_y, s_y = (ex.Poly(1 + i) for i in range(7))
    x, y, o, d_x, a_x, d_a, a_a = (ex.Poly(8 + i) for i in range(7))
    u, v = c_x + s_x * x, c_y + s_y * y
    unit = 1 << (table_xor.l // 2)
    constraints = [q_xor * (d_x - u * unit - v), q_xor * (o - a_x), q_and * (d_a - u * unit - v), q_and * (o - a_a),
                   q_lin * (o - u - v)]
    info = CircuitInfo(num_vars, [0], preprocess_polys, [7], [0], constraints, [], permutations, 4)
    info.lasso_lookups = [LassoLookup(table_xor, 12, [11]), LassoLookup(table_and, 14, [13])]
    return info



def rand_vanilla_plonk_with_lasso_circuit(num_vars, rng, spec):
    """-> (CircuitInfo, instances, witness tables [w_l, w_r, w_o, d_0.., a]); about half of the rows are lookups"""
    size, c = 1 << num_vars, spec.c
    rf = lambda: rng.randrange(P)
    polys = [[0] * size for _ in range(11 + c)]
    instances = [rf() for _ in range(num_vars)]
    polys[0] = instance_polys(num_vars, [instances])[0]
    instance_rows = set(BooleanHypercube(num_vars).iter()[:num_vars + 1])
    perm = _Permutation()
    for poly in (7, 8, 9):
        perm.copy((poly, 1), (poly, 1))
    for idx in range(size - 1):
        use_copy = rng.getrandbits(1) == 0 and idx > 1
        if use_copy:
            l_copy = (rng.randrange(7, 10), rng.randrange(1, idx))
            r_copy = (rng.randrange(7, 10), rng.randrange(1, idx))
            perm.copy(l_copy, (7, idx))
            perm.copy(r_copy, (8, idx))
            w_l, w_r = polys[l_copy[0]][l_copy[1]], polys[r_copy[0]][r_copy[1]]
        else:
            w_l, w_r = rf(), rf()
        q_c = rf()
        if use_copy or idx in instance_rows or rng.getrandbits(1) == 0:
            if rng.getrandbits(1) == 0:
                vals = [(1, 1), (2, 1), (4, P - 1), (5, q_c), (7, w_l), (8, w_r), (9, (w_l + w_r + q_c + polys[0][idx]) % P)]
            else:
                vals = [(3, 1), (4, P - 1), (5, q_c), (7, w_l), (8, w_r), (9, (w_l * w_r + q_c + polys[0][idx]) % P)]
        else:  # a lookup row: w_o is the table's value at the chunk indices
            dims = [rng.randrange(1 << spec.l) for _ in range(c)]
            out = spec.g_eval([lasso.subtable_entry(kind, dims[j], spec.l) for j, kind in spec.memories])
            vals = [(6, 1), (7, w_l), (8, w_r), (9, out), (10 + c, out)] + [(10 + j, dims[j]) for j in range(c)]
        for poly, v in vals:
            polys[poly][idx] = v
    info = vanilla_plonk_with_lasso_circuit_info(num_vars, len(instances), polys[1:7], perm.into_cycles(), spec)
    return info, [instances], polys[7:]
