"""Lasso lookup argument (Surge sum-check + offline memory checking).  TEST INFRASTRUCTURE ONLY.

NO REFERENCE CODE EXISTS for this layer: the snapshot under /root/reference mentions Lasso only
in README.md:1-9.  This file is the build's own specification of the protocol,
designed from the Lasso paper (eprint 2023/1216) on top of the reference's building blocks
(sum_check.py, gkr.py, kzg.py restate those).  The HIP prover must reproduce these proof bytes.

Protocol (transcript order == proof layout), for N = 2^n lookups into a table decomposed into
c chunks of l bits (subtable size M = 2^l), alpha memories (chunk j(i), subtable t(i)):

 0. C   n, l, c, alpha                                  (common_field_element, domain separation)
 1. W   commitments  a | dim_0..c-1 | read_ts_0..c-1 | E_0..alpha-1 | final_cts_0..c-1
        Every committed poly is zero-padded to nv = max(n, l) variables and committed with eqs[nv]
        (padding with zeros does not change the MSM; f_pad(x || 0..0) = f(x)), so that ONE
        batch_open serves all of them.
 2. S   r[0..n)
 3. W   v = a(r)
 4.     Surge: ClassicSumCheck<EvaluationsProver>, expression eq_0 * g(E_0..E_alpha-1), ys=[r],
        claim v  ->  point r_z ;   W  E_i(r_z) for every i
 5. S   gamma, tau
 6.     leaves (fingerprint h(a,v,t) = a*gamma^2 + v*gamma + t - tau), per memory i:
          RS_i[k] = h(dim_j[k], E_i[k], read_ts_j[k])      WS_i[k] = RS_i[k] + 1       (N leaves)
          Init_i[m] = h(m, T_t[m], 0)                      Final_i[m] = Init_i[m] + final_cts_j[m]
        grand-product GKR (gkr.prove_grand_product) over
          [RS_0, WS_0, .., RS_{alpha-1}, WS_{alpha-1}, Init_0, Final_0, .., Init_{alpha-1}, Final_{alpha-1}]
        -> points r_N (n vars) and r_M (l vars).  Verifier: Init_i*WS_i == RS_i*Final_i (roots).
 7. W   dim_j(r_N) (c) | read_ts_j(r_N) (c) | E_i(r_N) (alpha) | final_cts_j(r_M) (c)
 8.     ONE batch_open (reference pcs/multilinear.rs:134-235) over nv variables:
        polys [a, dim.., read_ts.., E.., final_cts..], points [r, r_z, r_N, r_M] each padded with zeros
        to nv coordinates, evals [(a,r)] + [(E_i,r_z)] + [(dim_j,r_N)] + [(read_ts_j,r_N)] + [(E_i,r_N)]
        + [(final_cts_j,r_M)].
"""
from .field import R_MOD as P
from . import expression as ex
from . import sum_check as sc
from . import gkr
from . import kzg
from .poly import evaluate, eq_xy_eval, identity_eval


class LassoError(Exception):
    pass


# ------------------------------------------------------------------ decomposable tables
SUBTABLE_IDENTITY, SUBTABLE_AND, SUBTABLE_XOR = 0, 1, 2


def subtable_entry(kind, m, l):
    h = l // 2
    x, y = m >> h, m & ((1 << h) - 1)
    if kind == SUBTABLE_IDENTITY:
        return m
    if kind == SUBTABLE_AND:
        return x & y
    if kind == SUBTABLE_XOR:
        return x ^ y
    raise ValueError(kind)


def subtable_mle_eval(kind, point):
    """MLE of the subtable at `point` (l variables, variable i = index bit i); index = x || y with
    y in the low l/2 bits."""
    l = len(point)
    if kind == SUBTABLE_IDENTITY:
        return identity_eval(point)
    h = l // 2
    acc = 0
    for i in range(h):
        yi, xi = point[i], point[h + i]
        if kind == SUBTABLE_AND:
            term = xi * yi
        else:
            term = xi + yi - 2 * xi * yi
        acc = (acc + (term << i)) % P
    return acc


class TableSpec:
    """memories: list of (chunk j, subtable kind); g = sum_m coeff_m * prod_{i in mono_m} E_i."""

    def __init__(self, name, num_chunks, chunk_bits, memories, g_terms):
        self.name, self.c, self.l = name, num_chunks, chunk_bits
        self.memories = list(memories)
        self.g_terms = [(coeff % P, tuple(mono)) for coeff, mono in g_terms]

    @property
    def alpha(self):
        return len(self.memories)

    def g_expression(self):
        terms = []
        for coeff, mono in self.g_terms:
            e = ex.Poly(mono[0])
            for i in mono[1:]:
                e = e * ex.Poly(i)
            terms.append(e * coeff)
        return ex.sum_exprs(terms)

    def g_eval(self, vals):
        acc = 0
        for coeff, mono in self.g_terms:
            t = coeff
            for i in mono:
                t = t * vals[i] % P
            acc = (acc + t) % P
        return acc


def range_table(num_chunks=2, chunk_bits=16):
    """value < 2^(c*l): limbs looked up in the identity subtable, a = sum_j 2^(l*j) * limb_j."""
    return TableSpec("range", num_chunks, chunk_bits,
                     [(j, SUBTABLE_IDENTITY) for j in range(num_chunks)],
                     [(1 << (chunk_bits * j), (j,)) for j in range(num_chunks)])


def bitwise_table(kind, num_chunks=4, chunk_bits=16):
    """AND / XOR of two (c*l/2)-bit operands: chunk j = x_j || y_j (l/2 bits each),
    a = sum_j 2^(l/2*j) * T[x_j || y_j]."""
    name = "and" if kind == SUBTABLE_AND else "xor"
    return TableSpec(name, num_chunks, chunk_bits, [(j, kind) for j in range(num_chunks)],
                     [(1 << (chunk_bits // 2 * j), (j,)) for j in range(num_chunks)])


# ------------------------------------------------------------------ witness
def witness(spec, dims):
    """dims: c lists of N chunk indices < 2^l.  Returns dict of the committed polys."""
    c, l = spec.c, spec.l
    N, M = len(dims[0]), 1 << l
    read_ts, final_cts = [], []
    for j in range(c):
        cnt = [0] * M
        rts = [0] * N
        for k, d in enumerate(dims[j]):
            rts[k] = cnt[d]
            cnt[d] += 1
        read_ts.append(rts)
        final_cts.append(cnt)
    E = [[subtable_entry(kind, dims[j][k], l) for k in range(N)] for j, kind in spec.memories]
    a = [spec.g_eval([E[i][k] for i in range(spec.alpha)]) for k in range(N)]
    return dict(a=a, dim=[list(d) for d in dims], read_ts=read_ts, final_cts=final_cts, E=E)


def _fingerprint(a, v, t, gamma, tau):
    return (a * gamma % P * gamma + v * gamma + t - tau) % P


def _check_shape(spec, n):
    if n < 1 or spec.l < 1:
        raise LassoError("need at least one variable")


# ------------------------------------------------------------------ commitment framing
# A committed column that is identically zero commits to the group identity: the high-limb `dim` of a 32-bit range
# check whose values are all below 2^16, `read_ts` when the indices are pairwise distinct, `E` of an AND whose operands
# never share a bit.  The reference's transcript cannot encode the identity (write_commitment fails on it,
# util/transcript.rs:172-179,216-219), and valid witnesses must stay provable, so the Lasso argument frames its
# commitments itself, with calls every TranscriptWrite offers: ONE field element whose bit i says that commitment i is
# the identity, then the other commitments in order.
def write_commitments(transcript, comms):
    mask = sum(1 << i for i, cm in enumerate(comms) if cm is None)
    transcript.write_field_element(mask)
    transcript.write_commitments([cm for cm in comms if cm is not None])


def read_commitments(transcript, count):
    mask = transcript.read_field_element()
    if mask >> count:
        raise LassoError("commitment mask out of range")
    return [None if (mask >> i) & 1 else transcript.read_commitment() for i in range(count)]


# ------------------------------------------------------------------ prover
def _trim(pcs, p, nv):
    return p.trim(nv) if pcs is kzg else p


def argue(spec, w, transcript):
    """Steps 2-7 of the argument, everything between the commitments and the opening: Surge sum-check, memory-checking
    grand products, evaluations.  `w`: the witness dict (a, dim, read_ts, E, final_cts; unpadded).  Returns the four
    points and the claimed evaluations; the caller opens them (standalone: `prove`; inside HyperPlonk:
    oracle/pyref/hyperplonk.py, where a and dim are witness columns of the circuit)."""
    c, l, alpha = spec.c, spec.l, spec.alpha
    N, M = len(w["a"]), 1 << l
    n = N.bit_length() - 1
    r = transcript.squeeze_challenges(n)
    v = evaluate(w["a"], r)
    transcript.write_field_element(v)
    surge = ex.EqXY(0) * spec.g_expression()
    vp = sc.VirtualPolynomial(surge, w["E"], [], [r])
    r_z, e_rz = sc.prove(sc.EvaluationsProver, n, vp, v, transcript)
    transcript.write_field_elements(e_rz)

    gamma = transcript.squeeze_challenge()
    tau = transcript.squeeze_challenge()
    leaves_n, leaves_l = [], []
    for i, (j, kind) in enumerate(spec.memories):
        rs = [_fingerprint(w["dim"][j][k], w["E"][i][k], w["read_ts"][j][k], gamma, tau) for k in range(N)]
        ws = [(x + 1) % P for x in rs]
        init = [_fingerprint(m, subtable_entry(kind, m, l), 0, gamma, tau) for m in range(M)]
        final = [(init[m] + w["final_cts"][j][m]) % P for m in range(M)]
        leaves_n += [rs, ws]
        leaves_l += [init, final]
    _, claims = gkr.prove_grand_product(leaves_n + leaves_l, transcript)
    r_N, r_M = claims[0][1], claims[2 * alpha][1]

    dim_e = [evaluate(t, r_N) for t in w["dim"]]
    rts_e = [evaluate(t, r_N) for t in w["read_ts"]]
    e_e = [evaluate(t, r_N) for t in w["E"]]
    fc_e = [evaluate(t, r_M) for t in w["final_cts"]]
    transcript.write_field_elements(dim_e + rts_e + e_e + fc_e)
    return (r, r_z, r_N, r_M), (v, e_rz, dim_e, rts_e, e_e, fc_e)


def prove(pp, spec, dims, transcript, pcs=kzg):
    """pcs: `kzg` (pp = its Params) or `zeromorph` (pp = a trimmed ProverParam covering 2^max(n, l) coefficients)"""
    c, l, alpha = spec.c, spec.l, spec.alpha
    N = len(dims[0])
    n = N.bit_length() - 1
    assert N == 1 << n and all(len(d) == N for d in dims) and len(dims) == c
    _check_shape(spec, n)
    w = witness(spec, dims)
    transcript.common_field_elements([n, l, c, alpha])
    nv = max(n, l)
    polys = [_pad(p, nv) for p in [w["a"]] + w["dim"] + w["read_ts"] + w["E"] + w["final_cts"]]
    write_commitments(transcript, [pcs.commit(_trim(pcs, pp, nv), p) for p in polys])

    pts, vals = argue(spec, w, transcript)
    evals = _evals(spec, *vals)
    points = [_pad_point(pt, nv) for pt in pts]
    pcs.batch_open(_trim(pcs, pp, nv), nv, polys, points, evals, transcript)
    return transcript


def _pad(evals, nv):
    return list(evals) + [0] * ((1 << nv) - len(evals))


def _pad_point(pt, nv):
    return list(pt) + [0] * (nv - len(pt))


def _evals(spec, v, e_rz, dim_e, rts_e, e_e, fc_e):
    c, alpha = spec.c, spec.alpha
    out = [kzg.Evaluation(0, 0, v)]
    out += [kzg.Evaluation(1 + 2 * c + i, 1, e_rz[i]) for i in range(alpha)]
    out += [kzg.Evaluation(1 + j, 2, dim_e[j]) for j in range(c)]
    out += [kzg.Evaluation(1 + c + j, 2, rts_e[j]) for j in range(c)]
    out += [kzg.Evaluation(1 + 2 * c + i, 2, e_e[i]) for i in range(alpha)]
    out += [kzg.Evaluation(1 + 2 * c + alpha + j, 3, fc_e[j]) for j in range(c)]
    return out


# ------------------------------------------------------------------ verifier
def check(spec, n, transcript):
    """The verifier's side of `argue`: reads the messages, checks Surge and the memory-checking identities, returns the
    four points and the claimed evaluations that remain to be checked against the commitments."""
    c, l, alpha = spec.c, spec.l, spec.alpha
    r = transcript.squeeze_challenges(n)
    v = transcript.read_field_element()
    surge = ex.EqXY(0) * spec.g_expression()
    x_eval, r_z = sc.verify(sc.Evaluations, n, ex.degree(surge), v, transcript)
    e_rz = transcript.read_field_elements(alpha)
    if x_eval != eq_xy_eval(r_z, r) * spec.g_eval(e_rz) % P:
        raise LassoError("Surge sum-check final evaluation mismatch")

    gamma = transcript.squeeze_challenge()
    tau = transcript.squeeze_challenge()
    roots, claims = gkr.verify_grand_product([n] * (2 * alpha) + [l] * (2 * alpha), transcript)
    for i in range(alpha):
        rs, ws = roots[2 * i], roots[2 * i + 1]
        init, final = roots[2 * alpha + 2 * i], roots[2 * alpha + 2 * i + 1]
        if init * ws % P != rs * final % P:
            raise LassoError("memory %d: Init*WS != RS*Final" % i)
    r_N, r_M = claims[0][1], claims[2 * alpha][1]

    vals = transcript.read_field_elements(3 * c + alpha)
    dim_e, rts_e = vals[:c], vals[c:2 * c]
    e_e, fc_e = vals[2 * c:2 * c + alpha], vals[2 * c + alpha:]
    id_M = identity_eval(r_M)
    for i, (j, kind) in enumerate(spec.memories):
        rs = _fingerprint(dim_e[j], e_e[i], rts_e[j], gamma, tau)
        init = _fingerprint(id_M, subtable_mle_eval(kind, r_M), 0, gamma, tau)
        want = [rs, (rs + 1) % P, init, (init + fc_e[j]) % P]
        got = [claims[2 * i][0], claims[2 * i + 1][0],
               claims[2 * alpha + 2 * i][0], claims[2 * alpha + 2 * i + 1][0]]
        if want != got:
            raise LassoError("memory %d: leaf claim mismatch" % i)
    return (r, r_z, r_N, r_M), (v, e_rz, dim_e, rts_e, e_e, fc_e)


def verify(vp, spec, n, transcript, pcs=kzg):
    c, l, alpha = spec.c, spec.l, spec.alpha
    _check_shape(spec, n)
    transcript.common_field_elements([n, l, c, alpha])
    nv = max(n, l)
    comms = read_commitments(transcript, 1 + 3 * c + alpha)
    pts, vals = check(spec, n, transcript)
    evals = _evals(spec, *vals)
    points = [_pad_point(pt, nv) for pt in pts]
    pcs.batch_verify(_trim(pcs, vp, nv), nv, comms, points, evals, transcript)
    if transcript.pos != len(transcript.stream):
        raise LassoError("trailing bytes in proof")
