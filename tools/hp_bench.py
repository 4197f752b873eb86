"""HyperPlonk + LogUp prove time on one GPU for a synthetic circuit of the reference's
`vanilla_plonk_with_lookup` shape (halo2_lasso_amd.synthetic; SURVEY.md §8d C5 substitute).  Development aid:
prints per-kernel times from lh_profile and checks the proof with the host verifier.  The judged line comes from
`bench.py --workload hyperplonk`.

usage: python tools/hp_bench.py <log_rows> [reps]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import halo2_lasso_amd as hl  # noqa: E402
from halo2_lasso_amd import hyperplonk as hp, synthetic  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ctx = hl.Context(0)
    t0 = time.perf_counter()
    circ = synthetic.vanilla_plonk_with_lookup(ctx, k)
    print("circuit 2^%d rows built in %.1f s (%d copies, %d lookups)" % (k, time.perf_counter() - t0, circ.num_copies,
                                                                        circ.num_lookups), flush=True)
    t0 = time.perf_counter()
    rng = np.random.default_rng(k)
    ss = [int(v) for v in rng.integers(1, 1 << 62, size=k)]
    pcs_pp, pcs_vp = hl.MultilinearKzg.setup(ctx, ss), hl.MultilinearKzgVerifierParams.setup(ss)
    pp, vp = synthetic.prover_param(pcs_pp, circ, pcs_vp)
    print("setup + preprocess %.1f s" % (time.perf_counter() - t0), flush=True)

    best = None
    for rep in range(reps):
        tr = hl.Keccak256Transcript()
        ctx.sync()
        t0 = time.perf_counter()
        hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, tr)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
        proof = tr.into_proof()
        print("hyperplonk vanilla+lookup 2^%d: %.1f ms, proof %d B" % (k, dt, len(proof)), flush=True)
    t0 = time.perf_counter()
    hp.HyperPlonk.verify(vp, circ.instances, hl.Keccak256Transcript.from_proof(proof))
    print("verified by the host verifier in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)

    hl.profile_enable(ctx, True)
    hp.HyperPlonk.prove(pp, circ.instances, circ.d_witness, hl.Keccak256Transcript())
    recs = hl.profile_read(ctx)
    hl.profile_enable(ctx, False)
    agg = {}
    for r in recs:
        a = agg.setdefault(r["name"], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += r["ms"]
        a[2] += r["bytes"]
    total = sum(a[1] for a in agg.values())
    print("profiled kernels: %.1f ms total" % total)
    for name, (cnt, ms, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-28s %4d launches %8.3f ms  %7.1f GB/s" % (name, cnt, ms, by / ms / 1e6 if ms else 0))
    print(json.dumps({"workload": "hyperplonk vanilla_plonk_with_lookup", "log_rows": k, "prove_ms": round(best, 2),
                      "proof_bytes": len(proof)}))


if __name__ == "__main__":
    main()
