#!/usr/bin/env python3
"""bench.py -- Lasso prove time on MI355X (BASELINE.json metric), one JSON line on stdout.

Step = one Lasso prove of the workload with the lookup indices and the SRS already resident in HBM.  Default
workload = BASELINE.json configs[2], the configuration north_star states its target on: 2^24 AND lookups (32-bit
operands as 4 chunks of 8+8 bits into the 2^16-entry AND subtable, Surge), BN254, multilinear-KZG openings.
`--log-n 20 --table range` is configs[1].

`--gpus N` (one rank per GPU: launched by torch.distributed.run as the driver does, or - when WORLD_SIZE is not set -
bench.py starts the N ranks itself before anything touches a GPU and relays rank 0's line): ONE proof of the same
2^log-n lookups sharded over the N GPUs (tables and SRS split on mid index bits; the same prover as on
one GPU runs on every shard, per-round partial sums and partial commitments cross over RCCL on the prover's stream):
strong scaling, value = ms per proof.  `--mode replicas` instead lets every rank prove its own batch (weak scaling, no
data-path collective).

Extra objects on the line:
  roofline      the kernel with the largest total time in a separately profiled prove: SURVEY.md §8(d) algorithmic
                bytes / HIP-event time (ctx stream), against 8 TB/s HBM
  alu           the same launches against the Fr-multiplication peak (the field arithmetic is integer-ALU bound)
  cpu_baseline  the C++ oracle (reference algorithms, all host cores) on a bounded sample, proof bytes compared
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SEED_BASE = 0x4C4153534F00  # SURVEY.md §8d (same constant as halo2-lasso_amd/dist.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=0,
                    help="log2 of the number of lookups per proof (default 24: BASELINE.json configs[2], the config "
                         "north_star's target is stated on; --log-n 20 --table range is configs[1]; hyperplonk: 20)")
    ap.add_argument("--table", default="and", choices=["range", "and", "xor"])
    ap.add_argument("--pcs", default="mkzg", choices=["mkzg", "zeromorph"],
                    help="polynomial commitment scheme of the Lasso workload: multilinear KZG (default, the metric's "
                         "configuration) or Zeromorph over univariate KZG")
    ap.add_argument("--workload", default="lasso", choices=["lasso", "hyperplonk"],
                    help="'lasso' (default, BASELINE metric) or 'hyperplonk': HyperPlonk + LogUp prove of a synthetic "
                         "vanilla_plonk_with_lookup circuit of 2^log-n rows (SURVEY.md §8d C5 substitute)")
    ap.add_argument("--lookup", default="logup", choices=["logup", "lasso"],
                    help="--workload hyperplonk: the circuit's lookup argument: 'logup' (the reference's, a 3-column "
                         "vanilla_plonk_with_lookup circuit) or 'lasso' (north_star's HyperPlonk + Lasso: vanilla gates + "
                         "a 32-bit --table lookup proven by Lasso inside HyperPlonk::prove, the configs[4] stand-in)")
    ap.add_argument("--circuit", default="vanilla", choices=["vanilla", "keccak"],
                    help="--workload hyperplonk --lookup lasso: 'keccak' = BASELINE.json configs[4]: Keccak-f[1600] "
                         "permutations (35013 rows each) packed into 2^log-n rows, every byte XOR / AND a Lasso lookup "
                         "(halo2-lasso_amd/keccak_circuit.py); 'vanilla' = vanilla gates + one 32-bit --table lookup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bind-host", action="store_true",
                    help="bind the process to the CPUs on its GPU's NUMA node (Context.bind_host; default: wherever the "
                         "scheduler puts it - on the shared hosts of the development pool the binding measured within the noise)")
    ap.add_argument("--cpu-sample-log-n", type=int, default=0, help="force the CPU sample size")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-inflight", action="store_true",
                    help="skip the extra throughput figure (two independent proofs in flight on one GPU)")
    ap.add_argument("--no-extra", action="store_true",
                    help="N>1: skip the extra objects (configs[3] 2^26 sharded, replicas)")
    ap.add_argument("--mode", default="sharded", choices=["replicas", "sharded"],
                    help="N>1: 'sharded' (default) = ONE proof of 2^log-n lookups split over the N GPUs (strong "
                         "scaling, SURVEY.md §8e, partial sums over RCCL); 'replicas' = one independent proof per GPU "
                         "(weak scaling, no data-path collective)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group, add up the ranks and print {n_gpus, sum} - no GPU work "
                         "(tests/test_dist.py: the launch path of --gpus N on a CPU-only machine)")
    args = ap.parse_args()
    if not args.log_n:
        args.log_n = 20 if args.workload == "hyperplonk" else 24
    return args


def make_table(hl, kind):
    if kind == "range":
        return hl.LassoTable.range(2, 16), "2^%d range-check Lasso lookup (32-bit values, 2x16-bit identity subtable)"
    k = hl.SUBTABLE_AND if kind == "and" else hl.SUBTABLE_XOR
    return hl.LassoTable.bitwise(k, 4, 16), "2^%d " + kind.upper() + " Lasso lookup (32-bit operands, 4 chunks of 8+8 bits)"


def gen_dims(table, n, rank):
    from halo2_lasso_amd import dist as hdist
    rng = np.random.Generator(np.random.PCG64(hdist.batch_seed(n, rank)))
    return [rng.integers(0, 1 << table.l, size=1 << n, dtype=np.uint32) for _ in range(table.c)]


def trapdoor(nv):
    rng = np.random.Generator(np.random.PCG64(SEED_BASE))
    return [int.from_bytes(rng.bytes(31), "little") + 1 for _ in range(nv)]


def aggregate(recs):
    by = {}
    for r in recs:
        a = by.setdefault(r["name"], dict(name=r["name"], launches=0, ms=0.0, bytes=0.0, muls=0.0, items=0.0, big=None))
        a["launches"] += 1
        a["ms"] += r["ms"]
        a["bytes"] += r["bytes"]
        a["muls"] += r["muls"]
        a["items"] += r["items"]
        if a["big"] is None or r["bytes"] > a["big"]["bytes"]:
            a["big"] = r
    return sorted(by.values(), key=lambda a: -a["ms"])


PMC_KERNEL_NAMES = {"msm_accumulate0": "msm_accumulate0_kernel", "msm_bucket_reduce": "msm_segment_reduce_kernel",
                    "sc_round_pp<bind>": "sc_round_pp_kernel<true>", "sc_round_pp<first>": "sc_round_pp_kernel<false>",
                    "sc_round_rw<bind>": "sc_round_rw_kernel<4, true>", "gkr_resident": "gkr_resident_kernel",
                    "msm_accumulate_levels": "msm_accumulate_n_kernel", "lincomb": "lincomb_kernel",
                    "tree_up": "tree_up_kernel", "sc_round_open<bind>": "sc_round_open_kernel<true>",
                    "sc_round_open<first>": "sc_round_open_kernel<false>"}


def pmc_traffic(profile_name, log_n, table_kind, world=1, launches_per_proof=None):
    """HBM bytes per launch of the kernel (average over its launches, like `achieved`) from the PMC passes committed under
    profiles/ (tools/pmc_extract.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of the same
    workload, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for streaming reads).  Counters cannot be collected
    from inside this process, so the figure is the recorded one; it is used only when the record is of THIS workload
    and the kernel's number of launches per proof matches this run's (a changed launch structure means a stale
    record: null then)."""
    if world != 1:
        return None, None
    import glob
    import re
    m = re.match(r"sc_round<(\d),(bind|first)>(/lds|/tp)?$", profile_name)
    if m:
        kern = "sc_round_%skernel<%s, %s>" % ("lds_" if m.group(3) == "/lds" else "", m.group(1),
                                              "true" if m.group(2) == "bind" else "false")
    else:
        kern = PMC_KERNEL_NAMES.get(profile_name)
    if not kern:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_2p%d*.json" % (table_kind, log_n))), reverse=True):
        recs = json.load(open(path))["kernels"]
        rec = recs.get(kern) or recs.get(kern.split("<")[0])  # (records of earlier rounds: the kernel was not a template)
        if not rec or "hbm_bytes_per_launch_avg" not in rec:
            continue
        if launches_per_proof is not None and abs(rec["launches_per_proof"] - launches_per_proof) > 1e-9:
            continue
        note = "" if rec.get("calibrated", True) else ("; gather kernel: a 64-byte gather counts as its whole 128-byte line, "
                                                        "profiles/r05_gather_calib.txt")
        return rec["hbm_bytes_per_launch_avg"], "profiles/%s (%s, average over %d launches%s)" % (
            os.path.basename(path), kern, rec["launches_in_run"], note)
    return None, None


def fr_mul_peak(hl, ctx):
    """measured peak Fr multiplications / s (two independent chains per thread, full occupancy, 256 products per
    element so that the loads and stores do not count)"""
    n, iters = 1 << 22, 256
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 1 << 60, size=4 * n, dtype=np.uint64).tobytes()
    a, b, out = ctx.upload(raw), ctx.upload(raw[::-1]), ctx.alloc(32 * n)
    best = 0.0
    for _ in range(3):
        ctx.lib.lh_fr_mul_chain(ctx.h, a.ptr, b.ptr, n, iters, out.ptr)
        ctx.sync()
        t = time.perf_counter()
        ctx.lib.lh_fr_mul_chain(ctx.h, a.ptr, b.ptr, n, iters, out.ptr)
        ctx.sync()
        best = max(best, n * iters / (time.perf_counter() - t))
    return best


def cpu_baseline(hl, ctx, pp, table, kind, args, gpu_proof_fn):
    unbind_host()
    co = use_native_oracle()
    cores = co.num_threads()
    zm = isinstance(pp, hl.ZeromorphProverParam)
    if zm:
        srs = C.create_string_buffer(64 * pp.params.size)
        hl._check(ctx.lib.lh_usrs_download(ctx.h, pp.params.h, srs))
    else:
        srs_nv = pp.num_vars
        srs = C.create_string_buffer(64 * ((2 << srs_nv) - 1))
        hl._check(ctx.lib.lh_srs_download(ctx.h, pp.h, srs))
    tc = table.to_c()

    def run(n):
        dims = gen_dims(table, n, 0)
        tr = co.Transcript()
        t = time.perf_counter()
        if zm:
            co.lasso_prove_zm(tr, srs.raw, pp.poly_size, tc, n, [d.tobytes() for d in dims])
        else:
            co.lasso_prove(tr, srs, srs_nv, tc, n, [d.tobytes() for d in dims])
        return (time.perf_counter() - t) * 1e3, tr.into_proof(), dims

    n = args.cpu_sample_log_n
    if not n:
        # calibrate on 2^16, then take the largest sample expected to stay under ~25 s
        ms16, _, _ = run(min(16, args.log_n))
        n = min(16, args.log_n)
        while n < args.log_n and ms16 * (1 << (n + 1 - 16)) < 25e3:
            n += 1
        # at least 2^21 lookups: from there on the GPU prover takes the route of the full-size proof (column-wise top
        # quotient, packed pairs, derived commitments), so `proof_bytes_equal_gpu` covers the route that was timed
        n = max(n, min(21, args.log_n))
    ms, proof, dims = run(n)
    same = gpu_proof_fn(n, dims) == proof
    frac = "the full workload" if n == args.log_n else "1/%d of the workload's lookups" % (1 << (args.log_n - n))
    return {"value": round(ms, 2), "unit": "ms", "cores": cores, "kind": "port",
            "sample": ("one Lasso prove of " + kind + " at 2^%d lookups (" + frac + "), same SRS and PRNG stream as the "
                       "GPU run; C++ oracle = reference algorithms, chunk-per-thread") % n,
            "sample_log_n": n, "proof_bytes_equal_gpu": bool(same)}


# The integer-ALU reference of the field arithmetic (csrc/ff.cuh) is the MEASURED chain (lh_fr_mul_chain: two independent
# product chains per thread).  Rounds 1-4 printed a modelled "ceiling" next to it (~296 VALU instructions per product at a
# 4-cycle wave64 issue: 133 G products/s); the measured chain sits at 125-138 G/s, i.e. it could beat the model - a ceiling
# the chip beats is not one, so it is gone: `alu.frac` is against the measured chain only.


# --bind-host: one process per GPU, bound to the CPUs on the GPU's NUMA node (what numactl does in a deployment): the
# proving thread and the device exchange a few hundred small messages per proof, and each is longer across the socket
# interconnect (tools/numa_ab.sh on a quiet host: 2^20 lookups 8.09 local / 8.26 remote ms; on a host shared with other
# jobs the difference drowns).  The CPU baseline runs on EVERY core the process was given: the binding is undone before it.
_HOST_BINDING = {"before": None}


def bind_host(ctx, args):
    if not args.bind_host:
        return {"bound": False}
    bus, local = ctx.host_cpus()
    before = ctx.bind_host()
    _HOST_BINDING["before"] = before
    if before is None:
        return {"bound": False, "device": bus, "why": "no local CPU list, or already inside it"}
    return {"bound": True, "device": bus, "cpus": len(local & before), "of": len(before)}


def unbind_host():
    if _HOST_BINDING["before"] is not None:
        os.sched_setaffinity(0, _HOST_BINDING["before"])
        _HOST_BINDING["before"] = None


def dominant(aggs):
    """the kernel the roofline is priced on: the largest total time among the kernels that move data (the resident
    sum-check tail keeps its tables in LDS: no algorithmic bytes, latency-bound by construction)"""
    moving = [a for a in aggs if a["bytes"] > 0]
    return max(moving or aggs, key=lambda a: a["ms"])


def live_accumulate(recs):
    """the live records of the timed region (lh_profile_enable(ctx, 2): a HIP-event pair around every msm_accumulate0
    launch on its own stream, nothing synchronised) -> {launches, ms, bytes, muls, items, batches, batch_ms}: `ms` adds the
    launches' SPANS (the two halves of a pipelined batch run at the same time: what a kernel trace adds up too), `batch_ms`
    the batches' spans (first start to last end: the time the chip spent on them)"""
    one = [r for r in recs if r["name"] == "msm_accumulate0"]
    bat = [r for r in recs if r["name"] == "msm_accumulate0/batch"]
    if not one or not bat:
        return None
    return {"launches": len(one), "ms": sum(r["ms"] for r in one), "bytes": sum(r["bytes"] for r in one),
            "muls": sum(r["muls"] for r in one), "items": sum(r["items"] for r in one), "batches": len(bat),
            "batch_ms": sum(r["ms"] for r in bat)}


def roofline_objects(hl, ctx, aggs, traffic=(None, None), live=None, steps=1):
    """`roofline`: the kernel with the largest total time in the profiled prove, priced in SURVEY.md §8(d)'s
    algorithmic bytes (sum over its launches) / its HIP-event time (sum over its launches) against 8 TB/s HBM;
    `alu`: the same launches against the measured Fr-multiplication peak.  When that kernel is the bucket accumulation and
    `live` (live_accumulate) holds its launches of the TIMED region, the durations are those - measured while the proofs ran,
    on the streams the launches went to - and the profiled prove only supplies the table of the other kernels."""
    tot = sum(a["ms"] for a in aggs) or 1.0
    peak_mul = fr_mul_peak(hl, ctx)
    dom = dominant(aggs)
    top = max(aggs, key=lambda a: a["ms"])
    share = dom["ms"] / tot
    live_note = None
    if live and dom["name"] == "msm_accumulate0":
        dom = dict(dom, ms=live["ms"], launches=live["launches"], bytes=live["bytes"], muls=live["muls"], items=live["items"])
        live_note = ("durations: HIP events around every launch of the %d TIMED proofs, on the stream each launch went to, "
                     "nothing synchronised; a pipelined MSM batch runs its two halves at the same time on two streams, so a "
                     "launch's duration is its span (a kernel trace shows the same) and the chip's rate is `alu.frac_chip`"
                     % steps)
    avg_ms = dom["ms"] / dom["launches"]
    ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
    roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(ach, 1), "peak": 8000.0,
            "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic[0], "traffic_source": traffic[1],
            "launches": dom["launches"], "avg_launch_ms": round(avg_ms, 4),
            "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
            "items_per_launch": dom["items"] / dom["launches"],
            "share_of_profiled_prove": round(share, 3),
            "accounting": "SURVEY.md 8(d): MSM 96 B per point with a 32-byte scalar, 68 B with a u32 scalar, summed "
                          "over the jobs of the batch; sum-check 96 B per bound entry; see DESIGN.md section 3"}
    if live_note:
        roof["timing"] = live_note
        if live["batch_ms"] > 0:  # the same bytes against the time the chip spent on the batches (overlapping launches counted once)
            chip = live["bytes"] / (live["batch_ms"] * 1e-3) / 1e9
            roof["achieved_chip"], roof["frac_chip"] = round(chip, 1), round(chip / 8000.0, 4)
    if top["name"] != dom["name"]:
        roof["note"] = ("largest share of this prove: %s (%.0f %%, rounds resident in LDS: no HBM traffic, latency-bound); "
                        "the roofline is priced on the largest kernel that moves data" % (top["name"], 100.0 * top["ms"] / tot))
    mul_rate = dom["muls"] / (dom["ms"] * 1e-3) if dom["ms"] > 0 else 0.0
    alu = {"bound": "int32-mul", "kernel": dom["name"], "achieved": round(mul_rate / 1e9, 2),
           "peak": round(peak_mul / 1e9, 2), "unit": "G Fr-mul/s", "frac": round(mul_rate / peak_mul, 4),
           "peak_source": "measured (lh_fr_mul_chain: two independent product chains per thread, 256 products per "
                          "element)",
           "counting": "`achieved` counts a mixed addition as 10 products (the two of Y3 share a reduction); ~296 VALU "
                       "instructions per product, each a ~4-cycle wave64 issue (profiles/r04_ubench_mul_fp64.txt)"}
    if live_note and live["batch_ms"] > 0:
        # the same products against the time the chip spent on the batches (launches that overlap counted once)
        chip = live["muls"] / (live["batch_ms"] * 1e-3)
        alu["achieved_chip"] = round(chip / 1e9, 2)
        alu["frac_chip"] = round(chip / peak_mul, 4)
        alu["batch_ms_per_proof"] = round(live["batch_ms"] / max(steps, 1), 3)
    kernels = [{"name": a["name"], "launches": a["launches"], "ms": round(a["ms"], 3),
                "GBps_all": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1) if a["ms"] > 0 else 0.0,
                "GBps_largest": round(a["big"]["bytes"] / (a["big"]["ms"] * 1e-3) / 1e9, 1) if a["big"]["ms"] > 0 else 0.0}
               for a in aggs[:12]]
    return roof, alu, kernels


def use_native_oracle():
    from oracle import cpu_oracle as co
    # a library tuned for this host if the toolchain is here; else the portable prebuilt one
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "cpu"), "NATIVE=1"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        native = os.path.join(ROOT, "oracle", "_build", "liboracle_cpu_native.so")
        if os.path.exists(native):
            co.LIB_PATH = native
            co._lib = None
    except Exception:
        pass
    return co


def make_hp_circuit(ctx, k, args, seed=None):
    from halo2_lasso_amd import synthetic
    if args.lookup == "lasso" and args.circuit == "keccak":
        return synthetic.keccak_f(ctx, k, seed=seed)
    if args.lookup == "lasso":
        return synthetic.vanilla_plonk_with_lasso(ctx, k, kind=args.table, seed=seed)
    return synthetic.vanilla_plonk_with_lookup(ctx, k, seed=seed)


def hyperplonk_cpu_baseline(hl, ctx, args, trap, gpu_proof_fn):
    """the C++ oracle's HyperPlonk restatement on the same synthetic circuit (bounded sample), proof bytes compared"""
    unbind_host()
    from halo2_lasso_amd import synthetic
    from oracle.pyref import hyperplonk as o_hp
    co = use_native_oracle()

    def run(k):
        circ = make_hp_circuit(ctx, k, args)
        pp = hl.MultilinearKzg.setup(ctx, trap[:k])
        srs = C.create_string_buffer(64 * ((2 << k) - 1))
        hl._check(ctx.lib.lh_srs_download(ctx.h, pp.h, srs))
        if args.lookup == "lasso" and args.circuit == "keccak":
            from oracle.pyref import lasso as o_lasso
            o_info = o_hp.keccak_circuit_info(k, [[]] * 7, [[(8, 1)], [(9, 1)], [(10, 1)]],
                                              o_lasso.bitwise_table(o_lasso.SUBTABLE_XOR, 1, 16),
                                              o_lasso.bitwise_table(o_lasso.SUBTABLE_AND, 1, 16))
            lasso_lookups = [(lk.table.to_c(), lk.output_poly, lk.chunk_polys) for lk in circ.info.lasso_lookups]
            perm_idx = [8, 9, 10]
        elif args.lookup == "lasso":
            from oracle.pyref import lasso as o_lasso
            spec = o_lasso.range_table(2, 16) if args.table == "range" else o_lasso.bitwise_table(
                o_lasso.SUBTABLE_AND if args.table == "and" else o_lasso.SUBTABLE_XOR, 4, 16)
            o_info = o_hp.vanilla_plonk_with_lasso_circuit_info(k, 0, [[]] * 6, [[(7, 1)], [(8, 1)], [(9, 1)]], spec)
            lk = circ.info.lasso_lookups[0]
            lasso_lookups, perm_idx = [(lk.table.to_c(), lk.output_poly, lk.chunk_polys)], [7, 8, 9]
        else:
            o_info = o_hp.vanilla_plonk_with_lookup_circuit_info(k, 0, [[]] * 9, [[(10, 1)], [(11, 1)], [(12, 1)]])
            lasso_lookups, perm_idx = [], [10, 11, 12]
        num_z, expression = o_hp.compose(o_info)
        lookups = [[(co.flatten_expression(i), co.flatten_expression(t)) for i, t in lk] for lk in o_info.lookups]
        perm = [p.buf.read() for p in circ.d_permutation]
        tr = co.Transcript()
        t = time.perf_counter()
        co.hyperplonk_prove(tr, srs, k, k, [0], [a.tobytes() for a in circ.h_preprocess], len(circ.h_witness), 0, lookups,
                            perm_idx, perm, num_z, co.flatten_expression(expression), [[]],
                            [a.tobytes() for a in circ.h_witness], lasso_lookups=lasso_lookups)
        ms = (time.perf_counter() - t) * 1e3
        return ms, tr.into_proof() == gpu_proof_fn(pp, circ)

    k = args.cpu_sample_log_n
    if not k:
        k = min(16 if args.lookup == "lasso" else 14, args.log_n)  # a Lasso circuit has at least 2^16 rows (the subtable)
        ms, _ = run(k)
        while k < args.log_n and ms * 2.2 < 25e3:
            k += 1
            ms *= 2.2
    ms, same = run(k)
    frac = "the full workload" if k == args.log_n else "1/%d of the workload's rows" % (1 << (args.log_n - k))
    return {"value": round(ms, 2), "unit": "ms", "cores": co.num_threads(), "kind": "port",
            "sample": ("one HyperPlonk prove of the same synthetic circuit at 2^%d rows (" + frac + "); C++ oracle = "
                       "reference algorithms (expression evaluator per point, evaluate-then-bind, Pippenger per thread "
                       "chunk)") % k,
            "sample_log_n": k, "proof_bytes_equal_gpu": bool(same)}


def main_hyperplonk(args, hdist, dist, rank, local_rank, world):
    import halo2_lasso_amd as hl
    from halo2_lasso_amd import hyperplonk as hp, synthetic
    ctx = hl.Context(int(os.environ.get("LH_DEVICE", local_rank)))
    host_binding = bind_host(ctx, args)
    k = args.log_n
    trap = trapdoor(k)
    pcs_pp = hl.MultilinearKzg.setup(ctx, trap)
    # N > 1, --mode sharded (default), a circuit whose lookups are Lasso lookups: ONE proof split over the N GPUs
    # (lh_hyperplonk_prove_sharded: every rank holds its rows of every poly; strong scaling).  LogUp circuits do not
    # shard (global sort-merge join) and run as replicas, as does --mode replicas.
    sharded = world > 1 and args.mode == "sharded" and args.lookup == "lasso"
    circ = make_hp_circuit(ctx, k, args, seed=hdist.batch_seed(k, 0 if sharded else rank) & 0xffffffff)
    pp = synthetic.prover_param(pcs_pp, circ)
    ctx.sync()
    transport = None
    if sharded:
        rho = world.bit_length() - 1
        assert 1 << rho == world, "sharded mode needs a power-of-two number of GPUs"
        shard_bit = max(16 - rho, min(10, k - rho - 1), 1)  # (the replicated 2^16-entry subtables: shard_bit + rho >= 16)
        pp_local = hp.HyperPlonk.shard_param(pp, rank, world, shard_bit)
        wit_local = [hl.shard_poly(p, rank, world, shard_bit) for p in circ.d_witness]
        transport = hdist.attach_sharded(ctx, dist, shard_bit)

    def prove(p=pp, c=circ, single=False):
        tr = hl.Keccak256Transcript()
        if sharded and not single:
            hp.HyperPlonk.prove_sharded(pp_local, c.instances, wit_local, tr)
        else:
            hp.HyperPlonk.prove(p, c.instances, c.d_witness, tr)
        return tr

    def barrier():
        ctx.sync()
        if dist is not None and dist.get_backend() == "nccl":
            import torch
            torch.cuda.synchronize()
        hdist.barrier(dist)

    for _ in range(args.warmup):
        prove()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr = prove()
    ctx.sync()
    elapsed = hdist.max_over_ranks(dist, time.perf_counter() - t0)
    hdist.barrier(dist)
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    if rank == 0:
        out = {
            "metric": "hyperplonk_prove_time_ms", "value": round(ms_per_step / (1 if sharded else world), 3), "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": False, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u256 (BN254 Fr/Fq, 8x u32 Montgomery)", "data": "synthetic",
            "config": {"workload": ("HyperPlonk + Lasso prove of a Keccak-f[1600] circuit (BASELINE configs[4]), 2^%d rows: %d "
                                    "permutations of 24 rounds, 35013 byte-operation rows each, every XOR / AND a Lasso "
                                    "lookup into a 2^16-entry subtable (15 polys, %d copy constraints)"
                                    % (k, circ.num_permutations, circ.num_copies))
                       if args.lookup == "lasso" and args.circuit == "keccak" else
                       ("HyperPlonk + Lasso prove (BASELINE configs[4] stand-in), 2^%d rows: vanilla gates on "
                                    "half of the rows, a 32-bit %s lookup proven by Lasso inside HyperPlonk::prove on "
                                    "the other half (%d polys, %d copy constraints)"
                                    % (k, args.table.upper(), 7 + len(circ.h_witness), circ.num_copies))
                       if args.lookup == "lasso" else
                       ("HyperPlonk + LogUp prove, vanilla_plonk_with_lookup circuit, 2^%d rows (13 polys, one "
                        "3-column lookup on 1/4 of the rows, %d copy constraints, degree-5 zero-check)"
                        % (k, circ.num_copies)),
                       "rows": 1 << k, "proofs_per_step": 1 if sharded else world, "pcs": "multilinear KZG (BN254)",
                       "proof_bytes": len(tr.into_proof()),
                       "parallelism": ("1 proof sharded over %d GPUs (row index bits [%d, %d)), transport %s"
                                       % (world, shard_bit, shard_bit + world.bit_length() - 1, transport)) if sharded
                       else "1 proof per GPU" if world > 1 else "1 GPU"},
            "rows_per_s": round((1 << k) * (1 if sharded else world) / (ms_per_step / 1e3)),
            "host_binding": host_binding,
        }
    if not args.no_profile and (sharded or rank == 0):
        # (sharded: every rank takes part in the collectives of the profiled prove, rank 0 records)
        if rank == 0:
            hl.profile_enable(ctx, True)
        prove()
        ctx.sync()
        if rank == 0:
            aggs = aggregate(hl.profile_read(ctx))
            hl.profile_enable(ctx, False)
            out["roofline"], out["alu"], out["kernels"] = roofline_objects(hl, ctx, aggs)
    if sharded:
        stats = hl.comm_stats(ctx)
        if rank == 0:
            out["comm_collectives_per_run"] = stats
            out["sharded_proof_equals_single_gpu"] = prove(single=True).into_proof() == tr.into_proof()
        if not args.no_extra:
            # TWO sharded proofs in flight per rank (main(): sharded_two_in_flight): a second ctx with its own communicator, a
            # host thread each - at this size a sharded proof IS its latency-bound floor, which a second proof fills
            try:
                import threading
                ctx2 = hl.Context(int(os.environ.get("LH_DEVICE", local_rank)))
                pp2 = hp.HyperPlonk.rebind_param(pp_local, ctx2)
                transport2 = hdist.attach_sharded(ctx2, dist, shard_bit, group=hdist.flight_group(dist))

                def flight(p, n_proofs):
                    for _ in range(n_proofs):
                        hp.HyperPlonk.prove_sharded(p, circ.instances, wit_local, hl.Keccak256Transcript())
                tif_steps = max(1, min(args.steps, 5))
                elapsed2 = None
                for n_proofs in (1, tif_steps):
                    th = [threading.Thread(target=flight, args=(p, n_proofs)) for p in (pp_local, pp2)]
                    ctx2.sync()
                    barrier()
                    t0 = time.perf_counter()
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    ctx.sync(), ctx2.sync()
                    elapsed2 = hdist.max_over_ranks(dist, time.perf_counter() - t0)
                    hdist.barrier(dist)
                hl.detach_comm(ctx2)
                if rank == 0:
                    out["sharded_two_in_flight"] = {"ms_per_proof": round(elapsed2 * 1e3 / (2 * tif_steps), 3),
                                                    "steps_per_flight": tif_steps, "transport": transport2}
                del pp2, ctx2
            except Exception as e:
                if rank == 0:
                    out["sharded_two_in_flight"] = {"error": "%s: %s" % (type(e).__name__, e)}
        hl.detach_comm(ctx)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            def gpu_proof(pcs, c):
                return prove(synthetic.prover_param(pcs, c), c).into_proof()
            try:
                out["cpu_baseline"] = hyperplonk_cpu_baseline(hl, ctx, args, trap, gpu_proof)
            except Exception as e:
                if os.environ.get("LH_BENCH_DEBUG"):
                    raise
                out["cpu_baseline"] = {"value": None, "unit": "ms", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "unavailable: %s" % e}
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    from halo2_lasso_amd import dist as hdist
    rank, local_rank, world = hdist.env_rank()
    dist = hdist.init()
    # which attempt of the job this process is (sharded_fallback below): the first, the sharded retry with the personalised
    # exchange staged through all-gathers, or the replicas successor
    stage = "replicas" if "LH_BENCH_MODE_FALLBACK" in os.environ else "a2a" if "LH_BENCH_A2A_RETRY" in os.environ else "first"
    if stage in os.environ.get("LH_BENCH_TEST_RAISE", "").split(","):  # (tests/test_dist.py: the fallback chain on CPU)
        raise RuntimeError("forced failure of the %s attempt (LH_BENCH_TEST_RAISE)" % stage)
    if args.rendezvous_only:
        total = hdist.max_over_ranks(dist, float(rank))
        if rank == 0:
            line = {"n_gpus": world, "max_rank": int(total), "rendezvous": "ok"}
            if stage != "first":
                line.update(stage=stage, mode=args.mode, a2a=os.environ.get("LH_COMM_A2A", ""))
            emit(line)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload == "hyperplonk":
        return main_hyperplonk(args, hdist, dist, rank, local_rank, world)

    import halo2_lasso_amd as hl
    ctx = hl.Context(int(os.environ.get("LH_DEVICE", local_rank)))  # LH_DEVICE: several ranks on one GPU (tests)
    host_binding = bind_host(ctx, args)
    n = args.log_n
    table, desc = make_table(hl, args.table)
    zm = args.pcs == "zeromorph"
    sharded = args.mode == "sharded" and world > 1
    assert not (zm and sharded), "the sharded mode is implemented for multilinear KZG"

    def setup(nv):
        if zm:
            return hl.Zeromorph.trim(hl.Zeromorph.setup(ctx, trapdoor(1)[0], 1 << nv), 1 << nv)
        return hl.MultilinearKzg.setup(ctx, trapdoor(nv))

    def shard_geometry(tb, nn):
        rho = world.bit_length() - 1
        assert 1 << rho == world, "sharded mode needs a power-of-two number of GPUs"
        # the lowest shard bit the replicated subtables allow (2^l <= 2^(shard_bit + rho)): everything with more than
        # shard_bit + rho variables - tree levels, quotient levels - then stays sharded; when a sum-check's residual
        # tables travel is decided by their size (option shard_exchange_log), not by shard_bit
        return max(tb.l - rho, min(10, nn - rho - 1), 1)

    def load_columns(tb, nn, shard_bit):
        """device columns of this rank: its shard of the (rank-0 seeded) batch when sharded, else its own batch"""
        cols = gen_dims(tb, nn, 0 if sharded else rank)
        if sharded:
            cols = [hl.shard_of(c, rank, world, shard_bit) for c in cols]
        return [ctx.upload(c.tobytes()) for c in cols]

    pp = setup(max(n, table.l))
    transport = None
    shard_bit = shard_geometry(table, n) if sharded else 0
    d_dims = load_columns(table, n, shard_bit)
    ctx.sync()

    def prove(nn=n, bufs=None, single=False, p=None, tb=None):
        tr = hl.Keccak256Transcript()
        if sharded and not single:
            hl.lasso_prove_sharded(p or pp, tb or table, nn, bufs or d_dims, tr)
        else:
            hl.lasso_prove(p or pp, tb or table, nn, bufs or d_dims, tr)
        return tr

    def barrier():
        ctx.sync()
        if dist is not None and dist.get_backend() == "nccl":
            import torch
            torch.cuda.synchronize()
        hdist.barrier(dist)

    live_box = {}

    def timed(steps, warmup, live=False, **kw):
        for _ in range(warmup):
            prove(**kw)
        if live:  # event pairs around the accumulation launches of the timed proofs (no synchronisation: lasso_hip.h)
            hl.profile_enable(ctx, 2)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr = prove(**kw)
        ctx.sync()
        if dist is not None and dist.get_backend() == "nccl":
            import torch
            torch.cuda.synchronize()
        elapsed = hdist.max_over_ranks(dist, time.perf_counter() - t0)
        hdist.barrier(dist)
        if live:
            live_box["acc"] = live_accumulate(hl.profile_read(ctx))
            hl.profile_enable(ctx, 0)
        return elapsed * 1e3 / max(steps, 1), tr

    replicas = None
    watchdog = None
    headline = {}  # rank 0: the sharded line once it exists (the extras that follow run collectives too)
    if sharded:
        # N independent replicas first (no data-path collective: nothing in it can wait for another rank's GPU) - the extra
        # object of the sharded line
        if not args.no_extra:
            extra_steps = max(1, min(args.steps, 3))
            own = [ctx.upload(c.tobytes()) for c in gen_dims(table, n, rank)]
            msr, trr = timed(extra_steps, 1, bufs=own, single=True)
            del own
            replicas = {"ms_per_step": round(msr, 3), "proofs_per_step": world, "scaling": "weak",
                        "ms_per_proof": round(msr / world, 3), "lookups_per_s": round((1 << n) * world / (msr / 1e3))}

        def give_up():
            # a collective that never completes cannot be caught as an exception.  Two timers: the HEADLINE one
            # (LH_BENCH_SHARDED_TIMEOUT seconds, default 240: replicas done, communicator attached, warm-up and the timed
            # proofs) - when it fires the run FAILS with exit code 3 (the other ranks sit in the same collective; their
            # timers do the same) and what goes out is the replicas figure taken BEFORE the sharded proof started, marked
            # as a fallback, next to the error: a process with a kernel stuck on the GPU is not trusted with new
            # measurements.  Once the headline is measured the EXTRAS timer (LH_BENCH_EXTRAS_TIMEOUT, default 420: the
            # profiled prove, the single-GPU re-prove for the equality check, configs[3] with its 2^26 SRS) takes over: a
            # slow or stuck extra cannot fail the headline - its line goes out as measured, with `extras_error`, exit code 0.
            if headline.get("done"):
                if rank == 0:
                    late = dict(headline["line"])
                    late["extras_error"] = ("an extra object after the headline did not complete within "
                                            "LH_BENCH_EXTRAS_TIMEOUT seconds; the headline above was measured before it")
                    late.setdefault("replicas", replicas)
                    emit(late)
                os._exit(0)
            if rank == 0:
                err = "the sharded proof did not complete within LH_BENCH_SHARDED_TIMEOUT seconds; exit code 3"
                sys.stderr.write(json.dumps({"error": err, "replicas_measured_before": replicas}) + "\n")
                sys.stderr.flush()
                if replicas is not None:
                    emit({
                        "metric": "lasso_prove_time_ms", "value": replicas["ms_per_proof"], "unit": "ms", "n_gpus": world,
                        "steps": max(1, min(args.steps, 3)), "warmup": 1, "ms_per_step": replicas["ms_per_step"],
                        "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
                        "dtype": "u256 (BN254 Fr/Fq, 8x u32 Montgomery)", "data": "synthetic",
                        "config": {"workload": desc % n, "lookups_per_proof": 1 << n, "proofs_per_step": world,
                                   "pcs": "multilinear KZG (BN254)", "parallelism": "1 proof per GPU"},
                        "lookups_per_s": replicas["lookups_per_s"],
                        "mode_fallback": {"ran": "replicas (measured before the sharded proof was started)",
                                          "sharded_error": err}})
            os._exit(3)
        import threading
        watchdog = threading.Timer(float(os.environ.get("LH_BENCH_SHARDED_TIMEOUT", "240")), give_up)
        watchdog.daemon = True
        watchdog.start()
        transport = hdist.attach_sharded(ctx, dist, shard_bit)

    ms_per_step, tr = timed(args.steps, args.warmup, live=not args.no_profile)
    phases = hl.lasso_last_timing(ctx)
    proof = tr.into_proof()
    proofs_per_step = 1 if sharded else world

    out = None
    if rank == 0:
        out = {
            "metric": "lasso_prove_time_ms", "value": round(ms_per_step / proofs_per_step, 3), "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": False, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u256 (BN254 Fr/Fq, 8x u32 Montgomery)", "data": "synthetic",
            "config": {"workload": desc % n, "lookups_per_proof": 1 << n, "proofs_per_step": proofs_per_step,
                       "pcs": "Zeromorph over univariate KZG (BN254)" if zm else "multilinear KZG (BN254)",
                       "proof_bytes": len(proof),
                       "parallelism": ("1 proof sharded over %d GPUs (index bits [%d, %d))" % (world, shard_bit, shard_bit + world.bit_length() - 1)
                                       if sharded else "1 proof per GPU" if world > 1 else "1 GPU")},
            "lookups_per_s": round((1 << n) * proofs_per_step / (ms_per_step / 1e3)),
            "phases_ms": {k: round(v, 3) for k, v in phases.items()},
            "host_binding": host_binding,
        }
        if sharded:
            out["config"]["transport"] = transport
        if os.environ.get("LH_BENCH_MODE_FALLBACK"):
            out["mode_fallback"] = {"ran": "replicas", "sharded_error": os.environ["LH_BENCH_MODE_FALLBACK"]}
        if os.environ.get("LH_BENCH_A2A_RETRY"):
            out["transport_retry"] = {"ran": "sharded, personalised exchange staged through all-gathers (LH_COMM_A2A=allgather)",
                                      "first_attempt_error": os.environ["LH_BENCH_A2A_RETRY"]}
    if watchdog is not None:
        # the headline is measured: from here on a stuck extra must not take it along (give_up above)
        watchdog.cancel()
        headline["line"] = out
        headline["done"] = True
        watchdog = threading.Timer(float(os.environ.get("LH_BENCH_EXTRAS_TIMEOUT", "420")), give_up)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_profile:
        # a separately profiled prove (every instrumented launch synchronised); sharded: every rank takes part in the
        # collectives, rank 0 records
        if rank == 0:
            hl.profile_enable(ctx, True)
        if sharded or rank == 0:
            if sharded:
                hl.comm_phase_stats(ctx, reset=True)
            prove()
            ctx.sync()
            if sharded and rank == 0:
                # collectives and bytes (this rank's contribution) of ONE proof by phase: what a measured scaling curve is
                # read against, next to the per-rank compute profile (profiles/*_sharded_rank_ms.json)
                out["comm_by_phase_one_proof"] = hl.comm_phase_stats(ctx, reset=True)
        if rank == 0:
            aggs = aggregate(hl.profile_read(ctx))
            hl.profile_enable(ctx, False)
            dom = dominant(aggs)
            out["roofline"], out["alu"], out["kernels"] = roofline_objects(
                hl, ctx, aggs, pmc_traffic(dom["name"], n, args.table, world, dom["launches"]), live_box.get("acc"), args.steps)
    if world > 1:
        # device memory per rank after the headline's proofs (the workspace arena's high-water mark; SRS and lookup columns are
        # on top of it): gathered over the control plane
        mem = hdist.gather_objects(dist, hl.memory_stats(ctx))
        if rank == 0:
            out["memory_by_rank"] = [{"arena_high_water_mib": round(m["arena_high_water_bytes"] / 2**20, 1),
                                      "device_used_mib": round((m["device_total_bytes"] - m["device_free_bytes"]) / 2**20, 1)}
                                     for m in mem]
    if sharded:
        if rank == 0:
            out["comm_round"] = {0: "all-gather + sum-and-publish kernel", 1: "one all-reduce of u64 lanes into host memory",
                                 2: "one all-reduce of u64 lanes, then a copy"}[hl.get_option(ctx, "comm_round")]
        # the sharded proof against the single-GPU prover on the same lookups (rank 0 holds the whole batch for it)
        stats = hl.comm_stats(ctx)
        if rank == 0:
            out["comm_collectives_per_run"] = stats
            if n <= 26:
                full = [ctx.upload(c.tobytes()) for c in gen_dims(table, n, 0)]
                out["sharded_proof_equals_single_gpu"] = prove(bufs=full, single=True).into_proof() == proof
                del full
    if sharded and not args.no_extra:
        # ---- the first lease of a multi-GPU node has to choose by itself: A/B of the two switches whose best setting depends
        # on what a collective costs on the links - how a sharded round's partial sums are combined (comm_round 0: all-gather +
        # publish kernel, 1: one all-reduce into host memory; the attach-time probe turns 1 into 2 where the fabric refuses
        # host memory) and when a sum-check's residual tables travel (shard_exchange_log 19 / 20: two collectives fewer per
        # sum-check for ~2 % more replicated compute).  3 timed proofs each; when the best setting beats the default by more
        # than 2 % the headline is timed AGAIN with it - same K steps, same barriers - and that is the line's `value` (the
        # first timing stays in the object as `default_ms_per_proof`).
        ab_steps = max(1, min(args.steps, 3))
        default_cfg = (hl.get_option(ctx, "comm_round"), hl.get_option(ctx, "shard_exchange_log"))
        ab = {"steps": ab_steps, "default": {"comm_round": default_cfg[0], "shard_exchange_log": default_cfg[1]}, "runs": []}
        best = (ms_per_step, default_cfg)
        try:
            for cr in (0, 1):
                for xl in (19, 20):
                    hl.set_option(ctx, "comm_round", cr)
                    hl.set_option(ctx, "shard_exchange_log", xl)
                    s0 = hl.comm_stats(ctx)
                    hl.comm_phase_stats(ctx, reset=True)
                    ms_ab, _ = timed(ab_steps, 1)
                    s1 = hl.comm_stats(ctx)
                    by_phase = hl.comm_phase_stats(ctx, reset=True)
                    runs = ab_steps + 1
                    ab["runs"].append({"comm_round": cr, "shard_exchange_log": xl, "ms_per_proof": round(ms_ab, 3),
                                       "device_collectives_per_proof": (s1["device"] - s0["device"]) // runs,
                                       "host_collectives_per_proof": (s1["host"] - s0["host"]) // runs,
                                       "bytes_contributed_per_proof": sum(v["bytes"] for v in by_phase.values()) // runs})
                    if ms_ab < best[0]:
                        best = (ms_ab, (cr, xl))
            ab["best"] = {"comm_round": best[1][0], "shard_exchange_log": best[1][1], "ms_per_proof": round(best[0], 3)}
            # (every rank sees the same max-over-ranks timings: the same choice everywhere)
            if best[1] != default_cfg and best[0] < 0.98 * ms_per_step:
                hl.set_option(ctx, "comm_round", best[1][0])
                hl.set_option(ctx, "shard_exchange_log", best[1][1])
                ms2, tr2 = timed(args.steps, args.warmup)
                ab["headline_retimed_with_best"] = True
                if rank == 0:
                    assert tr2.into_proof() == proof, "a route option changed the proof bytes"
                    out["default_ms_per_proof"] = out["value"]
                    out["value"] = out["ms_per_step"] = round(ms2, 3)
                    out["lookups_per_s"] = round((1 << n) / (ms2 / 1e3))
                    out["phases_ms"] = {k: round(v, 3) for k, v in hl.lasso_last_timing(ctx).items()}
                    out["comm_round"] = {0: "all-gather + sum-and-publish kernel", 1: "one all-reduce of u64 lanes into host memory",
                                         2: "one all-reduce of u64 lanes, then a copy"}[best[1][0]]
            else:
                hl.set_option(ctx, "comm_round", default_cfg[0])
                hl.set_option(ctx, "shard_exchange_log", default_cfg[1])
                ab["headline_retimed_with_best"] = False
        except Exception as e:
            ab["error"] = "%s: %s" % (type(e).__name__, e)
            hl.set_option(ctx, "comm_round", default_cfg[0])
            hl.set_option(ctx, "shard_exchange_log", default_cfg[1])
        if rank == 0:
            out["ab"] = ab
        # ---- TWO sharded proofs in flight per rank: a second ctx on the same device (own stream, arena, communicator), a host
        # thread each; the latency-bound stretches of one proof - resident rounds, MSM tails, collectives and the waits for
        # peers - run under the streaming kernels of the other (what two_proofs_in_flight measures on one GPU, for shards)
        try:
            import threading
            ctx2 = hl.Context(int(os.environ.get("LH_DEVICE", local_rank)))
            pp2 = pp.view(ctx2)
            for name in ("comm_round", "shard_exchange_log"):
                hl.set_option(ctx2, name, hl.get_option(ctx, name))
            group2 = hdist.flight_group(dist)
            transport2 = hdist.attach_sharded(ctx2, dist, shard_bit, group=group2)
            flights = [(pp, ctx), (pp2, ctx2)]

            def flight(p, k):
                for _ in range(k):
                    hl.lasso_prove_sharded(p, table, n, d_dims, hl.Keccak256Transcript())
            tif_steps = max(1, min(args.steps, 5))
            elapsed2 = None
            for k in (1, tif_steps):
                th = [threading.Thread(target=flight, args=(p, k)) for p, _ in flights]
                ctx2.sync()
                barrier()
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                ctx.sync(), ctx2.sync()
                elapsed2 = hdist.max_over_ranks(dist, time.perf_counter() - t0)
                hdist.barrier(dist)
            hl.detach_comm(ctx2)
            if rank == 0:
                ms2f = elapsed2 * 1e3 / (2 * tif_steps)
                out["sharded_two_in_flight"] = {"ms_per_proof": round(ms2f, 3), "steps_per_flight": tif_steps,
                                                "lookups_per_s": round((1 << n) / (ms2f / 1e3)), "transport": transport2}
            del pp2, ctx2
        except Exception as e:
            if rank == 0:
                out["sharded_two_in_flight"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1 and not args.no_extra:
        # extra objects next to the headline: BASELINE.json configs[3] (2^26 range-check lookups, one proof sharded over
        # the N GPUs) and N independent replicas of the headline workload (weak scaling, no data-path collective)
        extra_steps = max(1, min(args.steps, 3))
        n3 = int(os.environ.get("LH_BENCH_CONFIG3_LOG_N", "26"))  # (smaller in the tests)
        # (an extra that fails the same way on every rank - out of memory, an SRS too small - is reported in its object and
        # must not take the headline with it)
        if sharded and not (n == n3 and args.table == "range") and not zm:
            try:
                t26, _ = make_table(hl, "range")
                sb26 = shard_geometry(t26, n3)
                hl.detach_comm(ctx)
                pp26 = setup(n3)
                cols26 = load_columns(t26, n3, sb26)
                hdist.attach_sharded(ctx, dist, sb26)
                ms26, tr26 = timed(extra_steps, 1, nn=n3, bufs=cols26, p=pp26, tb=t26)
                if rank == 0:
                    out["config3_2p26_range_sharded"] = {"workload": "2^%d range-check Lasso lookup, one proof sharded over %d GPUs"
                                                                     % (n3, world),
                                                         "ms_per_proof": round(ms26, 3), "steps": extra_steps,
                                                         "lookups_per_s": round((1 << n3) / (ms26 / 1e3)),
                                                         "proof_bytes": len(tr26.into_proof())}
                del cols26, pp26
            except Exception as e:
                if rank == 0:
                    out["config3_2p26_range_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}
            hl.detach_comm(ctx)
            hdist.attach_sharded(ctx, dist, shard_bit)
        if sharded and replicas is not None and rank == 0:
            out["replicas"] = replicas
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if not args.no_inflight and not sharded and world == 1:
            # throughput with TWO independent proofs in flight on the same GPU (two contexts = two streams, one host
            # thread each): a single proof leaves the chip idle during its latency-bound rounds and MSM tails
            import threading
            ctx2 = hl.Context(int(os.environ.get("LH_DEVICE", local_rank)))
            # the SRS is device memory: shared, owned by `pp`
            if zm:
                shared = hl.UnivariateKzgParams(ctx2, pp.params.h)
                pp2 = hl.ZeromorphProverParam(shared, pp.poly_size)
            else:
                shared = pp2 = pp.view(ctx2)

            def worker(p, k):
                for _ in range(k):
                    hl.lasso_prove(p, table, n, d_dims, hl.Keccak256Transcript())
            for k in (1, args.steps):
                th = [threading.Thread(target=worker, args=(p, k)) for p in (pp, pp2)]
                ctx.sync(), ctx2.sync()
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                ctx.sync(), ctx2.sync()
                dt = time.perf_counter() - t0
            shared.h = None
            out["two_proofs_in_flight"] = {"ms_per_proof": round(dt * 1e3 / (2 * args.steps), 3),
                                           "lookups_per_s": round((1 << n) * 2 * args.steps / dt)}
        if not args.no_cpu_baseline and not sharded and world == 1:  # the contract: rank 0 at N = 1 only
            def gpu_proof(nn, dims):
                bufs = [ctx.upload(d.tobytes()) for d in dims]
                return prove(nn, bufs).into_proof()
            try:
                out["cpu_baseline"] = cpu_baseline(hl, ctx, pp, table, args.table, args, gpu_proof)
            except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "ms", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "unavailable: %s" % e}
        emit(out)
    if sharded:
        hl.detach_comm(ctx)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


_REAL_STDOUT = None  # the process's stdout as it was started with, once claim_stdout() has moved fd 1 to stderr


def claim_stdout():
    """bench.py's contract is ONE JSON line on stdout.  Libraries print there too (gloo notes every connected rank, RCCL may
    print a banner): from here on file descriptor 1 is stderr - for this process, its libraries and whatever it starts -
    and only emit() writes to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:  # (imported as a module: no claim)
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def spawn_ranks(n, argv, extra_env=None):
    """start the n ranks of a job as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run
    sets them; rank 0's stdout is this process's stdout) and return the first non-zero exit code, else 0.  The caller has
    not touched a GPU - or is on its way out."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=_REAL_STDOUT if r == 0 else subprocess.DEVNULL))
    rc = 0
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in procs:  # a dead rank leaves its peers in a collective: stop them (exact PIDs we started)
                    q.terminate()
        time.sleep(0.1)
    return rc


def successor_env(extra):
    """environment of a fresh successor job started by every rank of a failed one: same RANK / WORLD_SIZE, the rendezvous
    one step of 17 ports further up, and NO agent store - under torch.distributed.run every worker inherits
    TORCHELASTIC_USE_AGENT_STORE=True, which makes every rank (rank 0 included) a TCPStore CLIENT of a store the
    launcher's agent hosts on MASTER_PORT; nothing listens on the new port, so the successor's rank 0 must host it."""
    env = dict(os.environ)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + 17)
    env.update(extra)
    return env


def sharded_fallback(err):
    """A sharded run that failed the same way on every rank (an exception, not a hang) is re-run in FRESH processes - this
    one has live contexts and a process group in an unknown state (a new child, never a re-exec of a process that touched
    the GPU).  First once more as a sharded proof with the personalised exchange staged through all-gathers
    (LH_COMM_A2A=allgather: the grouped ncclSend / ncclRecv path is the one piece of the transport no one-GPU box can
    run), then as N independent replicas; the line says which.  Every rank starts its own successor and leaves with its
    exit code."""
    import traceback
    traceback.print_exc()
    what = "%s: %s" % (type(err).__name__, err)
    argv = [a for a in sys.argv[1:]]
    if os.environ.get("LH_COMM_A2A") != "allgather" and "LH_BENCH_A2A_RETRY" not in os.environ:
        env = successor_env({"LH_COMM_A2A": "allgather", "LH_BENCH_A2A_RETRY": what})
        return subprocess.call([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=_REAL_STDOUT)
    env = successor_env({"LH_BENCH_MODE_FALLBACK": what})
    return subprocess.call([sys.executable, os.path.abspath(__file__)] + argv + ["--mode", "replicas"], env=env, stdout=_REAL_STDOUT)


if __name__ == "__main__":
    _args = parse()
    claim_stdout()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process only starts the ranks (no GPU call before this point)
        sys.exit(spawn_ranks(_args.gpus, sys.argv[1:]))
    try:
        main()
    except Exception as e:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and _args.mode == "sharded" and "LH_BENCH_MODE_FALLBACK" not in os.environ:
            sys.stdout.flush()
            os._exit(sharded_fallback(e))  # (no teardown of this process's half-finished job: the successor spoke for it)
        raise
