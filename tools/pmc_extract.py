"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv (two separate passes over the same program) -> JSON of
per-kernel HBM traffic: per-launch average (what bench.py's roofline.traffic reports next to the per-launch `achieved`),
total over the run, and the largest launch.  Corrections per MI355X_MICROARCH.md §HBM: counters are in KB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B per lane), WRITE_SIZE is exact for
16-B-per-lane streaming stores.  Gather kernels are flagged: for msm_accumulate0's shape (a random 64-byte record per lane)
the same formula gives the LINE traffic - a 64-byte gather counts, and costs, as its whole 128-byte line
(tools/gather_calib.sh, profiles/r05_gather_calib.txt) - i.e. twice the bytes the lanes use.

usage: pmc_extract.py FETCH.csv WRITE.csv "workload description" proofs_in_run OUT.json"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.match(r"(?:void )?(?:lh::)?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def load(path, counter):
    acc = collections.defaultdict(lambda: dict(n=0, total=0.0, big=0.0, grid=0))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        v = float(r["Counter_Value"])
        a["n"] += 1
        a["total"] += v
        if v > a["big"]:
            a["big"], a["grid"] = v, int(r["Grid_Size"])
    return acc


GATHER = {"msm_accumulate0_kernel", "msm_derived_gather_kernel", "rotate_gather_kernel"}
fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
proofs = int(sys.argv[4])
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, dict(n=0, total=0.0, big=0.0, grid=0)), write.get(k, dict(n=0, total=0.0, big=0.0, grid=0))
    n = max(f["n"], w["n"])
    out[k] = {"launches_in_run": n, "launches_per_proof": n / proofs,
              "hbm_bytes_per_launch_avg": int((2.0 * f["total"] + w["total"]) * 1024 / max(n, 1)),
              "hbm_bytes_per_proof": int((2.0 * f["total"] + w["total"]) * 1024 / proofs),
              "largest_launch": {"grid_threads": f["grid"] or w["grid"], "FETCH_SIZE_KB_raw": f["big"],
                                 "WRITE_SIZE_KB": w["big"], "hbm_bytes_corrected": int((2.0 * f["big"] + w["big"]) * 1024)},
              # (every kernel is calibrated since round 5: wide streams by the guide's x2, the gather kernels by
              # tools/gather_calib.sh - a 64-byte gather reads 0.967 counter-KB per KB of records, i.e. with x2 the 128-byte
              # lines it really moves, 2.07 bytes of HBM traffic per byte the lanes use)
              "calibrated": True,
              "calibration": ({"by": "tools/gather_calib.sh", "counter_per_record_byte": 0.967, "hbm_per_record_byte": 1.934}
                              if k.split("<")[0] in GATHER else {"by": "MI355X_MICROARCH.md", "counter_per_stream_byte": 0.5})}
json.dump({"note": "hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB, the streaming-read correction of MI355X_MICROARCH.md; "
                   "gather kernels (calibration.by = tools/gather_calib.sh): the figure is their 128-byte-line traffic, about twice the 64-byte records the lanes use (profiles/r05_gather_calib.txt)",
           "workload": sys.argv[3], "proofs_in_run": proofs, "kernels": out}, open(sys.argv[5], "w"), indent=1)
print("wrote", sys.argv[5], len(out), "kernels")
