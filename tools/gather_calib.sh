#!/bin/bash
# FETCH_SIZE of the three access shapes of tools/ubench/gather_calib.hip against their known byte counts (GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/gcal
tools/ubench/gather_calib.bin
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/gcal -- tools/ubench/gather_calib.bin > $O/gcal.log 2>&1
F=$(find $O/gcal -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'P'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
n = 1 << 24
known = {"stream64": n * 64 + 0, "gather64": n * 64 + n * 4, "gather128": n * 128 + n * 4}
for k, v in acc.items():
    if k not in known:
        continue
    kb = sum(v) / len(v)
    print("%-10s FETCH_SIZE %.1f MB (counter x 1 KB) against %.1f MB read by the lanes: ratio %.3f" % (k, kb * 1024 / 1e6, known[k] / 1e6, kb * 1024 / known[k]))
P
rm -rf $O/gcal
