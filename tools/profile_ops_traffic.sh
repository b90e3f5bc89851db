#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of every operator at full size: tools/prof_ops.py.
# Run on the GPU box through gpurun; corrections as in profiles/traffic_l1box.json (FETCH_SIZE x2 for 128-byte streaming reads, KiB).
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_traffic; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/prof_ops.py > "$OUT/fetch.log" 2>&1 || { echo "fetch run failed"; tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/prof_ops.py > "$OUT/write.log" 2>&1 || { echo "write run failed"; tail -5 "$OUT/write.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
print("%-86s %6s %12s %12s %12s" % ("kernel", "calls", "read GB", "write GB", "total GB"))
for k in sorted(fe, key=lambda k: -sum(fe[k]) / len(fe[k])):
    if k.startswith("void at::") or "rocclr" in k: continue
    r = sum(fe[k]) / len(fe[k]) * 1024 * 2 / 1e9
    w = (sum(wr[k]) / len(wr[k]) * 1024 / 1e9) if k in wr else float("nan")
    if r + (w if w == w else 0) < 0.01: continue
    print("%-86s %6d %12.3f %12.3f %12.3f" % (k[:86], len(fe[k]), r, w, r + w))
PY
