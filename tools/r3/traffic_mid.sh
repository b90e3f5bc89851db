#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace --stats) and HBM traffic (--pmc FETCH_SIZE / WRITE_SIZE, separate passes, the
# corrections of tools/profile_ops_traffic.sh) of tools/r3/prof_mid.py.  Run on the GPU box through gpurun.
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3/prof_mid; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/r3/prof_mid.py > "$OUT/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$OUT/stats.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/r3/prof_mid.py > "$OUT/fetch.log" 2>&1 || { echo "fetch run failed"; tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/r3/prof_mid.py > "$OUT/write.log" 2>&1 || { echo "write run failed"; tail -5 "$OUT/write.log"; exit 1; }
python3 - <<PY | tee "$OUT/summary.txt"
import csv, glob, collections
def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
st = {}
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): st[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3)
print("%-96s %5s %9s %9s %10s %10s" % ("kernel (n = 4e6 unless noted)", "calls", "avg us", "min us", "read MB", "write MB"))
for k in sorted(st, key=lambda k: -st[k][1]):
    if k.startswith("void at::") or "rocclr" in k: continue
    r = (sum(fe[k]) / len(fe[k]) * 1024 * 2 / 1e6) if k in fe else float("nan")
    w = (sum(wr[k]) / len(wr[k]) * 1024 / 1e6) if k in wr else float("nan")
    print("%-96s %5d %9.1f %9.1f %10.1f %10.1f" % (k[:96], st[k][0], st[k][1], st[k][2], r, w))
PY
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
