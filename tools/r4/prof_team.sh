#!/bin/bash
# rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the team form and the chunked psi(y):
# tools/prof_ops.py with SPX_OPS=team.  -> gpurun_out/r4/prof_team_{stats,traffic}.txt
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1 SPX_OPS="${SPX_OPS:-team}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4/prof_team; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/prof_ops.py > "$OUT/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$OUT/stats.log"; exit 1; }
python3 - <<PY > gpurun_out/r4/prof_team_stats.txt
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("void at::") : continue
    print("%-100s calls=%s avg=%.1f us min=%.1f max=%.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
cat gpurun_out/r4/prof_team_stats.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/prof_ops.py > "$OUT/fetch.log" 2>&1 || { echo "fetch run failed"; tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/prof_ops.py > "$OUT/write.log" 2>&1 || { echo "write run failed"; tail -5 "$OUT/write.log"; exit 1; }
python3 - <<PY > gpurun_out/r4/prof_team_traffic.txt
import csv, glob, collections
def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
print("# HBM traffic per launch (FETCH_SIZE x 2 x 1024 B: the guide's gfx950 correction for 128-byte streaming reads; WRITE_SIZE x 1024 B)")
print("%-96s %6s %12s %12s %12s" % ("kernel", "calls", "read GB", "write GB", "total GB"))
for k in sorted(fe, key=lambda k: -sum(fe[k]) / len(fe[k])):
    if k.startswith("void at::") or "rocclr" in k: continue
    r = sum(fe[k]) / len(fe[k]) * 1024 * 2 / 1e9
    w = (sum(wr[k]) / len(wr[k]) * 1024 / 1e9) if k in wr else float("nan")
    if r + (w if w == w else 0) < 0.01: continue
    print("%-96s %6d %12.3f %12.3f %12.3f" % (k[:96], len(fe[k]), r, w, r + w))
PY
cat gpurun_out/r4/prof_team_traffic.txt
