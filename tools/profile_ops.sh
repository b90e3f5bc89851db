#!/bin/bash
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_ops; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/prof_ops.py > "$OUT/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$OUT/stats.log"; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("void at::") : continue
    print("%-90s calls=%s avg=%.1f us min=%.1f max=%.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
