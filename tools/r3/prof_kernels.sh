#!/bin/bash
# rocprofv3 kernel trace of one python tool; prints per-kernel averages.  usage: tools/r3/prof_kernels.sh <tag> <script> [env assignments are inherited]
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
OUT=gpurun_out/r3/prof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$@" > "$OUT/run.log" 2>&1 || { echo "profiled run failed"; tail -5 "$OUT/run.log"; exit 1; }
python3 - <<PY > "$OUT/kernels.txt"
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("void at::") : continue
    print("%-100s calls=%s avg=%.1f us min=%.1f max=%.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
grep -v amdgpu.ids "$OUT/run.log" | tail -12
cat "$OUT/kernels.txt"
python3 tools/r3/trace_cases.py "$OUT/stats" > "$OUT/cases.txt" 2>&1; cat "$OUT/cases.txt"
rm -rf "$OUT/stats"
