#!/bin/bash
# kernel timeline (start offsets, durations, gaps) of the last call of an operator: tools/trace_gaps.sh <op>
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/trace; rm -rf "$OUT"; mkdir -p "$OUT"
SPX_OPS="$1" rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 tools/prof_ops.py > "$OUT/log" 2>&1 || { tail -5 "$OUT/log"; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if not r["Kernel_Name"].startswith("void at::")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last call = last 40 kernels at most; find the last init kernel
idx = max(i for i, r in enumerate(rows) if "init" in r["Kernel_Name"] or i == 0)
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:70]))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
PY
