#!/bin/bash
# PMC pass for one operator: tools/pmc_ops.sh <ops> <counters...>
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OPS="$1"; shift
OUT=gpurun_out/pmc_ops; rm -rf "$OUT"; mkdir -p "$OUT"
SPX_OPS="$OPS" rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc" -- python3 tools/prof_ops.py > "$OUT/pmc.log" 2>&1 || { echo "pmc run failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("void at::"): continue
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: "%.4g" % (sum(v)/len(v)) for c, v in d.items()})
PY
