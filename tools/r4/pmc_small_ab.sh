#!/bin/bash
# VALU / lane-utilisation counters of the Binf register-tile kernels on small groups, for A/B builds: tools/r4/pmc_small_ab.sh libA.so libB.so ...
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1 SPX_N=${SPX_N:-32000000} SPX_GS=${SPX_GS:-2,4,8,16}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "$@"; do
  OUT=gpurun_out/r4b/pmc_${lib%.so}; rm -rf "$OUT"; mkdir -p "$OUT"
  SPX_LIB_NAME=$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT" -- python3 tools/r4/binf_small_ab.py > "$OUT.log" 2>&1 || { echo "$lib run failed"; tail -5 "$OUT.log"; }
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_group_reg" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$lib")
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
        print("  %-50s VALU instructions per wave %.0f, lanes active per VALU cycle %.1f of 64, wave cycles per wave %.0f" % (k, m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 4, m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]))
PY
done
