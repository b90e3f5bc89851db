#!/bin/bash
# (1) WRITE_SIZE per dispatch of k_b2_coop on a fresh context; (2) VALU / lane-utilisation counters of the Binf kernel on groups of 8
set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4/pmc2; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/b2w" -- python3 tools/r4/b2_first_call.py > "$OUT/b2w.log" 2>&1 || { echo "b2 run failed"; tail -5 "$OUT/b2w.log"; }
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/b2w/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_b2_coop" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    print("k_b2_coop WRITE_SIZE per dispatch (GB):", ["%.3f" % (float(r["Counter_Value"]) * 1024 / 1e9) for r in rows])
PY
SPX_OPS=binf rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/g8" -- python3 tools/prof_ops.py > "$OUT/g8.log" 2>&1 || { echo "g8 run failed"; tail -5 "$OUT/g8.log"; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g8/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_group_reg" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: "%.4g" % v for c, v in m.items()})
    if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
        print("   lanes active per VALU cycle: %.1f of 64;  VALU instructions per wave: %.0f" % (m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 4 * 1, m["SQ_INSTS_VALU"] / m["SQ_WAVES"]))
PY
