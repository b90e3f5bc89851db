"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into a short text summary."""
import csv, glob, os, sys, json
root = sys.argv[1]
def find(sub, pat):
    r = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return r[0] if r else None
st = find("stats", "*kernel_stats.csv")
if st:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(st)))
    for r in rows[:12]:
        print({k: r[k][:110] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
res = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print("no counter file for", ctr); continue
    vals = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == ctr:
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    print("== %s per dispatch (raw counter units) ==" % ctr)
    for k, v in vals.items():
        print("  %-90s n=%d mean=%.6g" % (k[:90], len(v), sum(v) / len(v)))
        if "k_sep_" in k and "OpL1Box" in k:
            res[ctr] = sum(v) / len(v)
if res:
    print("raw", json.dumps(res))
