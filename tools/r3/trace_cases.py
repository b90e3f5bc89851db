"""Per-case kernel times from a rocprofv3 kernel trace of tools/r3/topr_ties.py: the trace is split at the reference calls
(k_sel_coop<.., false> alone = the exact select with key 2 = 0); within a case, the median duration of every kernel."""
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]
cases, cur = [], None
for r in rows:
    name = short(r["Kernel_Name"])
    if name.startswith("at::") or name.startswith("__amd"):
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if name.startswith("k_sel_coop") and (cur is None or cur.get("_last") != "k_s2_tail"):
        pass
    if name.startswith("k_sel_coop"):
        cur = {"_ref": d, "_last": name}
        cases.append(cur)
        continue
    if cur is None:
        continue
    cur.setdefault(name, []).append(d)
    cur["_last"] = name.split("<")[0]
for i, c in enumerate(cases):
    parts = ["%s %.1f" % (k, statistics.median(v)) for k, v in c.items() if not k.startswith("_")]
    tot = sum(statistics.median(v) for k, v in c.items() if not k.startswith("_"))
    print("case %2d | exact select %.0f us | pipeline kernels (median us): %s | sum %.0f" % (i, c["_ref"], "; ".join(parts), tot))
