"""Per-kernel summary of a rocprofv3 --kernel-trace csv: calls, avg/min/max duration, and the average gap to the previous kernel."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = collections.OrderedDict(); prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    if name.startswith("void at::") or "elementwise" in name or "distribution" in name: prev_end = int(r["End_Timestamp"]); continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = st.setdefault(name[:90], [0, 0.0, 1e30, 0.0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3; d[2] = min(d[2], (e - s) / 1e3); d[3] = max(d[3], (e - s) / 1e3)
    if prev_end is not None: d[4] += max(0, (s - prev_end)) / 1e3
    prev_end = e
for k, d in st.items():
    print("%-92s calls=%4d avg=%8.1f us min=%8.1f max=%8.1f  avg gap before=%6.1f us" % (k, d[0], d[1] / d[0], d[2], d[3], d[4] / d[0]))
