set -uo pipefail
export TMPDIR=/tmp SPX_NO_BUILD=1 SPX_KINDS=continuous SPX_RS=5e7
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3/pmc_compact; rm -rf "$OUT"; mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 tools/r3/topr_ties.py > "$OUT/$c.log" 2>&1 || { echo "$c run failed"; tail -5 "$OUT/$c.log"; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for c, mul in (("FETCH_SIZE", 2048), ("WRITE_SIZE", 1024)):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "k_s2" in k: print(c, k, "calls", len(v), "MB per call %.2f" % (sum(v) / len(v) * mul / 1e6))
PY
